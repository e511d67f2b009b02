cd $GRAFT_REPO_ROOT
for v in 1 2 4 24; do
cp build/alt/libmdpp_p$v.so mdp_playground_amd/csrc/libmdpp_hip.so
echo "== parts $v"; python3 tools/repro_cfg5.py 2>&1 | grep "^4096 0 0\|^65536 0 0" | cut -c1-90
done
