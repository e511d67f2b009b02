echo "== cfg5 philox"; timeout 400 python tools/ablate.py run mdpp_continuous_fast.hip cfg5 philox c2p0 c0p2 c1p3 c0p0 c3p1
echo "== shipped"; for w in "cfg2 numpy" "cfg2 philox" "cfg2_noise philox"; do timeout 100 python tools/ablate.py run mdpp_discrete_lean.hip $w shipped; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unaligned_ticks or specialised or discrete_philox or lean" 2>&1 | tail -4
