echo "== cfg2_noise philox, quiet producers"; timeout 400 python tools/ablate.py run mdpp_discrete_quiet.hip cfg2_noise philox disable=NO_LEAN q000 qh3e1o2 qh2e1o3 qh0e3o2 qh0e3o0 qh0e0o3
echo "== cfg2_noise numpy (quiet E/O)"; timeout 400 python tools/ablate.py run mdpp_discrete_quiet.hip cfg2_noise numpy q000 qh0e3o2 qh0e3o0 qh0e0o3
echo "== cfg2_irr numpy (quiet trio)"; timeout 400 python tools/ablate.py run mdpp_discrete_quiet.hip cfg2_irr numpy disable=NO_LEAN q000 qh3e1o2 qh0e3o2 qh0e3o0 qh2e1o3
echo "== cfg5 numpy"; timeout 400 python tools/ablate.py run mdpp_continuous_fast.hip cfg5 numpy np00 np30 np03 np21
