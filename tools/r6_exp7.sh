# Round 6, experiment 7: the 2-D / surrogate launches' output rows written through (option "tune" bit 1 = 2: round 5's plain stores)
cd /root/repo; export TMPDIR=/tmp; out=gpurun_out/r6; mkdir -p $out
python tools/ab2d.py tune 2 0 20 2>&1 | grep ms/step | tee $out/ab2d_wt.txt
python tools/ab_force.py tune 2 0 3 2>&1 | grep -E "ms per|difference" | tee $out/abforce_wt.txt
timeout 1500 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_force.py -x -q 2>&1 | tail -3
