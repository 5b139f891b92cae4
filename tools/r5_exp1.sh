# round 5, experiment 1: conv2d_ws_kernel without the k-group reduction (ws_nosplit)
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r5e1; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity_2d.py -x -q -m gpu > $out/test2d.txt 2>&1; tail -3 $out/test2d.txt
timeout 300 python tools/ab2d.py ws_nosplit 0 1 20 > $out/ab_nosplit.txt 2>&1; cat $out/ab_nosplit.txt
cd /tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o t -- python3 /root/repo/tools/prof2d.py 64 2 10 > /tmp/kt.log 2>&1
cd /root/repo
python3 tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $out/kstats_cfg5.txt
python3 tools/trace_gaps.py $(find /tmp/kt -name "*.db" | head -1) stem7 > $out/cfg5_step_launches.txt
head -30 $out/kstats_cfg5.txt
timeout 600 python tools/bench_force.py 64 2 5 > $out/force.txt 2>&1; tail -3 $out/force.txt
