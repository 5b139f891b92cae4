# ForceUnet after the cluster GroupNorm derivative: tests, A/B, kernel trace
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r4; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_force.py tests/test_gpu_range.py -m gpu -x -q 2>&1 | tail -n 3
timeout 300 python tools/ab_force.py gn_on_load 0 1 2 2>&1 | grep "per gradient\|difference"
cd /tmp; rm -rf /tmp/ktf; rocprofv3 --kernel-trace -d /tmp/ktf -o t -- python3 /root/repo/tools/bench_force.py 64 2 10 > /tmp/ktf.log 2>&1; tail -n 2 /tmp/ktf.log
cd /root/repo; python3 tools/rocprof_summary.py $(find /tmp/ktf -name "*.db" | head -1) > $out/kstats_force.txt; head -n 14 $out/kstats_force.txt | cut -c1-150
