cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r5e8; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity_2d.py -x -q -m gpu > $out/test2d.txt 2>&1; tail -3 $out/test2d.txt
timeout 300 python tools/ab2d.py la_h16 0 1 20 > $out/ab_la.txt 2>&1; cat $out/ab_la.txt
timeout 1500 python -m pytest tests/test_gpu_force.py -x -q -m gpu > $out/testforce.txt 2>&1; tail -5 $out/testforce.txt
timeout 900 python -m pytest tests/test_gpu_paths.py -x -q -m gpu -k "second_chain or timeout" > $out/testpaths.txt 2>&1; tail -5 $out/testpaths.txt
cd /tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o t -- python3 /root/repo/tools/prof2d.py 64 2 10 > /tmp/kt.log 2>&1
cd /root/repo
python3 tools/trace_gaps.py $(find /tmp/kt -name "*.db" | head -1) stem7 > $out/cfg5_step_launches.txt; head -1 $out/cfg5_step_launches.txt; grep "la2d" $out/cfg5_step_launches.txt
timeout 600 python tools/bench_force.py 64 2 5 > $out/force.txt 2>&1; tail -3 $out/force.txt
