# per-launch durations and idle gaps of one cfg5 reverse step (graph replay, as bench.py runs it)
cd /tmp; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r4; mkdir -p $out; rm -rf /tmp/kt5
rocprofv3 --kernel-trace -d /tmp/kt5 -o t -- python3 /root/repo/tools/prof2d.py 64 2 10 > /tmp/kt5.log 2>&1; tail -n 1 /tmp/kt5.log
cd /root/repo; python3 tools/trace_gaps.py $(find /tmp/kt5 -name "*.db" | head -1) stem7 > $out/gaps_cfg5.txt; cat $out/gaps_cfg5.txt | cut -c1-100
