cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kf
rocprofv3 --kernel-trace -d /tmp/kf -o cf -- python3 /root/repo/tools/bench_force.py 64 2 > /tmp/kf.log 2>&1
cd /root/repo; mkdir -p gpurun_out/r2x
python3 tools/rocprof_summary.py $(find /tmp/kf -name "*.db" | head -1) gpurun_out/r2x/kstats_force.txt | cut -c1-150 | head -30
