# ForceUnet: gradient-call time + kernel trace (quick)
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r4; mkdir -p $out
python tools/bench_force.py 64 2 10 2>&1 | tail -n 2
cd /tmp; rm -rf /tmp/ktf; rocprofv3 --kernel-trace -d /tmp/ktf -o t -- python3 /root/repo/tools/bench_force.py 64 2 10 > /tmp/ktf.log 2>&1
cd /root/repo; python3 tools/rocprof_summary.py $(find /tmp/ktf -name "*.db" | head -1) > $out/kstats_force.txt; grep "fu_conv_kernel\|fu_stem" $out/kstats_force.txt | cut -c1-150
