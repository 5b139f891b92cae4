# Round-4 GPU pass: 'after' phase tables (cfg2, cfg3; profiling build of the current tree) + the cfg5 kernel trace
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r4; mkdir -p $out
CINDM_LIB_VARIANT=prof timeout 300 python tools/phase_table.py cfg2 40 > $out/phase_table_cfg2.txt 2> $out/pt.err; head -n 3 $out/phase_table_cfg2.txt
CINDM_LIB_VARIANT=prof timeout 300 python tools/phase_table.py cfg3 40 > $out/phase_table_cfg3.txt 2>> $out/pt.err; tail -n 1 $out/phase_table_cfg3.txt
cd /tmp; rm -rf /tmp/kt5; rocprofv3 --kernel-trace -d /tmp/kt5 -o t -- python3 /root/repo/tools/prof2d.py 64 2 10 > /tmp/kt5.log 2>&1
cd /root/repo; (echo "# rocprofv3 --kernel-trace -- python3 tools/prof2d.py 64 2 10"; python3 tools/rocprof_summary.py $(find /tmp/kt5 -name "*.db" | head -1)) > $out/kstats_cfg5.txt; head -n 30 $out/kstats_cfg5.txt | cut -c1-150
