# Round-4 GPU pass: 'before' phase table (the tree of the round's first commit, profiling build), the 2-D tests, cfg5 timings + kernel trace
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r4; mkdir -p $out
if [ -d _before ]; then
    (cd _before && CINDM_LIB_VARIANT=prof timeout 300 python tools/phase_table.py cfg2 40 > $out/phase_table_cfg2_before.txt 2> $out/phase_table_before.err; tail -n 2 $out/phase_table_before.err; head -n 3 $out/phase_table_cfg2_before.txt)
fi
timeout 900 python -m pytest tests/test_gpu_parity_2d.py tests/test_gpu_paths.py -m gpu -x -q -k "2d" > $out/gpu_tests_2d.txt 2>&1; tail -n 4 $out/gpu_tests_2d.txt
python tools/prof2d.py 64 2 40 2>&1 | tail -n 1
cd /tmp; rm -rf /tmp/kt5; rocprofv3 --kernel-trace -d /tmp/kt5 -o t -- python3 /root/repo/tools/prof2d.py 64 2 10 > /tmp/kt5.log 2>&1
cd /root/repo; (echo "# rocprofv3 --kernel-trace -- python3 tools/prof2d.py 64 2 10"; python3 tools/rocprof_summary.py $(find /tmp/kt5 -name "*.db" | head -1)) > $out/kstats_cfg5.txt; head -n 24 $out/kstats_cfg5.txt | cut -c1-140
CINDM_LIB_VARIANT=prof timeout 300 python tools/phase_table.py cfg2 40 > $out/phase_table_cfg2.txt 2> $out/phase_table_cfg2.err; head -n 3 $out/phase_table_cfg2.txt
python bench.py --no-extra-workloads --no-cpu-baseline 2>/dev/null | cut -c1-300
