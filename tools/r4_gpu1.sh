# Round-4 GPU pass 1 (through gpurun): the GPU suite, the default bench line (all workloads), the in-replay phase tables.
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/r4; mkdir -p $out
what=${1:-all}
if [ "$what" = all ] || [ "$what" = tests ]; then
    timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; tail -n 15 $out/gpu_tests.txt
fi
if [ "$what" = all ] || [ "$what" = bench ]; then
    ( time timeout 600 python bench.py ) > $out/bench_default.json 2> $out/bench_default.err; tail -n 4 $out/bench_default.err; cut -c1-600 $out/bench_default.json
fi
if [ "$what" = all ] || [ "$what" = phases ]; then
    for w in cfg2 cfg3; do
        CINDM_LIB_VARIANT=prof timeout 300 python tools/phase_table.py $w 40 > $out/phase_table_$w.txt 2> $out/phase_table_$w.err; tail -n 3 $out/phase_table_$w.err; head -n 30 $out/phase_table_$w.txt
    done
fi
