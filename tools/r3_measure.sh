# Round-3 measurements on the GPU box (run through gpurun): bench lines of every workload, rocprofv3 kernel traces and the
# two separate PMC passes per workload; results land in gpurun_out/meas3/ and are copied into profiles/ by hand.
#   usage: bash tools/r3_measure.sh [1d|2d|force|all]
what=${1:-all}
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/meas3; mkdir -p $out
pmc() {   # pmc <tag> <steps incl. warm-up> <command...>: FETCH_SIZE and WRITE_SIZE in separate passes -> $out/pmc_traffic_<tag>.json
    tag=$1; steps=$2; shift 2
    cd /tmp; rm -rf /tmp/pf_$tag /tmp/pw_$tag
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$tag -- "$@" > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$tag -- "$@" > /dev/null 2>&1
    cd /root/repo; python3 tools/pmc_traffic.py /tmp/pf_$tag /tmp/pw_$tag $steps > $out/pmc_traffic_$tag.json
}
ktrace() {   # ktrace <tag> <command...>: per-kernel stats of a kernel trace -> $out/kstats_<tag>.txt
    tag=$1; shift
    cd /tmp; rm -rf /tmp/kt_$tag
    rocprofv3 --kernel-trace -d /tmp/kt_$tag -o t -- "$@" > /tmp/kt_$tag.log 2>&1
    cd /root/repo
    (echo "# rocprofv3 --kernel-trace -- $*"; tail -1 /tmp/kt_$tag.log | cut -c1-300; python3 tools/rocprof_summary.py $(find /tmp/kt_$tag -name "*.db" | head -1)) > $out/kstats_$tag.txt
}
if [ "$what" = 1d ] || [ "$what" = all ]; then
    for w in cfg2 cfg3 cfg4; do
        pmc $w 22 python3 /root/repo/tools/prof_wl.py $w 20
        cp $out/pmc_traffic_$w.json profiles/r03_pmc_traffic_$w.json          # bench.py reads the committed name (hash-checked)
    done
    cp profiles/r03_pmc_traffic_cfg2.json profiles/r03_pmc_traffic_cfg2-ddim250.json 2>/dev/null
    ktrace bench_cfg2 python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline
    ktrace cfg3 python3 /root/repo/tools/prof_wl.py cfg3 50
    ktrace cfg4 python3 /root/repo/tools/prof_wl.py cfg4 50
    for w in cfg2 cfg3 cfg4 cfg2-ddim250; do
        python bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err; cut -c1-200 $out/bench_$w.json
    done
fi
if [ "$what" = 2d ] || [ "$what" = all ]; then
    pmc cfg5 7 python3 /root/repo/tools/prof2d.py 64 2 5
    cp $out/pmc_traffic_cfg5.json profiles/r03_pmc_traffic_cfg5.json
    ktrace cfg5 python3 /root/repo/tools/prof2d.py 64 2 10
    python bench.py --workload cfg5 > $out/bench_cfg5.json 2> $out/bench_cfg5.err; cut -c1-200 $out/bench_cfg5.json
fi
if [ "$what" = force ] || [ "$what" = all ]; then
    pmc force 0 python3 /root/repo/tools/bench_force.py 64 2 3
    cp $out/pmc_traffic_force.json profiles/r03_pmc_traffic_force.json        # the guided bench line adds the surrogate's bytes (hash-checked)
    ktrace force python3 /root/repo/tools/bench_force.py 64 2 10
    python bench.py --workload cfg5g --steps 1 --warmup 1 > $out/bench_cfg5g.json 2> $out/bench_cfg5g.err; cut -c1-200 $out/bench_cfg5g.json
fi
