# Round-6 measurements on the GPU box (run through gpurun): rocprofv3 kernel traces, the two separate PMC passes per workload
# (cfg2-ddim250 gets its OWN pass this round), the in-replay phase tables of the profiling build, and the bench lines.
# Results land in gpurun_out/meas6/; the PMC summaries are also placed in profiles/ on the box so the bench lines of the same
# call carry them (hash-checked); the copies judged are made by hand from gpurun_out/meas6/ into profiles/r06_*.
#   usage: bash tools/r6_measure.sh [1d|2d|force|phases|bench|all]
what=${1:-all}
cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/meas6; mkdir -p $out
pmc() {   # pmc <tag> <steps incl. warm-up> <command...>: FETCH_SIZE and WRITE_SIZE in separate passes -> $out/r06_pmc_traffic_<tag>.json
    tag=$1; steps=$2; shift 2
    cd /tmp; rm -rf /tmp/pf_$tag /tmp/pw_$tag
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$tag -- "$@" > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw_$tag -- "$@" > /dev/null 2>&1
    cd /root/repo; python3 tools/pmc_traffic.py /tmp/pf_$tag /tmp/pw_$tag $steps > $out/r06_pmc_traffic_$tag.json
    cp $out/r06_pmc_traffic_$tag.json profiles/r06_pmc_traffic_$tag.json
}
ktrace() {   # ktrace <tag> <command...>: per-kernel stats of a kernel trace -> $out/r06_kernel_stats_<tag>.txt
    tag=$1; shift
    cd /tmp; rm -rf /tmp/kt_$tag
    rocprofv3 --kernel-trace -d /tmp/kt_$tag -o t -- "$@" > /tmp/kt_$tag.log 2>&1
    cd /root/repo
    (echo "# rocprofv3 --kernel-trace -- $*"; tail -1 /tmp/kt_$tag.log | cut -c1-300; python3 tools/rocprof_summary.py $(find /tmp/kt_$tag -name "*.db" | head -1)) > $out/r06_kernel_stats_$tag.txt
}
if [ "$what" = 1d ] || [ "$what" = all ]; then
    for w in cfg2 cfg3 cfg4 cfg2-ddim250 cfg2-b1024 cfg1-gpu cfg2-f32mfma; do pmc $w 22 python3 /root/repo/tools/prof_wl.py $w 20; done
    ktrace bench_cfg2 python3 /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-workloads
    ktrace cfg3 python3 /root/repo/tools/prof_wl.py cfg3 50
    ktrace cfg4 python3 /root/repo/tools/prof_wl.py cfg4 50
    ktrace cfg2-ddim250 python3 /root/repo/tools/prof_wl.py cfg2-ddim250 50
fi
if [ "$what" = 2d ] || [ "$what" = all ]; then
    pmc cfg5 7 python3 /root/repo/tools/prof2d.py 64 2 5
    ktrace cfg5 python3 /root/repo/tools/prof2d.py 64 2 10
    python3 tools/trace_gaps.py $(find /tmp/kt_cfg5 -name "*.db" | head -1) stem7 > $out/r06_cfg5_step_launches.txt
fi
if [ "$what" = force ] || [ "$what" = all ]; then
    pmc force 0 python3 /root/repo/tools/bench_force.py 64 2 3
    ktrace force python3 /root/repo/tools/bench_force.py 64 2 10
fi
if [ "$what" = phases ] || [ "$what" = all ]; then
    for w in cfg2 cfg3; do
        CINDM_LIB_VARIANT=prof timeout 300 python tools/phase_table.py $w 40 > $out/r06_phase_table_$w.txt 2> $out/phase_$w.err; head -n 2 $out/r06_phase_table_$w.txt; tail -n 1 $out/r06_phase_table_$w.txt
    done
fi
if [ "$what" = bench ] || [ "$what" = all ]; then
    python bench.py > $out/r06_bench_default.json 2> $out/bench_default.err; cut -c1-400 $out/r06_bench_default.json
    python - <<'PY'
import json
l = json.load(open("/root/repo/gpurun_out/meas6/r06_bench_default.json"))
print("roofline", l.get("roofline")); print("cpu_baseline", l.get("cpu_baseline"))
for name, w in (l.get("workloads") or {}).items():
    if isinstance(w, dict):
        print(name, {k: w.get(k) for k in ("value", "us_per_reverse_step", "rel_err")}, (w.get("roofline") or {}).get("frac"), w.get("hbm_bytes_per_step"), (w.get("cpu_baseline") or {}).get("value"))
PY
fi
