# LDS bank conflicts per kernel (SQ counters): bash tools/r5_ldsconf.sh <tag> [python3 <script> args...] -> gpurun_out/<tag>/lds_conflicts.txt
tag=${1:-r5e3}; shift; cmd=${@:-python3 /root/repo/tools/prof2d.py 64 2 3}; cd /root/repo; export TMPDIR=/tmp; out=/root/repo/gpurun_out/$tag; mkdir -p $out
cd /tmp; rm -rf /tmp/pc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc -- $cmd > /tmp/pc.log 2>&1
cd /root/repo
python3 - <<'PY' > $out/lds_conflicts.txt
import csv, glob, collections
f = glob.glob("/tmp/pc/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": cnt[k] += 1
print(f"{'kernel':70s} {'launches':>8s} {'LDS_IDX_ACTIVE':>16s} {'BANK_CONFLICT':>16s} {'conflict %':>10s} {'INSTS_LDS':>14s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
    a, c = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)
    if a: print(f"{k:70s} {cnt[k]:8d} {a:16.0f} {c:16.0f} {100 * c / a:10.1f} {v.get('SQ_INSTS_LDS', 0):14.0f}")
PY
cat $out/lds_conflicts.txt | head -30; tail -3 /tmp/pc.log
