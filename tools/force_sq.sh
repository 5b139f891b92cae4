# SQ-level counters of the ForceUnet kernels (one PMC pass per counter group; see /opt/skills/guides/MI355X_MICROARCH.md)
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/sq$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sq$i -- python3 /root/repo/tools/bench_force.py 64 2 > /tmp/sq$i.log 2>&1 || tail -3 /tmp/sq$i.log
done
cd /root/repo
python3 - <<'PY'
import csv, glob
from collections import defaultdict
want = ["fu_conv_kernel<1, 4, 0>", "fu_la_bwd_fused", "conv2d_ws_kernel<0, 0>", "fu_gn_silu_kernel"]
for i in range(1, 5):
    acc = defaultdict(lambda: defaultdict(float)); best = {}
    for f in glob.glob(f"/tmp/sq{i}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = next((w for w in want if w in row["Kernel_Name"]), None)
            if not k: continue
            g = int(row.get("Grid_Size", "0") or 0)
            # keep the largest-grid launch of each kernel (the 64 x 64 level), last occurrence
            key = (k, row["Counter_Name"])
            if g >= best.get(key, (0, 0))[0]: best[key] = (g, float(row["Counter_Value"]))
    for (k, c), (g, v) in sorted(best.items()): print(f"{k:28s} grid {g:>10d}  {c:28s} {v:.4g}")
PY
