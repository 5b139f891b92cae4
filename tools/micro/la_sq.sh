# SQ counters of the LinearAttention backward passes in the timing harness (one PMC pass per group)
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/lsq$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/lsq$i -- /root/repo/tools/micro/la_bwd.bin 768 1024 64 > /tmp/lsq$i.log 2>&1 || tail -3 /tmp/lsq$i.log
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("/tmp/lsq*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        k = "pass A" if "bwd_a" in n else ("pass B" if "bwd_b" in n else None)
        if k: acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    for c, v in sorted(acc[k].items()): print(f"{k}  {c:34s} {sum(v) / len(v):.5g}  (n={len(v)})")
PY
