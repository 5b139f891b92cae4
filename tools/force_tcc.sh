# L2 (TCC) and vector-L1 (TCP) counters of the ForceUnet gradient's fp32 1x1 convolutions (one PMC pass per group), largest launch of each kind
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_WRITEBACK_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/tc$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/tc$i -- python3 /root/repo/tools/bench_force.py 64 2 > /tmp/tc$i.log 2>&1 || tail -3 /tmp/tc$i.log
done
cd /root/repo
python3 - <<'PY'
import csv, glob
want = ["fu_conv_kernel<1, 4, 1>", "fu_conv_kernel<1, 4, 2>", "fu_conv_kernel<1, 4, 0>", "fu_la_bwd_a_kernel<64", "conv2d_ws_kernel<0, 0>"]
for i in range(1, 5):
    best = {}
    for f in glob.glob(f"/tmp/tc{i}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = next((w for w in want if w in row["Kernel_Name"]), None)
            if not k: continue
            g = int(row.get("Grid_Size", "0") or 0)
            key = (k, row["Counter_Name"])
            if g >= best.get(key, (0, 0))[0]: best[key] = (g, float(row["Counter_Value"]))
    for (k, c), (g, v) in sorted(best.items()): print(f"{k:28s} grid {g:>10d}  {c:30s} {v:.4g}")
PY
