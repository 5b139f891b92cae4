# memory-side bytes per ForceUnet kernel (two separate PMC passes over tools/bench_force.py; see tools/pmc_traffic.py)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pf_fetch /tmp/pf_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_fetch -- python3 /root/repo/tools/bench_force.py 64 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pf_write -- python3 /root/repo/tools/bench_force.py 64 2 > /dev/null 2>&1
cd /root/repo; mkdir -p gpurun_out/r2x
python3 tools/pmc_traffic.py /tmp/pf_fetch /tmp/pf_write 0 > gpurun_out/r2x/pmc_force.json
python3 - <<'PY'
import csv, glob
# the largest launches of the 1x1 kernel: per-dispatch WRITE_SIZE / FETCH_SIZE (KiB)
for d, name in (("/tmp/pf_write", "WRITE_SIZE"), ("/tmp/pf_fetch", "FETCH_SIZE")):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name and "fu_conv_kernel<1, 4, 0>" in row["Kernel_Name"]:
                vals.append((float(row["Counter_Value"]) * 1024 / 1e9, row.get("Grid_Size", row.get("Grid_Size_X", ""))))
    vals.sort(reverse=True)
    print(name, "GB per launch, top 6:", [(round(v, 2), g) for v, g in vals[:6]])
PY
