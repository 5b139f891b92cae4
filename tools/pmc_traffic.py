"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv), as
MI355X_MICROARCH.md's HBM section prescribes: the counters are in KiB per dispatch; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is (uncalibrated per the guide).
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [steps] > profiles/r02_pmc_traffic.json"""
import csv
import glob
import json
import sys
from collections import defaultdict


def collect(d, name):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != name:
                continue
            k = row["Kernel_Name"]
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        n, tot = fetch[k]
        wn, wtot = write.get(k, [0, 0.0])
        rd = 2.0 * tot * 1024 / n                      # gfx950 correction: x2
        wr = wtot * 1024 / wn if wn else 0.0
        out[k] = {"launches": n, "fetch_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "hbm_bytes_per_launch": round(rd + wr)}
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # reverse steps the traced command ran (incl. warm-up)
    per_step = round(sum(v["launches"] * v["hbm_bytes_per_launch"] for v in out.values()) / steps) if steps else None
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cindm_amd import build as _b
    # the hash of the sources the measured library was built from: bench.py drops these numbers when it loads another one
    json.dump({"bytes_per_step": per_step, "steps_traced": steps, "source_hash": _b.embedded_hash() or _b.source_hash(), "note": "FETCH_SIZE (KiB) x2 (gfx950: reports half of wide coalesced reads) + WRITE_SIZE (KiB), mean per launch; "
                       "Infinity-Cache hits are included in these memory-side counters", "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
