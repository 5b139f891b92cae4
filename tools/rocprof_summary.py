"""Summarise a rocprofv3 --kernel-trace (rocpd sqlite) result as a per-kernel stats table
(the same content `--stats` prints): python tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = list(con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    lines = [f"{'kernel':72s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}"]
    for r in rows:
        lines.append(f"{r[0][:72]:72s} {r[1]:8d} {r[2] / 1e6:10.3f} {r[3] / 1e3:9.2f} {r[4] / 1e3:8.2f} {r[5] / 1e3:8.2f} {100 * r[2] / tot:6.2f}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
