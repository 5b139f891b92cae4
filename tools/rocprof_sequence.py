"""Per-launch sequence of the last complete 2-D forward in a rocprofv3 --kernel-trace (rocpd sqlite) result:
    python tools/rocprof_sequence.py <results.db> [first-kernel substring = stem7]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    first = sys.argv[2] if len(sys.argv) > 2 else "stem7"
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    gy = gx.replace("x", "y") if gx else None
    q = f"select name, start, end{', ' + gx + ', ' + gy if gx else ''} from kernels order by start"
    rows = list(con.execute(q))
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if len(starts) < 2:
        print("need two forwards in the trace"); return
    a, b = starts[-2], starts[-1]
    tot = 0.0
    for r in rows[a:b]:
        d = (r[2] - r[1]) / 1e3
        tot += d
        g = f"{r[3]}x{r[4]}" if gx else ""
        print(f"{d:9.2f} us  {g:>14s}  {r[0][:110]}")
    print(f"sum {tot:.1f} us over {b - a} launches; wall {(rows[b][1] - rows[a][1]) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
