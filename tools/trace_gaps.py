"""Per-launch view of one replayed reverse step from a rocprofv3 --kernel-trace (rocpd sqlite) result: every kernel of the LAST
complete step in start order with its duration and the idle time before it, plus the step's totals.
    python tools/trace_gaps.py <results.db> <first kernel substring> [last kernel substring]
The step is delimited by two consecutive launches of the kernel whose name contains the first substring."""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = list(con.execute("select name, start, end from kernels order by start"))
    key = sys.argv[2]
    marks = [i for i, r in enumerate(rows) if key in r[0]]
    if len(marks) < 3:
        raise SystemExit(f"fewer than three launches of a kernel matching {key!r}")
    a, b = marks[-3], marks[-2]                      # the last step both of whose ends are inside the trace
    step = rows[a:b]
    busy = sum(r[2] - r[1] for r in step)
    wall = rows[b][1] - rows[a][1]
    print(f"# {len(step)} launches; kernels {busy / 1e3:.1f} us + idle {(wall - busy) / 1e3:.1f} us = {wall / 1e3:.1f} us start to start")
    print(f"{'kernel':64s} {'dur_us':>9s} {'gap_us':>8s}")
    prev_end = rows[a - 1][2] if a else step[0][1]
    for r in step:
        print(f"{r[0][:64]:64s} {(r[2] - r[1]) / 1e3:9.2f} {(r[1] - prev_end) / 1e3:8.2f}")
        prev_end = r[2]


if __name__ == "__main__":
    main()
