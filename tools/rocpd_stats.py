"""Summarise a rocprofv3 (ROCm 7.x rocpd sqlite) kernel trace: per-kernel calls / total / avg / share, like `--stats`.
Usage: python tools/rocpd_stats.py <results.db> [skip_first_n_dispatches_fraction]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows:
        short = name.split("(")[0]
        a = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':60s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'share':>7s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:60]:60s} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:9.2f} {a[3]:9.2f} {100*a[1]/tot:6.2f}%")
    print(f"{'TOTAL':60s} {sum(a[0] for a in agg.values()):7d} {tot:12.1f}")


if __name__ == "__main__":
    main()
