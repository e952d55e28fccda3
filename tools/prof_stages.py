"""Per-stage kernel durations of the last CBIR search in a rocprofv3 rocpd .db (prefilter / rescore / select)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kv = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')") if r[0].startswith('kernels')][0]
rows = list(cur.execute(f"select name,start,end,grid_x from {kv} order by start"))
# the prefilter search is timed 1+3 times before the exact_scan ones: take the last complete prefilter search
names = ["prefilter", "rescore", "select_kernel", "rank_wave", "boot_thr"]
last_boot = max(i for i, r in enumerate(rows) if 'cast_rows' in r[0])
seq = rows[last_boot:]
end = next((i for i, r in enumerate(seq) if 'score_filter' in r[0]), len(seq))
seq = seq[:end]
for n in names:
    d = [round((r[2] - r[1]) / 1e3) for r in seq if n in r[0]]
    print(n, len(d), sum(d), d)
print("span_us", (seq[-1][2] - seq[0][1]) / 1e3)
