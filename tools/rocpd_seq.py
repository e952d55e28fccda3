"""The dispatch sequence of the LAST step in a rocprofv3 rocpd .db: every launch between the last two `marker` kernels (default: the optimizer's sgd_step_kernel on the flat
buffer = one per step), in order, with its grid and duration; consecutive identical (kernel, grid) runs keep their own lines so the layer structure stays readable.
Usage: python tools/rocpd_seq.py <results.db> [marker-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
kv = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')") if r[0].startswith('kernels')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kv})")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
rows = list(cur.execute(f"select name,start,end,{gx}" + (f",{wx}" if wx else ",0") + f" from {kv} order by start"))
marker = sys.argv[2] if len(sys.argv) > 2 else "sgd_step_kernel"
idx = [i for i, r in enumerate(rows) if marker in r[0]]
big = [i for i in idx if (rows[i][2] - rows[i][1]) > 0.5 * max(rows[j][2] - rows[j][1] for j in idx)]      # the flat-buffer launch, not the small-tensor ones
a, b = big[-2] + 1, big[-1] + 1
seq = rows[a:b]
print(f"# {len(seq)} launches, kernel time {sum(r[2]-r[1] for r in seq)/1e6:.3f} ms, span {(seq[-1][2]-seq[0][1])/1e6:.3f} ms")
for n, s, e, g, w in seq:
    print(f"{(e-s)/1e3:9.2f} us  grid {g:>9} wg {w:>5}  {n.split('(')[0][:90]}")
