"""HBM traffic per launch of the dominant kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, each in its own run, csv output).
FETCH_SIZE / WRITE_SIZE are in KB (TCC_EA0 requests x 64 B / 1024); on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section): doubled here.  WRITE_SIZE is reported as counted (uncalibrated).
usage: python tools/pmc_traffic.py <fetch_csv> <write_csv> <kernel substring> [out.json] [gemm calls in the profiled run]
A GEMM call of bench.py's roofline ("launch": one vdk_gemm_bf16_nt call, one HIP event pair) may be TWO kernels (whole tile rounds on the persistent kernel, the remaining rows on the
256x128 kernel): with the call count given, bytes are per CALL, like `achieved`."""
import csv, json, sys
def per_kernel(path, sub):
    tot, n = {}, {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if sub not in k: continue
        key = k.split("(")[0][:60]
        tot[key] = tot.get(key, 0.0) + float(r["Counter_Value"]); n[key] = n.get(key, 0) + 1
    return tot, n
f, fn = per_kernel(sys.argv[1], sys.argv[3]); w, wn = per_kernel(sys.argv[2], sys.argv[3])
TF = sum(f.values()); TW = sum(w.values()); NK = sum(fn.values())
N = int(sys.argv[5]) if len(sys.argv) > 5 else NK
for k in sorted(f, key=lambda k: -f[k]):
    print(f"{k:62s} launches {fn[k]:5d}  fetch {2 * f[k] / fn[k] / 1024:9.2f} MB/launch (x2 corrected)  write {w.get(k, 0) / max(wn.get(k, 1), 1) / 1024:9.2f} MB/launch")
print(f"ALL '{sys.argv[3]}': kernel launches {NK}, GEMM calls {N}, fetch(x2) {2 * TF / N / 1024:.2f} MB/launch, write {TW / N / 1024:.2f} MB/launch, total {(2 * TF + TW) / N / 1024:.2f} MB/launch")
if len(sys.argv) > 4:
    json.dump({"kernel_filter": sys.argv[3], "launches": N, "kernel_launches": NK, "fetch_bytes_per_launch_x2": 2 * TF / N * 1024, "write_bytes_per_launch": TW / N * 1024,
               "bytes_per_launch": (2 * TF + TW) / N * 1024,
               "note": "L2 memory-side (fabric) requests: Infinity-Cache hits are counted; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE as counted"},
              open(sys.argv[4], "w"), indent=1)
