"""L2 memory-side traffic and MFMA-busy cycles of ONE CBIR search (10 k x 1 M x 128, k = 100) from rocprofv3 PMC passes:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d <out>/fetch -o p --output-format csv -- python tools/cbir_bench.py --searches 3
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  -d <out>/write -o p --output-format csv -- python tools/cbir_bench.py --searches 3
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d <out>/sq -o p --output-format csv -- python tools/cbir_bench.py --searches 3
    python tools/pmc_cbir.py <fetch_csv> <write_csv> <sq_csv|-> <searches incl. warm-up> profiles/r02_cbir_pmc.json

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md §HBM); both sit on the L2's memory side, so
Infinity-Cache hits are included: an upper bound of the HBM bytes."""
import csv, json, sys


def table(path, counter=None):
    tot = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("cbir") and "cbir" not in k:
            continue
        if counter and r.get("Counter_Name") != counter:
            continue
        a = tot.setdefault(k, [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return tot


fetch, write = table(sys.argv[1]), table(sys.argv[2])
nsearch = int(sys.argv[4])
per_kernel = {}
for k in sorted(set(fetch) | set(write)):
    f = 2 * fetch.get(k, [0, 0])[0] * 1024 / nsearch; w = write.get(k, [0, 0])[0] * 1024 / nsearch
    per_kernel[k] = {"launches_per_search": fetch.get(k, write.get(k))[1] / nsearch, "fetch_bytes_x2": f, "write_bytes": w}
    print(f"{k:48s} {per_kernel[k]['launches_per_search']:6.1f} launches/search  fetch {f / 1e6:9.1f} MB  write {w / 1e6:8.1f} MB")
out = {"bytes_per_search": sum(v["fetch_bytes_x2"] + v["write_bytes"] for v in per_kernel.values()), "per_kernel": per_kernel, "searches_profiled": nsearch,
       "note": "L2 memory-side (fabric) requests incl. Infinity-Cache hits; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE as counted"}
if sys.argv[3] != "-":
    busy = table(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES"); act = table(sys.argv[3], "GRBM_GUI_ACTIVE")
    for k in busy:
        if k in act and act[k][0] > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (4 per CU x 256 CUs); GRBM_GUI_ACTIVE is reported per XCD and rocprofv3 SUMS the 8 XCDs, so the
            # kernel's duration in clocks is act / 8 and the per-SIMD busy fraction is busy / (1024 * act / 8) = busy / (128 * act)
            out.setdefault("mfma_busy", {})[k] = {"mfma_busy_cycles": busy[k][0], "gui_active_cycles_summed_over_8_xcds": act[k][0],
                                                  "busy_per_simd_frac": busy[k][0] / (act[k][0] * 128)}
print("total per search: %.1f MB" % (out["bytes_per_search"] / 1e6))
json.dump(out, open(sys.argv[5], "w"), indent=1)
