"""Effective shader clock per kernel from one rocprofv3 PMC pass: `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -- <cmd>`.
effective clock = GRBM_GUI_ACTIVE / kernel wall time (MI355X_MICROARCH.md, "DVFS give-back"): the part clocks to its power budget, so an MFMA-dense kernel on random data runs well
under the 2.4 GHz the 2.5 PFLOP/s peak is quoted at.  The counter is summed over the XCDs by the tool (8 on the MI355X): divided by 8 here.
usage: python tools/pmc_clock.py <counter_collection.csv> [out.json] [kernel substring ...]"""
import csv, json, sys

def main():
    path = sys.argv[1]
    outp = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    subs = sys.argv[3:] or ["gemm_w4", "attn_s", "ln_", "sgd_step"]
    agg = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
            continue
        k = r["Kernel_Name"].split("(")[0][:64]
        if not any(s in k for s in subs):
            continue
        try:
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        except (KeyError, ValueError):
            continue
        a = agg.setdefault(k, [0.0, 0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
    rows = []
    for k, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        raw = cyc / ns if ns > 0 else 0.0                    # GHz if the counter were one clock domain
        per8 = raw / 8.0
        ghz = per8 if raw > 4.0 else raw                     # (summed over 8 XCDs -> 8 x the shader clock; a sane clock is 0.5 .. 2.5 GHz)
        rows.append({"kernel": k, "launches": n, "avg_us": ns / n / 1e3, "effective_clock_GHz": round(ghz, 3), "counter_per_ns_raw": round(raw, 3)})
        print(f"{k:66s} launches {n:5d}  avg {ns / n / 1e3:9.2f} us  effective clock {ghz:5.3f} GHz")
    if outp:
        json.dump({"source": path, "note": "effective clock = GRBM_GUI_ACTIVE / kernel wall time (per XCD); the bf16 / fp16 MFMA peak of 2.5 PFLOP/s is quoted at 2.4 GHz", "kernels": rows},
                  open(outp, "w"), indent=1)

if __name__ == "__main__":
    main()
