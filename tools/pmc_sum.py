"""Per-kernel averages of every counter in rocprofv3 counter_collection csv files.  usage: pmc_sum.py <kernel substring> <csv> [<csv> ...]"""
import csv, sys
sub = sys.argv[1]
acc = {}
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"])
        a = acc.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for (kn, cn), (t, n) in sorted(acc.items()):
    print(f"{kn:50s} {cn:28s} {t / n:16.1f}  (n={n})")
