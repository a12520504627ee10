"""per-kernel means of the PMC counters in a rocprofv3 --pmc counter_collection.csv:  python tools/diag/sq_of.py <csv> <kernel substring>"""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print(f"{k:32s} {s / n:16.1f}   ({n} dispatches)")
