"""Per-kernel averages of a rocprofv3 --pmc run: python tools/pmc_report.py <dir> [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        k = r["Kernel_Name"]
        if flt in k:
            dur[k][0] += 1
            dur[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k in sorted(acc):
    n, t = dur.get(k, [0, 0])
    print(k[:110], f"| launches {n} avg {t / max(n, 1) / 1e3:.1f} us")
    for c, (m, s) in sorted(acc[k].items()):
        print(f"    {c:28s} {s / m:16.1f}")
