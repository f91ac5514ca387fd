"""sum a rocprofv3 --pmc counter per kernel: python tools/pmc_sum.py <dir-or-csv> -> kernel, calls, total, per call"""
import csv, glob, os, sys
from collections import defaultdict
src = sys.argv[1]
files = [src] if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
tot, cnt = defaultdict(float), defaultdict(int)
for f in files:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:70], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(tot, key=lambda k: -tot[k])[:12]:
    print(f"{k[0]:72s} {k[1]:12s} calls {cnt[k]:5d} total {tot[k]:.4e} per call {tot[k]/cnt[k]:.4e}")
