"""per kernel: the dispatch with the largest counter value (= the level-0 launch) from a rocprofv3 --pmc csv"""
import csv, glob, os, sys
from collections import defaultdict
src = sys.argv[1]
f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
best = defaultdict(float)
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"].split("(")[0][:64], r["Counter_Name"])
    best[k] = max(best[k], float(r["Counter_Value"]))
for k in sorted(best, key=lambda k: -best[k])[:40]:
    print(f"{k[0]:66s} {k[1]:11s} max_per_dispatch_KB {best[k]:.4e}")
