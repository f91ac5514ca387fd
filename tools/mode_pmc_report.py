"""per solver of tools/mode_pmc.py: mean duration and mean counter values of its relax27_plane launches, from the
counter_collection.csv of a rocprofv3 --pmc pass.   python tools/mode_pmc_report.py DIR [launches per solver = 20]"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict

src = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
disp = OrderedDict()
for r in csv.DictReader(open(f)):
    if "relax27_plane" not in r["Kernel_Name"]:
        continue
    d = disp.setdefault(int(r["Dispatch_Id"]), {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6, "c": {}})
    d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
ids = list(disp)
for g in range(0, len(ids), per):
    grp = [disp[i] for i in ids[g:g + per]]
    tot = defaultdict(float)
    for d in grp:
        for k, v in d["c"].items():
            tot[k] += v
    print("solver %d: %2d launches, mean %.3f ms  " % (g // per, len(grp), sum(d["dur"] for d in grp) / len(grp))
          + "  ".join("%s %.4e" % (k, tot[k] / len(grp)) for k in sorted(tot)))
