"""One V-cycle out of a rocprofv3 kernel trace (csv): the kernels between the last two level-0 residual launches, grouped by
name -- launches, busy time, and the idle time between kernels.   usage: cycle_breakdown.py kernel_trace.csv [marker]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "residual27_rows<256"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
top = max(dur(rows[i]) for i in idx)
idx = [i for i in idx if dur(rows[i]) > top // 2]  # the finest level's launches of the marker kernel
a, b = idx[-2], idx[-1]
cyc = rows[a:b]
t0, t1 = int(cyc[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = defaultdict(lambda: [0, 0])
tot = 0
for r in cyc:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void cedar_amd::", "").replace("cedar_amd::", "")
    busy[name][0] += 1
    busy[name][1] += d
    tot += d
print("cycle: %.3f ms wall, %.3f ms in kernels (%d launches), %.3f ms between kernels" % ((t1 - t0) / 1e6, tot / 1e6, len(cyc), (t1 - t0 - tot) / 1e6))
for name, (n, d) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-60s %5d %9.3f ms" % (name[:60], n, d / 1e6))
