"""per-V-cycle kernel time of the DistSolver3 part of a tools/dist_overhead.py trace (rocprofv3 --kernel-trace csv):
everything between the first relax27_rows launch and the last launch before the resident solver's first relax27_plane"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
first_plane = next((i for i, r in enumerate(rows) if 'relax27_plane' in r['Kernel_Name']), len(rows))
first_rows = next(i for i, r in enumerate(rows) if 'relax27_rows' in r['Kernel_Name'])
last = max(i for i, r in enumerate(rows[:first_plane]) if 'relax27_rows' in r['Kernel_Name'] or 'box_copy' in r['Kernel_Name'])
seg = rows[first_rows:last + 1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = r['Kernel_Name'].split('(')[0].replace('void cedar_amd::', '').replace('cedar_amd::', '')
    agg[n][0] += 1
    agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
print("%d cycles: span %.2f ms/cycle, busy %.2f ms/cycle" % (ncyc, span / ncyc, sum(v[1] for v in agg.values()) / ncyc))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"{n[:64]:64s} {v[0] / ncyc:7.1f} calls/cycle {v[1] / ncyc:7.3f} ms/cycle")
