"""which allocation decides the sweep time: (a) only the solver (its SOR, coarse levels) is re-created, (b) only the operator is moved"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)


def sweep(s):
    s.time_relax(x, b, 2)
    return round(s.time_relax(x, b, 6) / 6, 4)


keep = []
for rep in range(6):
    if rep % 2:
        keep.append(capi.DeviceArray((rep * 53, 1024, 1024)))  # push the solver's allocations somewhere else
    s = capi.Solver(so, share_operator=True)
    print(json.dumps({"case": "solver re-created", "rep": rep, "ms_per_sweep": sweep(s)}), flush=True)
    s.close()
for k in keep:
    k.free()
for rep in range(5):
    so2 = capi.DeviceArray(so.shape)
    so2.copy_from(so)
    s = capi.Solver(so2, share_operator=True)
    print(json.dumps({"case": "operator moved", "rep": rep, "so_ptr": hex(so2.ptr), "ms_per_sweep": sweep(s)}), flush=True)
    s.close()
    if rep % 2 == 0:
        so2.free()
    else:
        keep.append(so2)  # keep it so that the next copy lands elsewhere
