"""does the relax sweep time depend on where the arrays were allocated?  Same process, the operator and the solver
created and destroyed several times (optionally with a dummy allocation of a varying size in front)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
for rep in range(8):
    pad = capi.DeviceArray((rep * 37 + 1, 1024, 1024)) if rep % 2 else None   # shifts the following allocations
    so, b = capi.gallery("fe3", (n, n, n))
    x = capi.DeviceArray(b.shape)
    s = capi.Solver(so, share_operator=True)
    s.time_relax(x, b, 3)
    ms = [s.time_relax(x, b, 6) / 6 for _ in range(3)]
    print(json.dumps({"rep": rep, "pad_GB": 0 if pad is None else (rep * 37 + 1) * 8 / 1024, "so_ptr": hex(so.ptr), "ms_per_sweep": [round(v, 4) for v in ms]}), flush=True)
    s.close(); so.free(); b.free(); x.free()
    if pad is not None:
        pad.free()
