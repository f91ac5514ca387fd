"""The four-launch (row-class) order of the 27-point sweep -- what rank grids with an x / y split run -- on the Cedar layout
against the row-interleaved solve copy, and the plane-fused order beside them; several solver allocations each (the sweep
time has placement modes, so minima and medians are what to compare).   python tools/fourpass_ilv_ab.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi

capi.lib.cedar_amd_solver_time_relax.restype = C.c_float
res = {}
for rep in range(5):
    for frun, ilv in ((0, 0), (0, 320), (8, 0), (8, 320)):
        os.environ["CEDAR_AMD_FRUN"] = str(frun)
        os.environ["CEDAR_AMD_ILV"] = str(ilv)
        so, b = capi.gallery("fe3", (512, 512, 512))
        s = capi.Solver(so, share_operator=True)
        x = capi.DeviceArray(b.shape)
        x.zero()
        capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 2)
        ms = capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 8) / 8
        res.setdefault((frun, ilv), []).append(ms)
        s.close(); so.free(); b.free(); x.free()
for (frun, ilv), v in res.items():
    v = sorted(v)
    print("%-28s %-12s sweeps %s ms   min %.3f median %.3f" % ("four row-class launches" if frun == 0 else "plane-fused, runs of 8",
                                                                "solve copy" if ilv else "Cedar layout", " ".join("%.3f" % t for t in v), v[0], v[len(v) // 2]), flush=True)
