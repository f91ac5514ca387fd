"""Workload for counter passes over the two modes of the 512^3 relax sweep: NALLOC solvers built one after the other in one
process (each on a fresh operator; the previous one released first), 10 plane-fused sweeps each (20 relax27_plane launches
per solver).  Prints the sweep time of each solver (HIP events).  Run under
    rocprofv3 --pmc <counters> -d DIR -o out --output-format csv -- python3 tools/mode_pmc.py
and summarise with tools/mode_pmc_report.py DIR."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi

NALLOC = int(sys.argv[1]) if len(sys.argv) > 1 else 6
capi.lib.cedar_amd_solver_time_relax.restype = C.c_float
keep = []
for a in range(NALLOC):
    so, b = capi.gallery("fe3", (512, 512, 512))
    s = capi.Solver(so, share_operator=True)
    x = capi.DeviceArray(b.shape)
    x.zero()
    ms = capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 10) / 10
    print("solver %d: %.3f ms per sweep" % (a, ms), flush=True)
    if a % 2 == 0:  # every other solver stays alive a little longer so that the next one lands elsewhere
        keep.append((s, so, b, x))
    else:
        s.close(); so.free(); b.free(); x.free()
        for t in keep:
            t[0].close(); t[1].free(); t[2].free(); t[3].free()
        keep = []
