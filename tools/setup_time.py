"""Set-up time of the device-resident 512^3 27-point solver, several solvers in one process (fresh allocations:
the first ones pay for the mapping of new device memory, later ones reuse released blocks), with the row-sum
Galerkin product and with the fused one-stage launch.
    python tools/setup_time.py [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from cedar_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for rows in ("1", "0"):
    os.environ["CEDAR_AMD_GALERKIN_ROWS"] = rows
    keep = []
    for rep in range(5):
        so, b = capi.gallery("fe3", (n, n, n))
        capi.sync()
        t0 = time.perf_counter()
        s = capi.Solver(so, share_operator=True)
        capi.sync()
        t1 = time.perf_counter()
        A3 = s.array(3, "A")
        print("rows=%s solver %d (%s): set-up %.1f ms   level-3 operator checksum %.17g"
              % (rows, rep, "previous one still alive" if rep in (1,) else "previous one released", (t1 - t0) * 1e3,
                 float(np.sum(A3 * (np.arange(A3.size).reshape(A3.shape) % 7)))), flush=True)
        if rep == 0:
            keep = [s, so, b]  # solver 1 is built while solver 0 lives
        else:
            s.close(); so.free(); b.free()
            if rep == 1:
                keep[0].close(); keep[1].free(); keep[2].free(); keep = []
