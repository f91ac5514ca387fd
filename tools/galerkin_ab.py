"""3D Galerkin product, level 0 of a 27-point (and 7-point) operator: the fused one-stage launch against the row-sum
product (CEDAR_AMD_GALERKIN_ROWS=1) for several slab sizes (CEDAR_AMD_GALERKIN_SLAB), through the drop-in entry
point on device arrays; checks that the coarse operators are the same bit for bit.
    python tools/galerkin_ab.py [n]        # default 512"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from cedar_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = capi.Kernels()
for kind, nst in (("fe3", 14), ("poisson3", 4)):
    so, _ = capi.gallery(kind, (n, n, n), with_rhs=False)
    nc = (n - 1) // 2 + 1
    gc = (nc + 2, nc + 2, nc + 2)
    ci = capi.DeviceArray((26,) + gc)
    ci.zero()
    K.setup_interp3(so, ci)
    ref = None
    for rows, slab in ((0, 0), (1, 2), (1, 4), (1, 8), (1, 16), (1, 32), (1, 1000)):
        os.environ["CEDAR_AMD_GALERKIN_ROWS"] = str(rows)
        os.environ["CEDAR_AMD_GALERKIN_SLAB"] = str(slab)
        soc = capi.DeviceArray((14,) + gc)
        soc.zero()
        K.galerkin3(so, soc, ci)  # warm-up (scratch allocation)
        capi.sync()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            K.galerkin3(so, soc, ci)
            capi.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        h = soc.numpy()
        same = "reference" if ref is None else ("same bits" if np.array_equal(h, ref) else "DIFFERENT max %.3e" % np.max(np.abs(h - ref)))
        if ref is None:
            ref = h
        print("%-8s n=%d  %-22s %8.2f ms (min of 3: %s)  %s" % (kind, n, "fused one-stage" if not rows else "row sums, slab %d" % slab,
                                                                   min(ts), " ".join("%.2f" % t for t in ts), same), flush=True)
        soc.free()
