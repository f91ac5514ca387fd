"""3D Galerkin product by row sums: stage 1 with the operator read as aligned pairs (CEDAR_AMD_GALERKIN_PAIRS=1,
the default) against 8-byte loads (=0); checks the coarse operators bit for bit against the one-stage kernels.
    python tools/galerkin_pairs_ab.py [n]        # default 512"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from cedar_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = capi.Kernels()
for kind in ("fe3", "poisson3"):
    so, _ = capi.gallery(kind, (n, n, n), with_rhs=False)
    nc = (n - 1) // 2 + 1
    gc = (nc + 2, nc + 2, nc + 2)
    ci = capi.DeviceArray((26,) + gc)
    ci.zero()
    K.setup_interp3(so, ci)
    ref = None
    for rows, pairs in ((0, 0), (1, 0), (1, 1), (1, 0), (1, 1)):
        os.environ["CEDAR_AMD_GALERKIN_ROWS"] = str(rows)
        os.environ["CEDAR_AMD_GALERKIN_PAIRS"] = str(pairs)
        soc = capi.DeviceArray((14,) + gc)
        soc.zero()
        K.galerkin3(so, soc, ci)
        capi.sync()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            K.galerkin3(so, soc, ci)
            capi.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        h = soc.numpy()
        same = "reference" if ref is None else ("same bits" if np.array_equal(h, ref) else "DIFFERENT max %.3e" % np.max(np.abs(h - ref)))
        if ref is None:
            ref = h
        print("%-8s n=%d  %-28s %8.2f ms (%s)  %s" % (kind, n, "one-stage" if not rows else "row sums, pairs=%d" % pairs,
                                                        min(ts), " ".join("%.2f" % t for t in ts), same), flush=True)
        soc.free()
