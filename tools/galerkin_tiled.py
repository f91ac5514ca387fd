"""LDS-tiled vs slot-kernel 3D Galerkin product: identical coarse operators? set-up time at n^3"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import problems as pb
from cedar_amd import capi
K = capi.Kernels()
# bit-identity on small random operators (27- and 7-point fine operators, odd / even extents)
for shape, nst in (((9, 8, 7), 14), ((12, 12, 12), 14), ((13, 9, 10), 4), ((6, 6, 6), 4), ((33, 20, 17), 14), ((20, 33, 40), 14), ((17, 12, 31), 4)):
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    gc = pb.coarse_shape(g)
    so = pb.random_op(g, nst, 5, zero_ghost=False)
    ci = pb.uniform((26,) + gc, 6, -1, 1)
    out = {}
    for ts, mb in (("0", "2048"), ("1", "2048"), ("1", "1")):   # 1 MB: slabs of 4 coarse planes
        os.environ["CEDAR_AMD_GALERKIN_TILED"] = ts
        os.environ["CEDAR_AMD_GALERKIN_SCRATCH_MB"] = mb
        soc = np.zeros((14,) + gc)
        K.galerkin3(so, soc, ci)
        out[ts + mb] = soc
    print(shape, nst, "identical:", np.array_equal(out["02048"], out["12048"]), np.array_equal(out["02048"], out["11"]),
          "max diff", float(np.max(np.abs(out["02048"] - out["12048"]))), float(np.max(np.abs(out["02048"] - out["11"]))))
os.environ["CEDAR_AMD_GALERKIN_SCRATCH_MB"] = "2048"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
so, b = capi.gallery("fe3", (n, n, n))
for ts in ("0", "1", "0", "1"):
    os.environ["CEDAR_AMD_GALERKIN_TILED"] = ts
    capi.sync(); t0 = time.time()
    s = capi.Solver(so, share_operator=True)
    capi.sync(); t1 = time.time()
    A3 = s.array(3, "A")
    chk = float(np.sum(A3 * (np.arange(A3.size).reshape(A3.shape) % 7)))
    print(json.dumps({"n": n, "tiled": ts, "setup_ms": (t1 - t0) * 1e3, "checksum": chk}), flush=True)
    s.close()
