"""Plane relaxation: V-cycle time of the device-resident 3D solver (relaxation plane-xy / plane-xyz, default plane
configuration: one line-xy V(2,1) cycle per plane) against the CPU oracle on the same problem, and the effect of the
number of side streams the plane solves of one colour are spread over (CEDAR_AMD_PLANE_STREAMS).
    python tools/planes_bench.py [n ...]          # default 64 128"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]


def one(n, relax, streams):
    os.environ["CEDAR_AMD_PLANE_STREAMS"] = str(streams)
    from cedar_amd import capi
    import problems as pb
    so = pb.diag_diffusion3(n, n, n, 1.0, 1e-2, 1e-4)
    b = pb.rhs3(n, n, n)
    t0 = time.perf_counter()
    s = capi.Solver(so, relax=relax)
    capi.sync()
    t_setup = time.perf_counter() - t0
    xd, bd = capi.DeviceArray.from_numpy(np.zeros_like(b)), capi.DeviceArray.from_numpy(b)
    s.vcycle(xd, bd)  # records the plane graphs
    capi.sync()
    ms = capi.lib.cedar_amd_solver_time_vcycles(s.h, capi._vp(xd), capi._vp(bd), 3) / 3
    x = np.zeros_like(b)
    h = s.solve(b, x)
    return t_setup, ms, h


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        n, relax, streams = int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
        from cedar_amd import capi
        capi.lib.cedar_amd_solver_time_vcycles.restype = __import__("ctypes").c_float
        ts, ms, h = one(n, relax, streams)
        print("    %-9s n=%3d streams=%2d  set-up %6.2f s  V-cycle %8.2f ms  cycles to 1e-8: %d  (last rel. residual %.1e)"
              % (relax, n, streams, ts, ms, len(h) - 1, h[-1]), flush=True)
        sys.exit(0)
    sizes = [int(a) for a in sys.argv[1:]] or [64, 128]
    for n in sizes:
        for relax in ("plane-xy", "plane-xyz"):
            for streams in (1, 8, 32):
                subprocess.run([sys.executable, __file__, "--one", str(n), relax, str(streams)], check=True)
        if n <= 64:
            from pyoracle import Oracle
            import problems as pb
            O = Oracle()
            so, b = pb.diag_diffusion3(n, n, n, 1.0, 1e-2, 1e-4), pb.rhs3(n, n, n)
            for relax in ("plane-xy", "plane-xyz"):
                ml = O.ml_create(so, relax=relax)
                x = np.zeros_like(b)
                t0 = time.perf_counter()
                ml.vcycle(x, b)
                dt = time.perf_counter() - t0
                ml.close()
                print("    %-9s n=%3d CPU oracle (1 core)                     V-cycle %8.2f ms" % (relax, n, dt * 1e3), flush=True)
