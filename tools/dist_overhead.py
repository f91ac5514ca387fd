"""What one rank of the domain-decomposed 3D solver costs beside its messages: DistSolver3 on ONE GPU with a
transport that talks to itself (every message of a (px,py,pz) rank grid is packed, copied device-to-device in place of
the RCCL send/recv, and unpacked -- the numbers in the ghosts are meaningless, the work per V-cycle is that of one
rank of the grid) against the resident C solver on the same 27-point problem.  Prints host enqueue time and time per
V-cycle with the GPU drained.   usage: dist_overhead.py n [pxXpyXpz [overlap_min]]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
from cedar_amd.dist import DistSolver3, GpuBackend, Topology

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pg = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (1, 1, 1)
overlap_min = int(sys.argv[3]) if len(sys.argv) > 3 else 96  # levels with fewer points per direction exchange in order
world = pg[0] * pg[1] * pg[2]


class SelfComm:
    """same interface as cedar_amd.comm.NativeComm; peers are this rank"""
    name = "self-talking mock"

    def __init__(self, world):
        self.rank, self.world = 0, world

    def p2p(self, sends, recvs):
        for (_, sa, so, sc), (_, ra, ro, rc) in zip(sends, recvs):
            capi.lib.cedar_amd_memcpy_d2d(ra.ptr + 8 * ro, sa.ptr + 8 * so, 8 * min(sc, rc))

    def allgather(self, send, count, recv):
        for r in range(self.world):
            capi.lib.cedar_amd_memcpy_d2d(recv.ptr + 8 * r * count, send.ptr, 8 * count)

    def allreduce_sum(self, v):
        return v * self.world


capi.set_device(0)
be = GpuBackend(SelfComm(world), 0)
g = (n + 2, n + 2, n + 2)
A, b = be.zeros((14,) + g), be.zeros(g)
pp = (C.c_double * 6)(0.0, 0.0, 0.0, float(n), float(n), float(n))
capi.lib.cedar_amd_gallery(112, A.ptr, b.ptr, n, n, n, pp)
centre = tuple(min(1, pg[d] - 1) for d in range(3))
topo = Topology(centre[2] * pg[0] * pg[1] + centre[1] * pg[0] + centre[0], world, pg)
ds = DistSolver3(be, topo, A, overlap_min=overlap_min)
x = be.zeros(g)


def run(f, k):
    for _ in range(2):
        f()
    capi.lib.cedar_amd_device_sync()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    t1 = time.perf_counter()
    capi.lib.cedar_amd_device_sync()
    t2 = time.perf_counter()
    return (t1 - t0) / k * 1e3, (t2 - t0) / k * 1e3


only = os.environ.get("DIST_OVERHEAD_ONLY", "")  # "native": just the native driver (for a profiler run)
if only == "native":
    ds.vcycle(x, b)
host, total = (0.0, 0.0) if only == "native" else run(lambda: ds.vcycle(x, b), 5)
print(json.dumps({"n": n, "solver": "DistSolver3, one rank of %dx%dx%d%s" % (pg + (" (self-talking mock)" if world > 1 else "",)),
                  "levels_distributed": len(ds.levels), "overlap_min": overlap_min, "host_ms_per_vcycle": host, "ms_per_vcycle": total}), flush=True)
# the same rank of the same grid on the native driver (cedar_amd_dist3_*, orchestration below the C ABI)
from cedar_amd.dist3 import DistSolver3 as Native
A2 = be.zeros((14,) + g)
capi.lib.cedar_amd_gallery(112, A2.ptr, b.ptr, n, n, n, pp)
dn = Native("loopback", topo.rank, world, A2, pgrid=pg, overlap_min=overlap_min)
xn = be.zeros(g)
host, total = run(lambda: dn.vcycle(xn, b), 5)
print(json.dumps({"n": n, "solver": "cedar_amd_dist3 (native driver), one rank of %dx%dx%d%s" % (pg + (" (loop-back transport)" if world > 1 else "",)),
                  "overlap_min": overlap_min, "host_ms_per_vcycle": host, "ms_per_vcycle": total}), flush=True)
dn.close()
if only == "native":
    sys.exit(0)
s = capi.Solver(A, share_operator=True)
xs = be.zeros(g)
host, total = run(lambda: capi.lib.cedar_amd_solver_vcycle(s.h, xs.ptr, b.ptr), 5)
print(json.dumps({"n": n, "solver": "resident C solver (hipGraph)", "host_ms_per_vcycle": host, "ms_per_vcycle": total}), flush=True)
