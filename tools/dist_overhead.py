"""host-side cost of the Python orchestration of the domain-decomposed solver: DistSolver3 on ONE rank (no
neighbours, so no messages) against the resident C solver on the same 27-point problem -- enqueue time per
V-cycle (host only) and time per V-cycle with the GPU drained"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C
import torch
from cedar_amd import capi
from cedar_amd.dist import DistSolver3, GpuBackend, Topology

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pg = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (1, 1, 1)
dev = torch.device("cuda", 0)
capi.set_device(0)
g = (n + 2, n + 2, n + 2)
A = torch.zeros((14,) + g, dtype=torch.float64, device=dev)
b = torch.zeros(g, dtype=torch.float64, device=dev)
pp = (C.c_double * 6)(0.0, 0.0, 0.0, float(n), float(n), float(n))
capi.lib.cedar_amd_gallery(112, A.data_ptr(), b.data_ptr(), n, n, n, pp)
world = pg[0] * pg[1] * pg[2]
if world > 1:
    # a rank that talks to itself: the messages of a (px,py,pz) rank grid are packed, copied device-to-device in
    # place of the RCCL send/recv, and unpacked.  The numbers in the ghosts are meaningless, the work per V-cycle
    # (launches, packs, host orchestration) is that of one rank of the grid.
    import torch.distributed as dist
    from cedar_amd import dist as cd

    def _p2p(self, sends, recvs):
        for (_, s_), (_, r_) in zip(sends, recvs):
            r_.copy_(s_)
    cd.Halo._p2p = _p2p
    dist.is_initialized = lambda: True
    dist.get_backend = lambda *a: "nccl"
    dist.all_reduce = lambda t, *a, **k: None

    def _ag(parts, src, *a, **k):
        for p_ in parts:
            p_.copy_(src)
    dist.all_gather = _ag
    centre = tuple(min(1, pg[d] - 1) for d in range(3))
    rank = centre[2] * pg[0] * pg[1] + centre[1] * pg[0] + centre[0]
    topo = Topology(rank, world, pg)
else:
    topo = Topology(0, 1, (1, 1, 1))
ds = DistSolver3(GpuBackend(dev), topo, A)
x = torch.zeros_like(b)


def run(f, k):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / k * 1e3, (t2 - t0) / k * 1e3


host, total = run(lambda: ds.vcycle(x, b), 5)
print(json.dumps({"n": n, "solver": "DistSolver3, one rank of %dx%dx%d%s" % (pg + (" (self-talking mock)" if world > 1 else "",)), "levels_distributed": len(ds.levels), "host_ms_per_vcycle": host,
                  "ms_per_vcycle": total}), flush=True)
s = capi.Solver(A, share_operator=True)
xs = torch.zeros_like(b)
host, total = run(lambda: capi.lib.cedar_amd_solver_vcycle(s.h, xs.data_ptr(), b.data_ptr()), 5)
print(json.dumps({"n": n, "solver": "resident C solver (hipGraph)", "host_ms_per_vcycle": host, "ms_per_vcycle": total}), flush=True)
