"""2D domain-decomposed solver with the HIP kernels: ranks share the one GPU of the test box, gloo
transport (RCCL refuses two ranks per device).  Same criterion as tests/test_dist2d_cpu.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_dist2d_cpu import build_global

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, outdir):
    for p in (HERE, ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import problems as pb
        from cedar_amd._torch_dist import GpuBackend
        from cedar_amd.dist import Topology
        from cedar_amd.dist2d import DistSolver2
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        kind, n, pgrid, relax, agg = case
        topo = Topology(rank, world, (pgrid[0], pgrid[1], 1))
        gn = tuple(n[d] * pgrid[d] for d in range(2))
        gso, gb = build_global(pb, kind, gn)
        ci, cj = topo.coord[:2]
        sl = (slice(cj * n[1], cj * n[1] + n[1] + 2), slice(ci * n[0], ci * n[0] + n[0] + 2))
        m = pb.interior_mask(tuple(s.stop - s.start for s in sl)).astype(np.float64)
        A = torch.from_numpy(np.ascontiguousarray(gso[(slice(None),) + sl]) * m).to(dev)
        b = torch.from_numpy(np.ascontiguousarray(gb[sl]) * m).to(dev)
        x = torch.zeros_like(b)
        s = DistSolver2(GpuBackend(dev), topo, A, relax=relax, max_iter=5, agglomerate_below=agg)
        h = s.solve(b, x)
        np.save(os.path.join(outdir, f"x{rank}.npy"), x.cpu().numpy())
        if rank == 0:
            np.save(os.path.join(outdir, "hist.npy"), np.array(h))
    finally:
        dist.destroy_process_group()


CASES = [
    ("rand9", (32, 24), (2, 2), "point", 4),
    ("poisson5", (32, 32), (1, 2), "point", 4),
    ("stretch5", (600, 40), (2, 1), "line-x", 8),   # lines longer than one wavefront tile, cut in two
    ("aniso9", (32, 32), (2, 2), "line-xy", 4),
    ("rand9", (128, 128), (2, 2), "line-xy", 64),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{'x'.join(map(str, c[1]))}-p{'x'.join(map(str, c[2]))}-{c[3]}")
def test_2d_ranks_on_one_gpu_equal_single_domain(case, tmp_path, oracle):
    import problems as pb
    kind, n, pgrid, relax, agg = case
    world = pgrid[0] * pgrid[1]
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    gn = tuple(n[d] * pgrid[d] for d in range(2))
    gso, gb = build_global(pb, kind, gn)
    ml = oracle.ml_create(gso, relax=relax)
    x = np.zeros_like(gb)
    want = ml.solve(gb, x, maxiter=5)
    ml.close()
    got = np.load(tmp_path / "hist.npy")
    assert len(got) == len(want)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 if relax != "point" else 1e-14)
    px, py = pgrid
    for r in range(world):
        ci, cj = r % px, r // px
        xr = np.load(tmp_path / f"x{r}.npy")
        ref = x[cj * n[1]:cj * n[1] + n[1] + 2, ci * n[0]:ci * n[0] + n[0] + 2]
        assert np.max(np.abs(xr[1:-1, 1:-1] - ref[1:-1, 1:-1])) <= 1e-11 * np.max(np.abs(x))
