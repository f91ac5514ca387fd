"""2D domain-decomposed solver (cedar_amd/dist2d.py) under gloo on CPU with the oracle as compute back
end: the N-rank residual history equals the single-domain history on the same global problem -- the
reference's own criterion for its MPI solvers (test/2d/mpi/test_relax.cc, test/3d/mpi/test_relax.cc:56-59).
Point relaxation (9- and 5-point) and line relaxation in x, y and both, with the lines cut by the rank
grid (distributed tridiagonal solves), 2 and 4 ranks, every level distributed or gathered early."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def build_global(pb, kind, gn):
    g = (gn[1] + 2, gn[0] + 2)
    if kind == "rand9":
        return pb.random_op(g, 5, 77), pb.uniform(g, 78, -1, 1) * pb.interior_mask(g)
    if kind == "poisson5":
        return pb.poisson2(*gn), pb.rhs2(*gn)
    if kind == "aniso9":
        return pb.aniso9(*gn), pb.rhs2(*gn)
    if kind == "stretch5":
        return pb.diag_diffusion2(gn[0], gn[1], 1.0, 1e-2), pb.rhs2(*gn)
    raise ValueError(kind)


def _worker(rank, world, port, case, outdir):
    for p in (HERE, ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import problems as pb
        from cedar_amd.dist import Topology
        from dist2d_torch import DistSolver2
        from dist_cpu_backend import CpuBackend
        kind, n, pgrid, relax, agg = case
        topo = Topology(rank, world, (pgrid[0], pgrid[1], 1))
        gn = tuple(n[d] * pgrid[d] for d in range(2))
        gso, gb = build_global(pb, kind, gn)
        ci, cj = topo.coord[:2]
        sl = (slice(cj * n[1], cj * n[1] + n[1] + 2), slice(ci * n[0], ci * n[0] + n[0] + 2))
        A = torch.from_numpy(np.ascontiguousarray(gso[(slice(None),) + sl]))
        m = torch.from_numpy(pb.interior_mask(A.shape[1:]).astype(np.float64))
        A *= m  # the solver must fill ghost layers itself
        b = torch.from_numpy(np.ascontiguousarray(gb[sl])) * m
        x = torch.zeros_like(b)
        s = DistSolver2(CpuBackend(), topo, A, relax=relax, max_iter=6, agglomerate_below=agg)
        h = s.solve(b, x)
        np.save(os.path.join(outdir, f"x{rank}.npy"), x.numpy())
        if rank == 0:
            np.save(os.path.join(outdir, "hist.npy"), np.array(h))
            np.save(os.path.join(outdir, "nlev.npy"), np.array([s.nlev_global, s.la]))
    finally:
        dist.destroy_process_group()


# (operator, local extents, rank grid, relaxation, agglomerate_below)
CASES = [
    ("rand9", (16, 12), (1, 2), "point", 2),
    ("rand9", (16, 16), (2, 1), "point", 2),
    ("rand9", (16, 8), (2, 2), "point", 2),
    ("poisson5", (16, 16), (2, 2), "point", 4),
    ("rand9", (32, 32), (2, 2), "point", 16),
    ("stretch5", (32, 16), (2, 1), "line-x", 2),
    ("stretch5", (32, 16), (1, 2), "line-x", 2),
    ("aniso9", (16, 32), (1, 2), "line-y", 2),
    ("aniso9", (16, 16), (2, 2), "line-xy", 2),
    ("rand9", (32, 16), (2, 2), "line-xy", 8),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{'x'.join(map(str, c[1]))}-p{'x'.join(map(str, c[2]))}-{c[3]}-agg{c[4]}")
def test_distributed_2d_equals_single_domain(case, tmp_path, oracle):
    import problems as pb
    kind, n, pgrid, relax, agg = case
    world = pgrid[0] * pgrid[1]
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    gn = tuple(n[d] * pgrid[d] for d in range(2))
    gso, gb = build_global(pb, kind, gn)
    ml = oracle.ml_create(gso, relax=relax)
    x = np.zeros_like(gb)
    want = ml.solve(gb, x, maxiter=6)
    nlev = ml.nlevels()
    ml.close()
    got = np.load(tmp_path / "hist.npy")
    assert int(np.load(tmp_path / "nlev.npy")[0]) == nlev
    assert len(got) == len(want)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 if relax != "point" else 1e-14)
    px, py = pgrid
    for r in range(world):
        ci, cj = r % px, r // px
        xr = np.load(tmp_path / f"x{r}.npy")
        ref = x[cj * n[1]:cj * n[1] + n[1] + 2, ci * n[0]:ci * n[0] + n[0] + 2]
        assert np.max(np.abs(xr[1:-1, 1:-1] - ref[1:-1, 1:-1])) <= 1e-11 * np.max(np.abs(x))
