"""Multi-rank path on CPU (gloo, world_size 2 / 4 / 8): the domain-decomposed solver of
cedar_amd/dist.py with the oracle as compute backend must reproduce the single-domain
oracle run on the same global problem -- the reference's own criterion for its MPI flavour
(test/3d/mpi/test_relax.cc:56-59: |norm_mpi - norm_ser| < 1e-10; here: whole residual history
to 1e-10 and the solution to 1e-12)."""
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
        from cedar_amd.dist import DistSolver3, Topology
        from dist_cpu_backend import CpuBackend
        kind, n, pgrid, agg = case[:4]
        chain = len(case) > 4 and case[4]
        topo = Topology(rank, world, pgrid)
        gn = tuple(n[d] * topo.p[d] for d in range(3))
        gso, gb = build_global(pb, kind, gn)
        ci, cj, ck = topo.coord
        sl = (slice(ck * n[2], ck * n[2] + n[2] + 2), slice(cj * n[1], cj * n[1] + n[1] + 2),
              slice(ci * n[0], ci * n[0] + n[0] + 2))
        A = torch.from_numpy(np.ascontiguousarray(gso[(slice(None),) + sl]))
        # the solver must fill ghost layers itself: wipe what the slicing brought along
        m = torch.from_numpy(pb.interior_mask(A.shape[1:]).astype(np.float64))
        A *= m
        b = torch.from_numpy(np.ascontiguousarray(gb[sl])) * m
        x = torch.zeros_like(b)
        # overlap_min=4: the interior-rows-first / deferred y-z halo ordering on every level that has an interior
        s = DistSolver3(CpuBackend(), topo, A, max_iter=6, agglomerate_below=agg, overlap_min=4, chain=chain)
        h = s.solve(b, x)
        np.save(os.path.join(outdir, f"x{rank}.npy"), x.numpy())
        if rank == 0:
            np.save(os.path.join(outdir, "hist.npy"), np.array(h))
            np.save(os.path.join(outdir, "nlev.npy"), np.array([s.nlev_global, s.la]))
    finally:
        dist.destroy_process_group()


def build_global(pb, kind, gn):
    g = (gn[2] + 2, gn[1] + 2, gn[0] + 2)
    if kind == "fe27":
        return pb.fe3(*gn), pb.rhs3(*gn)
    if kind == "rand27":
        return pb.random_op(g, 14, 77), pb.uniform(g, 78, -1, 1) * pb.interior_mask(g)
    if kind == "poisson7":
        return pb.poisson3(*gn), pb.rhs3(*gn)
    raise ValueError(kind)


# (operator, local extents, rank grid, agglomerate_below): 2 = every level distributed down to the
# coarsest direct solve; 32 = gather after the first coarsening (single-domain solver takes over)
CASES = [
    ("fe27", (8, 8, 8), (2, 1, 1), 2),
    ("rand27", (8, 6, 5), (2, 1, 1), 2),
    ("rand27", (16, 8, 8), (2, 1, 1), 4),
    ("rand27", (8, 8, 4), (2, 2, 1), 2),
    ("poisson7", (8, 8, 8), (2, 1, 1), 2),
    ("rand27", (4, 4, 4), (2, 2, 2), 2),
    ("rand27", (8, 8, 8), (2, 2, 2), 32),
    ("rand27", (6, 8, 8), (1, 2, 2), 2),
    ("rand27", (10, 12, 16), (1, 1, 2), 4),
    ("rand27", (9, 7, 8), (1, 1, 4), 2),
    ("fe27", (8, 8, 8), (1, 1, 4), 4),
    # the boundary-first chain of rank grids with an x / y split (what the native driver runs where a level takes the
    # partial-sum sweep), stated on the CPU backend: the points next to a neighbouring rank ahead, stage by stage, the rest
    # of a k-parity afterwards -- neighbour on the low / high / both sides in x and y, with and without a z split
    ("rand27", (8, 8, 4), (2, 2, 1), 2, True),
    ("rand27", (16, 8, 6), (2, 1, 1), 2, True),
    ("rand27", (8, 8, 4), (3, 1, 1), 2, True),
    ("rand27", (8, 8, 4), (1, 3, 1), 2, True),
    ("rand27", (8, 16, 4), (1, 2, 2), 2, True),
    ("rand27", (8, 8, 8), (2, 2, 2), 32, True),
    ("fe27", (16, 16, 8), (2, 2, 1), 4, True),
    ("rand27", (8, 8, 4), (3, 2, 1), 2, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{'x'.join(map(str, c[1]))}-p{'x'.join(map(str, c[2]))}-agg{c[3]}" + ("-chain" if len(c) > 4 else ""))
def test_distributed_equals_single_domain(case, tmp_path, oracle):
    import problems as pb
    kind, n, pgrid, agg = case[:4]
    world = pgrid[0] * pgrid[1] * pgrid[2]
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    gn = tuple(n[d] * pgrid[d] for d in range(3))
    gso, gb = build_global(pb, kind, gn)
    ml = oracle.ml_create(gso)
    x = np.zeros_like(gb)
    want = ml.solve(gb, x, maxiter=6)
    nlev = ml.nlevels()
    ml.close()
    got = np.load(tmp_path / "hist.npy")
    assert int(np.load(tmp_path / "nlev.npy")[0]) == nlev
    assert len(got) == len(want)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-14)
    px, py, pz = pgrid
    for r in range(world):
        ci, cj, ck = r % px, (r // px) % py, r // (px * py)
        xr = np.load(tmp_path / f"x{r}.npy")
        ref = x[ck * n[2]:ck * n[2] + n[2] + 2, cj * n[1]:cj * n[1] + n[1] + 2, ci * n[0]:ci * n[0] + n[0] + 2]
        own = (slice(1, -1),) * 3
        assert np.max(np.abs(xr[own] - ref[own])) <= 1e-12 * np.max(np.abs(x))


def test_rank_grid_and_neighbours():
    from cedar_amd.dist import Topology, rank_grid
    assert rank_grid(1) == (1, 1, 1) and rank_grid(2) == (1, 1, 2)
    assert rank_grid(4) == (1, 1, 4) and rank_grid(8) == (2, 2, 2) and rank_grid(12) == (2, 2, 3)
    t = Topology(5, 8)  # coord (1,0,1): rank = k*4 + j*2 + i (src/3d/util/topo.cc:82-84)
    assert t.coord == (1, 0, 1)
    nb = t.neighbours()
    assert len(nb) == 7 and nb[(-1, 0, 0)] == 4 and nb[(0, 1, 0)] == 7 and nb[(-1, 1, -1)] == 2
