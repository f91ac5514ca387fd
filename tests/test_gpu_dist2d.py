"""2D domain-decomposed solver with the HIP kernels, orchestrated below the C ABI (cedar_amd_dist2_*,
cedar_amd/csrc/dist2.cpp; distributed line relaxation: dist_lines.hip): 2 and 4 ranks share the one GPU of the test box
over the host-staged rehearsal transport handed in as the ABI's transport table (RCCL refuses two ranks per device), plus
the RCCL communicator itself with the one rank a one-GPU box allows.  NO torch in any rank process.  Criterion: the
decomposed run reproduces the single-domain history and solution (the reference's test/2d/mpi/test_poisson.cc /
test_lines.cc:45-126 compare parallel and serial results the same way), against the oracle on the global problem."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build_global(pb, kind, gn):
    """(operator, right-hand side) on the global grid gn = (nx, ny): the generators of tests/test_dist2d_cpu.py (which
    cannot be imported here: it brings torch)"""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, outdir, native_comm=False):
    for p in (HERE, ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    import problems as pb
    from cedar_amd import capi
    from cedar_amd.comm import NativeComm, SocketComm
    from cedar_amd.dist3 import DistSolver2
    capi.set_device(0)
    comm = NativeComm(rank, world) if native_comm else SocketComm(rank, world)
    try:
        kind, n, pgrid, relax, agg = case
        gn = tuple(n[d] * pgrid[d] for d in range(2))
        gso, gb = build_global(pb, kind, gn)
        ci, cj = rank % pgrid[0], rank // pgrid[0]
        sl = (slice(cj * n[1], cj * n[1] + n[1] + 2), slice(ci * n[0], ci * n[0] + n[0] + 2))
        m = pb.interior_mask(tuple(s.stop - s.start for s in sl)).astype(np.float64)
        A = capi.DeviceArray.from_numpy(np.ascontiguousarray(gso[(slice(None),) + sl]) * m)  # the solver fills the ghost layers
        b = capi.DeviceArray.from_numpy(np.ascontiguousarray(gb[sl]) * m)
        x = capi.DeviceArray(b.shape)
        s = DistSolver2(comm, rank, world, A, pgrid=pgrid, relax=relax, max_iter=5, agglomerate_below=agg)
        assert s.coord == (ci, cj)
        h = s.solve(b, x)
        s.close()
        assert "torch" not in sys.modules  # the point of cedar_amd_dist2_*: no torch in a 2D rank process
        np.save(os.path.join(outdir, f"x{rank}.npy"), x.numpy())
        if rank == 0:
            np.save(os.path.join(outdir, "hist.npy"), np.array(h))
    finally:
        comm.close()


def _spawn(target, world, args):
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=target, args=(r, world) + args) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(900)
    bad = [p.exitcode for p in ps if p.exitcode != 0]
    for p in ps:
        if p.is_alive():
            p.kill()
    assert not bad, f"rank processes failed: exit codes {bad}"


CASES = [
    ("rand9", (32, 24), (2, 2), "point", 4),
    ("poisson5", (32, 32), (1, 2), "point", 4),
    ("stretch5", (600, 40), (2, 1), "line-x", 8),   # lines longer than one wavefront tile, cut in two
    ("stretch5", (40, 96), (1, 2), "line-y", 8),    # y lines cut in two
    ("aniso9", (32, 32), (2, 2), "line-xy", 4),
    ("rand9", (128, 128), (2, 2), "line-xy", 64),
    ("aniso9", (64, 32), (4, 1), "line-x", 8),      # x lines cut by four ranks: carries composed over three segments
]


def _check(case, tmp_path, oracle):
    import problems as pb
    kind, n, pgrid, relax, agg = case
    world = pgrid[0] * pgrid[1]
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


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{'x'.join(map(str, c[1]))}-p{'x'.join(map(str, c[2]))}-{c[3]}")
def test_2d_ranks_on_one_gpu_equal_single_domain(case, tmp_path, oracle):
    kind, n, pgrid, relax, agg = case
    world = pgrid[0] * pgrid[1]
    _spawn(_worker, world, (_free_port(), case, str(tmp_path)))
    _check(case, tmp_path, oracle)


def test_2d_one_rank_over_rccl(tmp_path, oracle):
    """the same driver with the library's RCCL communicator (all-gather of the coarse level, norm all-reduce)"""
    case = ("aniso9", (48, 40), (1, 1), "line-xy", 8)
    _spawn(_worker, 1, (_free_port(), case, str(tmp_path), True))
    _check(case, tmp_path, oracle)
