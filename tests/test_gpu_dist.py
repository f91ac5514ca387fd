"""GPU multi-rank tests runnable on the one-GPU box.  No torch in any rank process.
* 2 / 4 ranks share device 0 and talk through the host-staged rehearsal transport (cedar_amd/comm.py SocketComm: RCCL
  refuses two ranks on one device); HIP kernels through the C ABI.  Same criterion as tests/test_dist_cpu.py: the
  decomposed run reproduces the single-domain history (the reference's test/3d/mpi/test_relax.cc:56-59).
* the RCCL transport itself (NativeComm -> cedar_amd_comm_* -> librccl) with the one rank a one-GPU box allows: a
  grouped self send/recv, all-reduce, all-gather, and a whole DistSolver3 solve on a 1x1x1 rank grid."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, pgrid, outdir, overlap_min, driver="python", agg=64, maxit=5):
    for p in (HERE, ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("CEDAR_AMD_TEST_STUCK_S", "600")), exit=True)  # a stuck rank says where
    import problems as pb
    from cedar_amd import capi
    from cedar_amd.comm import SocketComm
    from cedar_amd.dist import DistSolver3, GpuBackend, Topology
    assert "torch" not in sys.modules
    capi.set_device(0)
    comm = SocketComm(rank, world)
    try:
        be = GpuBackend(comm, 0)
        topo = Topology(rank, world, pgrid)
        gn = tuple(n[d] * topo.p[d] for d in range(3))
        g = (gn[2] + 2, gn[1] + 2, gn[0] + 2)
        gso = pb.random_op(g, 14, 77)
        gb = pb.uniform(g, 78, -1, 1) * pb.interior_mask(g)
        ci, cj, ck = topo.coord
        sl = (slice(ck * n[2], ck * n[2] + n[2] + 2), slice(cj * n[1], cj * n[1] + n[1] + 2),
              slice(ci * n[0], ci * n[0] + n[0] + 2))
        m = pb.interior_mask(tuple(s.stop - s.start for s in sl)).astype(np.float64)
        A = be.from_numpy(np.ascontiguousarray(gso[(slice(None),) + sl]) * m)
        b = be.from_numpy(np.ascontiguousarray(gb[sl]) * m)
        x = be.zeros(b.shape)
        if driver == "native":  # the orchestration below the C ABI (cedar_amd/csrc/dist3.cpp); Python hands over arrays + transport
            from cedar_amd.dist3 import DistSolver3 as Native
            s = Native(comm, rank, world, A, pgrid=pgrid, max_iter=maxit, overlap_min=overlap_min, agglomerate_below=agg)
            assert s.coord == topo.coord
            if rank == 0:
                open(os.path.join(outdir, "chain_levels.txt"), "w").write(str(s.chain_levels))
        else:
            s = DistSolver3(be, topo, A, max_iter=5, overlap_min=overlap_min)
        h = s.solve(b, x)
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


CASES = [((16, 12, 10), (2, 1, 1), 96), ((8, 8, 8), (2, 2, 1), 96),
         ((12, 10, 16), (1, 1, 2), 4), ((8, 8, 8), (1, 2, 2), 4),
         ((8, 8, 8), (2, 2, 1), 4), ((64, 64, 64), (1, 1, 2), 16), ((20, 16, 8), (1, 1, 4), 4),
         # 320 rows: the slab path runs the plane-fused kernel on level 0
         ((12, 320, 8), (1, 1, 2), 4),
         # 6.5e6 unknowns: plane-fused level 0 under a halo in flight, three
         # distributed levels, gathered 16x80x80 coarse problem
         ((64, 320, 160), (1, 1, 2), 32)]
IDS = ["2ranks-x", "4ranks-xy", "2ranks-z-overlap", "4ranks-yz-overlap", "4ranks-xy-overlap",
       "2ranks-z-64cubed-overlap", "4ranks-z-slabs-overlap", "2ranks-z-plane-fused", "2ranks-z-6M-unknowns"]


def _check_against_single_domain(n, pgrid, tmp_path, oracle, maxit=5):
    import problems as pb
    world = pgrid[0] * pgrid[1] * pgrid[2]
    gn = tuple(n[d] * pgrid[d] for d in range(3))
    g = (gn[2] + 2, gn[1] + 2, gn[0] + 2)
    gso = pb.random_op(g, 14, 77)
    gb = pb.uniform(g, 78, -1, 1) * pb.interior_mask(g)
    ml = oracle.ml_create(gso)
    x = np.zeros_like(gb)
    want = ml.solve(gb, x, maxiter=maxit)
    ml.close()
    got = np.load(tmp_path / "hist.npy")
    assert len(got) == len(want)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-14)
    px, py, pz = pgrid
    for r in range(world):
        ci, cj, ck = r % px, (r // px) % py, r // (px * py)
        xr = np.load(tmp_path / f"x{r}.npy")
        ref = x[ck * n[2]:ck * n[2] + n[2] + 2, cj * n[1]:cj * n[1] + n[1] + 2, ci * n[0]:ci * n[0] + n[0] + 2]
        own = (slice(1, -1),) * 3
        assert np.max(np.abs(xr[own] - ref[own])) <= 1e-12 * np.max(np.abs(x))


@pytest.mark.parametrize("n,pgrid,overlap_min", CASES, ids=IDS)
def test_native_driver_ranks_sharing_one_gpu_equal_single_domain(n, pgrid, overlap_min, tmp_path, oracle):
    """cedar_amd_dist3_* (the distributed V-cycle below the C ABI, cedar_amd/csrc/dist3.cpp) on 2 and 4 ranks sharing
    the GPU over the host-staged transport handed in as the ABI's transport table: the single-domain history and
    solution, on every rank grid shape, with and without the overlapped y/z halo, slab and row-class sweeps"""
    world = pgrid[0] * pgrid[1] * pgrid[2]
    _spawn(_worker, world, (_free_port(), n, pgrid, str(tmp_path), overlap_min, "native"))
    _check_against_single_domain(n, pgrid, tmp_path, oracle)


# Rank grids with an x / y split where the levels take the partial-sum sweep (runs of 2 rows so that boxes of a few thousand
# points qualify): boundary-first chain + one masked launch per k-parity (dist3.cpp chain_parity).  Every role of a rank:
# neighbour on the low / high / both sides in x and in y, with and without a z split.
CHAIN_CASES = [((16, 32, 8), (2, 2, 1)), ((16, 16, 8), (3, 1, 1)), ((8, 24, 8), (1, 3, 1)), ((16, 16, 8), (2, 1, 2)),
               ((8, 16, 8), (1, 2, 2)), ((32, 32, 16), (2, 2, 1)), ((8, 16, 8), (2, 2, 1))]
CHAIN_IDS = ["4ranks-xy", "3ranks-x", "3ranks-y", "4ranks-xz", "4ranks-yz", "4ranks-xy-two-chain-levels", "4ranks-xy-8-columns"]


# the production run length (16 rows from 224 rows per box on) on a 2 x 2 x 1 grid
CHAIN_CASES.append(((16, 256, 32), (2, 2, 1)))  # (z deep enough for a four-level hierarchy: the coarsest level is solved directly)
CHAIN_IDS.append("4ranks-xy-default-run-length")


# 128 rows per rank: below the 160 rows from which a single GPU takes the partial-sum sweep, but on a rank grid with an x / y
# split the driver registers it from 128 rows on (runs of 8 rows)
CHAIN_CASES.append(((16, 128, 16), (2, 2, 1)))
CHAIN_IDS.append("4ranks-xy-128-rows-default-runs")
DEFAULT_RUNS = {(16, 256, 32), (16, 128, 16)}
# (a 3 x 2 x 1 grid -- a rank with neighbours on both sides in x AND one in y -- runs on the CPU statement of the chain,
# tests/test_dist_cpu.py: six rank processes plus this one would exceed the six a one-GPU box lets share its card)
# rows of 512 points per rank: the 256-lane kernels of the production size (masks on lanes 254 / 255, dense column copy of a
# 514-point row), two chain levels; the gathered coarsest level has 4096 unknowns in a band of 1024
CHAIN_CASES.append(((512, 16, 16), (2, 1, 1)))
CHAIN_IDS.append("2ranks-x-512-point-rows")


@pytest.mark.parametrize("n,pgrid", CHAIN_CASES, ids=CHAIN_IDS)
def test_native_driver_boundary_first_chain_equals_single_domain(n, pgrid, tmp_path, oracle, monkeypatch):
    """the partial-sum sweep on rank grids with an x / y split: columns and rows next to a neighbouring rank relaxed ahead,
    stage by stage, the rest of a k-parity in one masked launch -- the single-domain history and solution"""
    if n not in DEFAULT_RUNS:
        monkeypatch.setenv("CEDAR_AMD_FRUN", "2")
    world = pgrid[0] * pgrid[1] * pgrid[2]
    # agglomerate_below = 4: (32, 32, 16) keeps levels 0 and 1 distributed (level 2, 8 x 8 x 4 per rank, is gathered), both on
    # the chain
    deep = {(32, 32, 16): 2, (512, 16, 16): 2}.get(n)
    maxit = 5
    _spawn(_worker, world, (_free_port(), n, pgrid, str(tmp_path), 96, "native", 4 if deep else 64, maxit))
    assert int(open(tmp_path / "chain_levels.txt").read()) == (deep or 1)
    _check_against_single_domain(n, pgrid, tmp_path, oracle, maxit)


# overlap_min = 4: the y/z halo of a row pass travels on a side HIP stream under the interior rows of
# the next pass wherever the level has an interior (the production default, 96, would leave these
# small grids on the in-order path, which the first two cases keep covering)
@pytest.mark.parametrize("n,pgrid,overlap_min", [CASES[0], CASES[3], CASES[4], CASES[7]], ids=[IDS[0], IDS[3], IDS[4], IDS[7]])
def test_two_ranks_one_gpu_equal_single_domain(n, pgrid, overlap_min, tmp_path, oracle):
    """the same orchestration in Python on the GPU backend (cedar_amd/dist.py: the statement of the algorithm that also
    runs on the CPU against the oracle, tests/test_dist_cpu.py)"""
    world = pgrid[0] * pgrid[1] * pgrid[2]
    _spawn(_worker, world, (_free_port(), n, pgrid, str(tmp_path), overlap_min))
    _check_against_single_domain(n, pgrid, tmp_path, oracle)


def _rccl_worker(rank, world, port, outdir):
    for p in (HERE, ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    import problems as pb
    from cedar_amd import capi
    from cedar_amd.comm import NativeComm, Stream
    from cedar_amd.dist import DistSolver3, GpuBackend, Topology
    assert "torch" not in sys.modules
    capi.set_device(0)
    comm = NativeComm(rank, world)
    out = {}
    # grouped self send/recv of two messages, on a side stream of the library
    a = capi.DeviceArray.from_numpy(np.arange(1000.0))
    r = capi.DeviceArray((1000,))
    st = Stream()
    main = capi.lib.cedar_amd_get_stream()
    capi.lib.cedar_amd_stream_wait(st.h, main)
    capi.lib.cedar_amd_set_stream(st.h)
    comm.p2p([(0, a, 0, 600), (0, a, 600, 400)], [(0, r, 0, 600), (0, r, 600, 400)])
    capi.lib.cedar_amd_set_stream(main)
    capi.lib.cedar_amd_stream_wait(main, st.h)
    capi.lib.cedar_amd_device_sync()
    out["p2p"] = bool(np.array_equal(r.numpy(), np.arange(1000.0)))
    out["sum"] = comm.allreduce_sum(3.5)
    g = capi.DeviceArray((1000,))
    comm.allgather(a, 1000, g)
    capi.sync()
    out["gather"] = bool(np.array_equal(g.numpy(), np.arange(1000.0)))
    # the whole distributed solver on a 1x1x1 rank grid over RCCL (all-gather of the coarse level, norm all-reduce)
    n = (24, 20, 16)
    g3 = (n[2] + 2, n[1] + 2, n[0] + 2)
    so = pb.random_op(g3, 14, 77)
    b = pb.uniform(g3, 78, -1, 1) * pb.interior_mask(g3)
    be = GpuBackend(comm, 0)
    s = DistSolver3(be, Topology(0, 1), be.from_numpy(so), max_iter=4, agglomerate_below=8)
    x = be.zeros(b.shape)
    out["hist"] = [float(v) for v in s.solve(be.from_numpy(b), x)]
    # and the native driver with the RCCL communicator handle (all-gather + all-reduce inside the library)
    from cedar_amd.dist3 import DistSolver3 as Native
    A2 = be.from_numpy(so)
    s2 = Native(comm, 0, 1, A2, max_iter=4, agglomerate_below=8)
    x2 = be.zeros(b.shape)
    out["hist_native"] = s2.solve(be.from_numpy(b), x2)
    s2.close()
    comm.close()
    import json
    json.dump(out, open(os.path.join(outdir, "rccl.json"), "w"))


def test_rccl_transport_one_rank(tmp_path, oracle):
    """ncclCommInitRank / grouped ncclSend+ncclRecv / ncclAllReduce / ncclAllGather through the library's C ABI"""
    import json
    import problems as pb
    _spawn(_rccl_worker, 1, (_free_port(), str(tmp_path)))
    out = json.load(open(tmp_path / "rccl.json"))
    assert out["p2p"] and out["gather"] and out["sum"] == 3.5
    n = (24, 20, 16)
    g3 = (n[2] + 2, n[1] + 2, n[0] + 2)
    so = pb.random_op(g3, 14, 77)
    b = pb.uniform(g3, 78, -1, 1) * pb.interior_mask(g3)
    ml = oracle.ml_create(so)
    x = np.zeros_like(b)
    want = ml.solve(b, x, maxiter=4)
    ml.close()
    np.testing.assert_allclose(out["hist"], want, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(out["hist_native"], want, rtol=1e-10, atol=1e-14)
