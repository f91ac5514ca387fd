"""Cedar's C interface (include/cedar/capi.h) with MORE THAN ONE RANK: bmgN_topo_create(nprocx, nprocy[, nprocz]) ->
bmgN_operator_set / _apply -> bmgN_solver_create / _run on the domain-decomposed drivers below the C ABI, as the reference
runs this interface on its MPI solvers (src/2d/interface/c/{topo,operator,solver}.cc).  Ranks share the one GPU of the
test box over the host-staged transport handed in with cedar_amd_bmg_set_transport (RCCL refuses two ranks per device;
without a table the interface bootstraps an RCCL communicator itself).  Every rank sets only the entries of the points
it owns, in global coordinates; results against the oracle on the global problem."""
import ctypes as C
import json
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


def _worker(rank, world, port, nd, n, pgrid, outdir):
    for p in (HERE, ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(outdir)  # config.json is read from the working directory, as in the reference
    import problems as pb
    from cedar_amd import capi
    from cedar_amd.comm import SocketComm
    from cedar_amd.dist3 import _table_from
    import test_bmg_capi as tb
    L = tb._lib()
    capi.set_device(0)
    comm = SocketComm(rank, world)
    try:
        tab, keep = _table_from(comm)
        L.cedar_amd_bmg_set_rank(rank, world)
        L.cedar_amd_bmg_set_transport(C.byref(tab))
        gn = tuple(n[d] * pgrid[d] for d in range(nd))
        if nd == 2:
            ent, so = tb.vertex_stencil_2d(gn[0], gn[1])
            ci, cj = rank % pgrid[0], rank // pgrid[0]
            lo = (ci * n[0], cj * n[1])
            # a rank sets every entry whose STORAGE location lies in its local array (owned points and ghost layer): the
            # symmetric layout stores E / N / NE / SE / NW entries at a neighbouring vertex (operator.cc:29-66), so e.g. the
            # NW slot of a rank's first owned point is named only through vertices of two other ranks -- the interface takes
            # ghost vertices (coordinates - is + 2 >= 0) exactly for that
            shift = {"C": (0, 0), "W": (0, 0), "S": (0, 0), "SW": (0, 0), "NW": (0, 1), "SE": (1, 0), "N": (0, 1), "NE": (1, 1), "E": (1, 0)}

            def stored_here(e):
                si, sj = e[0] + shift[e[2]][0] - lo[0] + 1, e[1] + shift[e[2]][1] - lo[1] + 1  # local array index incl. ghost
                return 0 <= si <= n[0] + 1 and 0 <= sj <= n[1] + 1 and e[0] - lo[0] + 1 >= 0 and e[1] - lo[1] + 1 >= 0
            mine = [e for e in ent if stored_here(e)]
            topo = L.bmg2_topo_create(0, gn[0], gn[1], (C.c_uint * pgrid[0])(*[n[0]] * pgrid[0]),
                                      (C.c_uint * pgrid[1])(*[n[1]] * pgrid[1]), pgrid[0], pgrid[1])
            op = L.bmg2_operator_create(topo)
            coords = (tb.Coord2 * len(mine))(*[tb.Coord2(i, j, tb.BMG2[d]) for i, j, d, _ in mine])
            vals = (C.c_double * len(mine))(*[v for *_, v in mine])
            L.bmg2_operator_set(op, len(mine), coords, vals)
            sl = (slice(lo[1], lo[1] + n[1]), slice(lo[0], lo[0] + n[0]))
            gx = pb.uniform((gn[1], gn[0]), 21, -1, 1)
            gb = pb.uniform((gn[1], gn[0]), 22, -1, 1)
            apply_, create, run = L.bmg2_operator_apply, L.bmg2_solver_create, L.bmg2_solver_run
            destroy_s, destroy_o = L.bmg2_solver_destroy, L.bmg2_operator_destroy
        else:
            so = pb.fe3(*gn)
            ci, cj, ck = rank % pgrid[0], (rank // pgrid[0]) % pgrid[1], rank // (pgrid[0] * pgrid[1])
            lo = (ci * n[0], cj * n[1], ck * n[2])
            ent = []
            for s in range(14):
                kk, jj, ii = np.nonzero(so[s])
                for k, j, i in zip(kk, jj, ii):
                    i0, j0, k0 = i - 1, j - 1, k - 1  # 0-based global vertex the slot is stored at (3D: dir names the slot)
                    if lo[0] - 1 <= i0 <= lo[0] + n[0] and lo[1] - 1 <= j0 <= lo[1] + n[1] and lo[2] - 1 <= k0 <= lo[2] + n[2]:
                        ent.append((i0, j0, k0, s, so[s, k, j, i] if s == 0 else -so[s, k, j, i]))
            topo = L.bmg3_topo_create(0, gn[0], gn[1], gn[2], (C.c_uint * pgrid[0])(*[n[0]] * pgrid[0]),
                                      (C.c_uint * pgrid[1])(*[n[1]] * pgrid[1]), (C.c_uint * pgrid[2])(*[n[2]] * pgrid[2]),
                                      pgrid[0], pgrid[1], pgrid[2])
            op = L.bmg3_operator_create(topo)
            coords = (tb.Coord3 * len(ent))(*[tb.Coord3(int(i), int(j), int(k), s) for i, j, k, s, _ in ent])
            vals = (C.c_double * len(ent))(*[float(v) for *_, v in ent])
            L.bmg3_operator_set(op, len(ent), coords, vals)
            sl = (slice(lo[2], lo[2] + n[2]), slice(lo[1], lo[1] + n[1]), slice(lo[0], lo[0] + n[0]))
            gx = pb.uniform(gn[::-1], 31, -1, 1)
            gb = pb.uniform(gn[::-1], 32, -1, 1)
            apply_, create, run = L.bmg3_operator_apply, L.bmg3_solver_create, L.bmg3_solver_run
            destroy_s, destroy_o = L.bmg3_solver_destroy, L.bmg3_operator_destroy
        assert topo and op
        x = np.ascontiguousarray(gx[sl])
        y = np.zeros_like(x)
        apply_(op, x.ctypes.data, y.ctypes.data)
        opp = C.c_void_p(op)
        slv = create(C.byref(opp))
        assert slv
        b = np.ascontiguousarray(gb[sl])
        sol = np.full_like(b, 7.0)
        run(slv, sol.ctypes.data, b.ctypes.data)
        np.save(os.path.join(outdir, f"y{rank}.npy"), y)
        np.save(os.path.join(outdir, f"x{rank}.npy"), sol)
        destroy_s(slv)
        destroy_o(op)
        assert "torch" not in sys.modules
    finally:
        comm.close()


def _spawn(world, args):
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_worker, args=(r, world) + args) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(600)
    bad = [p.exitcode for p in ps if p.exitcode != 0]
    for p in ps:
        if p.is_alive():
            p.kill()
    assert not bad, f"rank processes failed: exit codes {bad}"


@pytest.mark.parametrize("nd,n,pgrid", [(2, (24, 16), (2, 1)), (2, (16, 12), (2, 2)), (3, (12, 10, 8), (1, 1, 2)), (3, (8, 8, 8), (2, 2, 1))],
                         ids=["2d-2x1", "2d-2x2", "3d-1x1x2", "3d-2x2x1"])
def test_bmg_interface_on_several_ranks(nd, n, pgrid, tmp_path, oracle):
    import problems as pb
    import test_bmg_capi as tb
    world = int(np.prod(pgrid))
    json.dump({"solver": {"cycle": {"nrelax-pre": 2, "nrelax-post": 1}, "max-iter": 6, "tol": 1e-30}}, open(tmp_path / "config.json", "w"))
    _spawn(world, (_free_port(), nd, n, pgrid, str(tmp_path)))
    gn = tuple(n[d] * pgrid[d] for d in range(nd))
    if nd == 2:
        _, so = tb.vertex_stencil_2d(gn[0], gn[1])
        g = (gn[1] + 2, gn[0] + 2)
        gx, gb = pb.uniform((gn[1], gn[0]), 21, -1, 1), pb.uniform((gn[1], gn[0]), 22, -1, 1)
        inner = (slice(1, -1),) * 2
        matvec = oracle.matvec2
    else:
        so = pb.fe3(*gn)
        g = (gn[2] + 2, gn[1] + 2, gn[0] + 2)
        gx, gb = pb.uniform(gn[::-1], 31, -1, 1), pb.uniform(gn[::-1], 32, -1, 1)
        inner = (slice(1, -1),) * 3
        matvec = oracle.matvec3
    xg, yo = np.zeros(g), np.zeros(g)
    xg[inner] = gx
    matvec(so, xg, yo)
    bg, xo = np.zeros(g), np.zeros(g)
    bg[inner] = gb
    ml = oracle.ml_create(so)
    ml.solve(bg, xo, maxiter=6, tol=1e-30)
    ml.close()
    for r in range(world):
        if nd == 2:
            ci, cj = r % pgrid[0], r // pgrid[0]
            sl = (slice(cj * n[1], (cj + 1) * n[1]), slice(ci * n[0], (ci + 1) * n[0]))
        else:
            ci, cj, ck = r % pgrid[0], (r // pgrid[0]) % pgrid[1], r // (pgrid[0] * pgrid[1])
            sl = (slice(ck * n[2], (ck + 1) * n[2]), slice(cj * n[1], (cj + 1) * n[1]), slice(ci * n[0], (ci + 1) * n[0]))
        y, x = np.load(tmp_path / f"y{r}.npy"), np.load(tmp_path / f"x{r}.npy")
        np.testing.assert_array_equal(y, yo[inner][sl])  # operator_apply: ghost layer of x from the neighbours, same term order
        assert np.max(np.abs(x - xo[inner][sl])) <= 1e-11 * np.max(np.abs(xo))
