"""Cedar's C interface (include/cedar/capi.h: bmg2_*/bmg3_*) on the GPU library, driven the way a C
caller would (ctypes structs = the header's structs).  Reference behaviour:
src/{2d,3d}/interface/c/{topo,operator,solver}.cc.  The reference ships no test or example for this
interface, so parity is pinned on (i) the storage the calls must produce (vertex-based -> symmetric
layout, sign flip), (ii) operator_apply against the oracle's matvec and a dense assembly, (iii)
solver_run against the oracle's multilevel solve of the same stored operator."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import problems as pb
from test_edge_cases import dense_from_stencil

BMG2 = dict(C=0, W=1, S=2, SW=3, NW=4, SE=5, N=6, NE=7, E=8)


class Coord2(C.Structure):
    _fields_ = [("i", C.c_uint), ("j", C.c_uint), ("dir", C.c_int)]


class Coord3(C.Structure):
    _fields_ = [("i", C.c_uint), ("j", C.c_uint), ("k", C.c_uint), ("dir", C.c_int)]


def vertex_stencil_2d(nx, ny, seed=3):
    """a symmetric 9-point operator given vertex based with natural signs: returns the list of
    (i, j, dir, value) a C caller would pass and the same operator in Cedar's stored layout"""
    g = (ny + 2, nx + 2)
    so = pb.random_op(g, 5, seed)
    inner = pb.interior_mask(g)
    # drop couplings that leave the domain (a C caller never sets them)
    so[1] *= np.roll(inner, 1, axis=1)
    so[2] *= np.roll(inner, 1, axis=0)
    so[3] *= np.roll(inner, (1, 1), axis=(0, 1))
    so[4] *= np.roll(inner, 1, axis=0) * np.roll(inner, 1, axis=1)
    ent = []
    for j in range(ny):
        for i in range(nx):
            J, I = j + 1, i + 1
            ent.append((i, j, "C", so[0, J, I]))
            # every off-diagonal is set from BOTH ends with different directions, to exercise all 8
            if i > 0:
                ent.append((i, j, "W", -so[1, J, I]))
                ent.append((i - 1, j, "E", -so[1, J, I]))
            if j > 0:
                ent.append((i, j, "S", -so[2, J, I]))
                ent.append((i, j - 1, "N", -so[2, J, I]))
            if i > 0 and j > 0:
                ent.append((i, j, "SW", -so[3, J, I]))
                ent.append((i - 1, j - 1, "NE", -so[3, J, I]))
            if i > 0 and j > 0:
                # knw@(I,J) couples (I,J-1) with (I-1,J): NW of (I,J-1), SE of (I-1,J)
                ent.append((i, j - 1, "NW", -so[4, J, I]))
                ent.append((i - 1, j, "SE", -so[4, J, I]))
    return ent, so


def test_oracle_matvec_matches_dense_and_residual(oracle):
    for shape, nst in (((9, 7), 5), ((8, 8), 3), ((6, 5, 7), 14), ((5, 5, 5), 4)):
        g = tuple(s + 2 for s in shape[::-1])
        so = pb.random_op(g, nst, 11)
        q = pb.uniform(g, 12, -1, 1) * pb.interior_mask(g)
        y = np.zeros(g)
        r = np.zeros(g)
        if len(shape) == 2:
            oracle.matvec2(so, q, y)
            oracle.residual2(so, np.zeros(g), q, r)
        else:
            oracle.matvec3(so, q, y)
            oracle.residual3(so, np.zeros(g), q, r)
        inner = tuple(slice(1, -1) for _ in g)
        np.testing.assert_allclose(y[inner], -r[inner], rtol=0, atol=1e-13 * np.abs(y).max())
        full = so if nst in (5, 14) else np.concatenate([so, np.zeros((({3: 5, 4: 14}[nst]) - nst,) + g)])
        A = dense_from_stencil(full)
        np.testing.assert_allclose(y[inner].ravel(), A @ q[inner].ravel(), rtol=0, atol=1e-13 * np.abs(y).max())


def _lib():
    from cedar_amd import capi
    L = capi.lib
    for n in ("bmg2_topo_create", "bmg3_topo_create", "bmg2_operator_create", "bmg3_operator_create",
              "bmg2_solver_create", "bmg3_solver_create"):
        getattr(L, n).restype = C.c_void_p
    L.bmg2_topo_create.argtypes = [C.c_int, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_int, C.c_int]
    L.bmg3_topo_create.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                   C.POINTER(C.c_uint), C.c_int, C.c_int, C.c_int]
    for n in ("bmg2_operator_create", "bmg3_operator_create", "bmg2_operator_destroy", "bmg3_operator_destroy",
              "bmg2_solver_destroy", "bmg3_solver_destroy", "bmg2_operator_dump", "bmg3_operator_dump"):
        getattr(L, n).argtypes = [C.c_void_p]
    L.bmg2_operator_set.argtypes = [C.c_void_p, C.c_uint, C.POINTER(Coord2), C.POINTER(C.c_double)]
    L.bmg3_operator_set.argtypes = [C.c_void_p, C.c_uint, C.POINTER(Coord3), C.POINTER(C.c_double)]
    for n in ("bmg2_operator_apply", "bmg3_operator_apply", "bmg2_solver_run", "bmg3_solver_run"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.bmg2_solver_create.argtypes = [C.POINTER(C.c_void_p)]
    L.bmg3_solver_create.argtypes = [C.POINTER(C.c_void_p)]
    L.bmg_timer_save.argtypes = [C.c_char_p]
    L.cedar_amd_bmg_set_rank.argtypes = [C.c_int, C.c_int]
    L.cedar_amd_bmg_set_transport.argtypes = [C.c_void_p]
    return L


def test_topology_that_does_not_match_the_launched_ranks_is_refused():
    """a 2 x 1 process grid in a job of ONE rank (no RANK / WORLD_SIZE in the environment): reported and NULL.  With
    matching ranks the interface runs on the domain-decomposed drivers (tests/test_gpu_bmg_dist.py)."""
    L = _lib()
    one = (C.c_uint * 2)(8, 8)
    assert L.bmg2_topo_create(0, 16, 8, one, one, 2, 1) is None


@pytest.mark.gpu
def test_bmg2_interface_against_oracle(oracle, tmp_path, monkeypatch):
    L = _lib()
    nx, ny = 37, 22
    ent, so = vertex_stencil_2d(nx, ny)
    monkeypatch.chdir(tmp_path)
    json.dump({"solver": {"cycle": {"nrelax-pre": 2, "nrelax-post": 1}, "max-iter": 8, "tol": 1e-30}}, open("config.json", "w"))
    topo = L.bmg2_topo_create(0, nx, ny, (C.c_uint * 1)(nx), (C.c_uint * 1)(ny), 1, 1)
    op = L.bmg2_operator_create(topo)
    coords = (Coord2 * len(ent))(*[Coord2(i, j, BMG2[d]) for i, j, d, _ in ent])
    vals = (C.c_double * len(ent))(*[v for *_, v in ent])
    L.bmg2_operator_set(op, len(ent), coords, vals)
    # the reference flips the sign of the caller's off-diagonal values in place
    got = np.frombuffer(vals, dtype=np.float64)
    want = np.array([v if d == "C" else -v for *_, d, v in ent])
    np.testing.assert_array_equal(got, want)

    g = (ny + 2, nx + 2)
    x = pb.uniform((ny, nx), 21, -1, 1)
    y = np.zeros((ny, nx))
    L.bmg2_operator_apply(op, x.ctypes.data, y.ctypes.data)
    xg = np.zeros(g)
    xg[1:-1, 1:-1] = x
    yo = np.zeros(g)
    oracle.matvec2(so, xg, yo)
    np.testing.assert_array_equal(y, yo[1:-1, 1:-1])  # same term order => bit-identical

    b = pb.uniform((ny, nx), 22, -1, 1)
    opp = C.c_void_p(op)
    slv = L.bmg2_solver_create(C.byref(opp))
    sol = np.full((ny, nx), 7.0)  # run() must start from zero whatever the caller left in x
    L.bmg2_solver_run(slv, sol.ctypes.data, b.ctypes.data)
    bg = np.zeros(g)
    bg[1:-1, 1:-1] = b
    ml = oracle.ml_create(so)
    xo = np.zeros(g)
    ml.solve(bg, xo, maxiter=8, tol=1e-30)
    ml.close()
    assert np.max(np.abs(sol - xo[1:-1, 1:-1])) <= 1e-12 * np.max(np.abs(xo))
    L.bmg2_operator_dump(op)
    assert os.path.getsize("op0-0.txt") > 0
    L.bmg_timer_save(b"timings.json")
    t = json.load(open("timings.json"))
    assert t["solve"]["calls"] >= 1 and t["setup"]["seconds"] > 0
    L.bmg2_solver_destroy(slv)
    L.bmg2_operator_destroy(op)


@pytest.mark.gpu
def test_bmg3_interface_against_oracle(oracle, tmp_path, monkeypatch):
    L = _lib()
    n = (13, 10, 12)  # nx, ny, nz
    g = (n[2] + 2, n[1] + 2, n[0] + 2)
    so = pb.fe3(*n)
    monkeypatch.chdir(tmp_path)  # no config.json here: defaults V(2,1), 10 cycles, tol 1e-8
    topo = L.bmg3_topo_create(0, n[0], n[1], n[2], (C.c_uint * 1)(n[0]), (C.c_uint * 1)(n[1]), (C.c_uint * 1)(n[2]), 1, 1, 1)
    op = L.bmg3_operator_create(topo)
    ent = []
    for s in range(14):
        kk, jj, ii = np.nonzero(so[s])
        for k, j, i in zip(kk, jj, ii):
            # 3D: dir names the storage slot directly; coordinates 0-based => array index - 1
            ent.append((i - 1, j - 1, k - 1, s, so[s, k, j, i] if s == 0 else -so[s, k, j, i]))
    coords = (Coord3 * len(ent))(*[Coord3(int(i), int(j), int(k), s) for i, j, k, s, _ in ent])
    vals = (C.c_double * len(ent))(*[float(v) for *_, v in ent])
    L.bmg3_operator_set(op, len(ent), coords, vals)

    x = pb.uniform(n[::-1], 31, -1, 1)
    y = np.zeros(n[::-1])
    L.bmg3_operator_apply(op, x.ctypes.data, y.ctypes.data)
    xg = np.zeros(g)
    xg[1:-1, 1:-1, 1:-1] = x
    yo = np.zeros(g)
    oracle.matvec3(so, xg, yo)
    np.testing.assert_array_equal(y, yo[1:-1, 1:-1, 1:-1])

    b = pb.uniform(n[::-1], 32, -1, 1)
    opp = C.c_void_p(op)
    slv = L.bmg3_solver_create(C.byref(opp))
    sol = np.zeros(n[::-1])
    L.bmg3_solver_run(slv, sol.ctypes.data, b.ctypes.data)
    bg = np.zeros(g)
    bg[1:-1, 1:-1, 1:-1] = b
    ml = oracle.ml_create(so)
    xo = np.zeros(g)
    ml.solve(bg, xo, maxiter=10, tol=1e-8)
    ml.close()
    assert np.max(np.abs(sol - xo[1:-1, 1:-1, 1:-1])) <= 1e-11 * np.max(np.abs(xo))
    L.bmg3_solver_destroy(slv)
    L.bmg3_operator_destroy(op)
