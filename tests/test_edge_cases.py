"""Edge cases the reference's own tests exercise implicitly: tiny grids that give a one-level
hierarchy (direct solve only), ragged extents, odd/even mixes (the IICF1 = IIC ghost-column
quirk), early exit on tolerance.  CPU: the oracle against an independent dense solve of the
assembled operator; GPU: the device solver against the oracle."""
import numpy as np
import pytest

import problems as pb

SHAPES_2D = [(3, 3), (4, 3), (5, 5), (6, 9), (37, 6), (16, 33)]
SHAPES_3D = [(3, 3, 3), (4, 5, 3), (5, 5, 5), (6, 9, 7), (12, 5, 21)]


def slot_offsets(nd, nst):
    """slot s stored at P couples the points P+a and P+b; offsets in array (slowest-first) order"""
    if nd == 2:
        return {0: ((0, 0), (0, 0)), 1: ((0, 0), (0, -1)), 2: ((0, 0), (-1, 0)), 3: ((0, 0), (-1, -1)), 4: ((-1, 0), (0, -1))}
    # (di,dj,dk) tables of DESIGN.md section 2, reversed to (dk,dj,di)
    A3 = [(0, 0, 0)] * 5 + [(0, -1, 0)] + [(0, 0, 0)] + [(0, -1, 0), (0, -1, 0), (-1, -1, 0), (-1, 0, 0), (-1, 0, 0)] + [(0, 0, 0)] * 2
    B3 = [(0, 0, 0), (-1, 0, 0), (0, -1, 0), (0, 0, -1), (-1, -1, 0), (-1, 0, 0), (-1, 0, -1), (-1, 0, -1), (0, 0, -1),
          (0, 0, -1), (0, 0, -1), (0, -1, -1), (0, -1, -1), (-1, -1, -1)]
    return {s: (A3[s][::-1], B3[s][::-1]) for s in range(nst)}


def dense_from_stencil(so):
    """assemble the symmetric matrix of a Cedar stencil operator (positive off-diagonals stored)"""
    nd = so.ndim - 1
    g = so.shape[1:]
    n = [s - 2 for s in g]
    N = int(np.prod(n))
    idx = -np.ones(g, dtype=int)
    idx[tuple(slice(1, -1) for _ in g)] = np.arange(N).reshape(n)
    A = np.zeros((N, N))
    offs = slot_offsets(nd, so.shape[0])
    for s in range(so.shape[0]):
        a, b = offs[s]
        for p in np.ndindex(*g):
            v = so[(s,) + p]
            if v == 0.0:
                continue
            pa = tuple(p[d] + a[d] for d in range(nd))
            pb_ = tuple(p[d] + b[d] for d in range(nd))
            if any(q < 0 or q >= g[d] for d, q in enumerate(pa)) or any(q < 0 or q >= g[d] for d, q in enumerate(pb_)):
                continue
            ia, ib = idx[pa], idx[pb_]
            if ia < 0 or ib < 0:
                continue
            if s == 0:
                A[ia, ia] += v
            else:
                A[ia, ib] -= v
                A[ib, ia] -= v
    return A


def make(shape, seed=5):
    nd = len(shape)
    g = tuple(s + 2 for s in shape[::-1])
    so = pb.random_op(g, 5 if nd == 2 else 14, seed)
    # a valid Cedar operator carries no coupling across the physical boundary (the band
    # assembly of the coarse solve relies on it): zero every entry that reaches a ghost
    inner = pb.interior_mask(g)
    for s, (a, c) in slot_offsets(nd, so.shape[0]).items():
        for o in (a, c):
            so[s] *= np.roll(inner, tuple(-d for d in o), axis=tuple(range(nd)))
    b = pb.uniform(g, seed + 1, -1, 1) * pb.interior_mask(g)
    return so, b


@pytest.mark.parametrize("shape", SHAPES_2D + SHAPES_3D, ids=lambda s: "x".join(map(str, s)))
def test_oracle_solve_matches_dense_solve(oracle, shape):
    so, b = make(shape)
    ml = oracle.ml_create(so)
    x = np.zeros_like(b)
    h = ml.solve(b, x, maxiter=25, tol=1e-12)
    nlev = ml.nlevels()
    ml.close()
    A = dense_from_stencil(so)
    inner = tuple(slice(1, -1) for _ in b.shape)
    want = np.linalg.solve(A, b[inner].ravel()).reshape(b[inner].shape)
    assert h[-1] < 1e-10, (nlev, h)
    assert np.max(np.abs(x[inner] - want)) <= 1e-9 * np.max(np.abs(want))
    if min(shape) <= 4:
        assert nlev == 1 and len(h) == 2  # direct solve: converged after one "cycle"


def test_early_exit_on_tolerance(oracle):
    so, b = pb.poisson2(64, 64), pb.rhs2(64, 64)
    ml = oracle.ml_create(so)
    x = np.zeros_like(b)
    h = ml.solve(b, x, maxiter=10, tol=1e-3)
    ml.close()
    assert len(h) < 11 and h[-1] < 1e-3 and h[-2] >= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES_2D + SHAPES_3D, ids=lambda s: "x".join(map(str, s)))
def test_gpu_solver_matches_oracle_on_edge_shapes(oracle, shape):
    from cedar_amd import capi
    so, b = make(shape)
    ml = oracle.ml_create(so)
    xo = np.zeros_like(b)
    ho = ml.solve(b, xo, maxiter=6)
    s = capi.Solver(so, max_iter=6)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    assert s.nlevels() == ml.nlevels()
    s.close()
    ml.close()
    assert len(h) == len(ho)
    np.testing.assert_allclose(h, ho, rtol=1e-9, atol=1e-13)
    assert np.max(np.abs(x - xo)) <= 1e-11 * max(np.max(np.abs(xo)), 1e-300)


@pytest.mark.gpu
def test_gpu_early_exit_and_fcycle_direct():
    from cedar_amd import capi
    so, b = pb.poisson2(64, 64), pb.rhs2(64, 64)
    s = capi.Solver(so, tol=1e-3)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    s.close()
    assert len(h) < 11 and h[-1] < 1e-3 and h[-2] >= 1e-3


@pytest.mark.parametrize("shape", [(8, 6, 6), (9, 7, 5), (12, 4, 3), (6, 3, 4)], ids=str)
def test_oracle_row_class_parts_compose_for_every_face_mask(oracle, shape):
    """the checker's side of cedar_amd_relax3_pass_part / _planes: interior + shell (parts 1+2, plane parts 3+4) equal the
    whole colour for every set of faces that have a neighbouring rank, and a face without a neighbour moves its rows
    from the shell to the interior"""
    nx, ny, nz = shape
    g = (nz + 2, ny + 2, nx + 2)
    so = pb.random_op(g, 14, 71, zero_ghost=False)
    qf, q0 = pb.uniform(g, 72, -1, 1), pb.uniform(g, 73, -1, 1)
    sor = np.zeros((2,) + g)
    oracle.setup_recip3(so, sor)
    for pts in range(1, 9):
        whole = q0.copy()
        oracle.relax_colour3_part(so, qf, whole, sor, pts, 0)
        for sides in range(16):
            for pa, pb_ in ((1, 2), (3, 4)):
                parts = q0.copy()
                oracle.relax_colour3_part(so, qf, parts, sor, pts, pa | (sides << 4))
                first = parts.copy()
                oracle.relax_colour3_part(so, qf, parts, sor, pts, pb_ | (sides << 4))
                assert np.array_equal(parts, whole), (shape, pts, sides, pa)
                if sides and pa == 1:
                    # rows of the first / last owned row or plane are interior exactly when that face has no neighbour
                    jj, kk = 1 + ((pts - 1) // 2) % 2, 1 + ((pts - 1) // 4) % 2
                    if jj == 1 and kk + 2 <= nz and kk > 1 or (jj == 1 and not (sides & 4) and kk == 1 and nz > 1):
                        touched = not np.array_equal(first[kk, 1, :], q0[kk, 1, :])
                        assert touched == (not (sides & 1)) or nx < 1
