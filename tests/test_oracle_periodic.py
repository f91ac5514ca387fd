"""2D periodic boundary conditions: the oracle's restatement of the reference's
periodic branches against the reference's own Fortran (oracle/_ref), kernel by kernel on seeded
inputs with random ghost layers (each kernel is a deterministic function of its arrays).
Bit-exact except the dense Cholesky (MKL vs the unblocked netlib order) and what consumes it.
Skipped when oracle/_ref was not built (it cannot travel to the GPU box as source); the committed
goldens tests/golden/periodic2d.npz carry the same comparison there."""
import numpy as np
import pytest

import problems as pb

SHAPES = [(9, 9), (16, 16), (17, 12), (12, 17), (31, 20), (8, 8), (6, 7)]
IBCS = [1, 2, 3]


@pytest.fixture(scope="module")
def ref():
    try:
        from pyoracle import Ref
        return Ref()
    except (FileNotFoundError, OSError):
        pytest.skip("oracle/_ref not built")


def coarse(n):
    return int((n - 1) / 2.0 + 1)


def arrays(shape, nst, seed=3):
    nx, ny = shape
    g = (ny + 2, nx + 2)
    gc = (coarse(ny) + 2, coarse(nx) + 2)
    so = pb.random_op(g, nst, seed, zero_ghost=False)
    return g, gc, so


@pytest.mark.parametrize("ibc", IBCS)
@pytest.mark.parametrize("nst", [3, 5])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_periodic_kernels_match_reference(oracle, ref, shape, nst, ibc):
    g, gc, so = arrays(shape, nst)
    out = {}
    for name, B in (("o", oracle), ("r", ref)):
        sor = np.zeros((2,) + g)
        B.setup_recip2(so, sor)
        qf, q = pb.uniform(g, 11, -1, 1), pb.uniform(g, 12, -1, 1)
        for ud in (0, 1):
            B.relax2(so, qf, q, sor, ud, ibc=ibc)
            out[name, f"relax{ud}"] = q.copy()
        ci = np.zeros((8,) + gc)
        B.setup_interp2(so, ci, ibc=ibc)
        out[name, "interp"] = ci.copy()
        soc = np.zeros((5,) + gc)
        B.galerkin2(so, soc, ci, ibc=ibc)
        out[name, "galerkin"] = soc.copy()
        r, qc = pb.uniform(g, 13, -1, 1), np.zeros(gc)
        B.restrict2(r, qc, ci, ibc=ibc)
        out[name, "restrict_qc"], out[name, "restrict_q"] = qc.copy(), r.copy()
        x, xc, res = pb.uniform(g, 14, -1, 1), pb.uniform(gc, 15, -1, 1), pb.uniform(g, 16, -1, 1)
        B.interp_add2(x, xc, res, so, ci, ibc=ibc)
        out[name, "interp_add_q"], out[name, "interp_add_res"] = x.copy(), res.copy()
    for key in ("relax0", "relax1", "interp", "restrict_qc", "restrict_q", "interp_add_q", "interp_add_res"):
        assert np.array_equal(out["o", key], out["r", key]), (key, np.max(np.abs(out["o", key] - out["r", key])))
    a, b = out["o", "galerkin"], out["r", "galerkin"]
    assert np.max(np.abs(a - b)) <= 1e-13 * np.max(np.abs(b))


@pytest.mark.parametrize("ibc", IBCS)
@pytest.mark.parametrize("nst", [3, 5])
@pytest.mark.parametrize("shape", [(3, 3), (4, 3), (3, 5), (5, 4)], ids=lambda s: "x".join(map(str, s)))
def test_periodic_coarse_solve_matches_reference(oracle, ref, shape, nst, ibc):
    """dense assembly + DPOTRF/DPOTRS + mean removal + ghost wraps on a coarsest-grid-sized problem"""
    nx, ny = shape
    g = (ny + 2, nx + 2)
    so = pb.random_op(g, nst, 21, zero_ghost=False)
    so[0] *= 4.0  # keep the wrapped matrix positive definite
    n = nx * ny
    qf = pb.uniform(g, 22, -1, 1)
    res = {}
    for name, B in (("o", oracle), ("r", ref)):
        abd = np.zeros((n, n))
        B.setup_cg2(so, abd, ibc=ibc)
        q = pb.uniform(g, 23, -1, 1)
        B.solve_cg2(q, qf, abd, ibc=ibc)
        res[name] = (abd.copy(), q.copy())
    iu = np.triu_indices(n)
    ao, ar = res["o"][0].T[iu], res["r"][0].T[iu]  # column-major ABD(n,n): upper triangle
    assert np.max(np.abs(ao - ar)) <= 1e-13 * np.max(np.abs(ar))
    qo, qr = res["o"][1], res["r"][1]
    assert np.max(np.abs(qo - qr)) <= 1e-12 * np.max(np.abs(qr))
