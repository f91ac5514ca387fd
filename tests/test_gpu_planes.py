"""Plane relaxation on the GPU path: cedar_amd_planes_* (the plane_relax<rdir> kernel) and the
device-resident 3D solver with relaxation plane-xy / -xz / -yz / -xyz, against the reference's own known-answer test
(test/3d/test_planes.cc, restated in tests/test_oracle_planes.py) and against the oracle.  The plane solves go
through the 2D line / point kernels, whose agreement with the oracle is rounding-level (scan solves, Galerkin
association), hence tolerances instead of bit equality."""
import numpy as np
import pytest

import problems as pb
from test_oracle_planes import ANISO, DOWN, KAT, KAT_PLANE, UP, kat_check, varying_op

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from cedar_amd import capi
    assert capi.device_count() >= 1, "no GPU visible"
    return capi.Kernels()


@pytest.mark.parametrize("d,nx,ny,nz", KAT, ids=lambda v: str(v))
@pytest.mark.parametrize("nst", [4, 14])
def test_reference_known_answer_planes(K, oracle, d, nx, ny, nz, nst):
    kat_check(K, oracle, d, nx, ny, nz, nst)


@pytest.mark.parametrize("d", ["xy", "xz", "yz"])
@pytest.mark.parametrize("nst", [4, 14])
@pytest.mark.parametrize("plane", [None, dict(relax="point", max_iter=3, tol=1e-30), dict(relax="line-x", max_iter=2, tol=0.5)],
                         ids=["default", "point3", "linex-tol"])
def test_plane_sweeps_vs_oracle(K, oracle, d, nst, plane):
    """plane-dependent coefficients (every plane solver still uses the last plane's, like the reference), DOWN then UP;
    the third configuration stops some planes early on the tolerance"""
    nx, ny, nz = 21, 18, 15
    so = varying_op(nx, ny, nz, nst, 31)
    b = pb.uniform(so.shape[1:], 33, -1, 1)
    x1 = pb.uniform(so.shape[1:], 32, -1, 1)
    x2 = x1.copy()
    for ud in (DOWN, UP):
        K.relax_planes3(so, x1, b, d, ud, plane=plane)
        oracle.relax_planes3(so, x2, b, d, ud, plane=plane)
        assert np.max(np.abs(x1 - x2)) <= 1e-11 * np.max(np.abs(x2)), (d, nst, ud, np.max(np.abs(x1 - x2)))


@pytest.mark.parametrize("d", ["xy", "xz", "yz"])
@pytest.mark.parametrize("shape,nst", [((21, 18, 15), 14), ((40, 9, 12), 4), ((7, 70, 6), 14)], ids=str)
def test_batched_planes_equal_one_solver_per_plane(K, monkeypatch, d, shape, nst):
    """the planes of a colour as one batch through the 2D kernels (default) against one solver instance per pair of
    planes on side streams (CEDAR_AMD_PLANE_BATCH=0): same kernels on the same data, bit-identical"""
    nx, ny, nz = shape
    so = varying_op(nx, ny, nz, nst, 71)
    b, x0 = pb.uniform(so.shape[1:], 72, -1, 1), pb.uniform(so.shape[1:], 73, -1, 1)
    out = []
    for batch in ("1", "0"):
        monkeypatch.setenv("CEDAR_AMD_PLANE_BATCH", batch)
        x = x0.copy()
        for ud in (DOWN, UP):
            K.relax_planes3(so, x, b, d, ud)
        out.append(x)
    assert not np.array_equal(out[0], x0)
    assert np.array_equal(out[0], out[1])


def test_plane_relax_on_device_arrays(K):
    from cedar_amd import capi
    so = varying_op(12, 10, 9, 14, 5)
    b, x = pb.uniform(so.shape[1:], 6, -1, 1), pb.uniform(so.shape[1:], 7, -1, 1)
    want = x.copy()
    K.relax_planes3(so, want, b, "xz", DOWN)
    d = [capi.DeviceArray.from_numpy(a) for a in (so, x, b)]
    K.relax_planes3(d[0], d[1], d[2], "xz", DOWN)
    assert np.array_equal(d[1].numpy(), want)


SOLVES = dict(ANISO)
SOLVES["fe27_xy_33x20x18"] = (lambda: pb.fe3(33, 20, 18), "plane-xy")
SOLVES["aniso7_xyz_40x36x33"] = (lambda: pb.diag_diffusion3(40, 36, 33, 1.0, 1e-2, 1e-4), "plane-xyz")
SOLVES["aniso7_xyz_64"] = (lambda: pb.diag_diffusion3(64, 64, 64, 1.0, 1e-2, 1e-4), "plane-xyz")
SOLVES["fe27_xz_65x30x66"] = (lambda: pb.fe3(65, 30, 66), "plane-xz")


@pytest.mark.parametrize("name", list(SOLVES), ids=str)
def test_solver_with_plane_relaxation_vs_oracle(oracle, name):
    from cedar_amd import capi
    mk, relax = SOLVES[name]
    so = mk()
    nz, ny, nx = (n - 2 for n in so.shape[1:])
    b = pb.rhs3(nx, ny, nz)
    s = capi.Solver(so, relax=relax)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    ml = oracle.ml_create(so, relax=relax)
    assert s.nlevels() == ml.nlevels()
    xo = np.zeros_like(b)
    ho = ml.solve(b, xo, maxiter=10, tol=1e-8)
    ml.close()
    assert len(h) == len(ho) and h[-1] < 1e-5  # (plane-xz alone on an isotropic problem converges at 0.2 per cycle)
    np.testing.assert_allclose(h, ho, rtol=1e-8, atol=1e-13)
    assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo))


@pytest.mark.parametrize("shape,nst", [((9, 8, 7), 14), ((7, 7, 7), 4), ((6, 9, 13), 14), ((16, 5, 9), 4), ((5, 17, 6), 14)], ids=str)
def test_small_and_ragged_grids_with_plane_relaxation(oracle, shape, nst):
    """odd / tiny extents: planes whose 2D solver has one or two levels, colours of unequal size, batches of one plane"""
    from cedar_amd import capi
    nx, ny, nz = shape
    so = varying_op(nx, ny, nz, nst, 91)
    b = pb.uniform(so.shape[1:], 92, -1, 1) * pb.interior_mask(so.shape[1:])
    s = capi.Solver(so, relax="plane-xyz", max_iter=4)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    ml = oracle.ml_create(so, relax="plane-xyz")
    assert s.nlevels() == ml.nlevels()
    xo = np.zeros_like(b)
    ho = ml.solve(b, xo, maxiter=4, tol=1e-8)
    ml.close()
    assert len(h) == len(ho)
    np.testing.assert_allclose(h, ho, rtol=1e-8, atol=1e-13)
    assert np.max(np.abs(x - xo)) <= 1e-10 * max(np.max(np.abs(xo)), 1e-300)


def test_f_cycle_with_plane_relaxation_vs_oracle(oracle):
    from cedar_amd import capi
    so = pb.diag_diffusion3(20, 18, 17, 1.0, 1.0, 1e-3)
    b = pb.rhs3(20, 18, 17)
    s = capi.Solver(so, relax="plane-xy", cycle="f", max_iter=4)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    ml = oracle.ml_create(so, relax="plane-xy", cycle="f")
    xo = np.zeros_like(b)
    ho = ml.solve(b, xo, maxiter=4, tol=1e-8)
    ml.close()
    assert len(h) == len(ho)
    np.testing.assert_allclose(h, ho, rtol=1e-8, atol=1e-13)


def test_plane_relaxation_refusals(capfd):
    from cedar_amd import capi
    s = capi.Solver(pb.poisson2(20, 20), relax="plane-xy")  # 2D: falls back to point relaxation, loudly
    assert "relaxation must be" in capfd.readouterr().err
    del s
    s = capi.Solver(pb.poisson3(8, 8, 8), relax="plane-xy", plane=dict(relax="plane-xz"))
    assert "relaxation must be" in capfd.readouterr().err


def _random_plane_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        nx, ny, nz = (int(v) for v in rng.integers(5, 24, size=3))
        out.append((nx, ny, nz, int(rng.choice([4, 14])), str(rng.choice(["plane-xy", "plane-xz", "plane-yz", "plane-xyz"])),
                    str(rng.choice(["v", "f"]))))
    return out


@pytest.mark.parametrize("nx,ny,nz,nst,relax,cycle", _random_plane_cases(10, 4102026), ids=str)
def test_random_small_grids_with_plane_relaxation_follow_the_oracle(oracle, nx, ny, nz, nst, relax, cycle):
    """seeded random extents: plane counts of either parity, planes whose 2D hierarchy has one to three levels"""
    from cedar_amd import capi
    so = varying_op(nx, ny, nz, nst, 17)
    b = pb.uniform(so.shape[1:], 18, -1, 1) * pb.interior_mask(so.shape[1:])
    s = capi.Solver(so, relax=relax, cycle=cycle, max_iter=3)
    x = np.zeros_like(b)
    h = s.solve(b, x)
    ml = oracle.ml_create(so, relax=relax, cycle=cycle)
    assert s.nlevels() == ml.nlevels()
    s.close()
    xo = np.zeros_like(b)
    ho = ml.solve(b, xo, maxiter=3, tol=1e-8)
    ml.close()
    assert len(h) == len(ho)
    np.testing.assert_allclose(h, ho, rtol=1e-8, atol=1e-13)
    assert np.max(np.abs(x - xo)) <= 1e-9 * max(np.max(np.abs(xo)), 1e-300)
