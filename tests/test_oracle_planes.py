"""Plane relaxation: the oracle's restatement of include/cedar/3d/relax_planes.h + src/3d/relax_planes.cc.

1. The reference's own known-answer test (test/3d/test_planes.cc with test-planes-ser.json: point relaxation in the
   planes, 50 iterations, tol 1e-30, x = b = running index, one DOWN sweep): every even plane must then hold the exact
   solution of its 2D problem with the right-hand side copy_rhs gives -- the reference checks that against
   scipy/pyamg (test/3d/pyplanes.pyx) to 1e-8; restated here with scipy.
2. The property that pins the reference's copy_coeff behaviour (every plane solver built from the LAST plane's
   coefficients, relax_planes.h:80-160) on an operator whose coefficients vary from plane to plane.
3. Whole solves with plane relaxation as the smoother converge (anisotropic problems point relaxation cannot handle)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as sla

import cases
import problems as pb

DOWN, UP = 0, 1
KAT_PLANE = dict(relax="point", nrelax_pre=2, nrelax_post=1, max_iter=50, min_coarse=3, tol=1e-30)
# (direction, nx, ny, nz) of test/3d/test_planes.cc:17-19, :101-103, :185-187
KAT = [("xy", 31, 35, 5), ("xz", 31, 5, 35), ("yz", 5, 31, 35)]


def index_field(nx, ny, nz):
    """x(i,j,k) = k*nx*ny + j*nx + i on the interior (test_planes.cc:30-37)"""
    v = np.zeros((nz + 2, ny + 2, nx + 2))
    k, j, i = np.meshgrid(np.arange(1, nz + 1), np.arange(1, ny + 1), np.arange(1, nx + 1), indexing="ij")
    v[1:-1, 1:-1, 1:-1] = k * nx * ny + j * nx + i
    return v


def plane_view(a, d, ipl):
    return a[ipl] if d == "xy" else a[:, ipl] if d == "xz" else a[:, :, ipl]


def plane_matrix(so, d, ipl):
    """the 2D operator of plane ipl as a sparse matrix: diagonal = 3D diagonal, in-plane couplings only"""
    if d == "xy":
        c = {"c": so[0, ipl], "w": so[1, ipl], "s": so[2, ipl]}
        if so.shape[0] == 14:
            c.update(sw=so[4, ipl], nw=so[5, ipl])
    elif d == "xz":
        c = {"c": so[0, :, ipl], "w": so[1, :, ipl], "s": so[3, :, ipl]}
        if so.shape[0] == 14:
            c.update(sw=so[6, :, ipl], nw=so[10, :, ipl])
    else:
        c = {"c": so[0, :, :, ipl], "w": so[2, :, :, ipl], "s": so[3, :, :, ipl]}
        if so.shape[0] == 14:
            c.update(sw=so[12, :, :, ipl], nw=so[8, :, :, ipl])
    n2, n1 = c["c"].shape[0] - 2, c["c"].shape[1] - 2
    idx = lambda a, b: (b - 1) * n1 + (a - 1)
    A = sp.lil_matrix((n1 * n2, n1 * n2))
    for b in range(1, n2 + 1):
        for a in range(1, n1 + 1):
            A[idx(a, b), idx(a, b)] = c["c"][b, a]
            for key, (da, db), (ea, eb) in (("w", (0, 0), (-1, 0)), ("s", (0, 0), (0, -1)), ("sw", (0, 0), (-1, -1)),
                                            ("nw", (0, -1), (-1, 0))):
                if key not in c:
                    continue
                p, q = (a + da, b + db), (a + ea, b + eb)
                if min(p + q) < 1 or p[0] > n1 or q[0] > n1 or p[1] > n2 or q[1] > n2:
                    continue
                A[idx(*p), idx(*q)] = A[idx(*q), idx(*p)] = -c[key][b, a]
    return A.tocsr()


def kat_check(impl, oracle, d, nx, ny, nz, nst):
    so = pb.poisson3(nx, ny, nz) if nst == 4 else pb.fe3(nx, ny, nz)
    x, b = index_field(nx, ny, nz), index_field(nx, ny, nz)
    impl.relax_planes3(so, x, b, d, DOWN, plane=KAT_PLANE)
    n_normal = {"xy": nz, "xz": ny, "yz": nx}[d]
    shape2 = plane_view(x, d, 1).shape
    for ipl in range(2, n_normal + 1, 2):
        b2 = np.zeros(shape2)
        oracle.plane_rhs3(so, x, b, b2, d, ipl)
        exact = sla.spsolve(plane_matrix(so, d, ipl), b2[1:-1, 1:-1].ravel()).reshape(b2[1:-1, 1:-1].shape)
        assert np.max(np.abs(plane_view(x, d, ipl)[1:-1, 1:-1] - exact)) < 1e-8, (d, nst, ipl)


@pytest.mark.parametrize("d,nx,ny,nz", KAT, ids=lambda v: str(v))
@pytest.mark.parametrize("nst", [4, 14])
def test_reference_known_answer_planes(oracle, d, nx, ny, nz, nst):
    kat_check(oracle, oracle, d, nx, ny, nz, nst)


def varying_op(nx, ny, nz, nst, seed):
    so = pb.random_op((nz + 2, ny + 2, nx + 2), nst, seed)
    so[0] *= 1.5
    return so


@pytest.mark.parametrize("d", ["xy", "xz", "yz"])
@pytest.mark.parametrize("nst", [4, 14])
def test_every_plane_solver_uses_the_last_planes_coefficients(oracle, d, nst):
    """relax_planes.h:80-160: with plane-dependent coefficients the sweep equals the one on an operator whose in-plane
    coefficients were overwritten by the last plane's in every plane -- as far as the 2D solvers are concerned; the
    off-plane couplings of the right-hand side keep their own values"""
    nx, ny, nz = 9, 8, 7
    so = varying_op(nx, ny, nz, nst, 21)
    x0, b = pb.uniform(so.shape[1:], 22, -1, 1), pb.uniform(so.shape[1:], 23, -1, 1)
    x1 = x0.copy()
    oracle.relax_planes3(so, x1, b, d, DOWN)
    # by hand: red-black over the planes with ONE 2D solver built from the last plane
    inplane = {"xy": ((0, 1, 2, 4, 5) if nst == 14 else (0, 1, 2)), "xz": ((0, 1, 3, 6, 10) if nst == 14 else (0, 1, 3)),
               "yz": ((0, 2, 3, 12, 8) if nst == 14 else (0, 2, 3))}[d]
    n_normal = {"xy": nz, "xz": ny, "yz": nx}[d]
    so2 = np.ascontiguousarray(np.stack([plane_view(so[s], d, n_normal) for s in inplane]))
    ml2 = oracle.ml_create(so2, relax="line-xy", nrelax_pre=2, nrelax_post=1)
    x2 = x0.copy()
    for beg in (1, 2):
        for ipl in range(beg, n_normal + 1, 2):
            v = np.ascontiguousarray(plane_view(x2, d, ipl))
            b2 = np.zeros_like(v)
            oracle.plane_rhs3(so, x2, b, b2, d, ipl)
            ml2.solve(b2, v, maxiter=1, tol=1e-8)
            plane_view(x2, d, ipl)[...] = v
    ml2.close()
    assert np.array_equal(x1, x2)


ANISO = {
    # strong coupling in x and y: xy planes; in all directions alternately: xyz
    "aniso7_xy_24x20x17": (lambda: pb.diag_diffusion3(24, 20, 17, 1.0, 1.0, 1e-3), "plane-xy"),
    "aniso7_xz_17x12x21": (lambda: pb.diag_diffusion3(17, 12, 21, 1.0, 1e-3, 1.0), "plane-xz"),
    "aniso7_yz_12x17x20": (lambda: pb.diag_diffusion3(12, 17, 20, 1e-3, 1.0, 1.0), "plane-yz"),
    "fe27_xyz_16x15x14": (lambda: pb.fe3(16, 15, 14), "plane-xyz"),
}


@pytest.mark.parametrize("name", list(ANISO), ids=str)
def test_plane_relaxation_as_the_smoother(oracle, name):
    mk, relax = ANISO[name]
    so = mk()
    nz, ny, nx = (n - 2 for n in so.shape[1:])
    b = pb.rhs3(nx, ny, nz)
    ml = oracle.ml_create(so, relax=relax, nrelax_pre=2, nrelax_post=1)
    x = np.zeros_like(b)
    h = ml.solve(b, x, maxiter=10, tol=1e-8)
    ml.close()
    assert h[-1] < 1e-8 and len(h) <= 9, h
    if name.startswith("aniso7"):
        mp = oracle.ml_create(so, relax="point", nrelax_pre=2, nrelax_post=1)
        xp = np.zeros_like(b)
        hp = mp.solve(b, xp, maxiter=10, tol=1e-8)
        mp.close()
        assert hp[-1] > 100 * h[-1], (hp, h)  # point relaxation stalls on the anisotropy, planes do not
