"""Deterministic synthetic inputs shared by the golden generator, the tests and
bench.py.  Nothing here depends on a library RNG: pseudo-random fields come
from a splitmix64 counter hash evaluated in uint64 arithmetic, so the same
bits come out on every machine and numpy version.

Array convention: C-ordered float64, reversed Fortran shape, ghosts included:
2D stencil (nst, JJ, II); 3D stencil (nst, KK, JJ, II).

Gallery operators restate the reference's generators:
  poisson2 / diag_diffusion2 / fe2 : src/2d/gallery.cc:7-113
  poisson3 / diag_diffusion3 / fe3 : src/3d/gallery.cc:7-190
(C++ index i there is the 0-based index including the ghost, identical to the
last numpy axis used here.)
"""
import numpy as np

# 2D slots
KO, KW, KS, KSW, KNW = range(5)
# 3D slots
KP, KPW, KPS, KB, KPSW, KPNW, KBW, KBNW, KBN, KBNE, KBE, KBSE, KBS, KBSW = range(14)


def splitmix64(idx, seed):
    """uniform [0,1) doubles from integer counters (vectorised, exact)."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(shape, seed, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * splitmix64(np.arange(n, dtype=np.uint64), seed)).reshape(shape)


def interior_mask(shape):
    m = np.zeros(shape, dtype=bool)
    m[tuple(slice(1, -1) for _ in shape)] = True
    return m


# ---------------------------------------------------------------- gallery 2D
def poisson2(nx, ny):
    return diag_diffusion2(nx, ny, 1.0, 1.0)


def diag_diffusion2(nx, ny, dx, dy):
    so = np.zeros((3, ny + 2, nx + 2))
    hx, hy = 1.0 / (nx + 1), 1.0 / (ny + 1)
    xh, yh = hy / hx, hx / hy
    so[KS, 2:ny + 1, 1:nx + 1] = dy * yh
    so[KW, 1:ny + 1, 2:nx + 1] = dx * xh
    so[KO, 1:ny + 1, 1:nx + 1] = 2 * dx * xh + 2 * dy * yh
    return so


def fe2(nx, ny):
    so = np.zeros((5, ny + 2, nx + 2))
    so[KS, 2:ny + 1, 1:nx + 1] = 1.0
    so[KW, 1:ny + 1, 2:nx + 1] = 1.0
    so[KSW, 2:ny + 1, 2:nx + 1] = 1.0
    so[KNW, 2:ny + 1, 2:nx + 1] = 1.0
    so[KO, 1:ny + 1, 1:nx + 1] = 8.0
    return so


def rhs2(nx, ny):
    """examples/basic-2d-ser/poisson.cc:15-37"""
    hx, hy = 1.0 / (nx + 1), 1.0 / (ny + 1)
    b = np.zeros((ny + 2, nx + 2))
    i = np.arange(1, nx + 1)
    j = np.arange(1, ny + 1)
    x, y = (i * hx)[None, :], (j * hy)[:, None]
    b[1:-1, 1:-1] = 8 * (np.pi * np.pi) * np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y) * (hx * hy)
    return b


def exact2(nx, ny):
    hx, hy = 1.0 / (nx + 1), 1.0 / (ny + 1)
    x = (np.arange(nx + 2) * hx)[None, :]
    y = (np.arange(ny + 2) * hy)[:, None]
    return np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y)


def varcoef9(nx, ny, seed=12345, sigma=1.0):
    """9-pt bilinear-FE discretisation of -div(a grad u) on a uniform square
    grid, a = exp(sigma*g) piecewise constant per cell, g ~ U(-1,1) from the
    counter hash.  Cells are indexed by their upper
    right node (i,j), i in 1..nx+1, j in 1..ny+1 (0-based incl. ghost).
    Element matrix of the unit square for a=1: diag 2/3, edge -1/6, diagonal -1/3.
    Stored in Cedar's layout: positive off-diagonals, symmetric half stencil;
    couplings to Dirichlet boundary nodes are dropped like gallery::poisson does."""
    g = uniform((ny + 2, nx + 2), seed, -1.0, 1.0)
    a = np.exp(sigma * g)          # a[j,i] = cell whose upper-right node is (i,j)
    a[0, :] = 0.0
    a[:, 0] = 0.0
    so = np.zeros((5, ny + 2, nx + 2))
    A = lambda dj, di: np.roll(np.roll(a, -dj, axis=0), -di, axis=1)  # a[j+dj, i+di]
    # node (i,j) touches cells (i,j), (i+1,j), (i,j+1), (i+1,j+1)
    so[KO] = (2.0 / 3.0) * (a + A(0, 1) + A(1, 0) + A(1, 1))
    so[KW] = (1.0 / 6.0) * (a + A(1, 0))          # edge (i-1,j)-(i,j): cells (i,j),(i,j+1)
    so[KS] = (1.0 / 6.0) * (a + A(0, 1))          # edge (i,j-1)-(i,j): cells (i,j),(i+1,j)
    so[KSW] = (1.0 / 3.0) * a                     # diagonal (i-1,j-1)-(i,j): cell (i,j)
    so[KNW] = (1.0 / 3.0) * a                     # KNW at (i,j): (i,j-1)-(i-1,j): cell (i,j)
    m = interior_mask((ny + 2, nx + 2))
    so[KO] *= m
    mw = m & np.roll(m, 1, axis=1)
    ms = m & np.roll(m, 1, axis=0)
    so[KW] *= mw
    so[KS] *= ms
    so[KSW] *= m & np.roll(np.roll(m, 1, axis=0), 1, axis=1)
    so[KNW] *= np.roll(m, 1, axis=0) & np.roll(m, 1, axis=1)
    return so


def aniso9(nx, ny, eps=1e-4, cross=0.05):
    """9-pt anisotropic operator: -(dx u_xx + dy u_yy)
    with dx = eps on the left half, dy = eps on the right half, plus a small
    symmetric cross coupling so that all nine points are populated."""
    hx, hy = 1.0 / (nx + 1), 1.0 / (ny + 1)
    xh, yh = hy / hx, hx / hy
    i = np.arange(nx + 2)[None, :]
    left = (i <= (nx + 1) // 2)
    dxc = np.where(left, eps, 1.0) * np.ones((ny + 2, 1))
    dyc = np.where(left, 1.0, eps) * np.ones((ny + 2, 1))
    so = np.zeros((5, ny + 2, nx + 2))
    m = interior_mask((ny + 2, nx + 2))
    cw = dxc * xh
    cs = dyc * yh
    cd = cross * np.minimum(cw, cs)
    so[KW] = cw * (m & np.roll(m, 1, axis=1))
    so[KS] = cs * (m & np.roll(m, 1, axis=0))
    so[KSW] = cd * (m & np.roll(np.roll(m, 1, axis=0), 1, axis=1))
    so[KNW] = cd * (np.roll(m, 1, axis=0) & np.roll(m, 1, axis=1))
    E = lambda a: np.roll(a, -1, axis=1)
    N = lambda a: np.roll(a, -1, axis=0)
    # diagonal = sum of the eight couplings of the untruncated operator
    # (Dirichlet: couplings to boundary nodes still count in the diagonal)
    so[KO] = (cw + E(cw) + cs + N(cs) + cd + E(N(cd)) + E(cd) + N(cd)) * m
    return so


# ---------------------------------------------------------------- gallery 3D
def poisson3(nx, ny, nz):
    return diag_diffusion3(nx, ny, nz, 1.0, 1.0, 1.0)


def diag_diffusion3(nx, ny, nz, dx, dy, dz):
    so = np.zeros((4, nz + 2, ny + 2, nx + 2))
    hx, hy, hz = 1.0 / (nx + 1), 1.0 / (ny + 1), 1.0 / (nz + 1)
    xh, yh, zh = hy * hz / hx, hx * hz / hy, hx * hy / hz
    so[KPS, 1:nz + 1, 2:ny + 1, 1:nx + 1] = dy * yh
    so[KPW, 1:nz + 1, 1:ny + 1, 2:nx + 1] = dx * xh
    so[KB, 2:nz + 1, 1:ny + 1, 1:nx + 1] = dz * zh
    so[KP, 1:nz + 1, 1:ny + 1, 1:nx + 1] = 2.0 * dx * xh + 2.0 * dy * yh + 2.0 * dz * zh
    return so


def fe3(nx, ny, nz):
    so = np.zeros((14, nz + 2, ny + 2, nx + 2))
    K1, J1, I1 = slice(1, nz + 1), slice(1, ny + 1), slice(1, nx + 1)
    K2, J2, I2 = slice(2, nz + 1), slice(2, ny + 1), slice(2, nx + 1)
    so[KPW, K1, J1, I2] = 1.0
    so[KPS, K1, J2, I1] = 1.0
    so[KB, K2, J1, I1] = 1.0
    so[KPNW, K1, J2, I2] = 1.0
    so[KPSW, K1, J2, I2] = 1.0
    so[KBW, K2, J1, I2] = 1.0
    so[KBE, K2, J1, I2] = 1.0
    so[KBN, K2, J2, I1] = 1.0
    so[KBS, K2, J2, I1] = 1.0
    for s in (KBNW, KBNE, KBSE, KBSW):
        so[s, K2, J2, I2] = 1.0
    so[KP, K1, J1, I1] = 26.0
    return so


def rhs3(nx, ny, nz):
    """examples/basic-3d-ser/poisson.cc:12-39: 12 pi^2 sin sin sin * h^3"""
    hx, hy, hz = 1.0 / (nx + 1), 1.0 / (ny + 1), 1.0 / (nz + 1)
    b = np.zeros((nz + 2, ny + 2, nx + 2))
    x = (np.arange(1, nx + 1) * hx)[None, None, :]
    y = (np.arange(1, ny + 1) * hy)[None, :, None]
    z = (np.arange(1, nz + 1) * hz)[:, None, None]
    b[1:-1, 1:-1, 1:-1] = (12 * (np.pi * np.pi) * np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y)
                           * np.sin(2 * np.pi * z) * (hx * hy * hz))
    return b


# ---------------------------------------------------------------- random ops
def random_op(shape_g, nst, seed, zero_ghost=True):
    """positive off-diagonals in [0.5,1.5], diagonal in [4 nst, 6 nst]"""
    so = uniform((nst,) + tuple(shape_g), seed, 0.5, 1.5)
    so[0] = uniform(shape_g, seed + 1000, 2.0, 3.0) * (2 * nst)
    if zero_ghost:
        so *= interior_mask(shape_g)
    return so


def coarse_shape(shape_g):
    return tuple(int((n - 2 - 1) / 2.0 + 1) + 2 for n in shape_g)


# --------------------------------------------------------------------------
# periodic problems
# --------------------------------------------------------------------------
def ibc_of(per):
    """(periodic_x, periodic_y) -> ibc as BMG_get_bc maps the mask (src/2d/ftn/BMG_get_bc.f90:13-16 with
    include/cedar/2d/ftn/BMG_parameters_c.h:193-196): x -> 2, y -> 1, both -> 3"""
    return {(False, False): 0, (True, False): 2, (False, True): 1, (True, True): 3}[(bool(per[0]), bool(per[1]))]


def periodic_poisson2(nx, ny, per):
    """five-point operator of examples/basic-2d-ser/periodic.cc:17-84 (create_op): mesh widths from
    nx-1 / ny-1 in a periodic direction, W / S entries also on the first row / column there, ghost
    columns / rows filled with the periodic image"""
    so = np.zeros((3, ny + 2, nx + 2))
    mx, my = nx - (1 if per[0] else 0), ny - (1 if per[1] else 0)
    hx, hy = 1.0 / (mx + 1), 1.0 / (my + 1)
    xh, yh = hy / hx, hx / hy
    ibeg, jbeg = (1 if per[0] else 2), (1 if per[1] else 2)
    so[2, jbeg:ny + 1, 1:nx + 1] = 1.0 * yh
    so[1, 1:ny + 1, ibeg:nx + 1] = 1.0 * xh
    so[0, 1:ny + 1, 1:nx + 1] = 2 * xh + 2 * yh
    if per[0]:
        so[:, 1:ny + 1, ibeg - 1] = so[:, 1:ny + 1, nx]
        so[:, 1:ny + 1, nx + 1] = so[:, 1:ny + 1, ibeg]
    if per[1]:
        so[:, jbeg - 1, 1:nx + 1] = so[:, ny, 1:nx + 1]
        so[:, ny + 1, 1:nx + 1] = so[:, jbeg, 1:nx + 1]
    return so


def periodic_rhs2(nx, ny, per):
    """set_problem of the same example (:87-125)"""
    mx, my = nx - (1 if per[0] else 0), ny - (1 if per[1] else 0)
    hx, hy = 1.0 / (mx + 1), 1.0 / (my + 1)
    h2 = hx * hy
    b = np.zeros((ny + 2, nx + 2))
    i = np.arange(1, nx + 1)[None, :]
    j = np.arange(1, ny + 1)[:, None]
    b[1:ny + 1, 1:nx + 1] = 8 * (np.pi * np.pi) * np.sin(2 * np.pi * (i * hx)) * np.sin(2 * np.pi * (j * hy)) * h2
    if per[0]:
        b[:, 0] = b[:, nx]
        b[:, nx + 1] = b[:, 1]
    if per[1]:
        b[0, :] = b[ny, :]
        b[ny + 1, :] = b[1, :]
    return b


def periodic_random_op(nx, ny, nst, per, seed):
    """strictly diagonally dominant random operator (definite also when both directions wrap) whose
    ghost layers carry the periodic image in the wrapped directions and zeros on Dirichlet sides"""
    g = (ny + 2, nx + 2)
    so = random_op(g, nst, seed)
    if per[0]:
        so[:, :, 0] = so[:, :, nx]
        so[:, :, nx + 1] = so[:, :, 1]
    if per[1]:
        so[:, 0, :] = so[:, ny, :]
        so[:, ny + 1, :] = so[:, 1, :]
    return so


def ibc3_of(per):
    """(periodic_x, periodic_y, periodic_z) -> ibc of BMG_get_bc (src/2d/ftn/BMG_get_bc.f90:13-20 with
    src/3d/ftn/BMG_parameters_f90.h:345-356)"""
    return {(0, 0, 0): 0, (1, 0, 0): 2, (0, 1, 0): 1, (1, 1, 0): 3, (0, 0, 1): 5, (1, 0, 1): 6, (0, 1, 1): 7,
            (1, 1, 1): 8}[tuple(int(bool(p)) for p in per)]


def per3_of(ibc):
    return {0: (0, 0, 0), 2: (1, 0, 0), 1: (0, 1, 0), 3: (1, 1, 0), 5: (0, 0, 1), 6: (1, 0, 1), 7: (0, 1, 1),
            8: (1, 1, 1)}[ibc]


def wrap3(a, per):
    """periodic image into the ghost layers of a[..., k, j, i] (y, then x, then z, each over the full range of
    the other two)"""
    if per[1]:
        a[..., :, 0, :] = a[..., :, -2, :]
        a[..., :, -1, :] = a[..., :, 1, :]
    if per[0]:
        a[..., :, :, 0] = a[..., :, :, -2]
        a[..., :, :, -1] = a[..., :, :, 1]
    if per[2]:
        a[..., 0, :, :] = a[..., -2, :, :]
        a[..., -1, :, :] = a[..., 1, :, :]
    return a


def periodic_poisson3(nx, ny, nz, per):
    """seven-point operator of examples/basic-3d-ser/periodic.cc:15-125 (create_op): mesh widths from n-1 in a
    periodic direction, W / S / B entries also on the first column / row / plane there, ghost layers filled with
    the periodic image (x, then y, then z)"""
    so = np.zeros((4, nz + 2, ny + 2, nx + 2))
    mx, my, mz = nx - (1 if per[0] else 0), ny - (1 if per[1] else 0), nz - (1 if per[2] else 0)
    hx, hy, hz = 1.0 / (mx + 1), 1.0 / (my + 1), 1.0 / (mz + 1)
    xh, yh, zh = hy * hz / hx, hx * hz / hy, hx * hy / hz
    ibeg, jbeg, kbeg = (1 if per[0] else 2), (1 if per[1] else 2), (1 if per[2] else 2)
    so[2, 1:nz + 1, jbeg:ny + 1, 1:nx + 1] = 1.0 * yh
    so[1, 1:nz + 1, 1:ny + 1, ibeg:nx + 1] = 1.0 * xh
    so[3, kbeg:nz + 1, 1:ny + 1, 1:nx + 1] = 1.0 * zh
    so[0, 1:nz + 1, 1:ny + 1, 1:nx + 1] = 2.0 * xh + 2.0 * yh + 2.0 * zh
    if per[0]:
        so[:, :, :, 0] = so[:, :, :, nx]
        so[:, :, :, nx + 1] = so[:, :, :, 1]
    if per[1]:
        so[:, :, 0, :] = so[:, :, ny, :]
        so[:, :, ny + 1, :] = so[:, :, 1, :]
    if per[2]:
        so[:, 0, :, :] = so[:, nz, :, :]
        so[:, nz + 1, :, :] = so[:, 1, :, :]
    return so


def periodic_rhs3(nx, ny, nz, per):
    """set_problem of the same example (:128-193)"""
    mx, my, mz = nx - (1 if per[0] else 0), ny - (1 if per[1] else 0), nz - (1 if per[2] else 0)
    hx, hy, hz = 1.0 / (mx + 1), 1.0 / (my + 1), 1.0 / (mz + 1)
    h2 = hx * hy * hz
    b = np.zeros((nz + 2, ny + 2, nx + 2))
    i = np.arange(1, nx + 1)[None, None, :]
    j = np.arange(1, ny + 1)[None, :, None]
    k = np.arange(1, nz + 1)[:, None, None]
    b[1:-1, 1:-1, 1:-1] = (12 * (np.pi * np.pi) * np.sin(2 * np.pi * (i * hx)) * np.sin(2 * np.pi * (j * hy))
                           * np.sin(2 * np.pi * (k * hz))) * h2
    if per[0]:
        b[:, :, 0] = b[:, :, nx]
        b[:, :, nx + 1] = b[:, :, 1]
    if per[1]:
        b[:, 0, :] = b[:, ny, :]
        b[:, ny + 1, :] = b[:, 1, :]
    if per[2]:
        b[0, :, :] = b[nz, :, :]
        b[nz + 1, :, :] = b[1, :, :]
    return b


def periodic_random_op3(nx, ny, nz, nst, per, seed):
    """strictly diagonally dominant random operator with the periodic image in the ghost layers of the wrapped
    directions and zeros on Dirichlet sides"""
    so = random_op((nz + 2, ny + 2, nx + 2), nst, seed)
    return wrap3(so, per)
