"""Kernel-level and solve-level parity cases, written once and run against any
implementation object that exposes the method names of oracle/pyoracle.py
(`Ref` = the reference's Fortran, `Oracle` = the C restatement, and the
product's ctypes front-end `cedar_amd.capi.Kernels`).

`kernel_suite(impl, case)` returns {name: ndarray}; the golden files hold what
`Ref` returned (oracle/gen_golden.py), the tests compare the others to it.
"""
import numpy as np

import problems as pb

DOWN, UP = 0, 1

CASES_2D = [
    # (name, nx, ny, nst)
    ("r9x9_5", 9, 9, 3), ("r9x9_9", 9, 9, 5),
    ("r16x12_5", 16, 12, 3), ("r16x12_9", 16, 12, 5),
    ("r31x31_9", 31, 31, 5), ("r37x37_5", 37, 37, 3),
    ("r64x48_9", 64, 48, 5), ("r7x10_9", 7, 10, 5),
]
CASES_3D = [
    ("r9x9x9_7", 9, 9, 9, 4), ("r9x9x9_27", 9, 9, 9, 14),
    ("r8x6x10_7", 8, 6, 10, 4), ("r8x6x10_27", 8, 6, 10, 14),
    ("r12x7x9_27", 12, 7, 9, 14), ("r13x13x13_7", 13, 13, 13, 4),
    ("r13x13x13_27", 13, 13, 13, 14),
]


def _seed(name):
    return sum((i + 1) * ord(c) for i, c in enumerate(name)) % 100000


def kernel_suite_2d(impl, case):
    name, nx, ny, nst = case
    sd = _seed(name)
    g = (ny + 2, nx + 2)
    gc = pb.coarse_shape(g)
    so = pb.random_op(g, nst, sd)
    qf = pb.uniform(g, sd + 1, -1, 1)
    q0 = pb.uniform(g, sd + 2, -1, 1)
    out = {}
    sor = np.zeros((2,) + g)
    impl.setup_recip2(so, sor)
    out["recip"] = sor.copy()
    for ud in (DOWN, UP):
        q = q0.copy()
        impl.relax2(so, qf, q, sor, ud)
        impl.relax2(so, qf, q, sor, ud)
        out[f"relax{ud}"] = q
    r = np.zeros(g)
    impl.residual2(so, qf, q0, r)
    out["residual"] = r
    for d in "xy":
        sl = np.zeros((2,) + g)
        impl.setup_lines2(so, sl, d)
        out[f"setup_lines_{d}"] = sl.copy()
        for ud in (DOWN, UP):
            q = q0.copy()
            impl.relax_lines2(so, qf, q, sl, ud, d)
            out[f"relax_lines_{d}{ud}"] = q
    ci = np.zeros((8,) + gc)
    impl.setup_interp2(so, ci)
    out["interp"] = ci.copy()
    soc = np.zeros((5,) + gc)
    impl.galerkin2(so, soc, ci)
    out["galerkin"] = soc
    qc = np.zeros(gc)
    impl.restrict2(q0, qc, ci)
    out["restrict"] = qc
    qcx = pb.uniform(gc, sd + 3, -1, 1) * pb.interior_mask(gc)
    q, res = q0.copy(), qf.copy()
    impl.interp_add2(q, qcx, res, so, ci)
    out["interp_add_q"], out["interp_add_res"] = q, res
    if nx * ny <= 200:
        abd = np.zeros((nx * ny, nx + 2))
        impl.setup_cg2(so, abd)
        out["abd"] = abd.copy()
        x = np.zeros(g)
        impl.solve_cg2(x, qf, abd)
        out["solve_cg"] = x
    return out


def kernel_suite_3d(impl, case):
    name, nx, ny, nz, nst = case
    sd = _seed(name)
    g = (nz + 2, ny + 2, nx + 2)
    gc = pb.coarse_shape(g)
    so = pb.random_op(g, nst, sd)
    qf = pb.uniform(g, sd + 1, -1, 1)
    q0 = pb.uniform(g, sd + 2, -1, 1)
    out = {}
    sor = np.zeros((2,) + g)
    impl.setup_recip3(so, sor)
    out["recip"] = sor.copy()
    for ud in (DOWN, UP):
        q = q0.copy()
        impl.relax3(so, qf, q, sor, ud)
        impl.relax3(so, qf, q, sor, ud)
        out[f"relax{ud}"] = q
    r = np.zeros(g)
    impl.residual3(so, qf, q0, r)
    out["residual"] = r
    ci = np.zeros((26,) + gc)
    impl.setup_interp3(so, ci)
    out["interp"] = ci.copy()
    soc = np.zeros((14,) + gc)
    impl.galerkin3(so, soc, ci)
    out["galerkin"] = soc
    qc = np.zeros(gc)
    impl.restrict3(q0, qc, ci)
    out["restrict"] = qc
    qcx = pb.uniform(gc, sd + 3, -1, 1) * pb.interior_mask(gc)
    q, res = q0.copy(), qf.copy()
    impl.interp_add3(q, qcx, so, res, ci)
    out["interp_add_q"], out["interp_add_res"] = q, res
    if nx * ny * nz <= 500:
        abd = np.zeros((nx * ny * nz, nx * (ny + 1) + 2))
        impl.setup_cg3(so, abd)
        out["abd"] = abd.copy()
        x = np.zeros(g)
        impl.solve_cg3(x, qf, abd)
        out["solve_cg"] = x
    return out


# --------------------------------------------------------------------------
# Sweeps modelled on the reference's own relax tests
#   test/2d/test_relax.cc:14-54  Point5: 31^2 poisson, x=1, b=0, 7 DOWN + 7 UP
#   test/2d/test_relax.cc:57-97  Point9: 37^2 gallery::fe, 3 + 3
#   test/3d/mpi/test_relax.cc:11-60 : 50^3 7-pt, 5 x (DOWN, UP)  (here 20^3 + 27-pt too)
# --------------------------------------------------------------------------
def sweep_suite(impl):
    out = {}
    for nm, so, n in (("point5", pb.poisson2(31, 31), 7), ("point9", pb.fe2(37, 37), 3)):
        g = so.shape[1:]
        sor = np.zeros((2,) + g)
        impl.setup_recip2(so, sor)
        x, b = np.ones(g), np.zeros(g)
        for _ in range(n):
            impl.relax2(so, b, x, sor, DOWN)
        for _ in range(n):
            impl.relax2(so, b, x, sor, UP)
        out[nm] = x
    for nm, so in (("point7", pb.poisson3(20, 20, 20)), ("point27", pb.fe3(20, 18, 16))):
        g = so.shape[1:]
        sor = np.zeros((2,) + g)
        impl.setup_recip3(so, sor)
        x, b = np.ones(g), np.zeros(g)
        for _ in range(5):
            impl.relax3(so, b, x, sor, DOWN)
            impl.relax3(so, b, x, sor, UP)
        out[nm] = x
    return out


# --------------------------------------------------------------------------
# Full solves: residual-norm histories (multilevel.h:268-298)
# --------------------------------------------------------------------------
SOLVES = {
    # name: (operator builder, rhs builder, settings)
    "poisson5_400_v11": (lambda: pb.poisson2(400, 400), lambda: pb.rhs2(400, 400),
                         dict(relax="point", nrelax_pre=1, nrelax_post=1)),
    "poisson5_512_v21": (lambda: pb.poisson2(512, 512), lambda: pb.rhs2(512, 512),
                         dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "varcoef9_513_v21": (lambda: pb.varcoef9(513, 513), lambda: pb.rhs2(513, 513),
                         dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "varcoef9_200x120_v21": (lambda: pb.varcoef9(200, 120), lambda: pb.rhs2(200, 120),
                             dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "aniso9_512_linexy": (lambda: pb.aniso9(512, 512), lambda: pb.rhs2(512, 512),
                          dict(relax="line-xy", nrelax_pre=2, nrelax_post=1)),
    "stretch5_800x200_linex": (lambda: pb.diag_diffusion2(800, 200, 1.0, 1e-2), lambda: pb.rhs2(800, 200),
                               dict(relax="line-x", nrelax_pre=2, nrelax_post=1)),
    "stretch5_200x800_liney": (lambda: pb.diag_diffusion2(200, 800, 1e-2, 1.0), lambda: pb.rhs2(200, 800),
                               dict(relax="line-y", nrelax_pre=2, nrelax_post=1)),
    "fe27_65_v21": (lambda: pb.fe3(65, 65, 65), lambda: pb.rhs3(65, 65, 65),
                    dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "fe27_40x33x50_v21": (lambda: pb.fe3(40, 33, 50), lambda: pb.rhs3(40, 33, 50),
                          dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "poisson7_65_v21": (lambda: pb.poisson3(65, 65, 65), lambda: pb.rhs3(65, 65, 65),
                        dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "poisson7_64_v21": (lambda: pb.poisson3(64, 64, 64), lambda: pb.rhs3(64, 64, 64),
                        dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    # mid sizes (SURVEY 8c list): between the small goldens above and the full-size property tests
    "fe27_129_v21": (lambda: pb.fe3(129, 129, 129), lambda: pb.rhs3(129, 129, 129),
                     dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "varcoef9_1024_v21": (lambda: pb.varcoef9(1024, 1024), lambda: pb.rhs2(1024, 1024),
                          dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "aniso9_1024_linexy": (lambda: pb.aniso9(1024, 1024), lambda: pb.rhs2(1024, 1024),
                           dict(relax="line-xy", nrelax_pre=2, nrelax_post=1)),
    # F-cycles (include/cedar/cycle/fcycle.h), SURVEY section 8f-3
    "varcoef9_200x120_f21": (lambda: pb.varcoef9(200, 120), lambda: pb.rhs2(200, 120),
                             dict(relax="point", nrelax_pre=2, nrelax_post=1, cycle="f")),
    "fe27_40x33x50_f21": (lambda: pb.fe3(40, 33, 50), lambda: pb.rhs3(40, 33, 50),
                          dict(relax="point", nrelax_pre=2, nrelax_post=1, cycle="f")),
}

# absolute floor (in units of ||r0||) below which residual histories of two
# correctly-rounded implementations may differ: eps * cond-ish.  Default 1e-14.
HIST_ATOL = {"aniso9_512_linexy": 1e-12, "stretch5_800x200_linex": 1e-12, "stretch5_200x800_liney": 1e-12,
             "aniso9_1024_linexy": 1e-12}


# --------------------------------------------------------------------------
# 2D periodic boundary conditions: kernels with random ghosts, and full solves
# --------------------------------------------------------------------------
CASES_PER = [
    # (name, nx, ny, nst, ibc)
    ("p9x9_5_x", 9, 9, 3, 2), ("p9x9_9_y", 9, 9, 5, 1), ("p16x16_9_xy", 16, 16, 5, 3),
    ("p17x12_9_x", 17, 12, 5, 2), ("p12x17_5_xy", 12, 17, 3, 3), ("p31x20_9_y", 31, 20, 5, 1),
    ("p8x8_5_y", 8, 8, 3, 1), ("p6x7_9_xy", 6, 7, 5, 3), ("p64x48_9_x", 64, 48, 5, 2),
]


def kernel_suite_per(impl, case):
    name, nx, ny, nst, ibc = case
    sd = _seed(name)
    g = (ny + 2, nx + 2)
    gc = pb.coarse_shape(g)
    so = pb.random_op(g, nst, sd, zero_ghost=False)
    out = {}
    sor = np.zeros((2,) + g)
    impl.setup_recip2(so, sor)
    qf, q = pb.uniform(g, sd + 1, -1, 1), pb.uniform(g, sd + 2, -1, 1)
    for ud in (DOWN, UP):
        impl.relax2(so, qf, q, sor, ud, ibc=ibc)
        out[f"relax{ud}"] = q.copy()
    ci = np.zeros((8,) + gc)
    impl.setup_interp2(so, ci, ibc=ibc)
    out["interp"] = ci.copy()
    soc = np.zeros((5,) + gc)
    impl.galerkin2(so, soc, ci, ibc=ibc)
    out["galerkin"] = soc.copy()
    r, qc = pb.uniform(g, sd + 3, -1, 1), np.zeros(gc)
    impl.restrict2(r, qc, ci, ibc=ibc)
    out["restrict_qc"], out["restrict_q"] = qc.copy(), r.copy()
    x, xc, res = pb.uniform(g, sd + 4, -1, 1), pb.uniform(gc, sd + 5, -1, 1), pb.uniform(g, sd + 6, -1, 1)
    impl.interp_add2(x, xc, res, so, ci, ibc=ibc)
    out["interp_add_q"], out["interp_add_res"] = x.copy(), res.copy()
    # line relaxation: cyclic tridiagonals (Sherman-Morrison) in the wrapped direction
    q0 = pb.uniform(g, sd + 7, -1, 1)
    for d in "xy":
        sl = np.zeros((2,) + g)
        impl.setup_lines2(so, sl, d, ibc=ibc)
        out[f"setup_lines_{d}"] = sl.copy()
        q = q0.copy()
        for ud in (DOWN, UP):
            impl.relax_lines2(so, qf, q, sl, ud, d, ibc=ibc)
            out[f"relax_lines_{d}{ud}"] = q.copy()
    return out


CG_PER = [("c3x3_5_x", 3, 3, 3, 2), ("c4x3_9_y", 4, 3, 5, 1), ("c3x5_9_xy", 3, 5, 5, 3), ("c5x4_5_xy", 5, 4, 3, 3)]


def coarse_solve_per(impl, case):
    name, nx, ny, nst, ibc = case
    sd = _seed(name)
    g = (ny + 2, nx + 2)
    so = pb.random_op(g, nst, sd, zero_ghost=False)
    so[0] *= 4.0  # keep the wrapped matrix positive definite
    n = nx * ny
    abd = np.zeros((n, n))
    impl.setup_cg2(so, abd, ibc=ibc)
    q = pb.uniform(g, sd + 2, -1, 1)
    impl.solve_cg2(q, pb.uniform(g, sd + 1, -1, 1), abd, ibc=ibc)
    return {"abd_upper": abd.T[np.triu_indices(n)].copy(), "q": q}


SOLVES_PER = {
    # the reference's periodic example (examples/basic-2d-ser/periodic.cc with periodic-config.json:
    # V(1,1), point relaxation, periodic in x) at two sizes, and wrapped random operators
    "perpoisson5_x_300_v11": (lambda: pb.periodic_poisson2(300, 300, (True, False)),
                              lambda: pb.periodic_rhs2(300, 300, (True, False)),
                              dict(relax="point", nrelax_pre=1, nrelax_post=1, ibc=2)),
    "perpoisson5_y_128x96_v21": (lambda: pb.periodic_poisson2(128, 96, (False, True)),
                                 lambda: pb.periodic_rhs2(128, 96, (False, True)),
                                 dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=1)),
    "perrand9_xy_96x80_v21": (lambda: pb.periodic_random_op(96, 80, 5, (True, True), 5),
                              lambda: pb.periodic_rhs2(96, 80, (True, True)),
                              dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=3)),
    "perrand9_x_100x75_v21": (lambda: pb.periodic_random_op(100, 75, 5, (True, False), 6),
                              lambda: pb.periodic_rhs2(100, 75, (True, False)),
                              dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=2)),
    "perrand5_xy_64_v21": (lambda: pb.periodic_random_op(64, 64, 3, (True, True), 7),
                           lambda: pb.periodic_rhs2(64, 64, (True, True)),
                           dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=3)),
    # line relaxation on periodic problems
    "perpoisson5_x_200x60_linex": (lambda: pb.periodic_poisson2(200, 60, (True, False)),
                                   lambda: pb.periodic_rhs2(200, 60, (True, False)),
                                   dict(relax="line-x", nrelax_pre=2, nrelax_post=1, ibc=2)),
    "perpoisson5_y_60x200_liney": (lambda: pb.periodic_poisson2(60, 200, (False, True)),
                                   lambda: pb.periodic_rhs2(60, 200, (False, True)),
                                   dict(relax="line-y", nrelax_pre=2, nrelax_post=1, ibc=1)),
    "perrand9_xy_96x80_linexy": (lambda: pb.periodic_random_op(96, 80, 5, (True, True), 5),
                                 lambda: pb.periodic_rhs2(96, 80, (True, True)),
                                 dict(relax="line-xy", nrelax_pre=2, nrelax_post=1, ibc=3)),
    "perrand9_x_100x75_linexy": (lambda: pb.periodic_random_op(100, 75, 5, (True, False), 6),
                                 lambda: pb.periodic_rhs2(100, 75, (True, False)),
                                 dict(relax="line-xy", nrelax_pre=2, nrelax_post=1, ibc=2)),
    # periodic F-cycles (round 3): include/cedar/cycle/fcycle.h:49-83 with the periodic kernels
    "perrand9_xy_96x80_f21": (lambda: pb.periodic_random_op(96, 80, 5, (True, True), 5),
                              lambda: pb.periodic_rhs2(96, 80, (True, True)),
                              dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=3, cycle="f")),
    "perpoisson5_x_200x60_linex_f21": (lambda: pb.periodic_poisson2(200, 60, (True, False)),
                                       lambda: pb.periodic_rhs2(200, 60, (True, False)),
                                       dict(relax="line-x", nrelax_pre=2, nrelax_post=1, ibc=2, cycle="f")),
}


# --------------------------------------------------------------------------
# 3D periodic boundary conditions.  Even extents in the periodic directions, ny <= nz (the
# reference's periodic interpolation set-up mixes up the y and z extents otherwise, oracle/boxmg3_per.c).
# --------------------------------------------------------------------------
CASES_PER3 = [
    # (name, nx, ny, nz, nst, ibc)
    ("q8x10x12_27_x", 8, 10, 12, 14, 2), ("q8x10x12_27_y", 8, 10, 12, 14, 1), ("q8x10x12_27_z", 8, 10, 12, 14, 5),
    ("q6x4x8_27_xy", 6, 4, 8, 14, 3), ("q12x6x10_27_xz", 12, 6, 10, 14, 6), ("q8x10x12_27_yz", 8, 10, 12, 14, 7),
    ("q6x8x8_27_xyz", 6, 8, 8, 14, 8), ("q8x10x12_7_x", 8, 10, 12, 4, 2), ("q6x4x8_7_z", 6, 4, 8, 4, 5),
    ("q4x4x4_7_xyz", 4, 4, 4, 4, 8), ("q8x9x11_27_x", 8, 9, 11, 14, 2), ("q7x8x9_7_y", 7, 8, 9, 4, 1),
    ("q9x7x10_27_z", 9, 7, 10, 14, 5), ("q16x16x16_27_xyz", 16, 16, 16, 14, 8),
]


def kernel_suite_per3(impl, case, reference=False):
    """reference=True (golden generation): interp_add only for per_z, the one code for which the routine's ghost
    refresh is well defined by the source.  Restriction and Galerkin product use the implementation's own
    interpolation weights: they pin every weight that is ever read."""
    name, nx, ny, nz, nst, ibc = case
    sd = _seed(name)
    per = pb.per3_of(ibc)
    so = pb.periodic_random_op3(nx, ny, nz, nst, per, sd)
    g = so.shape[1:]
    gc = pb.coarse_shape(g)
    out = {}
    sor = np.zeros((2,) + g)
    impl.setup_recip3(so, sor)
    qf, q = pb.uniform(g, sd + 1, -1, 1), pb.uniform(g, sd + 2, -1, 1)
    for ud in (DOWN, UP):
        impl.relax3(so, qf, q, sor, ud, ibc=ibc)
        out[f"relax{ud}"] = q.copy()
    ci = np.zeros((26,) + gc)
    impl.setup_interp3(so, ci, ibc=ibc)
    out["interp_interior"] = ci[:, 1:-1, 1:-1, 1:-1].copy()
    soc = np.zeros((14,) + gc)
    impl.galerkin3(so, soc, ci, ibc=ibc)
    out["galerkin"] = soc.copy()
    r, qc = pb.uniform(g, sd + 3, -1, 1), np.zeros(gc)
    impl.restrict3(r, qc, ci, ibc=ibc)
    out["restrict_qc"], out["restrict_q"] = qc.copy(), r.copy()
    if ibc == 5 or not reference:
        x, xc, res = pb.uniform(g, sd + 4, -1, 1), pb.uniform(gc, sd + 5, -1, 1), pb.uniform(g, sd + 6, -1, 1)
        pb.wrap3(xc, per)
        impl.interp_add3(x, xc, so, res, ci, ibc=ibc)
        out["interp_add_q"], out["interp_add_res"] = x.copy(), res.copy()
    return out


# dense coarsest-grid solves: the codes and extents for which the reference assembles the periodic operator
# (per_x, per_y, per_z, per_yz; per_xy with nx = ny), oracle/boxmg3_per.c
CG_PER3 = [("d3x3x3_x", 3, 3, 3, 2), ("d4x3x5_y", 4, 3, 5, 1), ("d3x4x4_z", 3, 4, 4, 5), ("d4x3x4_yz", 4, 3, 4, 7),
           ("d4x4x3_xy", 4, 4, 3, 3), ("d5x4x3_x", 5, 4, 3, 2)]


def coarse_solve_per3(impl, case):
    name, nx, ny, nz, ibc = case
    sd = _seed(name)
    so = pb.periodic_random_op3(nx, ny, nz, 14, pb.per3_of(ibc), sd)
    g = so.shape[1:]
    n = nx * ny * nz
    abd = np.zeros((n, n))
    impl.setup_cg3(so, abd, ibc=ibc)
    q = pb.uniform(g, sd + 2, -1, 1)
    impl.solve_cg3(q, pb.uniform(g, sd + 1, -1, 1), abd, ibc=ibc)
    return {"abd_upper": abd.T[np.triu_indices(n)].copy(), "q": q}


def _per3(n, per, kind, seed=0):
    if kind == "poisson":
        return (lambda: pb.periodic_poisson3(n[0], n[1], n[2], per)), (lambda: pb.periodic_rhs3(n[0], n[1], n[2], per))
    return (lambda: pb.periodic_random_op3(n[0], n[1], n[2], 14, per, seed)), (lambda: pb.periodic_rhs3(n[0], n[1], n[2], per))


SOLVES_PER3 = {
    # examples/basic-3d-ser/periodic.cc (seven-point Poisson, V-cycle, point relaxation) and wrapped random
    # 27-point operators.  Goldens (reference kernels driven end to end) exist for per_z (ibc 5) only.
    "perpoisson7_z_32_v21": _per3((32, 32, 32), (0, 0, 1), "poisson") + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=5),),
    "perpoisson7_z_24x20x32_v11": _per3((24, 20, 32), (0, 0, 1), "poisson") + (dict(relax="point", nrelax_pre=1, nrelax_post=1, ibc=5),),
    "perrand27_z_20x24x32_v21": _per3((20, 24, 32), (0, 0, 1), "random", 11) + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=5),),
    "perpoisson7_x_32_v21": _per3((32, 32, 32), (1, 0, 0), "poisson") + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=2),),
    "perpoisson7_xy_32x32x24_v21": _per3((32, 32, 24), (1, 1, 0), "poisson") + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=3),),
    "perrand27_y_24x32x20_v21": _per3((24, 32, 20), (0, 1, 0), "random", 12) + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=1),),
    "perrand27_xz_32x20x24_v21": _per3((32, 20, 24), (1, 0, 1), "random", 13) + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=6),),
    "perrand27_yz_20x32x32_v11": _per3((20, 32, 32), (0, 1, 1), "random", 14) + (dict(relax="point", nrelax_pre=1, nrelax_post=1, ibc=7),),
    "perrand27_xyz_32_v21": _per3((32, 32, 32), (1, 1, 1), "random", 15) + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=8),),
    # periodic F-cycles (round 3); golden for per_z, the other code against the oracle
    "perrand27_z_20x24x32_f21": _per3((20, 24, 32), (0, 0, 1), "random", 11) + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=5, cycle="f"),),
    "perpoisson7_xy_32x32x24_f21": _per3((32, 32, 24), (1, 1, 0), "poisson") + (dict(relax="point", nrelax_pre=2, nrelax_post=1, ibc=3, cycle="f"),),
}
