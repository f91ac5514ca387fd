"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends for the two CPU checkers.

* ``Oracle``  : liboracle.so, the plain-C restatement (oracle/boxmg*.c).
* ``Ref``     : oracle/_ref/libcedar_ref.so, the reference's own Fortran
                compiled in the build container (only present where
                /root/reference existed at build time, or where the built
                .so travelled with the snapshot).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Arrays are C-ordered numpy float64 whose *reversed* shape is the
Fortran shape, i.e. a 2D stencil is (nst, JJ, II), a 3D grid function is
(KK, JJ, II); ghosts included.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
P = C.POINTER(C.c_double)
u = C.c_uint
DOWN, UP = 0, 1
RELAX = {"point": 0, "line-x": 1, "line-y": 2, "line-xy": 3, "plane-xy": 4, "plane-xz": 5, "plane-yz": 6, "plane-xyz": 7}
PLANE_DIR = {"xy": 0, "xz": 1, "yz": 2}


def plane_cfg(plane):
    """plane = dict(relax=, nrelax_pre=, nrelax_post=, max_iter=, min_coarse=, tol=) or None (the reference's default
    plane configuration, src/kernel_params.cc:72-78) -> (int[5] or None, tol)"""
    if plane is None:
        return None, 0.0
    cfg = (C.c_int * 5)(RELAX[plane.get("relax", "line-xy")], plane.get("nrelax_pre", 2), plane.get("nrelax_post", 1),
                        plane.get("max_iter", 1), plane.get("min_coarse", 3))
    return cfg, float(plane.get("tol", 1e-8))


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


class Oracle:
    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        self.L = L = C.CDLL(path)
        L.orc_l2_norm2.restype = C.c_double
        L.orc_l2_norm3.restype = C.c_double
        L.orc_inf_norm3.restype = C.c_double
        L.orc_ml_create.restype = C.c_void_p
        L.orc_ml_level_array.restype = P

    # ---------------- 2D ----------------
    def setup_recip2(self, so, sor):
        _, JJ, II = so.shape
        self.L.orc2_setup_recip(_p(so), _p(sor), u(II), u(JJ))

    def relax2(self, so, qf, q, sor, updown, ibc=0):
        nst, JJ, II = so.shape
        if ibc:
            self.L.orc2_relax_gs_per(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), int(nst == 3), updown, ibc)
        else:
            self.L.orc2_relax_gs(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), int(nst == 3), updown)

    def relax_colour2(self, so, qf, q, sor, pts):
        nst, JJ, II = so.shape
        self.L.orc2_relax_colour(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), int(nst == 3), pts)

    def relax_column2(self, so, qf, q, sor, i1, jb):
        nst, JJ, II = so.shape
        self.L.orc2_relax_column(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), i1, jb)

    def setup_interp2_ex(self, so, ci, phase_mask, lo):
        nst, JJ, II = so.shape
        _, JJC, IIC = ci.shape
        self.L.orc2_setup_interp_ex(_p(so), _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3), phase_mask, lo[0], lo[1])

    def setup_lines2(self, so, sor, d, ibc=0):
        _, JJ, II = so.shape
        if ibc:
            f = self.L.orc2_setup_lines_x_per if d == "x" else self.L.orc2_setup_lines_y_per
            f(_p(so), _p(sor), u(II), u(JJ), ibc)
            return
        f = self.L.orc2_setup_lines_x if d == "x" else self.L.orc2_setup_lines_y
        f(_p(so), _p(sor), u(II), u(JJ))

    def relax_lines2(self, so, qf, q, sor, updown, d, ibc=0):
        nst, JJ, II = so.shape
        if ibc:
            b = np.zeros(2 * JJ + II)
            f = self.L.orc2_relax_lines_x_per if d == "x" else self.L.orc2_relax_lines_y_per
            f(_p(so), _p(qf), _p(q), _p(sor), _p(b), u(II), u(JJ), int(nst == 3), updown, ibc)
            return
        if d == "x":
            self.L.orc2_relax_lines_x(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), int(nst == 3), updown)
        else:
            b = np.zeros(2 * JJ + II)
            self.L.orc2_relax_lines_y(_p(so), _p(qf), _p(q), _p(sor), _p(b), u(II), u(JJ), int(nst == 3), updown)

    def residual2(self, so, qf, q, res):
        nst, JJ, II = so.shape
        self.L.orc2_residual(_p(so), _p(qf), _p(q), _p(res), u(II), u(JJ), int(nst == 3))

    def matvec2(self, so, q, qf):
        nst, JJ, II = so.shape
        self.L.orc2_matvec(_p(so), _p(q), _p(qf), u(II), u(JJ), int(nst == 3))

    def matvec3(self, so, q, qf):
        nst, KK, JJ, II = so.shape
        self.L.orc3_matvec(_p(so), _p(q), _p(qf), u(II), u(JJ), u(KK), int(nst == 4))

    def restrict2(self, q, qc, ci, ibc=0):
        JJ, II = q.shape
        JJC, IIC = qc.shape
        if ibc:
            self.L.orc2_restrict_per(_p(q), _p(qc), _p(ci), u(II), u(JJ), u(IIC), u(JJC), ibc)
        else:
            self.L.orc2_restrict(_p(q), _p(qc), _p(ci), u(II), u(JJ), u(IIC), u(JJC))

    def interp_add2(self, q, qc, res, so, ci, ibc=0):
        JJ, II = q.shape
        JJC, IIC = qc.shape
        if ibc:
            self.L.orc2_interp_add_per(_p(q), _p(qc), _p(res), _p(so), _p(ci), u(IIC), u(JJC), u(II), u(JJ), ibc)
        else:
            self.L.orc2_interp_add(_p(q), _p(qc), _p(res), _p(so), _p(ci), u(IIC), u(JJC), u(II), u(JJ))

    def setup_interp2(self, so, ci, ibc=0):
        nst, JJ, II = so.shape
        _, JJC, IIC = ci.shape
        if ibc:
            self.L.orc2_setup_interp_per(_p(so), _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3), ibc)
        else:
            self.L.orc2_setup_interp(_p(so), _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3))

    def galerkin2(self, so, soc, ci, ibc=0):
        nst, JJ, II = so.shape
        _, JJC, IIC = ci.shape
        if ibc:
            self.L.orc2_galerkin_per(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3), ibc)
        else:
            self.L.orc2_galerkin(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3))

    def setup_cg2(self, so, abd, ibc=0):
        """abd: (n, nx+2) band storage, or (n, n) dense for a periodic ibc (2d/solver.h:110-114)"""
        nst, JJ, II = so.shape
        n2, n1 = abd.shape
        if ibc:
            return self.L.orc2_setup_cg_per(_p(so), u(II), u(JJ), nst, _p(abd), u(n1), ibc)
        return self.L.orc2_setup_cg(_p(so), u(II), u(JJ), nst, _p(abd), u(n1), u(n2))

    def solve_cg2(self, q, qf, abd, ibc=0):
        JJ, II = q.shape
        n2, n1 = abd.shape
        bbd = np.zeros(n2)
        if ibc:
            self.L.orc2_solve_cg_per(_p(q), _p(qf), u(II), u(JJ), _p(abd), _p(bbd), u(n1), ibc)
        else:
            self.L.orc2_solve_cg(_p(q), _p(qf), u(II), u(JJ), _p(abd), _p(bbd), u(n1), u(n2))

    # ---------------- 3D ----------------
    def setup_recip3(self, so, sor):
        _, KK, JJ, II = so.shape
        self.L.orc3_setup_recip(_p(so), _p(sor), u(II), u(JJ), u(KK))

    def relax3(self, so, qf, q, sor, updown, ibc=0):
        nst, KK, JJ, II = so.shape
        if ibc:
            self.L.orc3_relax_gs_per(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), u(KK), int(nst == 4), updown, ibc)
        else:
            self.L.orc3_relax_gs(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), u(KK), int(nst == 4), updown)

    def wrap3(self, a, ibc):
        """periodic ghost refresh of one array or a stack of arrays (y, x, z)"""
        KK, JJ, II = a.shape[-3:]
        self.L.orc3_wrap(_p(a), u(II), u(JJ), u(KK), int(a.size // (II * JJ * KK)), ibc)

    def relax_colour3(self, so, qf, q, sor, pts):
        nst, KK, JJ, II = so.shape
        self.L.orc3_relax_colour(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), u(KK), int(nst == 4), pts)

    def relax_colour3_part(self, so, qf, q, sor, pts, part):
        nst, KK, JJ, II = so.shape
        assert nst == 14
        self.L.orc3_relax_colour_part(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), u(KK), pts, part)

    def relax_column3(self, so, qf, q, sor, i1, jb, kb):
        nst, KK, JJ, II = so.shape
        self.L.orc3_relax_column(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), u(KK), i1, jb, kb)

    def setup_interp3_ex(self, so, ci, phase_mask, lo):
        nst, KK, JJ, II = so.shape
        _, KKC, JJC, IIC = ci.shape
        self.L.orc3_setup_interp_ex(_p(so), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), int(nst == 4),
                                    phase_mask, lo[0], lo[1], lo[2])

    def residual3(self, so, qf, q, res):
        nst, KK, JJ, II = so.shape
        self.L.orc3_residual(_p(so), _p(qf), _p(q), _p(res), u(II), u(JJ), u(KK), int(nst == 4))

    def restrict3(self, q, qc, ci, ibc=0):
        KK, JJ, II = q.shape
        KKC, JJC, IIC = qc.shape
        if ibc:
            self.L.orc3_restrict_per(_p(q), _p(qc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), ibc)
        else:
            self.L.orc3_restrict(_p(q), _p(qc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC))

    def interp_add3(self, q, qc, so, res, ci, ibc=0):
        KK, JJ, II = q.shape
        KKC, JJC, IIC = qc.shape
        if ibc:
            self.L.orc3_interp_add_per(_p(q), _p(qc), _p(so), _p(res), _p(ci), u(IIC), u(JJC), u(KKC), u(II), u(JJ), u(KK), ibc)
        else:
            self.L.orc3_interp_add(_p(q), _p(qc), _p(so), _p(res), _p(ci), u(IIC), u(JJC), u(KKC), u(II), u(JJ), u(KK))

    def setup_interp3(self, so, ci, ibc=0):
        nst, KK, JJ, II = so.shape
        _, KKC, JJC, IIC = ci.shape
        if ibc:
            self.L.orc3_setup_interp_per(_p(so), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), int(nst == 4), ibc)
        else:
            self.L.orc3_setup_interp(_p(so), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), int(nst == 4))

    def galerkin3(self, so, soc, ci, ibc=0):
        nst, KK, JJ, II = so.shape
        _, KKC, JJC, IIC = ci.shape
        if ibc:
            self.L.orc3_galerkin_per(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), int(nst == 4), ibc)
        else:
            self.L.orc3_galerkin(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), int(nst == 4))

    def setup_cg3(self, so, abd, ibc=0):
        """abd: (n, nx*(ny+1)+2) band storage, or (n, n) dense for a periodic ibc (3d/solver.h:118-121)"""
        nst, KK, JJ, II = so.shape
        n2, n1 = abd.shape
        if ibc:
            return self.L.orc3_setup_cg_per(_p(so), u(II), u(JJ), u(KK), nst, _p(abd), u(n1), ibc)
        return self.L.orc3_setup_cg(_p(so), u(II), u(JJ), u(KK), nst, _p(abd), u(n1), u(n2))

    def solve_cg3(self, q, qf, abd, ibc=0):
        KK, JJ, II = q.shape
        n2, n1 = abd.shape
        bbd = np.zeros(n2)
        if ibc:
            self.L.orc3_solve_cg_per(_p(q), _p(qf), u(II), u(JJ), u(KK), _p(abd), _p(bbd), u(n1), ibc)
        else:
            self.L.orc3_solve_cg(_p(q), _p(qf), u(II), u(JJ), u(KK), _p(abd), _p(bbd), u(n1), u(n2))

    def l2(self, v):
        if v.ndim == 2:
            return self.L.orc_l2_norm2(_p(v), u(v.shape[1]), u(v.shape[0]))
        return self.L.orc_l2_norm3(_p(v), u(v.shape[2]), u(v.shape[1]), u(v.shape[0]))

    # ---------------- multilevel ----------------
    # ---------------- plane relaxation ----------------
    def plane_rhs3(self, so, x, b, b2, d, ipl):
        nst, KK, JJ, II = so.shape
        self.L.orc3_plane_rhs(PLANE_DIR[d], nst, _p(so), _p(x), _p(b), _p(b2), u(II), u(JJ), u(KK), ipl)

    def relax_planes3(self, so, x, b, d, updown, plane=None):
        """kman->setup<plane_relax<d>>(so); kman->run<plane_relax<d>>(so, x, b, updown)"""
        nst, KK, JJ, II = so.shape
        cfg, tol = plane_cfg(plane)
        self.L.orc3_planes_create.restype = C.c_void_p
        h = C.c_void_p(self.L.orc3_planes_create(PLANE_DIR[d], _p(so), u(II), u(JJ), u(KK), nst, cfg, C.c_double(tol)))
        self.L.orc3_planes_relax(h, _p(so), _p(x), _p(b), updown)
        self.L.orc3_planes_destroy(h)

    def ml_create(self, so, relax="point", nrelax_pre=2, nrelax_post=1, min_coarse=3, num_levels=-1, cycle="v", ibc=0,
                  plane=None):
        nd = so.ndim - 1
        if nd == 2:
            nst, JJ, II = so.shape
            nx, ny, nz = II - 2, JJ - 2, 1
        else:
            nst, KK, JJ, II = so.shape
            nx, ny, nz = II - 2, JJ - 2, KK - 2
        self.L.orc_ml_create_ex.restype = C.c_void_p
        cfg, tol = plane_cfg(plane)
        h = self.L.orc_ml_create_ex(nd, u(nx), u(ny), u(nz), nst, _p(so), RELAX[relax],
                                    nrelax_pre, nrelax_post, min_coarse, num_levels, ibc, cfg, C.c_double(tol))
        if not h:
            raise ValueError("unknown boundary code %r" % (ibc,))
        m = MLHandle(self, h, nd)
        if cycle == "f":
            self.L.orc_ml_set_cycle(m.h, 1)
        return m


class MLHandle:
    def __init__(self, orc, h, nd):
        self.o, self.h, self.nd = orc, C.c_void_p(h), nd

    def nlevels(self):
        return self.o.L.orc_ml_nlevels(self.h)

    def dims(self, lvl):
        a, b, c = u(), u(), u()
        self.o.L.orc_ml_level_dims(self.h, lvl, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def array(self, lvl, what):
        n = C.c_size_t()
        ptr = self.o.L.orc_ml_level_array(self.h, lvl, what.encode(), C.byref(n))
        if not ptr or n.value == 0:
            return None
        flat = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()
        if what == "ABD":
            return flat
        nx, ny, nz = self.dims(lvl)
        shp = (ny + 2, nx + 2) if self.nd == 2 else (nz + 2, ny + 2, nx + 2)
        return flat.reshape((-1,) + shp)

    def vcycle(self, x, b):
        self.o.L.orc_ml_vcycle(self.h, _p(x), _p(b))

    def solve(self, b, x, maxiter=10, tol=1e-8):
        rel = np.zeros(maxiter + 1)
        n = self.o.L.orc_ml_solve(self.h, _p(b), _p(x), maxiter, C.c_double(tol), _p(rel))
        return rel[: n + 1]

    def close(self):
        if self.h:
            self.o.L.orc_ml_destroy(self.h)
            self.h = None


class Ref:
    """The reference's Fortran entry points (signatures: SURVEY.md section 8b)."""

    def __init__(self):
        path = os.path.join(HERE, "_ref", "libcedar_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.L = C.CDLL(path)

    def setup_recip2(self, so, sor):
        nst, JJ, II = so.shape
        self.L.BMG2_SymStd_SETUP_recip(_p(so), _p(sor), u(II), u(JJ), nst, 2)

    def relax2(self, so, qf, q, sor, updown, ibc=0):
        nst, JJ, II = so.shape
        self.L.BMG2_SymStd_relax_GS(1, _p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), 1, int(nst == 3), nst, 2, 1, updown, ibc)

    def setup_lines2(self, so, sor, d, ibc=0):
        nst, JJ, II = so.shape
        f = self.L.BMG2_SymStd_SETUP_lines_x if d == "x" else self.L.BMG2_SymStd_SETUP_lines_y
        f(_p(so), _p(sor), u(II), u(JJ), nst, ibc)

    def relax_lines2(self, so, qf, q, sor, updown, d, ibc=0):
        nst, JJ, II = so.shape
        b = np.zeros(2 * JJ + 2 * II)
        f = self.L.BMG2_SymStd_relax_lines_x if d == "x" else self.L.BMG2_SymStd_relax_lines_y
        f(1, _p(so), _p(qf), _p(q), _p(sor), _p(b), u(II), u(JJ), 1, int(nst == 3), nst, 1, updown, ibc)

    def residual2(self, so, qf, q, res):
        nst, JJ, II = so.shape
        i = lambda v: C.byref(C.c_int(v))
        self.L.BMG2_SymStd_residual(i(0), _p(so), _p(qf), _p(q), _p(res), C.byref(u(II)), C.byref(u(JJ)),
                                    i(0), i(int(nst == 3)), i(nst), i(0), i(0), i(0), i(0))

    def restrict2(self, q, qc, ci, ibc=0):
        JJ, II = q.shape
        JJC, IIC = qc.shape
        self.L.BMG2_SymStd_restrict(_p(q), _p(qc), _p(ci), II, JJ, IIC, JJC, ibc)

    def interp_add2(self, q, qc, res, so, ci, ibc=0):
        JJ, II = q.shape
        JJC, IIC = qc.shape
        self.L.BMG2_SymStd_interp_add(_p(q), _p(qc), _p(res), _p(so), _p(ci), u(IIC), u(JJC), u(II), u(JJ), so.shape[0], ibc)

    def setup_interp2(self, so, ci, ibc=0):
        nst, JJ, II = so.shape
        _, JJC, IIC = ci.shape
        soc = np.zeros((5, JJC, IIC))
        self.L.BMG2_SymStd_SETUP_interp_OI(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3), nst, ibc, 0)

    def galerkin2(self, so, soc, ci, ibc=0):
        nst, JJ, II = so.shape
        _, JJC, IIC = ci.shape
        self.L.BMG2_SymStd_SETUP_ITLI_ex(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3), nst, ibc)

    def setup_cg2(self, so, abd, ibc=0):
        nst, JJ, II = so.shape
        n2, n1 = abd.shape
        r = lambda v: C.byref(u(v))
        self.L.BMG2_SymStd_SETUP_cg_LU(_p(so), r(II), r(JJ), C.byref(C.c_int(nst)), _p(abd), r(n1), r(n2), C.byref(C.c_int(ibc)))

    def solve_cg2(self, q, qf, abd, ibc=0):
        JJ, II = q.shape
        n2, n1 = abd.shape
        bbd = np.zeros(n2)
        self.L.BMG2_SymStd_SOLVE_cg(_p(q), _p(qf), u(II), u(JJ), _p(abd), _p(bbd), u(n1), u(n2), ibc)

    def setup_recip3(self, so, sor):
        nst, KK, JJ, II = so.shape
        self.L.BMG3_SymStd_SETUP_recip(_p(so), _p(sor), u(II), u(JJ), u(KK), nst, 2)

    def relax3(self, so, qf, q, sor, updown, ibc=0):
        nst, KK, JJ, II = so.shape
        self.L.BMG3_SymStd_relax_GS(1, _p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), u(KK), int(nst == 4), nst, 2, 1, updown, ibc)

    def residual3(self, so, qf, q, res):
        nst, KK, JJ, II = so.shape
        self.L.BMG3_SymStd_residual(1, 1, int(nst == 4), _p(q), _p(qf), _p(so), _p(res), u(II), u(JJ), u(KK), nst)

    def restrict3(self, q, qc, ci, ibc=0):
        KK, JJ, II = q.shape
        KKC, JJC, IIC = qc.shape
        self.L.BMG3_SymStd_restrict(_p(q), _p(qc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), ibc)

    def interp_add3(self, q, qc, so, res, ci, ibc=0):
        """ibc: only 0 and 5 (per_z) are defined by the source: the x / y ghost loops of the routine
        (BMG3_SymStd_interp_add.f90:253-272) run over stale loop indices"""
        assert ibc in (0, 5)
        KK, JJ, II = q.shape
        KKC, JJC, IIC = qc.shape
        self.L.BMG3_SymStd_interp_add(_p(q), _p(qc), _p(so), _p(res), _p(ci), u(IIC), u(JJC), u(KKC), u(II), u(JJ), u(KK), so.shape[0], ibc)

    def setup_interp3(self, so, ci, ibc=0):
        nst, KK, JJ, II = so.shape
        _, KKC, JJC, IIC = ci.shape
        soc = np.zeros((14, KKC, JJC, IIC))
        yo = np.zeros((14, 2, JJ, II))
        self.L.BMG3_SymStd_SETUP_interp_OI(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC),
                                           int(nst == 4), nst, 1, ibc, _p(yo))

    def galerkin3(self, so, soc, ci, ibc=0):
        nst, KK, JJ, II = so.shape
        _, KKC, JJC, IIC = ci.shape
        f = self.L.BMG3_SymStd_SETUP_ITLI07_ex if nst == 4 else self.L.BMG3_SymStd_SETUP_ITLI27_ex
        f(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), ibc)

    def setup_cg3(self, so, abd, ibc=0):
        nst, KK, JJ, II = so.shape
        n2, n1 = abd.shape
        self.L.BMG3_SymStd_SETUP_cg_LU(_p(so), u(II), u(JJ), u(KK), nst, _p(abd), u(n1), u(n2), ibc)

    def solve_cg3(self, q, qf, abd, ibc=0):
        KK, JJ, II = q.shape
        n2, n1 = abd.shape
        bbd = np.zeros(n2)
        self.L.BMG3_SymStd_SOLVE_cg(_p(q), _p(qf), u(II), u(JJ), u(KK), _p(abd), _p(bbd), u(n1), u(n2), ibc)
