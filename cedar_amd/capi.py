"""ctypes bindings of include/cedar_amd.h.

`Kernels` exposes the BMG2_/BMG3_SymStd_* drop-ins under the method names the
parity suites use (tests/cases.py); numpy arrays are passed as host pointers and
staged through HBM by the library, `DeviceArray`s are passed as device pointers
and operated on in place.  `Solver` wraps the device-resident handle API.

There is no CPU fallback: a missing library is an ImportError, a missing GPU
makes every compute call abort inside HIP.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get("CEDAR_AMD_LIBPATH") or os.path.join(HERE, "lib", "libcedar_amd.so")  # override: A/B of two builds
if not os.path.exists(LIBPATH):
    # not built yet (fresh checkout): compile the HIP sources in-tree; still no CPU fallback
    import shutil
    import subprocess
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.call(["make", "-s", "-j8", "-C", os.path.join(HERE, "csrc")])
if not os.path.exists(LIBPATH):
    raise ImportError(
        f"{LIBPATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950); cedar_amd has no CPU fallback")

lib = C.CDLL(LIBPATH)
P = C.POINTER(C.c_double)
u = C.c_uint
DOWN, UP = 0, 1
RELAX = {"point": 0, "line-x": 1, "line-y": 2, "line-xy": 3, "plane-xy": 4, "plane-xz": 5, "plane-yz": 6, "plane-xyz": 7}
PLANE_DIR = {"xy": 0, "xz": 1, "yz": 2}

lib.cedar_amd_malloc.restype = C.c_void_p
lib.cedar_amd_malloc.argtypes = [C.c_size_t]
lib.cedar_amd_free.argtypes = [C.c_void_p]
lib.cedar_amd_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.cedar_amd_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.cedar_amd_memcpy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.cedar_amd_memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
lib.cedar_amd_l2norm.restype = C.c_double
lib.cedar_amd_l2norm.argtypes = [C.c_void_p, u, u, u]
lib.cedar_amd_version.restype = C.c_char_p
lib.cedar_amd_get_stream.restype = C.c_void_p
lib.cedar_amd_set_stream.argtypes = [C.c_void_p]


class Settings(C.Structure):
    _fields_ = [("relaxation", C.c_int), ("nrelax_pre", C.c_int), ("nrelax_post", C.c_int),
                ("num_levels", C.c_int), ("max_iter", C.c_int), ("tol", C.c_double),
                ("min_coarse", C.c_int), ("cycle", C.c_int), ("ibc", C.c_int),
                ("plane_relaxation", C.c_int), ("plane_nrelax_pre", C.c_int), ("plane_nrelax_post", C.c_int),
                ("plane_max_iter", C.c_int), ("plane_min_coarse", C.c_int), ("plane_tol", C.c_double)]


def plane_settings(st, plane):
    """fill the "plane-config" part of a Settings from plane = dict(relax=, nrelax_pre=, nrelax_post=, max_iter=,
    min_coarse=, tol=) or None (the reference's default: line-xy, one cycle per plane)"""
    plane = plane or {}
    st.plane_relaxation = RELAX[plane.get("relax", "line-xy")]
    st.plane_nrelax_pre, st.plane_nrelax_post = plane.get("nrelax_pre", 2), plane.get("nrelax_post", 1)
    st.plane_max_iter, st.plane_min_coarse = plane.get("max_iter", 1), plane.get("min_coarse", 3)
    st.plane_tol = float(plane.get("tol", 1e-8))
    return st


def device_count():
    return lib.cedar_amd_device_count()


def set_device(dev):
    if lib.cedar_amd_set_device(int(dev)) != 0:
        raise RuntimeError(f"hipSetDevice({dev}) failed")


def release_scratch():
    """frees the device scratch the library keeps between calls (cedar_amd_release_scratch)"""
    lib.cedar_amd_release_scratch()


def sync():
    lib.cedar_amd_sync()


class DeviceArray:
    """FP64 array resident in HBM (shape = reversed Fortran shape, like the numpy side)."""

    def __init__(self, shape):
        self.shape = tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape))
        self.ptr = lib.cedar_amd_malloc(self.size * 8)
        if not self.ptr:
            raise MemoryError("cedar_amd_malloc failed")

    @classmethod
    def from_numpy(cls, a):
        d = cls(a.shape)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.size
        lib.cedar_amd_memcpy_h2d(self.ptr, a.ctypes.data, self.size * 8)

    def numpy(self):
        out = np.empty(self.shape)
        lib.cedar_amd_memcpy_d2h(out.ctypes.data, self.ptr, self.size * 8)
        return out

    def zero(self):
        lib.cedar_amd_memset(self.ptr, 0, self.size * 8)

    def copy_from(self, other):
        assert other.size == self.size
        lib.cedar_amd_memcpy_d2d(self.ptr, other.ptr, self.size * 8)

    def free(self):
        if self.ptr:
            lib.cedar_amd_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _p(a):
    """host numpy array or DeviceArray -> pointer argument"""
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return C.cast(a.ptr, P)
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(P)


def l2norm(v):
    shp = v.shape
    KK = shp[0] if len(shp) == 3 else 1
    ptr = v.ptr if isinstance(v, DeviceArray) else v.ctypes.data
    return lib.cedar_amd_l2norm(ptr, shp[-1], shp[-2], KK)


class Kernels:
    """The BMG*_SymStd_* entry points, argument marshalling exactly as in the
    reference's binding classes (include/cedar/{2d,3d}/relax.h etc.)."""

    L = lib

    # ---- 2D
    def setup_recip2(self, so, sor):
        nst, JJ, II = so.shape
        lib.BMG2_SymStd_SETUP_recip(_p(so), _p(sor), u(II), u(JJ), nst, 2)

    # ibc: the boundary code the reference's bindings obtain from BMG_get_bc(per_mask) and hand to every
    # kernel (0 definite, 1 periodic in y, 2 in x, 3 in both; include/cedar/2d/relax.h:97 ...)
    def relax2(self, so, qf, q, sor, updown, ibc=0):
        nst, JJ, II = so.shape
        lib.BMG2_SymStd_relax_GS(1, _p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), 1, int(nst == 3), nst, 2, 1, updown, ibc)

    def relax2_psum(self, so, qf, q, sor, updown):
        """nine-point sweep with inter-row partial sums (cedar_amd_relax2_gs_psum); returns 1 if that path ran"""
        nst, JJ, II = so.shape
        assert nst == 5
        return lib.cedar_amd_relax2_gs_psum(_p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), updown)

    def setup_lines2(self, so, sor, d, ibc=0):
        nst, JJ, II = so.shape
        f = lib.BMG2_SymStd_SETUP_lines_x if d == "x" else lib.BMG2_SymStd_SETUP_lines_y
        f(_p(so), _p(sor), u(II), u(JJ), nst, ibc)

    def relax_lines2(self, so, qf, q, sor, updown, d, ibc=0):
        nst, JJ, II = so.shape
        f = lib.BMG2_SymStd_relax_lines_x if d == "x" else lib.BMG2_SymStd_relax_lines_y
        f(1, _p(so), _p(qf), _p(q), _p(sor), None, u(II), u(JJ), 1, int(nst == 3), nst, 1, updown, ibc)

    def residual2(self, so, qf, q, res):
        nst, JJ, II = so.shape
        i = lambda v: C.byref(C.c_int(v))
        lib.BMG2_SymStd_residual(i(0), _p(so), _p(qf), _p(q), _p(res), C.byref(u(II)), C.byref(u(JJ)),
                                 i(0), i(int(nst == 3)), i(nst), i(0), i(0), i(0), i(0))

    def matvec2(self, so, q, qf):
        nst, JJ, II = so.shape
        lib.cedar_amd_matvec2(_p(so), _p(q), _p(qf), u(II), u(JJ), nst)

    def matvec3(self, so, q, qf):
        nst, KK, JJ, II = so.shape
        lib.cedar_amd_matvec3(_p(so), _p(q), _p(qf), u(II), u(JJ), u(KK), nst)

    def restrict2(self, q, qc, ci, ibc=0):
        JJ, II = q.shape
        JJC, IIC = qc.shape
        lib.BMG2_SymStd_restrict(_p(q), _p(qc), _p(ci), II, JJ, IIC, JJC, ibc)

    def interp_add2(self, q, qc, res, so, ci, ibc=0):
        JJ, II = q.shape
        JJC, IIC = qc.shape
        lib.BMG2_SymStd_interp_add(_p(q), _p(qc), _p(res), _p(so), _p(ci), u(IIC), u(JJC), u(II), u(JJ), so.shape[0], ibc)

    def setup_interp2(self, so, ci, ibc=0):
        nst, JJ, II = so.shape
        _, JJC, IIC = ci.shape
        lib.BMG2_SymStd_SETUP_interp_OI(_p(so), None, _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3), nst, ibc, 0)

    def galerkin2(self, so, soc, ci, ibc=0):
        nst, JJ, II = so.shape
        _, JJC, IIC = ci.shape
        lib.BMG2_SymStd_SETUP_ITLI_ex(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(IIC), u(JJC), int(nst == 3), nst, ibc)

    def setup_cg2(self, so, abd, ibc=0):
        nst, JJ, II = so.shape
        n2, n1 = abd.shape
        r = lambda v: C.byref(u(v))
        lib.BMG2_SymStd_SETUP_cg_LU(_p(so), r(II), r(JJ), C.byref(C.c_int(nst)), _p(abd), r(n1), r(n2), C.byref(C.c_int(ibc)))

    def solve_cg2(self, q, qf, abd, ibc=0):
        JJ, II = q.shape
        n2, n1 = abd.shape
        bbd = np.zeros(n2)
        lib.BMG2_SymStd_SOLVE_cg(_p(q), _p(qf), u(II), u(JJ), _p(abd), _p(bbd), u(n1), u(n2), ibc)

    # ---- 3D
    def setup_recip3(self, so, sor):
        nst, KK, JJ, II = so.shape
        lib.BMG3_SymStd_SETUP_recip(_p(so), _p(sor), u(II), u(JJ), u(KK), nst, 2)

    def relax3(self, so, qf, q, sor, updown, ibc=0):
        nst, KK, JJ, II = so.shape
        lib.BMG3_SymStd_relax_GS(1, _p(so), _p(qf), _p(q), _p(sor), u(II), u(JJ), u(KK), int(nst == 4), nst, 2, 1, updown, ibc)

    def relax3_psum(self, so, qf, q, sor, updown):
        """27-point sweep with inter-plane partial sums (cedar_amd_relax3_gs_psum); returns 1 if that path ran"""
        nst, KK, JJ, II = so.shape
        assert nst == 14
        return lib.cedar_amd_relax3_gs_psum(_p(so), _p(qf), _p(q), _p(sor), C.cast(None, P), u(II), u(JJ), u(KK), updown)

    def residual3(self, so, qf, q, res):
        nst, KK, JJ, II = so.shape
        lib.BMG3_SymStd_residual(1, 1, int(nst == 4), _p(q), _p(qf), _p(so), _p(res), u(II), u(JJ), u(KK), nst)

    def restrict3(self, q, qc, ci, ibc=0):
        KK, JJ, II = q.shape
        KKC, JJC, IIC = qc.shape
        lib.BMG3_SymStd_restrict(_p(q), _p(qc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), ibc)

    def interp_add3(self, q, qc, so, res, ci, ibc=0):
        KK, JJ, II = q.shape
        KKC, JJC, IIC = qc.shape
        lib.BMG3_SymStd_interp_add(_p(q), _p(qc), _p(so), _p(res), _p(ci), u(IIC), u(JJC), u(KKC), u(II), u(JJ), u(KK), so.shape[0], ibc)

    def setup_interp3(self, so, ci, ibc=0):
        nst, KK, JJ, II = so.shape
        _, KKC, JJC, IIC = ci.shape
        lib.BMG3_SymStd_SETUP_interp_OI(_p(so), None, _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC),
                                        int(nst == 4), nst, 1, ibc, None)

    def galerkin3(self, so, soc, ci, ibc=0):
        nst, KK, JJ, II = so.shape
        _, KKC, JJC, IIC = ci.shape
        f = lib.BMG3_SymStd_SETUP_ITLI07_ex if nst == 4 else lib.BMG3_SymStd_SETUP_ITLI27_ex
        f(_p(so), _p(soc), _p(ci), u(II), u(JJ), u(KK), u(IIC), u(JJC), u(KKC), ibc)

    def setup_cg3(self, so, abd, ibc=0):
        nst, KK, JJ, II = so.shape
        n2, n1 = abd.shape
        lib.BMG3_SymStd_SETUP_cg_LU(_p(so), u(II), u(JJ), u(KK), nst, _p(abd), u(n1), u(n2), ibc)

    def solve_cg3(self, q, qf, abd, ibc=0):
        KK, JJ, II = q.shape
        n2, n1 = abd.shape
        bbd = np.zeros(n2)
        lib.BMG3_SymStd_SOLVE_cg(_p(q), _p(qf), u(II), u(JJ), u(KK), _p(abd), _p(bbd), u(n1), u(n2), ibc)

    def relax_planes3(self, so, x, b, d, updown, plane=None):
        """kman->setup<plane_relax<d>>(so); kman->run<plane_relax<d>>(so, x, b, updown) (cedar_amd_planes_*)"""
        nst, KK, JJ, II = so.shape
        st = None
        if plane is not None:
            st = Settings()
            lib.cedar_amd_default_settings(C.byref(st))
            st.relaxation = RELAX[plane.get("relax", "line-xy")]
            st.nrelax_pre, st.nrelax_post = plane.get("nrelax_pre", 2), plane.get("nrelax_post", 1)
            st.max_iter, st.min_coarse, st.tol = plane.get("max_iter", 1), plane.get("min_coarse", 3), float(plane.get("tol", 1e-8))
        lib.cedar_amd_planes_create.restype = C.c_void_p
        h = lib.cedar_amd_planes_create(PLANE_DIR[d], u(II - 2), u(JJ - 2), u(KK - 2), nst, _p(so), C.byref(st) if st else None)
        if not h:
            raise RuntimeError("cedar_amd_planes_create failed")
        h = C.c_void_p(h)
        lib.cedar_amd_planes_run(h, _p(so), _p(x), _p(b), updown)
        lib.cedar_amd_planes_destroy(h)

    def l2(self, v):
        return l2norm(v)


# ---------------------------------------------------------------- handle API
lib.cedar_amd_solver_create.restype = C.c_void_p
lib.cedar_amd_solver_create.argtypes = [C.c_int, u, u, u, C.c_int, C.c_void_p, C.c_int, C.POINTER(Settings)]
lib.cedar_amd_solver_destroy.argtypes = [C.c_void_p]
lib.cedar_amd_solver_nlevels.argtypes = [C.c_void_p]
lib.cedar_amd_solver_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(u), C.POINTER(u), C.POINTER(u)]
lib.cedar_amd_solver_get.restype = C.c_size_t
lib.cedar_amd_solver_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
lib.cedar_amd_solver_set.restype = C.c_size_t
lib.cedar_amd_solver_set.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]
lib.cedar_amd_solver_vcycle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.cedar_amd_solver_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.cedar_amd_solver_time_vcycles.restype = C.c_float
lib.cedar_amd_solver_time_vcycles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.cedar_amd_solver_time_relax.restype = C.c_float
lib.cedar_amd_solver_time_relax.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.cedar_amd_solver_time_op.restype = C.c_float
lib.cedar_amd_solver_time_op.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
lib.cedar_amd_gallery.argtypes = [C.c_int, C.c_void_p, C.c_void_p, u, u, u, C.c_void_p]


def _vp(a):
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    if hasattr(a, "data_ptr"):  # torch tensor (device or host), contiguous float64
        return a.data_ptr()
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


GALLERY = {"poisson2": (0, 3), "diag_diffusion2": (1, 3), "fe2": (2, 5),
           "poisson3": (10, 4), "diag_diffusion3": (11, 4), "fe3": (12, 14)}


def gallery(name, n, params=None, with_rhs=True, device=True):
    """build a gallery operator (+ example rhs) in HBM; returns (so, b)"""
    which, nst = GALLERY[name]
    n = tuple(int(v) for v in n)
    g = tuple(v + 2 for v in n[::-1])
    mk = DeviceArray if device else (lambda s: np.zeros(s))
    so, b = mk((nst,) + g), (mk(g) if with_rhs else None)
    pp = None
    if params is not None:
        pp = (C.c_double * len(params))(*params)
    nx, ny = n[0], n[1]
    nz = n[2] if len(n) == 3 else 1
    lib.cedar_amd_gallery(which, _vp(so), _vp(b), nx, ny, nz, pp)
    return so, b


class Solver:
    """cedar::cdr2::solver / cdr3::solver on the device (include/cedar_amd.h, handle API)."""

    def __init__(self, so, relax="point", nrelax_pre=2, nrelax_post=1, num_levels=-1,
                 max_iter=10, tol=1e-8, min_coarse=3, share_operator=False, cycle="v", ibc=0, plane=None):
        shp = so.shape
        self.nd = len(shp) - 1
        nst = shp[0]
        nx, ny = shp[-1] - 2, shp[-2] - 2
        nz = shp[1] - 2 if self.nd == 3 else 1
        self.shape = tuple(shp[1:])
        st = Settings(RELAX[relax], nrelax_pre, nrelax_post, num_levels, max_iter, tol, min_coarse,
                      {"v": 0, "f": 1}[cycle], ibc)
        plane_settings(st, plane)
        self.max_iter = max_iter
        self._so = so if share_operator else None  # keep the shared operator alive
        self.h = lib.cedar_amd_solver_create(self.nd, nx, ny, nz, nst, _vp(so), int(share_operator), C.byref(st))
        if not self.h:
            raise RuntimeError("cedar_amd_solver_create failed")

    def nlevels(self):
        return lib.cedar_amd_solver_nlevels(self.h)

    def dims(self, lvl):
        a, b, c = u(), u(), u()
        lib.cedar_amd_solver_level_dims(self.h, lvl, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def array(self, lvl, what):
        n = lib.cedar_amd_solver_get(self.h, lvl, what.encode(), None)
        if n == 0:
            return None
        out = np.empty(n)
        lib.cedar_amd_solver_get(self.h, lvl, what.encode(), out.ctypes.data)
        if what == "ABD":
            return out
        nx, ny, nz = self.dims(lvl)
        shp = (ny + 2, nx + 2) if self.nd == 2 else (nz + 2, ny + 2, nx + 2)
        return out.reshape((-1,) + shp)

    def set_array(self, lvl, what, a):
        """replace a set-up product of a level (cedar_amd_solver_set); a: numpy array of the product's full size"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        n = lib.cedar_amd_solver_get(self.h, lvl, what.encode(), None)
        assert n == a.size, (what, lvl, n, a.size)
        assert lib.cedar_amd_solver_set(self.h, lvl, what.encode(), a.ctypes.data) == n

    def vcycle(self, x, b):
        lib.cedar_amd_solver_vcycle(self.h, _vp(x), _vp(b))

    def solve(self, b, x):
        rel = np.zeros(self.max_iter + 1)
        n = lib.cedar_amd_solver_solve(self.h, _vp(b), _vp(x), rel.ctypes.data)
        return rel[: n + 1]

    def time_vcycles(self, x, b, n):
        return lib.cedar_amd_solver_time_vcycles(self.h, x.ptr, b.ptr, n)

    def time_relax(self, x, b, n):
        return lib.cedar_amd_solver_time_relax(self.h, x.ptr, b.ptr, n)

    def time_op(self, x, b, op, n):
        """n launches of a level-0 kernel: op 'residual' | 'restrict' | 'interp_add' (overwrites x); elapsed ms"""
        return lib.cedar_amd_solver_time_op(self.h, x.ptr, b.ptr, {"residual": 1, "restrict": 2, "interp_add": 3}[op], n)

    def close(self):
        if self.h:
            lib.cedar_amd_solver_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
