"""The domain-decomposed solvers as the library runs them: `cedar_amd_dist3_*` / `cedar_amd_dist2_*` (include/cedar_amd.h
sections 4 and 4b, cedar_amd/csrc/dist3.cpp, dist2.cpp).  The whole cycle -- sweeps by row class, halo exchanges on the main and the side stream,
x-face fix-ups, gathered coarse levels, norms -- is orchestrated in compiled code below the C ABI; this module only
hands over the rank's arrays and the transport.

Transport: a `NativeComm` (RCCL communicator owned by the library) is passed as a handle; any other comm object
(`SocketComm`, the rehearsal transport for ranks that share one GPU) is wrapped into the three-function transport table
of the ABI -- the counterpart of the reference's halo_exchanger plug-in (include/cedar/kernel.h:25-37).

cedar_amd/dist.py keeps the same orchestration in Python on an abstract backend: it is what runs on the CPU against the
oracle (tests/test_dist_cpu.py) and pins the algorithm; on the GPU it is no longer the product path.
"""
import ctypes as C

import numpy as np

from . import capi
from .comm import NativeComm

lib = capi.lib

_EXCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                    C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
_GATH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_ARED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class _Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("exchange", _EXCH), ("allgather", _GATH), ("allreduce_sum", _ARED)]


lib.cedar_amd_dist3_create.restype = C.c_void_p
lib.cedar_amd_dist3_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p,
                                       C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_void_p, C.c_int, C.c_int]
lib.cedar_amd_dist3_destroy.argtypes = [C.c_void_p]
lib.cedar_amd_dist3_nlevels.argtypes = [C.c_void_p]
lib.cedar_amd_dist3_distributed_levels.argtypes = [C.c_void_p]
lib.cedar_amd_dist3_chain_levels.argtypes = [C.c_void_p]
lib.cedar_amd_dist3_vcycle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.cedar_amd_dist3_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.cedar_amd_dist3_time_relax.restype = C.c_float
lib.cedar_amd_dist3_time_relax.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.cedar_amd_dist3_rank_grid.argtypes = [C.c_int, C.POINTER(C.c_int)]


lib.cedar_amd_dist2_create.restype = C.c_void_p
lib.cedar_amd_dist2_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p,
                                       C.c_uint, C.c_uint, C.c_int, C.c_void_p, C.c_int]
lib.cedar_amd_dist2_destroy.argtypes = [C.c_void_p]
lib.cedar_amd_dist2_nlevels.argtypes = [C.c_void_p]
lib.cedar_amd_dist2_vcycle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.cedar_amd_dist2_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.cedar_amd_dist2_time_relax.restype = C.c_float
lib.cedar_amd_dist2_time_relax.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.cedar_amd_dist2_rank_grid.argtypes = [C.c_int, C.POINTER(C.c_int)]


def rank_grid2(world):
    p = (C.c_int * 2)()
    lib.cedar_amd_dist2_rank_grid(world, p)
    return tuple(p)


def rank_grid(world):
    p = (C.c_int * 3)()
    lib.cedar_amd_dist3_rank_grid(world, p)
    return tuple(p)


class _Raw:
    """a device address dressed as the (array, offset, count) triples the Python comm objects take"""

    def __init__(self, ptr):
        self.ptr = int(ptr)


def _table_from(comm):
    """the three transport functions on top of a comm object with p2p / allgather / allreduce_sum (SocketComm)"""

    def exchange(_ctx, ns, speer, sbuf, scount, nr, rpeer, rbuf, rcount):
        try:
            comm.p2p([(speer[i], _Raw(sbuf[i]), 0, scount[i]) for i in range(ns) if scount[i]],
                     [(rpeer[i], _Raw(rbuf[i]), 0, rcount[i]) for i in range(nr) if rcount[i]])
            return 0
        except Exception as e:  # noqa: BLE001 -- reported through the return code, the C side prints
            print("transport exchange failed:", e, flush=True)
            return 1

    def allgather(_ctx, send, recv, count):
        try:
            comm.allgather(_Raw(send), int(count), _Raw(recv))
            return 0
        except Exception as e:  # noqa: BLE001
            print("transport allgather failed:", e, flush=True)
            return 1

    def allreduce(_ctx, vals, n):
        try:
            for i in range(n):
                vals[i] = comm.allreduce_sum(vals[i])
            return 0
        except Exception as e:  # noqa: BLE001
            print("transport allreduce failed:", e, flush=True)
            return 1

    fns = (_EXCH(exchange), _GATH(allgather), _ARED(allreduce))
    return _Transport(None, *fns), fns


class DistSolver3:
    """cdr3::mpi::solver on one rank's GPU: `A_local` = capi.DeviceArray (nst, nz+2, ny+2, nx+2)"""

    def __init__(self, comm, rank, world, A_local, pgrid=None, nrelax_pre=2, nrelax_post=1, min_coarse=3, max_iter=10,
                 tol=1e-8, agglomerate_below=64, overlap_min=96):
        self.comm, self.rank, self.world = comm, rank, world
        self._A = A_local  # must outlive the handle
        nst = A_local.shape[0]
        nz, ny, nx = (int(v) - 2 for v in A_local.shape[1:])
        st = capi.Settings(0, nrelax_pre, nrelax_post, -1, max_iter, tol, min_coarse, 0, 0)
        capi.plane_settings(st, None)
        self.max_iter = max_iter
        self.p = tuple(pgrid) if pgrid else rank_grid(world)
        pg = (C.c_int * 3)(*self.p)
        handle, table = None, None
        if comm == "loopback":  # measuring aid: one rank of the grid talking to itself (tools/dist_overhead.py)
            self._tab = _Transport()
            lib.cedar_amd_transport_loopback(C.byref(self._tab), world)
            table = C.byref(self._tab)
        elif isinstance(comm, NativeComm):
            handle = comm.h
        elif comm is not None and world > 1:
            self._tab, self._keep = _table_from(comm)  # keep the callbacks alive as long as the solver
            table = C.byref(self._tab)
        self.h = lib.cedar_amd_dist3_create(handle, table, rank, world, pg, A_local.ptr, nx, ny, nz, nst, C.byref(st),
                                            agglomerate_below, overlap_min)
        if not self.h:
            raise RuntimeError("cedar_amd_dist3_create failed")
        self.nlev_global = lib.cedar_amd_dist3_nlevels(self.h)
        self.chain_levels = lib.cedar_amd_dist3_chain_levels(self.h)
        self.coord = (rank % self.p[0], (rank // self.p[0]) % self.p[1], rank // (self.p[0] * self.p[1]))

    def vcycle(self, x, b):
        lib.cedar_amd_dist3_vcycle(self.h, x.ptr, b.ptr)

    def solve(self, b, x):
        rel = np.zeros(self.max_iter + 1)
        n = lib.cedar_amd_dist3_solve(self.h, b.ptr, x.ptr, rel.ctypes.data)
        return [float(v) for v in rel[: n + 1]]

    def time_relax(self, x, b, n):
        return float(lib.cedar_amd_dist3_time_relax(self.h, x.ptr, b.ptr, n))

    def close(self):
        if self.h:
            lib.cedar_amd_dist3_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class DistSolver2:
    """cdr2::mpi::solver on one rank's GPU (cedar_amd_dist2_*, cedar_amd/csrc/dist2.cpp): point relaxation or zebra line
    relaxation with the lines cut by the ranks; `A_local` = capi.DeviceArray (nst, ny+2, nx+2)"""

    def __init__(self, comm, rank, world, A_local, pgrid=None, relax="point", nrelax_pre=2, nrelax_post=1, min_coarse=3,
                 max_iter=10, tol=1e-8, agglomerate_below=64):
        self.comm, self.rank, self.world = comm, rank, world
        self._A = A_local
        nst = A_local.shape[0]
        ny, nx = (int(v) - 2 for v in A_local.shape[1:])
        st = capi.Settings(capi.RELAX[relax], nrelax_pre, nrelax_post, -1, max_iter, tol, min_coarse, 0, 0)
        capi.plane_settings(st, None)
        self.max_iter = max_iter
        self.p = tuple(pgrid) if pgrid else rank_grid2(world)
        pg = (C.c_int * 2)(*self.p)
        handle, table = None, None
        if comm == "loopback":
            self._tab = _Transport()
            lib.cedar_amd_transport_loopback(C.byref(self._tab), world)
            table = C.byref(self._tab)
        elif isinstance(comm, NativeComm):
            handle = comm.h
        elif comm is not None and world > 1:
            self._tab, self._keep = _table_from(comm)
            table = C.byref(self._tab)
        self.h = lib.cedar_amd_dist2_create(handle, table, rank, world, pg, A_local.ptr, nx, ny, nst, C.byref(st),
                                            agglomerate_below)
        if not self.h:
            raise RuntimeError("cedar_amd_dist2_create failed")
        self.nlev_global = lib.cedar_amd_dist2_nlevels(self.h)
        self.coord = (rank % self.p[0], rank // self.p[0])

    def vcycle(self, x, b):
        lib.cedar_amd_dist2_vcycle(self.h, x.ptr, b.ptr)

    def solve(self, b, x):
        rel = np.zeros(self.max_iter + 1)
        n = lib.cedar_amd_dist2_solve(self.h, b.ptr, x.ptr, rel.ctypes.data)
        return [float(v) for v in rel[: n + 1]]

    def time_relax(self, x, b, n):
        return float(lib.cedar_amd_dist2_time_relax(self.h, x.ptr, b.ptr, n))

    def close(self):
        if self.h:
            lib.cedar_amd_dist2_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
