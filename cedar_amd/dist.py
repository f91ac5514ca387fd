"""Domain-decomposed 3D BoxMG solver: one rank per GPU, halo exchange over RCCL / xGMI issued by the
library itself (cedar_amd/comm.py NativeComm -> cedar_amd_comm_* of include/cedar_amd.h).  No torch in a rank
process: arrays are `capi.DeviceArray`s, every array operation goes through the backend.

What this replaces in the reference: the MPI flavour's
Cartesian block decomposition with a one-cell ghost layer exchanged after every
colour of a sweep, after residual and after interp_add
(src/3d/ftn/mpi/BMG3_SymStd_relax_GS.f90:102-147, ..._residual.f90:130,
..._interp_add.f90:308), the stencil / interpolation ghost updates of the set-up
(..._SETUP_ITLI27_ex.f90:1803, ..._SETUP_interp_OI.f90:418-1074), the global
norm all-reduce (include/cedar/3d/mpi/grid_func.h:41) and the coarsest-grid
gather (include/cedar/3d/mpi/redist_solver.h).  MSG/MPI transport is replaced
wholesale by grouped point-to-point sends/receives to the (at most 26, on the
2x2x2 node 7) neighbouring ranks.

Design rule: *serial equivalence by construction*.  Every rank runs the serial
kernels on its local box (owned points + one ghost layer); ghost layers always
hold the owner's current values when a kernel reads them.  Because the local
extents stay even on every level (512 -> 256 -> ... -> 2 per rank) local and
global parities coincide, so colourings, coarse-point ownership and all index
ranges are those of the single-domain run on the global grid, and the residual
history equals the 1-rank history to rounding -- the reference's own criterion
(test/3d/mpi/test_relax.cc:56-59).

Two places need more than "exchange after the kernel":
  * the fused 27-point row pass relaxes two i-colours back to back; the second
    colour's last (UP) / first (DOWN) point of a row needs the x-neighbour's
    fresh first-colour value.  After the pass the one x-face is exchanged and
    that single column is recomputed (`relax_fixup`; the update of a point does
    not read its own old value, so the recomputation is exact).
  * the interpolation set-up skips coarse index 2 in the "between two coarse
    points" directions because index 1 is a physical boundary in the serial
    code; on a side with a neighbouring rank the lower bound becomes 2
    (`cedar_amd_setup_interp3_phase(..., ilo, jlo, klo)`) and CI ghosts are
    exchanged between the dependent phases.
Coarse levels are latency-bound (a 32^3 block per GPU is microseconds of work between
exchanges), so below `agglomerate_below` points per direction the level is assembled on every
rank by all-gather and the rest of the cycle runs redundantly on the single-domain solver
(one hipGraph replay on the GPU).  This replaces the reference's redistribution solver
(include/cedar/3d/mpi/redist_solver.h) and, at the last level, its coarsest-grid gather.

The orchestration is backend-agnostic.  A backend owns the arrays (opaque here except for `.shape`), the
kernels and the transport: `GpuBackend` (below) = HIP kernels through the C ABI on device arrays + a `comm`
object (NativeComm = RCCL; SocketComm = host-staged rehearsal for ranks sharing one GPU); the test-suite supplies
a CPU backend on the oracle + gloo so that the same code runs on CPU (tests/test_dist_cpu.py).
Backend interface: zeros(shape), buffer(n), fill_zero(a), box_copy(arr, nplanes, boxes, offs, buf, unpack),
p2p(sends, recvs) with items (peer, buffer, offset, count), allgather(send, count, recv), allreduce_sum(float),
side()/wait() for the overlapped exchange, and the kernels.
"""
import ctypes as C
import math

DOWN, UP = 0, 1


# ------------------------------------------------------------------ topology
def rank_grid(world):
    """1 -> 1x1x1, 2 -> 1x1x2, 4 -> 1x1x4 (z slabs: two contiguous faces per rank, and a whole k-parity of
    planes -- both row classes, the plane-fused kernel -- between two exchanges), 8 -> 2x2x2 (BASELINE
    config 5); otherwise the most cubic factorisation with pz >= py >= px."""
    if world <= 4:
        return (1, 1, world)
    best = None
    for pz in range(1, world + 1):
        if world % pz:
            continue
        for py in range(pz, world // pz + 1):
            if (world // pz) % py:
                continue
            px = world // pz // py
            if px < py:
                continue
            key = (px - pz, px)
            if best is None or key < best[0]:
                best = (key, (px, py, pz))
    return best[1][::-1]


class Topology:
    """rank = k*(px*py) + j*px + i  (src/3d/util/topo.cc:82-84)"""

    def __init__(self, rank, world, pgrid=None):
        self.rank, self.world = rank, world
        self.p = tuple(pgrid) if pgrid else rank_grid(world)
        px, py, pz = self.p
        assert px * py * pz == world
        self.coord = (rank % px, (rank // px) % py, rank // (px * py))

    def rank_of(self, c):
        px, py, pz = self.p
        return c[0] + px * (c[1] + py * c[2])

    def neighbours(self):
        out = {}
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if (dx, dy, dz) == (0, 0, 0):
                        continue
                    c = (self.coord[0] + dx, self.coord[1] + dy, self.coord[2] + dz)
                    if all(0 <= c[d] < self.p[d] for d in range(3)):
                        out[(dx, dy, dz)] = self.rank_of(c)
        return out

    def has(self, d, side):
        return 0 <= self.coord[d] + side < self.p[d]


# ------------------------------------------------------------------ halo exchange
def _rng(d, n, recv, has_minus, has_plus):
    """index range (start, stop) along one axis of extent n+2 for neighbour offset d.
    d != 0: send = the owned layer next to that side, recv = the ghost layer on that side.
    d == 0 (tangential): the owned cells, plus the ghost cell on every side that is a PHYSICAL
    boundary -- those ghosts hold values the serial kernels compute there for even extents
    (IICF1 = IIC) and must stay coherent across ranks; a ghost on a side with a neighbouring rank
    belongs to the diagonal neighbour's message, so the boxes of one exchange never overlap."""
    if d == 0:
        return (1 if has_minus else 0, n + 1 if has_plus else n + 2)
    if d < 0:
        return (0, 1) if recv else (1, 2)
    return (n + 1, n + 2) if recv else (n, n + 1)


class Halo:
    """ghost-layer exchange with the (up to 26) neighbouring ranks: pack (one launch) -> one grouped
    send/recv -> unpack (one launch).  Boxes are (i0, j0, k0, ni, nj, nk), 0-based incl. ghost."""

    def __init__(self, topo, n, backend):
        self.topo, self.n, self.be = topo, n, backend
        self.nb = []  # (offset, peer, send box, recv box, size, buffer offset)
        off = 0
        hm = [topo.has(d, -1) for d in range(3)]
        hp = [topo.has(d, +1) for d in range(3)]
        for o, peer in sorted(topo.neighbours().items()):
            sr = [_rng(o[d], n[d], False, hm[d], hp[d]) for d in range(3)]
            rr = [_rng(o[d], n[d], True, hm[d], hp[d]) for d in range(3)]
            sbox = (sr[0][0], sr[1][0], sr[2][0], sr[0][1] - sr[0][0], sr[1][1] - sr[1][0], sr[2][1] - sr[2][0])
            rbox = (rr[0][0], rr[1][0], rr[2][0], rr[0][1] - rr[0][0], rr[1][1] - rr[1][0], rr[2][1] - rr[2][0])
            size = sbox[3] * sbox[4] * sbox[5]
            self.nb.append((o, peer, sbox, rbox, size, off))
            off += size
        self.total = off
        self._buf = {}
        # neighbour groups: "x" = across an x face/edge/corner (offset has dx != 0), "yz" = the others.
        # The interior rows of a row pass read x ghosts but no y/z ghosts, so the "yz" group may still
        # be in flight while they run (DistSolver3._smooth).
        self.groups = {None: list(self.nb),
                       "x": [e for e in self.nb if e[0][0] != 0],
                       "yz": [e for e in self.nb if e[0][0] == 0]}
        self._tabs = {g: ([e[2] for e in nb], [e[3] for e in nb], [e[5] for e in nb]) for g, nb in self.groups.items() if nb}

    def _buffers(self, key, count):
        if key not in self._buf:
            self._buf[key] = (self.be.buffer(max(count, 1)), self.be.buffer(max(count, 1)))
        return self._buf[key]

    def exchange(self, arr, group=None):
        """fill every ghost cell that has an owner on another rank (group None), or only those owned
        by the neighbours of one group ("x" / "yz"); arr: (..., KK, JJ, II).  The groups use disjoint
        parts of the send/receive buffers, so one of each may be in flight at a time."""
        nb = self.groups[group]
        if not nb:
            return
        nplanes = 1
        for v in arr.shape[:-3]:
            nplanes *= int(v)
        sb, rb = self._buffers(nplanes, self.total * nplanes)
        sboxes, rboxes, offs = self._tabs[group]
        self.be.box_copy(arr, nplanes, sboxes, offs, sb, 0)
        self.be.p2p([(e[1], sb, e[5] * nplanes, e[4] * nplanes) for e in nb],
                    [(e[1], rb, e[5] * nplanes, e[4] * nplanes) for e in nb])
        self.be.box_copy(arr, nplanes, rboxes, offs, rb, 1)

    def exchange_sub(self, arr, send, recv, jpar_of, kpar):
        """what one stage of the boundary-first chain has changed (dist3.cpp halo_exchange_sub): of the messages selected
        by send(o) / recv(o) (o = neighbour offset) only the rows with index parity jpar_of(o) and the planes with index
        parity kpar (-1: all).  Local extents are even along split directions, so both ends of a message find the same
        boxes; an empty box is no message.  Boxes carry the row / plane step as entries 6, 7."""
        def restrict(box, jpar):
            i0, j0, k0, ni, nj, nk = box
            sj = sk = 1
            if jpar >= 0:
                f = j0 if (j0 & 1) == jpar else j0 + 1
                nj, j0, sj = max((j0 + nj - f + 1) // 2, 0), f, 2
            if kpar >= 0:
                f = k0 if (k0 & 1) == kpar else k0 + 1
                nk, k0, sk = max((k0 + nk - f + 1) // 2, 0), f, 2
            return (i0, j0, k0, ni, nj, nk, sj, sk), ni * nj * nk
        sb, rb = self._buffers(1, self.total)
        sboxes, soffs, sends, rboxes, roffs, recvs = [], [], [], [], [], []
        for o, peer, sbox, rbox, _size, off in self.nb:
            if send(o):
                bx, c = restrict(sbox, jpar_of(o))
                if c:
                    sboxes.append(bx); soffs.append(off); sends.append((peer, sb, off, c))
            if recv(o):
                bx, c = restrict(rbox, jpar_of(o))
                if c:
                    rboxes.append(bx); roffs.append(off); recvs.append((peer, rb, off, c))
        if sboxes:
            self.be.box_copy(arr, 1, sboxes, soffs, sb, 0)
        self.be.p2p(sends, recvs)
        if rboxes:
            self.be.box_copy(arr, 1, rboxes, roffs, rb, 1)

    def exchange_x(self, arr, to_minus):
        """x faces only (owned j,k).  to_minus: send the first owned column to the -x neighbour and
        receive the +x neighbour's into the high ghost column (UP order); else the mirror image."""
        nx, ny, nz = self.n
        t = self.topo
        if to_minus:
            send_to, send_col, recv_from, recv_col = -1, 1, +1, nx + 1
        else:
            send_to, send_col, recv_from, recv_col = +1, nx, -1, 0
        c = t.coord
        sb, rb = self._buffers(("x", to_minus), ny * nz)
        sends, recvs = [], []
        if t.has(0, send_to):
            self.be.box_copy(arr, 1, [(send_col, 1, 1, 1, ny, nz)], [0], sb, 0)
            sends.append((t.rank_of((c[0] + send_to, c[1], c[2])), sb, 0, ny * nz))
        if t.has(0, recv_from):
            recvs.append((t.rank_of((c[0] + recv_from, c[1], c[2])), rb, 0, ny * nz))
        self.be.p2p(sends, recvs)
        if recvs:
            self.be.box_copy(arr, 1, [(recv_col, 1, 1, 1, ny, nz)], [0], rb, 1)
        return bool(recvs)


# ------------------------------------------------------------------ GPU backend
class GpuBackend:
    """HIP kernels through the C ABI (include/cedar_amd.h) on capi.DeviceArray; transport = `comm`
    (cedar_amd/comm.py: NativeComm = RCCL issued by the library, SocketComm = one-GPU rehearsal)."""

    def __init__(self, comm, device=0):
        from . import capi
        self.capi, self.lib = capi, capi.lib
        self.comm = comm
        capi.set_device(device)
        self._side = None
        self._tabs = {}

    @staticmethod
    def _p(t):
        return C.c_void_p(t.ptr)

    @staticmethod
    def _dims(t):
        KK, JJ, II = t.shape[-3:]
        return C.c_uint(II), C.c_uint(JJ), C.c_uint(KK)

    # ---- arrays
    def zeros(self, shape):
        return self.capi.DeviceArray(shape)

    def buffer(self, n):
        return self.capi.DeviceArray((int(n),))

    def fill_zero(self, a):
        a.zero()

    def from_numpy(self, a):
        return self.capi.DeviceArray.from_numpy(a)

    def to_numpy(self, a):
        return a.numpy()

    def box_copy(self, arr, nplanes, boxes, offs, buf, unpack):
        key = (tuple(boxes), tuple(offs))
        tab = self._tabs.get(key)
        if tab is None:
            tab = self._tabs[key] = ((C.c_int * (6 * len(boxes)))(*[v for b in boxes for v in b]),
                                     (C.c_ulonglong * len(boxes))(*offs))
        KK, JJ, II = arr.shape[-3:]
        self.lib.cedar_amd_box_copy(self._p(arr), C.c_uint(II), C.c_uint(JJ), C.c_uint(KK), nplanes, len(boxes),
                                    tab[0], tab[1], self._p(buf), unpack)

    # ---- transport
    def p2p(self, sends, recvs):
        self.comm.p2p(sends, recvs)

    def allgather(self, send, count, recv):
        self.comm.allgather(send, count, recv)

    def allreduce_sum(self, v):
        return self.comm.allreduce_sum(v)

    # ---- kernels
    def relax_pass(self, A, b, x, sor, jb, kb, efirst, part=0, sides=0):
        """sides: faces of the box with a neighbouring rank (bit 0 -y, 1 +y, 2 -z, 3 +z; 0 = all), see cedar_amd.h"""
        self.lib.cedar_amd_relax3_pass_part(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims(x),
                                            jb, kb, int(efirst), part | (sides << 4))

    def relax_planes(self, A, b, x, sor, kb, up, part=0, sides=0):
        self.lib.cedar_amd_relax3_planes(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims(x), kb, int(up),
                                         part | (sides << 4))

    class _Side:
        """`with backend.side() as h:` issues the enclosed work (library launches and RCCL calls) on a
        side HIP stream ordered after everything already queued on the main stream; `backend.wait(h)`
        orders the main stream after it.  The main stream is the library's current stream (the null
        stream by default); the side stream is non-blocking, so the two really overlap."""

        def __init__(self, be):
            self.be = be

        def __enter__(self):
            be = self.be
            if be._side is None:
                from .comm import Stream
                be._side = Stream()
            self.main = be.lib.cedar_amd_get_stream()
            be.lib.cedar_amd_stream_wait(C.c_void_p(be._side.h), C.c_void_p(self.main))
            be.lib.cedar_amd_set_stream(C.c_void_p(be._side.h))
            return self

        def __exit__(self, *exc):
            self.be.lib.cedar_amd_set_stream(C.c_void_p(self.main))
            return False

    def side(self):
        return GpuBackend._Side(self)

    def wait(self, h):
        # everything queued on the side stream so far (= the work of the last `with side()` block)
        self.lib.cedar_amd_stream_wait(C.c_void_p(self.lib.cedar_amd_get_stream()), C.c_void_p(self._side.h))

    def relax_fixup(self, A, b, x, sor, icol, jb, kb):
        self.lib.cedar_amd_relax3_fixup(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims(x), icol, jb, kb)

    def relax_colour7(self, A, b, x, sor, pts):
        self.lib.cedar_amd_relax3_colour7(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims(x), pts)

    def recip(self, A, sor, solve_copy=False):
        self.lib.BMG3_SymStd_SETUP_recip(self._p(A), self._p(sor), *self._dims(sor), A.shape[0], 2)
        # solve_copy: register a row-interleaved solve copy of a 27-point level for the per-piece entry points
        # (cedar_amd_relax3_prepare; dropped when A is released through cedar_amd_free).  Only the slab decomposition
        # asks for it: its sweeps are the plane-fused passes of the single-GPU solver, where the copy is worth 7 %; on
        # rank grids with an x / y split (four row-class launches) it measured neutral
        # (profiles/r02_dist_orchestration_cost.log)
        if solve_copy and A.shape[0] == 14 and hasattr(A, "ptr"):
            self.lib.cedar_amd_relax3_prepare(self._p(A), self._p(sor), *self._dims(sor))

    def residual(self, A, x, b, r):
        nst = A.shape[0]
        self.lib.BMG3_SymStd_residual(1, 1, int(nst == 4), self._p(x), self._p(b), self._p(A), self._p(r), *self._dims(x), nst)

    def restrict(self, r, bc, P):
        self.lib.BMG3_SymStd_restrict(self._p(r), self._p(bc), self._p(P), *self._dims(r), *self._dims(bc), 0)

    def interp_add(self, x, xc, A, r, P):
        self.lib.BMG3_SymStd_interp_add(self._p(x), self._p(xc), self._p(A), self._p(r), self._p(P),
                                        *self._dims(xc), *self._dims(x), A.shape[0], 0)

    def interp_phase(self, A, P, phase, lo):
        nst = A.shape[0]
        self.lib.cedar_amd_setup_interp3_phase(self._p(A), self._p(P), *self._dims(A), *self._dims(P),
                                               int(nst == 4), nst, phase, lo[0], lo[1], lo[2])

    def galerkin(self, A, Ac, P):
        f = self.lib.BMG3_SymStd_SETUP_ITLI07_ex if A.shape[0] == 4 else self.lib.BMG3_SymStd_SETUP_ITLI27_ex
        f(self._p(A), self._p(Ac), self._p(P), *self._dims(A), *self._dims(Ac), 0)

    def make_serial(self, gA, pre, post, min_coarse, num_levels):
        """single-domain device-resident solver on the gathered level (V-cycle = one graph replay)"""
        capi = self.capi

        class _H:
            def __init__(h):
                h.s = capi.Solver(gA, nrelax_pre=pre, nrelax_post=post, min_coarse=min_coarse,
                                  num_levels=num_levels, share_operator=True)

            def vcycle(h, x, b):
                capi.lib.cedar_amd_solver_vcycle(h.s.h, x.ptr, b.ptr)
        return _H()

    def sumsq(self, r):
        v = self.capi.lib.cedar_amd_l2norm(r.ptr, r.shape[2], r.shape[1], r.shape[0])
        return v * v

    def sync(self):
        self.lib.cedar_amd_device_sync()


# ------------------------------------------------------------------ solver
class Level:
    pass


class DistSolver3:
    """cedar::cdr3::mpi::solver equivalent for Dirichlet problems, point relaxation, V(pre,post)."""

    def __init__(self, backend, topo, A_local, nrelax_pre=2, nrelax_post=1, min_coarse=3, max_iter=10, tol=1e-8,
                 agglomerate_below=64, overlap_min=96, chain=False):
        """A_local: (nst, nz+2, ny+2, nx+2) local part of the global operator (ghost layers are
        filled here by exchange; entries coupling to a neighbouring rank must be present)."""
        self.be, self.topo = backend, topo
        self._gbuf = {}
        self.pre, self.post, self.max_iter, self.tol = nrelax_pre, nrelax_post, max_iter, tol
        self.min_coarse = min_coarse
        self.overlap_min = overlap_min
        # chain: rank grids with an x / y split relax a k-parity as the native driver does where the level takes the
        # partial-sum sweep (dist3.cpp chain_parity): the points next to a neighbouring rank ahead, stage by stage, the rest
        # in one go.  Here it is the statement of that ORDER on a backend with relax_colour_masked (the CPU backend: every
        # update in the reference's arithmetic, so the run must equal the single-domain run like any other).
        self.chain = bool(chain) and (topo.p[0] > 1 or topo.p[1] > 1)
        # faces of the box with a neighbouring rank (bit 0 -y, 1 +y, 2 -z, 3 +z): only rows next to those wait for a halo
        self.sides = (int(topo.has(1, -1)) | int(topo.has(1, +1)) << 1 | int(topo.has(2, -1)) << 2 | int(topo.has(2, +1)) << 3)
        nst = A_local.shape[0]
        n = tuple(int(s) - 2 for s in A_local.shape[1:][::-1])
        p = topo.p
        gn = tuple(n[d] * p[d] for d in range(3))
        # number of levels from the GLOBAL extents (include/cedar/3d/solver.h:54-72)
        ng = 0
        while True:
            ng += 1
            if min((g - 1) // (1 << ng) + 1 for g in gn) < min_coarse:
                break
        self.nlev_global = ng
        # distributed levels 0..la; level la is gathered and handed to the single-domain solver
        la, m = ng - 1, n
        for l in range(1, ng):
            m = tuple(int((v - 1) / 2.0 + 1) if p[d] == 1 else v // 2 for d, v in enumerate(m))
            if min(m[d] for d in range(3) if p[d] > 1 or True) <= agglomerate_below:
                la = l
                break
        self.la = max(la, 1) if ng > 1 else 0
        self.levels = []
        for l in range(self.la + 1):
            L = Level()
            L.n = n
            for d in range(3):
                if p[d] > 1 and l < self.la:
                    assert n[d] % 2 == 0, f"level {l}: local extent {n[d]} in dim {d} must be even"
            shp = (n[2] + 2, n[1] + 2, n[0] + 2)
            L.halo = Halo(topo, n, backend)
            # halo of one row pass in flight under the interior rows of the next (side stream): only
            # where a pass is long enough to hide it and there is a y/z neighbour to talk to
            L.overlap = bool(L.halo.groups["yz"]) and min(n) >= overlap_min and hasattr(backend, "side")
            L.res = backend.zeros(shp)
            L.sor = backend.zeros((2,) + shp)
            if l == 0:
                L.A = A_local
                L.P = L.x = L.b = None
            else:
                L.A = backend.zeros((14,) + shp)
                L.P = backend.zeros((26,) + shp)
                L.x, L.b = backend.zeros(shp), backend.zeros(shp)
            self.levels.append(L)
            n = tuple(int((m - 1) / 2.0 + 1) if p[d] == 1 else m // 2 for d, m in enumerate(n))
        self._setup()

    # ---- set-up (multilevel.h:243-265 with the MPI flavour's ghost updates)
    def _setup(self):
        be, t = self.be, self.topo
        lo = tuple(2 if t.has(d, -1) else 3 for d in range(3))
        L0 = self.levels[0]
        L0.halo.exchange(L0.A)
        for l in range(len(self.levels) - 1):
            F, K = self.levels[l], self.levels[l + 1]
            for phase in range(3):
                be.interp_phase(F.A, K.P, phase, lo)
                K.halo.exchange(K.P)
            be.galerkin(F.A, K.A, K.P)
            K.halo.exchange(K.A)
            if isinstance(be, GpuBackend):
                be.recip(F.A, F.sor, solve_copy=(t.p[0] == 1 and t.p[1] == 1))
            else:
                be.recip(F.A, F.sor)
        # level la: assemble the global operator on every rank; the single-domain solver takes over
        Cl = self.levels[-1]
        self.cn = Cl.n
        p = t.p
        gshape = (Cl.n[2] * p[2] + 2, Cl.n[1] * p[1] + 2, Cl.n[0] * p[0] + 2)
        self.gA = be.zeros((Cl.A.shape[0],) + gshape)
        self._gather_into(Cl.A, self.gA)
        self.gx, self.gb = be.zeros(gshape), be.zeros(gshape)
        self.serial = be.make_serial(self.gA, self.pre, self.post, self.min_coarse, self.nlev_global - self.la)

    def _gather_into(self, local, glob):
        """all-gather the owned block of `local` (..., KK,JJ,II) into the global array: pack the owned box,
        one all-gather, unpack every rank's block at its place (one launch each way)"""
        be, t = self.be, self.topo
        nx, ny, nz = self.cn
        nplanes = 1
        for v in local.shape[:-3]:
            nplanes *= int(v)
        blk = nx * ny * nz
        key = ("gather", nplanes)
        if key not in self._gbuf:
            self._gbuf[key] = (be.buffer(blk * nplanes), be.buffer(blk * nplanes * t.world))
        sb, rb = self._gbuf[key]
        be.box_copy(local, nplanes, [(1, 1, 1, nx, ny, nz)], [0], sb, 0)
        if t.world == 1:
            rb = sb
        else:
            be.allgather(sb, blk * nplanes, rb)
        px, py, pz = t.p
        boxes = []
        for r in range(t.world):
            ci, cj, ck = r % px, (r // px) % py, r // (px * py)
            boxes.append((1 + ci * nx, 1 + cj * ny, 1 + ck * nz, nx, ny, nz))
        be.box_copy(glob, nplanes, boxes, [r * blk for r in range(t.world)], rb, 1)

    # ---- cycle (vcycle.h:57-115)
    def _smooth(self, L, x, b, updown, n):
        be, t = self.be, self.topo
        nst = L.A.shape[0]
        nx = L.n[0]
        pending = None  # handle of the y/z halo exchange in flight on the side stream
        for _ in range(n):
            if nst == 4:
                for c in range(2):
                    be.relax_colour7(L.A, b, x, L.sor, c if updown == UP else 1 - c)
                    L.halo.exchange(x)
                continue
            up = updown == UP
            if self.chain and L.n[0] >= 8 and L.n[1] >= 8:
                for c in range(2):
                    self._chain_parity(L, x, b, c if up else 1 - c, up)
                continue
            if t.p[0] == 1 and t.p[1] == 1:
                # slab decomposition: the second row class of a plane needs nothing from another rank, so the
                # unit between two exchanges is a whole k-parity (plane-fused kernel on big levels); its
                # planes next to a ghost plane wait for the previous parity's halo, the others do not
                for c in range(2):
                    kb = c if up else 1 - c
                    if L.overlap:
                        be.relax_planes(L.A, b, x, L.sor, kb, up, 1, self.sides)
                        if pending is not None:
                            be.wait(pending)
                            pending = None
                        be.relax_planes(L.A, b, x, L.sor, kb, up, 2, self.sides)
                        with be.side() as pending:
                            L.halo.exchange(x)
                    else:
                        be.relax_planes(L.A, b, x, L.sor, kb, up)
                        L.halo.exchange(x)
                continue
            for c in range(4):
                cc = c if up else 3 - c
                jb, kb = cc & 1, cc >> 1
                if L.overlap:
                    # interior rows first: they read no y/z ghost, whose exchange (previous pass) may
                    # still be running on the side stream; then join and do the shell rows
                    be.relax_pass(L.A, b, x, L.sor, jb, kb, up, 1, self.sides)
                    if pending is not None:
                        be.wait(pending)
                        pending = None
                    be.relax_pass(L.A, b, x, L.sor, jb, kb, up, 2, self.sides)
                else:
                    be.relax_pass(L.A, b, x, L.sor, jb, kb, up)
                if t.p[0] > 1:
                    # second i-colour at the x-boundary needs the neighbour's fresh first colour
                    if L.halo.exchange_x(x, to_minus=up):
                        be.relax_fixup(L.A, b, x, L.sor, nx if up else 1, jb, kb)
                if L.overlap:
                    L.halo.exchange(x, "x")  # x ghosts are read by every row of the next pass: not deferred
                    with be.side() as pending:
                        L.halo.exchange(x, "yz")
                else:
                    L.halo.exchange(x)
        if pending is not None:
            be.wait(pending)

    def _chain_parity(self, L, x, b, kb, up):
        """One k-parity of planes, boundary-first (cedar_amd/csrc/dist3.cpp chain_parity, same sets, same exchanges).
        Colours of a parity in sweep order: F rows c1, F rows c2, S rows c1, S rows c2 (c1 = the i-colour relaxed first).
        d = distance of a column from the x face.  Side Q (boundary column is c2): F c1 {d1,d3}, F c2 {d0,d2}, S c1 {d1},
        S c2 {d0}; side P (boundary column is c1): F c1 {d0,d2}, F c2 {d1}, S c1 {d0}.  y: the side whose boundary row is an
        F row: that row; the other side: its F row d1 and its S row d0.  Both colours of a row class run with side Q's
        ghost column still stale; its c2 column d0 is recomputed after the exchange (nobody has read it)."""
        import numpy as np
        be, t = self.be, self.topo
        nx, ny, nz = L.n
        jbF = 0 if up else 1
        jbS = 1 - jbF
        ib1 = 0 if up else 1  # i-colour relaxed first: offsets 1 + ib + 2a
        ib2 = 1 - ib1
        F1, F2, S1, S2, fixc = [], [], [], [], []
        for side in (0, 1):
            if not t.has(0, +1 if side else -1):
                continue
            is_p = (side == 0) == up
            col = (lambda d: nx - d) if side else (lambda d: 1 + d)
            if is_p:
                F1 += [col(0), col(2)]; F2 += [col(1)]; S1 += [col(0)]
            else:
                F1 += [col(1), col(3)]; F2 += [col(0), col(2)]; S1 += [col(1)]; S2 += [col(0)]; fixc += [col(0)]
        rowsF, rowS = [], None
        for side in (0, 1):
            if not t.has(1, +1 if side else -1):
                continue
            is_p = (side == 0) == up
            row = (lambda d: ny - d) if side else (lambda d: 1 + d)
            if is_p:
                rowsF.append(row(0))
            else:
                rowsF.append(row(1)); rowS = row(0)
        shape = (nz + 2, ny + 2, nx + 2)
        ks = slice(1 + kb, nz + 1, 2)

        def points(jb, cols, rows, ib):
            """mask: the listed columns in every row of class jb, and the listed rows whole -- colour ib only"""
            m = np.zeros(shape, dtype=bool)
            for c in cols:
                m[ks, 1 + jb:ny + 1:2, c] = True
            for r in rows:
                m[ks, r, 1:nx + 1] = True
            keep = np.zeros(shape, dtype=bool)
            keep[:, :, 1 + ib:nx + 1:2] = True
            return m & keep

        def colour(ib, jb):
            return 1 + ib + 2 * jb + 4 * kb

        xs = t.p[0] > 1
        jparF, jparS, kpar, qdir = (1 + jbF) & 1, (1 + jbS) & 1, (1 + kb) & 1, (1 if up else -1)
        layer = lambda o: o[2] == 0
        not_p = lambda o: o[2] == 0 and o[0] != -qdir
        not_q = lambda o: o[2] == 0 and o[0] != qdir
        to_p = lambda o: o[2] == 0 and o[0] == -qdir
        to_q = lambda o: o[2] == 0 and o[0] == qdir
        every = lambda o: True
        skip = np.zeros(shape, dtype=bool)
        # F rows: both colours, side Q's ghost column stale
        for ib, cols in ((ib1, F1), (ib2, F2)):
            m = points(jbF, cols, rowsF, ib)
            skip |= m
            be.relax_colour_masked(L.A, b, x, L.sor, colour(ib, jbF), m)
        L.halo.exchange_sub(x, layer, layer, lambda o: jparF, kpar)
        if xs:
            be.relax_colour_masked(L.A, b, x, L.sor, colour(ib2, jbF), points(jbF, fixc, [], ib2))
            # the recomputed column also sits in the boundary row a y neighbour has already received
            L.halo.exchange_sub(x, not_p, not_q, lambda o: jparF, kpar)
        # S rows
        rS = [rowS] if rowS is not None else []
        for ib, cols in ((ib1, S1), (ib2, S2)):
            m = points(jbS, cols, rS, ib)
            skip |= m
            be.relax_colour_masked(L.A, b, x, L.sor, colour(ib, jbS), m)
        if xs:
            L.halo.exchange_sub(x, to_p, to_q, lambda o: jparS, kpar)
            be.relax_colour_masked(L.A, b, x, L.sor, colour(ib2, jbS), points(jbS, fixc, [], ib2))
        # everything else of the parity; it reads no ghost cell of the rank's own z layer
        for jb in (jbF, jbS):
            for ib in (ib1, ib2):
                be.relax_colour_masked(L.A, b, x, L.sor, colour(ib, jb), ~skip)
        L.halo.exchange_sub(x, every, every, lambda o: jparS if o[2] == 0 else -1, kpar)

    def _coarse_solve(self, x, b):
        """levels la.. : gather the right-hand side, one single-domain cycle (or the direct solve when
        la is the coarsest level) from a zero initial guess, keep the own block + ghosts"""
        t = self.topo
        be = self.be
        self._gather_into(b, self.gb)
        be.fill_zero(self.gx)
        self.serial.vcycle(self.gx, self.gb)
        nx, ny, nz = self.cn
        ci, cj, ck = t.coord
        # own block plus ghost layer straight from the global solution
        full = (nx + 2) * (ny + 2) * (nz + 2)
        if "cs" not in self._gbuf:
            self._gbuf["cs"] = be.buffer(full)
        tmp = self._gbuf["cs"]
        be.box_copy(self.gx, 1, [(ci * nx, cj * ny, ck * nz, nx + 2, ny + 2, nz + 2)], [0], tmp, 0)
        be.box_copy(x, 1, [(0, 0, 0, nx + 2, ny + 2, nz + 2)], [0], tmp, 1)

    def _cycle(self, l, x, b):
        be = self.be
        L, K = self.levels[l], self.levels[l + 1]
        self._smooth(L, x, b, DOWN, self.pre)
        be.residual(L.A, x, b, L.res)
        L.halo.exchange(L.res)
        be.restrict(L.res, K.b, K.P)
        be.fill_zero(K.x)
        if l + 1 == len(self.levels) - 1:
            self._coarse_solve(K.x, K.b)
        else:
            self._cycle(l + 1, K.x, K.b)
        be.interp_add(x, K.x, L.A, L.res, K.P)
        L.halo.exchange(x)
        self._smooth(L, x, b, UP, self.post)

    def vcycle(self, x, b):
        if len(self.levels) == 1:
            self._coarse_solve(x, b)
        else:
            self._cycle(0, x, b)

    def _norm(self, r):
        s = self.be.sumsq(r)
        if self.topo.world > 1:
            s = self.be.allreduce_sum(s)
        return math.sqrt(s)

    def solve(self, b, x):
        """multilevel::solve (multilevel.h:277-298); returns [||r0||, rel_1, ...]"""
        L = self.levels[0]
        L.halo.exchange(x)
        self.be.residual(L.A, x, b, L.res)
        r0 = self._norm(L.res)
        hist = [r0]
        for _ in range(self.max_iter):
            self.vcycle(x, b)
            self.be.residual(L.A, x, b, L.res)
            rel = self._norm(L.res) / r0
            hist.append(rel)
            if rel < self.tol:
                break
        return hist
