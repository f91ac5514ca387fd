"""torch-based halo exchange and GPU backend of the 2D domain-decomposed solver (cedar_amd/dist2d.py).

The 2D multi-GPU solver keeps its per-line carry composition as tensor-level host code and
therefore still runs on torch tensors and torch.distributed ("nccl" = RCCL, or gloo for the one-GPU rehearsal).  The
3D solver of the headline configuration does not: cedar_amd/dist.py is torch-free and talks RCCL through the
library's own C ABI (cedar_amd/comm.py).  This module is the former transport of dist.py, kept for dist2d only.
A process that uses it must import torch BEFORE libcedar_amd.so is loaded (both bring a libamdhip64.so.7; the first
one mapped serves both -- DESIGN.md section 7).
"""
import ctypes as C

import torch
import torch.distributed as dist

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
    """ghost-layer exchange with the (up to 26) neighbouring ranks: pack -> grouped isend/irecv ->
    unpack.  Packing is one kernel launch through the C ABI when the backend offers `box_copy`
    (GPU), torch slicing otherwise (CPU tests)."""

    def __init__(self, topo, n, device, staged, backend=None):
        self.topo, self.n, self.device, self.staged = topo, n, device, staged
        self.be = backend if hasattr(backend, "box_copy_tab") else None
        self.nb = []  # (offset, peer, send box, recv box, size, buffer offset); box = (i0,j0,k0,ni,nj,nk)
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
        self._tabs = {}
        if self.be is not None:
            for g, nb in self.groups.items():
                if nb:
                    IntArr, OffArr = C.c_int * (6 * len(nb)), C.c_ulonglong * len(nb)
                    self._tabs[g] = (IntArr(*[v for e in nb for v in e[2]]), IntArr(*[v for e in nb for v in e[3]]),
                                     OffArr(*[e[5] for e in nb]))
        # x-face mini exchange (one box each way), see exchange_x
        nx, ny, nz = n
        self._xface = (1, 1, 1, 1, ny, nz)  # template: i0 is filled in per call

    def _buffers(self, nplanes):
        if nplanes not in self._buf:
            mk = lambda: torch.empty(max(self.total, 1) * nplanes, dtype=torch.float64, device=self.device)
            self._buf[nplanes] = (mk(), mk())
        return self._buf[nplanes]

    def _p2p(self, sends, recvs):
        """sends/recvs: lists of (peer, 1-D contiguous tensor)"""
        if not sends and not recvs:
            return
        if self.staged:  # gloo with device tensors (one-GPU rehearsal): stage through host memory
            hs = [(p, t.cpu()) for p, t in sends]
            hr = [(p, torch.empty(t.shape, dtype=t.dtype)) for p, t in recvs]
            ops = [dist.P2POp(dist.isend, t, p) for p, t in hs] + [dist.P2POp(dist.irecv, t, p) for p, t in hr]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for (_, t), (_, h) in zip(recvs, hr):
                t.copy_(h)
            return
        ops = [dist.P2POp(dist.isend, t, p) for p, t in sends] + [dist.P2POp(dist.irecv, t, p) for p, t in recvs]
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    @staticmethod
    def _view(arr, box):
        i0, j0, k0, ni, nj, nk = box
        return arr[..., k0:k0 + nk, j0:j0 + nj, i0:i0 + ni]

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
        sb, rb = self._buffers(nplanes)
        if self.be is not None:
            sboxes, rboxes, offs = self._tabs[group]
            self.be.box_copy_tab(arr, nplanes, len(nb), sboxes, offs, sb, 0)
        else:
            for o, peer, sbox, rbox, size, off in nb:
                sb[off * nplanes:(off + size) * nplanes].copy_(self._view(arr, sbox).reshape(-1))
        sends = [(e[1], sb[e[5] * nplanes:(e[5] + e[4]) * nplanes]) for e in nb]
        recvs = [(e[1], rb[e[5] * nplanes:(e[5] + e[4]) * nplanes]) for e in nb]
        self._p2p(sends, recvs)
        if self.be is not None:
            self.be.box_copy_tab(arr, nplanes, len(nb), rboxes, offs, rb, 1)
        else:
            for o, peer, sbox, rbox, size, off in nb:
                v = self._view(arr, rbox)
                v.copy_(rb[off * nplanes:(off + size) * nplanes].reshape(v.shape))

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
        key = ("x", to_minus)
        if key not in self._buf:
            mk = lambda: torch.empty(ny * nz, dtype=torch.float64, device=self.device)
            self._buf[key] = (mk(), mk())
        sb, rb = self._buf[key]
        sends, recvs = [], []
        if t.has(0, send_to):
            sb.copy_(arr[1:nz + 1, 1:ny + 1, send_col].reshape(-1))
            sends.append((t.rank_of((c[0] + send_to, c[1], c[2])), sb))
        if t.has(0, recv_from):
            recvs.append((t.rank_of((c[0] + recv_from, c[1], c[2])), rb))
        self._p2p(sends, recvs)
        if recvs:
            arr[1:nz + 1, 1:ny + 1, recv_col].copy_(rb.reshape(nz, ny))
        return bool(recvs)


# ------------------------------------------------------------------ GPU backend
class GpuBackend:
    """HIP kernels through the C ABI (include/cedar_amd.h) on torch CUDA tensors."""

    def __init__(self, device):
        from cedar_amd import capi
        self.capi, self.lib = capi, capi.lib
        self.device = device
        capi.set_device(device.index if device.index is not None else 0)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    @staticmethod
    def _dims(t):
        KK, JJ, II = t.shape[-3:]
        return C.c_uint(II), C.c_uint(JJ), C.c_uint(KK)

    def zeros(self, shape):
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    def relax_pass(self, A, b, x, sor, jb, kb, efirst, part=0, sides=0):
        """sides: faces of the box with a neighbouring rank (bit 0 -y, 1 +y, 2 -z, 3 +z; 0 = all), see cedar_amd.h"""
        self.lib.cedar_amd_relax3_pass_part(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims(x),
                                            jb, kb, int(efirst), part | (sides << 4))

    def relax_planes(self, A, b, x, sor, kb, up, part=0, sides=0):
        self.lib.cedar_amd_relax3_planes(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims(x), kb, int(up),
                                         part | (sides << 4))

    class _Side:
        """`with backend.side() as h:` issues the enclosed work (library launches and collectives) on a
        side HIP stream ordered after everything already queued on the main stream; `backend.wait(h)`
        orders the main stream after it.  The main stream is the null stream the library and torch share;
        torch's side streams are non-blocking, so the two really overlap."""

        def __init__(self, be):
            self.be = be

        def __enter__(self):
            be = self.be
            if be._side is None:
                be._side = torch.cuda.Stream(device=be.device)
            self.main = torch.cuda.current_stream(be.device)
            ev = torch.cuda.Event()
            ev.record(self.main)
            be._side.wait_event(ev)
            self.ctx = torch.cuda.stream(be._side)
            self.ctx.__enter__()
            self.prev = be.lib.cedar_amd_get_stream()
            be.lib.cedar_amd_set_stream(C.c_void_p(be._side.cuda_stream))
            return self

        def __exit__(self, *exc):
            be = self.be
            self.done = torch.cuda.Event()
            self.done.record(be._side)
            be.lib.cedar_amd_set_stream(C.c_void_p(self.prev))
            self.ctx.__exit__(*exc)
            return False

    _side = None

    def side(self):
        return GpuBackend._Side(self)

    def wait(self, h):
        torch.cuda.current_stream(self.device).wait_event(h.done)

    def relax_fixup(self, A, b, x, sor, icol, jb, kb):
        self.lib.cedar_amd_relax3_fixup(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims(x), icol, jb, kb)

    def relax_colour7(self, A, b, x, sor, pts):
        self.lib.cedar_amd_relax3_colour7(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims(x), pts)

    def recip(self, A, sor):
        self.lib.BMG3_SymStd_SETUP_recip(self._p(A), self._p(sor), *self._dims(sor), A.shape[0], 2)

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
                capi.lib.cedar_amd_solver_vcycle(h.s.h, x.data_ptr(), b.data_ptr())
        return _H()

    # ---- 2D (cedar_amd/dist2d.py)
    @staticmethod
    def _dims2(t):
        JJ, II = t.shape[-2:]
        return C.c_uint(II), C.c_uint(JJ)

    def relax_pass2(self, A, b, x, sor, jb, efirst):
        self.lib.cedar_amd_relax2_pass(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims2(x), jb, int(efirst))

    def relax_fixup2(self, A, b, x, sor, icol, jb):
        self.lib.cedar_amd_relax2_fixup(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims2(x), icol, jb)

    def relax_colour5(self, A, b, x, sor, jo):
        self.lib.cedar_amd_relax2_colour5(self._p(A), self._p(b), self._p(x), self._p(sor), *self._dims2(x), jo)

    def recip2(self, A, sor):
        self.lib.BMG2_SymStd_SETUP_recip(self._p(A), self._p(sor), *self._dims2(sor), A.shape[0], 2)

    def residual2(self, A, x, b, r):
        nst = A.shape[0]
        II, JJ = self._dims2(x)
        i = lambda v: C.byref(C.c_int(v))
        self.lib.BMG2_SymStd_residual(i(0), self._p(A), self._p(b), self._p(x), self._p(r), C.byref(II), C.byref(JJ),
                                      i(0), i(int(nst == 3)), i(nst), i(0), i(0), i(0), i(0))

    def restrict2(self, r, bc, P):
        JJ, II = r.shape
        JJC, IIC = bc.shape
        self.lib.BMG2_SymStd_restrict(self._p(r), self._p(bc), self._p(P), II, JJ, IIC, JJC, 0)

    def interp_add2(self, x, xc, A, r, P):
        self.lib.BMG2_SymStd_interp_add(self._p(x), self._p(xc), self._p(r), self._p(A), self._p(P),
                                        *self._dims2(xc), *self._dims2(x), A.shape[0], 0)

    def interp_phase2(self, A, P, phase, lo):
        nst = A.shape[0]
        self.lib.cedar_amd_setup_interp2_phase(self._p(A), self._p(P), *self._dims2(A), *self._dims2(P),
                                               int(nst == 3), nst, phase, lo[0], lo[1])

    def galerkin2(self, A, Ac, P):
        nst = A.shape[0]
        self.lib.BMG2_SymStd_SETUP_ITLI_ex(self._p(A), self._p(Ac), self._p(P), *self._dims2(A), *self._dims2(Ac),
                                           int(nst == 3), nst, 0)

    def make_serial2(self, gA, relax, pre, post, min_coarse, num_levels):
        capi = self.capi

        class _H:
            def __init__(h):
                h.s = capi.Solver(gA, relax=relax, nrelax_pre=pre, nrelax_post=post, min_coarse=min_coarse,
                                  num_levels=num_levels, share_operator=True)

            def vcycle(h, x, b):
                capi.lib.cedar_amd_solver_vcycle(h.s.h, x.data_ptr(), b.data_ptr())
        return _H()

    def sumsq2(self, r):
        v = self.capi.lib.cedar_amd_l2norm(r.data_ptr(), r.shape[1], r.shape[0], 1)
        return v * v

    def affine_lines(self, c, a, div, reverse):
        """y_i = a_i y_prev + c_i (/ div_i) per row of the (lines, n) tensors; returns y (new tensor)"""
        y = c.clone()
        nl, n = y.shape
        a = a.contiguous()
        self.lib.cedar_amd_affine_lines(self._p(y), self._p(a), self._p(div.contiguous()) if div is not None else None,
                                        nl, n, n, int(bool(reverse)))
        return y

    def lines_rhs2(self, A, b, x, d, lb):
        """(lines of colour lb, positions): b - (off-line part of A) x, computed by the library"""
        JJ, II = x.shape
        nl = ((JJ - 2 - lb + 1) // 2) if d == 0 else ((II - 2 - lb + 1) // 2)
        n = II - 2 if d == 0 else JJ - 2
        out = torch.empty((nl, n), dtype=torch.float64, device=self.device)
        self.lib.cedar_amd_lines_rhs2(self._p(A), self._p(b), self._p(x), self._p(out), *self._dims2(x), A.shape[0], d, lb)
        return out

    def lines_carry(self, y, p, c):
        nl, n = y.shape
        self.lib.cedar_amd_lines_carry(self._p(y), self._p(p), self._p(c.contiguous()), nl, n, n)
        return y

    def lines_store2(self, xs, x, d, lb):
        self.lib.cedar_amd_lines_store2(self._p(xs.contiguous()), self._p(x), *self._dims2(x), d, lb)

    def box_copy_tab(self, arr, nplanes, nboxes, boxes, offs, buf, unpack):
        KK, JJ, II = arr.shape[-3:]
        self.lib.cedar_amd_box_copy(self._p(arr), C.c_uint(II), C.c_uint(JJ), C.c_uint(KK), nplanes, nboxes,
                                    boxes, offs, self._p(buf), unpack)

    def sumsq(self, r):
        v = self.capi.lib.cedar_amd_l2norm(r.data_ptr(), r.shape[2], r.shape[1], r.shape[0])
        return v * v

    def sync(self):
        self.capi.sync()


