"""Domain-decomposed 2D BoxMG solver: one rank per GPU on a px x py rank grid, halo exchange over
torch.distributed (backend "nccl" = RCCL over xGMI).  SURVEY.md section 8f-4.

What this replaces in the reference: the 2D MPI flavour (include/cedar/2d/mpi/solver.h,
src/2d/ftn/mpi/BMG2_SymStd_relax_GS.f90, ..._residual.f90, ..._interp_add.f90,
..._SETUP_interp_OI.f90, ..._SETUP_ITLI_ex.f90) and its distributed line relaxation
(src/2d/ftn/mpi/BMG2_SymStd_relax_lines_x.f90 / _y.f90 with the multilevel tridiagonal solves of
include/cedar/2d/mpi/ml_relax.h).  Same design rule as cedar_amd/dist.py (the 3D solver, whose
Topology and Halo classes are reused on (1, JJ, II) views of the 2D arrays): *serial equivalence by
construction* -- local extents stay even on every distributed level so local and global parities
coincide, ghost layers hold the owner's current values whenever a kernel reads them, and the N-rank
residual history equals the single-domain history on the same global problem to rounding.

Point relaxation: the fused nine-point row pass relaxes both i-colours of a row class; with px > 1 the
second colour's boundary column needs the x-neighbour's fresh first colour (one x-face exchange + a
column fix-up, as in 3D).  Five-point operators relax one red-black colour per exchange.

Line relaxation (x lines cut by the px ranks of a row of the rank grid, y lines by the py ranks of a
column): the L D L^T factors of a whole line are computed segment by segment along the line (set-up,
a pipeline of px or py steps).  A solve is two first-order affine recurrences; each rank runs its
segment from a zero carry, the ranks of the line exchange (value leaving the segment, product of
the segment's multipliers) with one all-gather, every rank composes the carry entering its segment
and adds carry x (running product) -- the same decomposition the device kernels use between the
tiles of one line, with the rank in the role of the tile.  This re-associates the recurrences like
the single-GPU scan does (agreement with sequential DPTTRS to ~1e-12 of the line's maximum).

Coarse levels are gathered below `agglomerate_below` points per direction per rank and handed to the
single-domain device solver, as in 3D.
"""
import math

import torch
import torch.distributed as dist

from _torch_dist import Halo
from cedar_amd.dist import Topology

DOWN, UP = 0, 1


def rank_grid2(world):
    """1 -> 1x1, 2 -> 1x2, 4 -> 2x2, 8 -> 2x4: y is split first (y faces are contiguous rows and the
    fused row pass needs no x fix-up while px = 1)"""
    best = None
    for py in range(1, world + 1):
        if world % py:
            continue
        px = world // py
        if px > py:
            continue
        if best is None or py - px < best[0]:
            best = (py - px, (px, py))
    return best[1]


class Level:
    pass


def v3(t):
    """(…, JJ, II) -> (…, 1, JJ, II): the 3D halo machinery on a 2D array"""
    return t.unsqueeze(-3)


class DistSolver2:
    """cedar::cdr2::mpi::solver equivalent for Dirichlet problems: V(pre,post), point relaxation or
    line relaxation in x, y or both."""

    def __init__(self, backend, topo, A_local, relax="point", nrelax_pre=2, nrelax_post=1, min_coarse=3,
                 max_iter=10, tol=1e-8, agglomerate_below=64):
        """A_local: (nst, ny+2, nx+2) local part of the global operator; topo: Topology with pz = 1"""
        assert topo.p[2] == 1
        self.be, self.topo, self.relax = backend, topo, relax
        self.pre, self.post, self.max_iter, self.tol = nrelax_pre, nrelax_post, max_iter, tol
        self.min_coarse = min_coarse
        staged = dist.is_initialized() and dist.get_backend() == "gloo" and A_local.is_cuda
        n = (int(A_local.shape[2]) - 2, int(A_local.shape[1]) - 2)
        p = topo.p[:2]
        gn = tuple(n[d] * p[d] for d in range(2))
        ng = 0
        while True:  # include/cedar/2d/solver.h:57-73 on the GLOBAL extents
            ng += 1
            if min((g - 1) // (1 << ng) + 1 for g in gn) < min_coarse:
                break
        self.nlev_global = ng
        la, m = ng - 1, n
        for l in range(1, ng):
            m = tuple(int((v - 1) / 2.0 + 1) if p[d] == 1 else v // 2 for d, v in enumerate(m))
            if min(m) <= agglomerate_below:
                la = l
                break
        self.la = max(la, 1) if ng > 1 else 0
        self.levels = []
        for l in range(self.la + 1):
            L = Level()
            L.n = n
            for d in range(2):
                if p[d] > 1 and l < self.la:
                    assert n[d] % 2 == 0, f"level {l}: local extent {n[d]} in dim {d} must be even"
            shp = (n[1] + 2, n[0] + 2)
            L.halo = Halo(topo, (n[0], n[1], -1), A_local.device, staged, backend)
            L.res = backend.zeros(shp)
            L.sor = backend.zeros((2,) + shp)
            L.sor_y = backend.zeros((2,) + shp) if relax == "line-xy" else None
            if l == 0:
                L.A, L.P, L.x, L.b = A_local, None, None, None
            else:
                L.A = backend.zeros((5,) + shp)
                L.P = backend.zeros((8,) + shp)
                L.x, L.b = backend.zeros(shp), backend.zeros(shp)
            self.levels.append(L)
            n = tuple(int((v - 1) / 2.0 + 1) if p[d] == 1 else v // 2 for d, v in enumerate(n))
        self._line_groups()
        self._setup()

    # ---- process groups of the ranks that share a line (a row / a column of the rank grid)
    def _line_groups(self):
        t = self.topo
        px, py = t.p[:2]
        self.row_group = self.col_group = None
        if self.relax == "point" or t.world == 1:
            return
        # every rank must create every group (torch.distributed semantics)
        for j in range(py):
            ranks = [t.rank_of((i, j, 0)) for i in range(px)]
            g = dist.new_group(ranks) if px > 1 else None
            if j == t.coord[1]:
                self.row_group = g
        for i in range(px):
            ranks = [t.rank_of((i, j, 0)) for j in range(py)]
            g = dist.new_group(ranks) if py > 1 else None
            if i == t.coord[0]:
                self.col_group = g

    # ---- set-up (multilevel.h:243-265 with the MPI flavour's ghost updates)
    def _setup(self):
        be, t = self.be, self.topo
        lo = tuple(2 if t.has(d, -1) else 3 for d in range(2))
        L0 = self.levels[0]
        L0.halo.exchange(v3(L0.A))
        for l in range(len(self.levels) - 1):
            F, K = self.levels[l], self.levels[l + 1]
            for phase in range(2):
                be.interp_phase2(F.A, K.P, phase, lo)
                K.halo.exchange(v3(K.P))
            be.galerkin2(F.A, K.A, K.P)
            K.halo.exchange(v3(K.A))
            self._setup_relax(F)
        Cl = self.levels[-1]
        self.cn = Cl.n
        p = t.p
        gshape = (Cl.n[1] * p[1] + 2, Cl.n[0] * p[0] + 2)
        self.gA = be.zeros((Cl.A.shape[0],) + gshape)
        self._gather_into(Cl.A, self.gA)
        self.gx, self.gb = be.zeros(gshape), be.zeros(gshape)
        self.serial = be.make_serial2(self.gA, self.relax, self.pre, self.post, self.min_coarse, self.nlev_global - self.la)

    def _setup_relax(self, F):
        be = self.be
        if self.relax == "point":
            be.recip2(F.A, F.sor)
            return
        if self.relax in ("line-x", "line-xy"):
            F.lx = DistLines(be, self.topo, 0, self.row_group, F.A, F.sor, F.n)
        if self.relax == "line-y":
            F.ly = DistLines(be, self.topo, 1, self.col_group, F.A, F.sor, F.n)
        if self.relax == "line-xy":
            F.ly = DistLines(be, self.topo, 1, self.col_group, F.A, F.sor_y, F.n)

    def _gather_into(self, local, glob):
        t = self.topo
        nx, ny = self.cn
        own = local[..., 1:ny + 1, 1:nx + 1].contiguous()
        if t.world == 1:
            parts = [own]
        else:
            staged = own.is_cuda and dist.get_backend() == "gloo"
            src = own.cpu() if staged else own
            parts = [torch.empty_like(src) for _ in range(t.world)]
            dist.all_gather(parts, src)
        px = t.p[0]
        for r, blk in enumerate(parts):
            ci, cj = r % px, r // px
            glob[..., 1 + cj * ny:1 + (cj + 1) * ny, 1 + ci * nx:1 + (ci + 1) * nx].copy_(blk)

    # ---- the x-face mini exchange of the fused row pass
    def _exchange_x(self, L, x, to_minus):
        t = self.topo
        nx, ny = L.n
        c = t.coord
        send_to, send_col, recv_from, recv_col = (-1, 1, +1, nx + 1) if to_minus else (+1, nx, -1, 0)
        sends, recvs = [], []
        rb = None
        if t.has(0, send_to):
            sends.append((t.rank_of((c[0] + send_to, c[1], 0)), x[1:ny + 1, send_col].contiguous()))
        if t.has(0, recv_from):
            rb = torch.empty(ny, dtype=torch.float64, device=x.device)
            recvs.append((t.rank_of((c[0] + recv_from, c[1], 0)), rb))
        L.halo._p2p(sends, recvs)
        if rb is not None:
            x[1:ny + 1, recv_col].copy_(rb)
        return rb is not None

    # ---- cycle
    def _smooth(self, L, x, b, updown, n):
        be, t = self.be, self.topo
        nst = L.A.shape[0]
        nx = L.n[0]
        down = updown == DOWN
        for _ in range(n):
            if self.relax != "point":
                # multilevel.h:165-222: pre = DOWN sweeps (line-xy: x then y), post = UP (y then x)
                order = {"line-x": "x", "line-y": "y", "line-xy": "xy" if down else "yx"}[self.relax]
                for d in order:
                    (L.lx if d == "x" else L.ly).relax(x, b, updown, L.halo)
                continue
            if nst == 3:
                for c in range(2):
                    be.relax_colour5(L.A, b, x, L.sor, 2 + c if down else 3 - c)
                    L.halo.exchange(v3(x))
                continue
            for c in range(2):
                jb = c if down else 1 - c  # DOWN: rows J = 2,4,.. first, even 1-based i first
                be.relax_pass2(L.A, b, x, L.sor, jb, down)
                if t.p[0] > 1 and self._exchange_x(L, x, to_minus=down):
                    be.relax_fixup2(L.A, b, x, L.sor, nx if down else 1, jb)
                L.halo.exchange(v3(x))

    def _coarse_solve(self, x, b):
        t = self.topo
        self._gather_into(b, self.gb)
        self.gx.zero_()
        self.serial.vcycle(self.gx, self.gb)
        nx, ny = self.cn
        ci, cj = t.coord[:2]
        x.copy_(self.gx[cj * ny:cj * ny + ny + 2, ci * nx:ci * nx + nx + 2])

    def _cycle(self, l, x, b):
        be = self.be
        L, K = self.levels[l], self.levels[l + 1]
        self._smooth(L, x, b, DOWN, self.pre)
        be.residual2(L.A, x, b, L.res)
        L.halo.exchange(v3(L.res))
        be.restrict2(L.res, K.b, K.P)
        K.x.zero_()
        if l + 1 == len(self.levels) - 1:
            self._coarse_solve(K.x, K.b)
        else:
            self._cycle(l + 1, K.x, K.b)
        be.interp_add2(x, K.x, L.A, L.res, K.P)
        L.halo.exchange(v3(x))
        self._smooth(L, x, b, UP, self.post)

    def vcycle(self, x, b):
        if len(self.levels) == 1:
            self._coarse_solve(x, b)
        else:
            self._cycle(0, x, b)

    def _norm(self, r):
        s = torch.tensor([self.be.sumsq2(r)], dtype=torch.float64)
        if self.topo.world > 1:
            if dist.get_backend() == "nccl":
                s = s.to(r.device)
            dist.all_reduce(s)
        return math.sqrt(float(s.item()))

    def solve(self, b, x):
        """multilevel::solve (multilevel.h:277-298); returns [||r0||, rel_1, ...]"""
        L = self.levels[0]
        L.halo.exchange(v3(x))
        self.be.residual2(L.A, x, b, L.res)
        r0 = self._norm(L.res)
        hist = [r0]
        for _ in range(self.max_iter):
            self.vcycle(x, b)
            self.be.residual2(L.A, x, b, L.res)
            rel = self._norm(L.res) / r0
            hist.append(rel)
            if rel < self.tol:
                break
        return hist


class DistLines:
    """Zebra line relaxation along direction d (0 = x lines, 1 = y lines) with the lines cut by the
    ranks of `group` (None: the direction is not split, every line is local).

    Factors (set-up): d'_i = d_i - e_{i-1}^2 / d'_{i-1}, e'_i = e_i / d'_i along the whole line
    (DPTTRF, BMG2_SymStd_SETUP_lines_x.f90:68-87); the rank owning segment r waits for the last pivot of
    segment r-1.  Stored like the serial factors (sor planes: pivots, scaled couplings) plus the running
    products of the multipliers of both sweeps, which the solves need to apply a carry:
        forward  y_i = rhs_i - e'_{i-1} y_{i-1}     =>  y_i = y0_i + pf_i * y_in,  pf_i = prod_{k<=i} (-e'_{k-1})
        backward x_i = y_i / d'_i - e'_i x_{i+1}    =>  x_i = x0_i + pb_i * x_in,  pb_i = prod_{k>=i} (-e'_k)
    """

    def __init__(self, be, topo, d, group, A, sor, n):
        self.be, self.topo, self.d, self.group, self.A, self.sor, self.n = be, topo, d, group, A, sor, n
        self.nseg = topo.p[d]
        self.seg = topo.coord[d]
        self._setup()

    # arrays are handled as (line index, position along the line): x lines = rows of the (JJ, II)
    # array, y lines = columns (a transposed view; the backends work on contiguous copies)
    def _lines(self, a):
        return a if self.d == 0 else a.transpose(-1, -2)

    def _setup(self):
        be, t, d = self.be, self.topo, self.d
        nx, ny = self.n
        npos = self.n[d]               # unknowns of a line segment
        A = self.A
        diag = self._lines(A[0])[1:-1, 1:-1].contiguous()                      # (lines, npos)
        # coupling between position i-1 and i is stored at i (KW for x lines, KS for y lines), incl. i = first owned
        cpl = self._lines(A[1 if d == 0 else 2])[1:-1, 1:npos + 2].contiguous()  # (lines, npos+1): positions 1..npos+1
        off = -cpl                                                              # e_{i-1} of position i (reference sign)
        nl = diag.shape[0]
        dev = diag.device
        piv_in = torch.zeros(nl, dtype=torch.float64, device=dev)              # d'_{last} of the previous segment
        has_prev, has_next = t.has(d, -1), t.has(d, +1)
        c = list(t.coord)
        if has_prev:
            c2 = list(c); c2[d] -= 1
            self._recv(piv_in, t.rank_of(tuple(c2)))
        # sequential recurrence along the segment (vectorised over the lines)
        dp = torch.empty_like(diag)
        es = torch.zeros_like(diag)    # e'_{i-1}: scaled coupling to the previous unknown (0 at the very first of the line)
        prev = piv_in
        for i in range(npos):
            e = off[:, i]
            if i == 0 and not has_prev:
                dp[:, 0] = diag[:, 0]
            else:
                en = e / prev
                es[:, i] = en
                dp[:, i] = diag[:, i] - en * e
            prev = dp[:, i]
        if has_next:
            c2 = list(c); c2[d] += 1
            self._send(prev.contiguous(), t.rank_of(tuple(c2)))
        # scaled coupling leaving the segment to the right: e'_{npos} = e_{npos} / d'_{npos}
        self.e_out = (off[:, npos] / prev) if has_next else torch.zeros(nl, dtype=torch.float64, device=dev)
        self.dp, self.es = dp, es
        # running products for the carries
        self.pf = torch.cumprod(-es, dim=1) if has_prev else torch.zeros_like(es)
        e_next = torch.cat([es[:, 1:], self.e_out[:, None]], dim=1)            # e'_i of position i (coupling to i+1)
        self.e_next = e_next
        self.pb = torch.flip(torch.cumprod(torch.flip(-e_next, [1]), dim=1), [1]) if has_next else torch.zeros_like(es)

    def _send(self, tns, peer):
        tt = tns.cpu() if (tns.is_cuda and dist.get_backend() == "gloo") else tns
        dist.send(tt, peer)

    def _recv(self, tns, peer):
        if tns.is_cuda and dist.get_backend() == "gloo":
            h = torch.empty(tns.shape, dtype=tns.dtype)
            dist.recv(h, peer)
            tns.copy_(h)
        else:
            dist.recv(tns, peer)

    def _gather(self, v):
        """all-gather of a (lines, k) tensor over the ranks of the line -> list indexed by segment"""
        if self.group is None:
            return [v]
        staged = v.is_cuda and dist.get_backend() == "gloo"
        src = v.cpu() if staged else v.contiguous()
        parts = [torch.empty_like(src) for _ in range(self.nseg)]
        dist.all_gather(parts, src, group=self.group)
        return [p.to(v.device) for p in parts] if staged else parts

    # _rhs / _store: the same pieces in plain tensor operations for back ends without the library (CPU tests)
    def _rhs(self, x, b, lb):
        """right-hand sides of the lines of colour lb (0-based interior parity): b - (off-line part of A) x,
        reference term order (relax_lines_x.f90:104-109 / relax_lines_y.f90:103-107); -> (lines, positions)"""
        A = self.A
        nx, ny = self.n
        nine = A.shape[0] == 5
        KW, KS, KSW, KNW = 1, 2, 3, 4
        if self.d == 0:
            r, rm, rp = slice(1 + lb, ny + 1, 2), slice(lb, ny, 2), slice(2 + lb, ny + 2, 2)
            c, cm, cp = slice(1, nx + 1), slice(0, nx), slice(2, nx + 2)
            s = b[r, c] + A[KS][r, c] * x[rm, c]
            s = s + A[KS][rp, c] * x[rp, c]
            if nine:
                s = s + A[KSW][r, c] * x[rm, cm]
                s = s + A[KNW][r, cp] * x[rm, cp]
                s = s + A[KNW][rp, c] * x[rp, cm]
                s = s + A[KSW][rp, cp] * x[rp, cp]
            return s.contiguous()
        c, cm, cp = slice(1 + lb, nx + 1, 2), slice(lb, nx, 2), slice(2 + lb, nx + 2, 2)
        r, rm, rp = slice(1, ny + 1), slice(0, ny), slice(2, ny + 2)
        s = b[r, c] + A[KW][r, c] * x[r, cm]
        s = s + A[KW][r, cp] * x[r, cp]
        if nine:
            s = s + A[KSW][r, c] * x[rm, cm]
            s = s + A[KNW][r, cp] * x[rm, cp]
            s = s + A[KNW][rp, c] * x[rp, cm]
            s = s + A[KSW][rp, cp] * x[rp, cp]
        return s.transpose(0, 1).contiguous()

    def _store(self, x, xs, lb):
        nx, ny = self.n
        if self.d == 0:
            x[1 + lb:ny + 1:2, 1:nx + 1] = xs
        else:
            x[1:ny + 1, 1 + lb:nx + 1:2] = xs.transpose(0, 1)

    def relax(self, x, b, updown, halo):
        """one zebra sweep: DOWN relaxes lines 3,5,.. then 2,4,.. (1-based), UP the reverse
        (relax_lines_x.f90:82-97); halo exchange after each colour"""
        be = self.be
        for c in range(2):
            lb = (1 - c) if updown == DOWN else c          # 0-based interior line parity
            sel = slice(lb, None, 2)
            if self.dp[sel].shape[0] == 0:
                halo.exchange(v3(x))
                continue
            rhs = be.lines_rhs2(self.A, b, x, self.d, lb) if hasattr(be, "lines_rhs2") else self._rhs(x, b, lb)
            dp, es, en, pf, pb = (v[sel].contiguous() for v in (self.dp, self.es, self.e_next, self.pf, self.pb))
            # forward sweep from a zero carry, then the carry entering this segment
            y = be.affine_lines(rhs, -es, None, False)
            parts = self._gather(torch.stack([y[:, -1], pf[:, -1]], dim=1))
            y_in = torch.zeros_like(y[:, 0])
            for r in range(self.seg):                      # compose the segments to the left
                y_in = parts[r][:, 0] + parts[r][:, 1] * y_in
            if self.seg > 0:
                y = be.lines_carry(y, pf, y_in) if hasattr(be, "lines_carry") else y + pf * y_in[:, None]
            # backward sweep from a zero carry, then the carry entering from the right
            xs = be.affine_lines(y, -en, dp, True)
            parts = self._gather(torch.stack([xs[:, 0], pb[:, 0]], dim=1))
            x_in = torch.zeros_like(xs[:, 0])
            for r in range(self.nseg - 1, self.seg, -1):
                x_in = parts[r][:, 0] + parts[r][:, 1] * x_in
            if self.seg < self.nseg - 1:
                xs = be.lines_carry(xs, pb, x_in) if hasattr(be, "lines_carry") else xs + pb * x_in[:, None]
            if hasattr(be, "lines_store2"):
                be.lines_store2(xs, x, self.d, lb)
            else:
                self._store(x, xs, lb)
            halo.exchange(v3(x))
