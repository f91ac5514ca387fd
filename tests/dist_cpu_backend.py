"""TEST INFRASTRUCTURE: CPU backend for cedar_amd.dist.DistSolver3 built on the oracle, so that
the distributed orchestration (topology, halo plans, pass sequencing, coarse gather) runs under
gloo on CPU.  Mirrors cedar_amd.dist.GpuBackend method for method."""
import numpy as np
import torch

from pyoracle import Oracle


class CpuBackend:
    def __init__(self):
        self.O = Oracle()
        self.device = torch.device("cpu")

    @staticmethod
    def _n(t):
        a = t.numpy()
        assert a.flags["C_CONTIGUOUS"]
        return a

    def zeros(self, shape):
        return torch.zeros(shape, dtype=torch.float64)

    # ---- array / transport interface of cedar_amd.dist (gloo on CPU tensors)
    def buffer(self, n):
        return torch.zeros(int(n), dtype=torch.float64)

    def fill_zero(self, a):
        a.zero_()

    @staticmethod
    def _view(arr, box):
        i0, j0, k0, ni, nj, nk = box[:6]
        sj, sk = (box[6], box[7]) if len(box) == 8 else (1, 1)  # row / plane step (cedar_amd_box_copy_strided)
        return arr[..., k0:k0 + nk * sk:sk, j0:j0 + nj * sj:sj, i0:i0 + ni]

    def box_copy(self, arr, nplanes, boxes, offs, buf, unpack):
        # buffer layout of cedar_amd_box_copy: box b at offs[b]*nplanes, plane-major inside
        for box, off in zip(boxes, offs):
            v = self._view(arr, box)
            size = box[3] * box[4] * box[5]
            seg = buf[off * nplanes:(off + size) * nplanes]
            if unpack:
                v.copy_(seg.reshape(v.shape))
            else:
                seg.copy_(v.reshape(-1))

    def p2p(self, sends, recvs):
        import torch.distributed as dist
        if not sends and not recvs:
            return
        ops = [dist.P2POp(dist.isend, b[o:o + c], p) for p, b, o, c in sends] + \
              [dist.P2POp(dist.irecv, b[o:o + c], p) for p, b, o, c in recvs]
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def allgather(self, send, count, recv):
        import torch.distributed as dist
        parts = [recv[r * count:(r + 1) * count] for r in range(dist.get_world_size())]
        dist.all_gather(parts, send[:count])

    def allreduce_sum(self, v):
        import torch.distributed as dist
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    def relax_pass(self, A, b, x, sor, jb, kb, efirst, part=0, sides=0):
        # the fused device kernel = the two i-colours of the row class back to back, no exchange between;
        # part 1 / 2 = interior rows / shell rows of the class (cedar_amd_relax3_pass_part)
        for ib in ((0, 1) if efirst else (1, 0)):
            self.O.relax_colour3_part(self._n(A), self._n(b), self._n(x), self._n(sor), 1 + ib + 2 * jb + 4 * kb, part | (sides << 4))

    def relax_planes(self, A, b, x, sor, kb, up, part=0, sides=0):
        # cedar_amd_relax3_planes: both row classes of the planes of parity kb in sweep order
        opart = {0: 0, 1: 3, 2: 4}[part]
        for jb in ((0, 1) if up else (1, 0)):
            for ib in ((0, 1) if up else (1, 0)):
                self.O.relax_colour3_part(self._n(A), self._n(b), self._n(x), self._n(sor), 1 + ib + 2 * jb + 4 * kb, opart | (sides << 4))

    def relax_colour_masked(self, A, b, x, sor, colour, mask):
        """the points of `colour` (1..8) where `mask` holds: points of one colour do not couple, so the colour's update of
        the whole box, kept only where asked, is what relaxing those points alone gives (statement of the boundary-first
        chain, cedar_amd/dist.py _chain_parity)"""
        y = x.clone()
        self.O.relax_colour3_part(self._n(A), self._n(b), self._n(y), self._n(sor), colour, 0)
        m = torch.from_numpy(mask)
        x[m] = y[m]

    class _Side:  # CPU: no streams, the "side" work simply runs in program order
        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    def side(self):
        return CpuBackend._Side()

    def wait(self, h):
        pass

    def relax_fixup(self, A, b, x, sor, icol, jb, kb):
        self.O.relax_column3(self._n(A), self._n(b), self._n(x), self._n(sor), icol + 1, jb, kb)

    def relax_colour7(self, A, b, x, sor, pts):
        self.O.relax_colour3(self._n(A), self._n(b), self._n(x), self._n(sor), pts)

    def recip(self, A, sor):
        self.O.setup_recip3(self._n(A), self._n(sor))

    def residual(self, A, x, b, r):
        self.O.residual3(self._n(A), self._n(b), self._n(x), self._n(r))

    def restrict(self, r, bc, P):
        self.O.restrict3(self._n(r), self._n(bc), self._n(P))

    def interp_add(self, x, xc, A, r, P):
        self.O.interp_add3(self._n(x), self._n(xc), self._n(A), self._n(r), self._n(P))

    def interp_phase(self, A, P, phase, lo):
        self.O.setup_interp3_ex(self._n(A), self._n(P), 1 << phase, lo)

    def galerkin(self, A, Ac, P):
        self.O.galerkin3(self._n(A), self._n(Ac), self._n(P))

    def make_serial(self, gA, pre, post, min_coarse, num_levels):
        O = self.O

        class _H:
            def __init__(h):
                h.ml = O.ml_create(gA.numpy(), nrelax_pre=pre, nrelax_post=post, min_coarse=min_coarse, num_levels=num_levels)

            def vcycle(h, x, b):
                h.ml.vcycle(x.numpy(), b.numpy())
        return _H()

    # ---- 2D (cedar_amd/dist2d.py)
    def relax_pass2(self, A, b, x, sor, jb, efirst):
        for ib in ((0, 1) if efirst else (1, 0)):
            self.O.relax_colour2(self._n(A), self._n(b), self._n(x), self._n(sor), ib + 2 * jb)

    def relax_fixup2(self, A, b, x, sor, icol, jb):
        self.O.relax_column2(self._n(A), self._n(b), self._n(x), self._n(sor), icol + 1, jb)

    def relax_colour5(self, A, b, x, sor, jo):
        self.O.relax_colour2(self._n(A), self._n(b), self._n(x), self._n(sor), jo)

    def recip2(self, A, sor):
        self.O.setup_recip2(self._n(A), self._n(sor))

    def residual2(self, A, x, b, r):
        self.O.residual2(self._n(A), self._n(b), self._n(x), self._n(r))

    def restrict2(self, r, bc, P):
        self.O.restrict2(self._n(r), self._n(bc), self._n(P))

    def interp_add2(self, x, xc, A, r, P):
        self.O.interp_add2(self._n(x), self._n(xc), self._n(r), self._n(A), self._n(P))

    def interp_phase2(self, A, P, phase, lo):
        self.O.setup_interp2_ex(self._n(A), self._n(P), 1 << phase, lo)

    def galerkin2(self, A, Ac, P):
        self.O.galerkin2(self._n(A), self._n(Ac), self._n(P))

    def make_serial2(self, gA, relax, pre, post, min_coarse, num_levels):
        O = self.O

        class _H:
            def __init__(h):
                h.ml = O.ml_create(gA.numpy(), relax=relax, nrelax_pre=pre, nrelax_post=post, min_coarse=min_coarse,
                                   num_levels=num_levels)

            def vcycle(h, x, b):
                h.ml.vcycle(x.numpy(), b.numpy())
        return _H()

    def sumsq2(self, r):
        v = self.O.l2(self._n(r))
        return v * v

    def affine_lines(self, c, a, div, reverse):
        """sequential recurrence along each row (the exact arithmetic the scan kernel re-associates)"""
        y = c.numpy().copy()
        an = a.numpy()
        if div is not None:
            y = y / div.numpy()
        n = y.shape[1]
        idx = range(n - 1, -1, -1) if reverse else range(n)
        prev = np.zeros(y.shape[0])
        for i in idx:
            prev = an[:, i] * prev + y[:, i]
            y[:, i] = prev
        return torch.from_numpy(y)

    def sumsq(self, r):
        v = self.O.l2(self._n(r))
        return v * v

    def sync(self):
        pass
