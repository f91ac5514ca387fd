#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/* by running the
reference's own Fortran kernels (oracle/_ref/libcedar_ref.so, built from
/root/reference by oracle/Makefile) on the deterministic inputs of
tests/cases.py.  Run in the build container only:

    make -C oracle all && python oracle/gen_golden.py

The multilevel orchestration around the Fortran (the reference does it in
C++ that cannot be compiled here: needs Boost and nlohmann/json) is restated
below from include/cedar/multilevel.h:243-298, include/cedar/cycle/vcycle.h:57-115
and include/cedar/{2d,3d}/solver.h; every arithmetic step is a call into the
reference's Fortran.  Norms follow grid_func::lp_norm<2> (sequential sum,
i fastest) via a cumulative sum.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
import problems as pb  # noqa: E402
from pyoracle import Ref, DOWN, UP  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def seq_l2(v):
    inner = v[tuple(slice(1, -1) for _ in v.shape)]
    return float(np.sqrt(np.cumsum((inner * inner).ravel())[-1]))


class RefML:
    """multilevel::setup + vcycle + solve composed from the Fortran kernels."""

    def __init__(self, R, so, relax="point", nrelax_pre=2, nrelax_post=1, min_coarse=3, cycle="v", ibc=0):
        """ibc != 0 (2D, point relaxation): the reference's periodic branches; the bindings pass the same
        code to every kernel (include/cedar/2d/relax.h:97, coarsen.h:57, src/2d/interp.cc:41-102 ...) and
        the coarsest operator is stored dense (include/cedar/2d/solver.h:110-114)"""
        self.R, self.relax, self.pre, self.post, self.cycle = R, relax, nrelax_pre, nrelax_post, cycle
        self.ibc = ibc
        # 3D: only per_z (5) can be driven end to end: the x / y ghost loops of BMG3_SymStd_interp_add.f90:253-272
        # are not well defined (pyoracle.Ref.interp_add3)
        assert ibc == 0 or so.ndim == 3 or ibc == 5
        self.nd = nd = so.ndim - 1
        n = [s - 2 for s in so.shape[1:]][::-1]  # nx, ny[, nz]
        ng = 0
        while True:  # 2d/solver.h:57-73
            ng += 1
            if min((m - 1) // (1 << ng) + 1 for m in n) < min_coarse:
                break
        self.nlev = ng
        self.A, self.P, self.x, self.b, self.res, self.SOR = [so], [None], [None], [None], [], []
        shp = so.shape[1:]
        for l in range(ng):
            self.res.append(np.zeros(shp))
            self.SOR.append([np.zeros((2,) + shp), np.zeros((2,) + shp)])
            if l + 1 < ng:
                shp = tuple(int((m - 2 - 1) / 2.0 + 1) + 2 for m in shp)
                self.A.append(np.zeros((14 if nd == 3 else 5,) + shp))
                self.P.append(np.zeros((26 if nd == 3 else 8,) + shp))
                self.x.append(np.zeros(shp))
                self.b.append(np.zeros(shp))
        for l in range(ng - 1):
            F, K, Pm = self.A[l], self.A[l + 1], self.P[l + 1]
            if nd == 2 and ibc:
                R.setup_interp2(F, Pm, ibc=ibc)
                R.galerkin2(F, K, Pm, ibc=ibc)
                if relax == "point":
                    R.setup_recip2(F, self.SOR[l][0])
                elif relax == "line-x":
                    R.setup_lines2(F, self.SOR[l][0], "x", ibc=ibc)
                elif relax == "line-y":
                    R.setup_lines2(F, self.SOR[l][0], "y", ibc=ibc)
                else:
                    R.setup_lines2(F, self.SOR[l][0], "x", ibc=ibc)
                    R.setup_lines2(F, self.SOR[l][1], "y", ibc=ibc)
            elif nd == 2:
                R.setup_interp2(F, Pm)
                R.galerkin2(F, K, Pm)
                if relax == "point":
                    R.setup_recip2(F, self.SOR[l][0])
                elif relax == "line-x":
                    R.setup_lines2(F, self.SOR[l][0], "x")
                elif relax == "line-y":
                    R.setup_lines2(F, self.SOR[l][0], "y")
                else:
                    R.setup_lines2(F, self.SOR[l][0], "x")
                    R.setup_lines2(F, self.SOR[l][1], "y")
            else:
                kw = dict(ibc=ibc) if ibc else {}
                R.setup_interp3(F, Pm, **kw)
                R.galerkin3(F, K, Pm, **kw)
                R.setup_recip3(F, self.SOR[l][0])
        C = self.A[-1]
        cs = [s - 2 for s in C.shape[1:]][::-1]
        if nd == 2 and ibc:
            self.abd = np.zeros((cs[0] * cs[1], cs[0] * cs[1]))
            R.setup_cg2(C, self.abd, ibc=ibc)
        elif nd == 2:
            self.abd = np.zeros((cs[0] * cs[1], cs[0] + 2))
            R.setup_cg2(C, self.abd)
        elif ibc:
            self.abd = np.zeros((cs[0] * cs[1] * cs[2], cs[0] * cs[1] * cs[2]))  # 3d/solver.h:118-121
            R.setup_cg3(C, self.abd, ibc=ibc)
        else:
            self.abd = np.zeros((cs[0] * cs[1] * cs[2], cs[0] * (cs[1] + 1) + 2))
            R.setup_cg3(C, self.abd)

    def _smooth(self, l, x, b, ud, n):
        R, A, S = self.R, self.A[l], self.SOR[l]
        for _ in range(n):
            if self.nd == 3:
                R.relax3(A, b, x, S[0], ud, **(dict(ibc=self.ibc) if self.ibc else {}))
            elif self.ibc and self.relax == "point":
                R.relax2(A, b, x, S[0], ud, ibc=self.ibc)
            elif self.ibc:
                kw = dict(ibc=self.ibc)
                if self.relax == "line-x":
                    R.relax_lines2(A, b, x, S[0], ud, "x", **kw)
                elif self.relax == "line-y":
                    R.relax_lines2(A, b, x, S[0], ud, "y", **kw)
                elif ud == DOWN:
                    R.relax_lines2(A, b, x, S[0], ud, "x", **kw)
                    R.relax_lines2(A, b, x, S[1], ud, "y", **kw)
                else:
                    R.relax_lines2(A, b, x, S[1], ud, "y", **kw)
                    R.relax_lines2(A, b, x, S[0], ud, "x", **kw)
            elif self.relax == "point":
                R.relax2(A, b, x, S[0], ud)
            elif self.relax == "line-x":
                R.relax_lines2(A, b, x, S[0], ud, "x")
            elif self.relax == "line-y":
                R.relax_lines2(A, b, x, S[0], ud, "y")
            elif ud == DOWN:
                R.relax_lines2(A, b, x, S[0], ud, "x")
                R.relax_lines2(A, b, x, S[1], ud, "y")
            else:
                R.relax_lines2(A, b, x, S[1], ud, "y")
                R.relax_lines2(A, b, x, S[0], ud, "x")

    def _residual(self, l, x, b, r):
        (self.R.residual2 if self.nd == 2 else self.R.residual3)(self.A[l], b, x, r)

    def _cycle(self, l, x, b):
        R = self.R
        self._smooth(l, x, b, DOWN, self.pre)
        self._residual(l, x, b, self.res[l])
        cx, cb, Pm = self.x[l + 1], self.b[l + 1], self.P[l + 1]
        kw = dict(ibc=self.ibc) if self.ibc else {}
        (R.restrict2 if self.nd == 2 else R.restrict3)(self.res[l], cb, Pm, **kw)
        cx[...] = 0.0
        if l + 1 == self.nlev - 1:
            (R.solve_cg2 if self.nd == 2 else R.solve_cg3)(cx, cb, self.abd, **kw)
        else:
            self._cycle(l + 1, cx, cb)
        if self.nd == 2:
            R.interp_add2(x, cx, self.res[l], self.A[l], Pm, **kw)
        else:
            R.interp_add3(x, cx, self.A[l], self.res[l], Pm, **kw)
        self._smooth(l, x, b, UP, self.post)

    def _fmg(self, l, x, b):
        """include/cedar/cycle/fcycle.h:49-83"""
        R = self.R
        kw = dict(ibc=self.ibc) if self.ibc else {}
        if l == self.nlev - 1:
            (R.solve_cg2 if self.nd == 2 else R.solve_cg3)(x, b, self.abd, **kw)
            return
        cx, cb, Pm = self.x[l + 1], self.b[l + 1], self.P[l + 1]
        (R.restrict2 if self.nd == 2 else R.restrict3)(b, cb, Pm, **kw)
        self._fmg(l + 1, cx, cb)
        x[...] = 0.0
        self.res[l][...] = 0.0
        if self.nd == 2:
            R.interp_add2(x, cx, self.res[l], self.A[l], Pm, **kw)
        else:
            R.interp_add3(x, cx, self.A[l], self.res[l], Pm, **kw)
        self._cycle(l, x, b)

    def vcycle(self, x, b):
        if self.nlev == 1:
            kw = dict(ibc=self.ibc) if self.ibc else {}
            (self.R.solve_cg2 if self.nd == 2 else self.R.solve_cg3)(x, b, self.abd, **kw)
        elif self.cycle == "f":
            self._fmg(0, x, b)
        else:
            self._cycle(0, x, b)

    def solve(self, b, x, maxiter=10, tol=1e-8):
        self._residual(0, x, b, self.res[0])
        r0 = seq_l2(self.res[0])
        hist = [r0]
        for _ in range(maxiter):
            self.vcycle(x, b)
            self._residual(0, x, b, self.res[0])
            rel = seq_l2(self.res[0]) / r0
            hist.append(rel)
            if rel < tol:
                break
        return hist


def main():
    os.makedirs(GOLD, exist_ok=True)
    R = Ref()
    k2 = {}
    for c in cases.CASES_2D:
        for k, v in cases.kernel_suite_2d(R, c).items():
            k2[f"{c[0]}/{k}"] = v
    np.savez_compressed(os.path.join(GOLD, "kernels2d.npz"), **k2)
    k3 = {}
    for c in cases.CASES_3D:
        for k, v in cases.kernel_suite_3d(R, c).items():
            k3[f"{c[0]}/{k}"] = v
    np.savez_compressed(os.path.join(GOLD, "kernels3d.npz"), **k3)
    np.savez_compressed(os.path.join(GOLD, "sweeps.npz"), **cases.sweep_suite(R))

    solves(R)
    main_periodic(R)
    main_periodic3(R)


def solves(R, only=None):
    """residual histories of full solves; `only`: regenerate just these entries of solves.json"""
    path = os.path.join(GOLD, "solves.json")
    hist = json.load(open(path)) if (only and os.path.exists(path)) else {}
    for name, (mk_op, mk_rhs, st) in cases.SOLVES.items():
        if only and name not in only:
            continue
        so, b = mk_op(), mk_rhs()
        ml = RefML(R, so, **st)
        x = np.zeros_like(b)
        h = ml.solve(b, x, maxiter=10, tol=1e-8)
        inner = x[tuple(slice(1, -1) for _ in x.shape)]
        hist[name] = {
            "settings": st,
            "nlevels": ml.nlev,
            "level_dims": [list(a.shape[1:][::-1]) for a in ml.A],
            "res0_l2": repr(h[0]),
            "rel_l2": [repr(v) for v in h[1:]],
            "x_l2": repr(seq_l2(x)),
            "x_sum": repr(float(np.cumsum(inner.ravel())[-1])),
        }
        print(name, ml.nlev, h[0], h[1:4], flush=True)
    with open(path, "w") as f:
        json.dump(hist, f, indent=1)


def main_periodic3(R):
    """3D periodic boundary conditions: outputs of the reference kernels that are well defined for the boundary
    code (cases.kernel_suite_per3), dense coarse solves, and per_z residual histories"""
    kp = {}
    for c in cases.CASES_PER3:
        for k, v in cases.kernel_suite_per3(R, c, reference=True).items():
            kp[f"{c[0]}/{k}"] = v
    for c in cases.CG_PER3:
        for k, v in cases.coarse_solve_per3(R, c).items():
            kp[f"{c[0]}/{k}"] = v
    np.savez_compressed(os.path.join(GOLD, "periodic3d.npz"), **kp)
    hist = {}
    for name, (mk_op, mk_rhs, st) in cases.SOLVES_PER3.items():
        if st["ibc"] != 5:
            continue
        so, b = mk_op(), mk_rhs()
        ml = RefML(R, so, **st)
        x = np.zeros_like(b)
        h = ml.solve(b, x, maxiter=10, tol=1e-8)
        inner = x[1:-1, 1:-1, 1:-1]
        hist[name] = {"settings": st, "nlevels": ml.nlev, "res0_l2": repr(h[0]), "rel_l2": [repr(v) for v in h[1:]],
                      "x_l2": repr(seq_l2(x)), "x_sum": repr(float(np.cumsum(inner.ravel())[-1]))}
        print(name, ml.nlev, h[0], h[1:4], flush=True)
    with open(os.path.join(GOLD, "solves_periodic3d.json"), "w") as f:
        json.dump(hist, f, indent=1)


def main_periodic(R):
    """2D periodic boundary conditions: kernel outputs, coarse solves and residual histories"""
    kp = {}
    for c in cases.CASES_PER:
        for k, v in cases.kernel_suite_per(R, c).items():
            kp[f"{c[0]}/{k}"] = v
    for c in cases.CG_PER:
        for k, v in cases.coarse_solve_per(R, c).items():
            kp[f"{c[0]}/{k}"] = v
    np.savez_compressed(os.path.join(GOLD, "periodic2d.npz"), **kp)
    hist = {}
    for name, (mk_op, mk_rhs, st) in cases.SOLVES_PER.items():
        so, b = mk_op(), mk_rhs()
        ml = RefML(R, so, **st)
        x = np.zeros_like(b)
        h = ml.solve(b, x, maxiter=10, tol=1e-8)
        inner = x[1:-1, 1:-1]
        hist[name] = {"settings": st, "nlevels": ml.nlev, "res0_l2": repr(h[0]), "rel_l2": [repr(v) for v in h[1:]],
                      "x_l2": repr(seq_l2(x)), "x_sum": repr(float(np.cumsum(inner.ravel())[-1]))}
        print(name, ml.nlev, h[0], h[1:4], flush=True)
    with open(os.path.join(GOLD, "solves_periodic.json"), "w") as f:
        json.dump(hist, f, indent=1)


# cases whose reference-built hierarchy is committed (small ones): tests upload it into the device-resident solver and run
# the SOLVE PHASE alone -- what is left of the late-cycle history deviation then belongs to the solve kernels
HIERARCHIES = {
    "fe27_24x20x17_v21": (lambda: pb.fe3(24, 20, 17), lambda: pb.rhs3(24, 20, 17), dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "fe27_40x33x50_v21": (lambda: pb.fe3(40, 33, 50), lambda: pb.rhs3(40, 33, 50), dict(relax="point", nrelax_pre=2, nrelax_post=1)),
    "varcoef9_72x50_v21": (lambda: pb.varcoef9(72, 50), lambda: pb.rhs2(72, 50), dict(relax="point", nrelax_pre=2, nrelax_post=1)),
}


def hierarchies(R):
    """the reference's set-up products of a few small cases (A and P of the coarse levels, relaxation data, the factored
    coarsest operator) and the residual history of 10 cycles on them"""
    for name, (mk_op, mk_rhs, st) in HIERARCHIES.items():
        so, b = mk_op(), mk_rhs()
        ml = RefML(R, so, **st)
        x = np.zeros_like(b)
        h = ml.solve(b, x, maxiter=10, tol=1e-8)
        out = {"hist": np.array(h), "abd": ml.abd, "nlev": np.array(ml.nlev)}
        for l in range(ml.nlev):
            if l > 0:
                out["A%d" % l] = ml.A[l]
                out["P%d" % l] = ml.P[l]
            if l < ml.nlev - 1:
                out["SOR0_%d" % l] = ml.SOR[l][0]
        np.savez_compressed(os.path.join(GOLD, "hier_%s.npz" % name), **out)
        print(name, ml.nlev, h[0], h[-1], flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hierarchy":
        hierarchies(Ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "periodic":  # regenerate only the periodic fixtures
        os.makedirs(GOLD, exist_ok=True)
        main_periodic(Ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "periodic3":
        main_periodic3(Ref())
    elif len(sys.argv) > 2 and sys.argv[1] == "solves":  # python gen_golden.py solves NAME [NAME ...]
        solves(Ref(), only=set(sys.argv[2:]))
    else:
        main()
