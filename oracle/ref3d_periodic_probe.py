"""TEST INFRASTRUCTURE (build container only: needs oracle/_ref, i.e. /root/reference).

Evidence for DESIGN.md section 6: the reference's 3D periodic path is not a well-defined parity target.
Runs the reference's own compiled BMG3_SymStd_interp_add (oracle/_ref/libcedar_ref.so) with a periodic boundary code
and reports which ghost cells of q it wrapped.  The wrap loops of src/3d/ftn/BMG3_SymStd_interp_add.f90:253-272 index
with variables other than their loop variables (`DO J = 1,KKF ... Q(I,1,K)`, `DO I = 1,IIF ... Q(1,J,K)`), so only one
stale column / row is wrapped per plane and, when the y block has run, the x block addresses row KKF+1 -- outside the
array for JJF <= KKF.  The array handed over here carries a guard plane behind it so that this write lands in memory we own.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "tests")]
import problems as pb  # noqa: E402
from pyoracle import Ref, _p, u  # noqa: E402

R = Ref()
nx, ny, nz = 8, 10, 6  # JJF > KKF: row KKF+1 exists
for ibc, name in ((2, "per_x"), (1, "per_y"), (5, "per_z"), (8, "per_xyz")):
    g = (nz + 2, ny + 2, nx + 2)
    gc = pb.coarse_shape(g)
    so = pb.random_op(g, 14, 11)
    ci = pb.uniform((26,) + gc, 12, 0, 0.5)
    qc = pb.uniform(gc, 13, -1, 1)
    res = pb.uniform(g, 14, -1, 1)
    buf = np.zeros((g[0] + 1,) + g[1:])  # one guard plane behind q
    q = buf[:-1]
    q[...] = pb.uniform(g, 15, -1, 1)
    q0 = q.copy()
    R.L.BMG3_SymStd_interp_add(_p(q), _p(qc), _p(so), _p(res), _p(ci), u(gc[2]), u(gc[1]), u(gc[0]), u(g[2]), u(g[1]), u(g[0]), 14, ibc)
    # what a complete wrap of that direction would have produced on the ghost faces
    chk = []
    if ibc in (2, 8):
        chk.append(("x ghost columns", np.array_equal(q[:, :, 0], q[:, :, -2]) and np.array_equal(q[:, :, -1], q[:, :, 1]),
                    int(np.count_nonzero(q[:, :, 0] != q0[:, :, 0]) + np.count_nonzero(q[:, :, -1] != q0[:, :, -1])), 2 * g[0] * g[1]))
    if ibc in (1, 8):
        chk.append(("y ghost rows", np.array_equal(q[:, 0, :], q[:, -2, :]) and np.array_equal(q[:, -1, :], q[:, 1, :]),
                    int(np.count_nonzero(q[:, 0, :] != q0[:, 0, :]) + np.count_nonzero(q[:, -1, :] != q0[:, -1, :])), 2 * g[0] * g[2]))
    if ibc in (5, 8):
        chk.append(("z ghost planes", np.array_equal(q[0], q[-2]) and np.array_equal(q[-1], q[1]),
                    int(np.count_nonzero(q[0] != q0[0]) + np.count_nonzero(q[-1] != q0[-1])), 2 * g[1] * g[2]))
    print("BMG3_SymStd_interp_add, %dx%dx%d, boundary code %d (%s): guard plane touched: %s" % (nx, ny, nz, ibc, name, bool(np.any(buf[-1] != 0))))
    for what, ok, changed, total in chk:
        print("    %-16s wrapped completely: %-5s  ghost cells the call changed: %d of %d" % (what, ok, changed, total))

# ---------------------------------------------------------------------------------------------------------------
# 2. BMG3_SymStd_SETUP_cg_LU, periodic branch (:200-619): the dense matrix the reference factors, rebuilt from its
#    Cholesky factor (U^T U), against the periodic operator assembled by walking the stencil (oracle/boxmg3_per.c).
# ---------------------------------------------------------------------------------------------------------------
from pyoracle import Oracle  # noqa: E402

O = Oracle()
print()
print("BMG3_SymStd_SETUP_cg_LU: entries of the factored dense matrix that are not the periodic operator's")
print("    (upper triangle; tolerance 1e-12; random 27-point operator with periodic ghost layers)")
for per in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 1, 0), (1, 0, 1), (1, 1, 1)]:
    row = []
    for n in [(4, 4, 4), (3, 3, 3), (4, 6, 4), (5, 4, 4), (4, 4, 5), (8, 8, 8)]:
        ibc = pb.ibc3_of(per)
        so = pb.periodic_random_op3(n[0], n[1], n[2], 14, per, 9)
        N = n[0] * n[1] * n[2]
        a_ref, a_orc = np.zeros((N, N)), np.zeros((N, N))
        R.setup_cg3(so, a_ref, ibc=ibc)
        O.setup_cg3(so, a_orc, ibc=ibc)
        Ur, Uo = np.triu(a_ref.T), np.triu(a_orc.T)
        bad = np.argwhere(np.triu(np.abs(Ur.T @ Ur - Uo.T @ Uo)) > 1e-12)
        row.append("%dx%dx%d: %d" % (n + (len(bad),)))
    print("    code %d %-8s %s" % (ibc, "per_" + "".join(c for c, p in zip("xyz", per) if p), "   ".join(row)))

# ---------------------------------------------------------------------------------------------------------------
# 3. BMG3_SymStd_SETUP_interp_OI, periodic branch (:808-2811): weights against the restatement (Dirichlet formulas,
#    loops started one coarse point earlier in a periodic direction, ghost refresh after every phase).  Compared
#    through what reads them: restriction of a random vector and the Galerkin product, both computed by the oracle
#    from the two sets of weights.
# ---------------------------------------------------------------------------------------------------------------
print()
print("BMG3_SymStd_SETUP_interp_OI: do the reference's weights restrict / coarsen like the restatement's?")
print("    (27-point; 'yes' = restriction of a random vector bit-identical and Galerkin product bit-identical)")
for per in [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]:
    row = []
    for n in [(8, 10, 12), (6, 4, 8), (12, 6, 10), (8, 8, 8), (16, 8, 4), (8, 6, 4), (10, 8, 6), (9, 8, 8)]:
        ibc = pb.ibc3_of(per)
        so = pb.periodic_random_op3(n[0], n[1], n[2], 14, per, 3)
        g = so.shape[1:]
        gc = pb.coarse_shape(g)
        c1, c2 = np.zeros((26,) + gc), np.zeros((26,) + gc)
        R.setup_interp3(so, c1, ibc=ibc)
        O.setup_interp3(so, c2, ibc=ibc)
        r = pb.uniform(g, 7, -1, 1)
        q1, q2 = np.zeros(gc), np.zeros(gc)
        O.restrict3(r.copy(), q1, c1, ibc=ibc)
        O.restrict3(r.copy(), q2, c2, ibc=ibc)
        s1, s2 = np.zeros((14,) + gc), np.zeros((14,) + gc)
        O.galerkin3(so, s1, c1, ibc=ibc)
        O.galerkin3(so, s2, c2, ibc=ibc)
        row.append("%dx%dx%d: %s" % (n + ("yes" if np.array_equal(q1, q2) and np.array_equal(s1, s2) else "NO",)))
    print("    code %d %-8s %s" % (ibc, "per_" + "".join(c for c, p in zip("xyz", per) if p), "  ".join(row)))
print("    -> agreement needs ny <= nz (the branch mixes up the y and z extents) and even extents in the periodic")
print("       directions (9x8x8 with x periodic: the reference pairs the two end points through the wrap; refused here).")
