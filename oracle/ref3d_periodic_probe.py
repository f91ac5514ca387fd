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
