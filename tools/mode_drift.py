"""Does the mode of a solver's relax sweep change while the solver lives?  Solver A is timed, then again after other device
memory has been allocated / released and a second solver B has been built beside it, B is timed, released, A timed again.
    python tools/mode_drift.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi

capi.lib.cedar_amd_solver_time_relax.restype = C.c_float


def mk():
    so, b = capi.gallery("fe3", (512, 512, 512))
    s = capi.Solver(so, share_operator=True)
    x = capi.DeviceArray(b.shape)
    x.zero()
    return s, so, b, x


def t(sv, tag):
    s, so, b, x = sv
    capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 2)
    print("%-44s %.3f ms per sweep" % (tag, capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 8) / 8), flush=True)


for rep in range(3):
    A = mk()
    t(A, "rep %d  A fresh" % rep)
    t(A, "rep %d  A again" % rep)
    big = capi.DeviceArray((3 * 10 ** 9,))
    t(A, "rep %d  A with 24 GB more allocated" % rep)
    big.free()
    t(A, "rep %d  A after releasing them" % rep)
    B = mk()
    t(A, "rep %d  A beside a second solver" % rep)
    t(B, "rep %d  B" % rep)
    B[0].close(); B[1].free(); B[2].free(); B[3].free()
    t(A, "rep %d  A after B is gone" % rep)
    x2 = capi.DeviceArray(A[2].shape)
    x2.zero()
    t((A[0], A[1], A[2], x2), "rep %d  A on another x" % rep)
    A[0].close(); A[1].free(); A[2].free(); A[3].free(); x2.free()
