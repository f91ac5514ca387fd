"""(kept under tests/ because it calls the oracle as the checker) ad-hoc first GPU run: parity of relax/residual vs the oracle + relax sweep timing"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import problems as pb
from pyoracle import Oracle
from cedar_amd import capi
O, K = Oracle(), capi.Kernels()
print(capi.lib.cedar_amd_version(), "devices", capi.device_count(), flush=True)
ok = True
for (nx, ny, nz, nst) in [(9, 9, 9, 14), (8, 6, 10, 14), (33, 20, 17, 14), (64, 64, 64, 14), (130, 40, 30, 14), (9, 9, 9, 4), (30, 21, 17, 4)]:
    g = (nz + 2, ny + 2, nx + 2)
    so = pb.random_op(g, nst, 7); qf = pb.uniform(g, 8, -1, 1); q0 = pb.uniform(g, 9, -1, 1)
    s1 = np.zeros((2,) + g); s2 = np.zeros((2,) + g)
    O.setup_recip3(so, s1); K.setup_recip3(so, s2)
    e = [np.array_equal(s1, s2)]
    for ud in (0, 1):
        a = q0.copy(); b = q0.copy()
        O.relax3(so, qf, a, s1, ud); K.relax3(so, qf, b, s2, ud)
        e.append(np.array_equal(a, b)); 
        if not e[-1]: print("  relax diff", np.max(np.abs(a - b)))
    a = np.zeros(g); b = np.zeros(g)
    O.residual3(so, qf, q0, a); K.residual3(so, qf, q0, b); e.append(np.array_equal(a, b))
    e.append(abs(O.l2(a) - K.l2(b)) <= 1e-13 * O.l2(a))
    print("3d", nx, ny, nz, nst, e, flush=True); ok &= all(e)
for (nx, ny, nst) in [(9, 9, 5), (16, 12, 5), (700, 33, 5), (9, 9, 3), (65, 30, 3)]:
    g = (ny + 2, nx + 2)
    so = pb.random_op(g, nst, 7); qf = pb.uniform(g, 8, -1, 1); q0 = pb.uniform(g, 9, -1, 1)
    s1 = np.zeros((2,) + g); s2 = np.zeros((2,) + g)
    O.setup_recip2(so, s1); K.setup_recip2(so, s2)
    e = [np.array_equal(s1, s2)]
    for ud in (0, 1):
        a = q0.copy(); b = q0.copy()
        O.relax2(so, qf, a, s1, ud); K.relax2(so, qf, b, s2, ud); e.append(np.array_equal(a, b))
    a = np.zeros(g); b = np.zeros(g)
    O.residual2(so, qf, q0, a); K.residual2(so, qf, q0, b); e.append(np.array_equal(a, b))
    print("2d", nx, ny, nst, e, flush=True); ok &= all(e)
print("PARITY", "OK" if ok else "FAIL", flush=True)

# timing: device-resident 27-pt relax sweeps
for n in ([256, 512] if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
    t0 = time.time()
    g = (n + 2, n + 2, n + 2)
    so = pb.fe3(n, n, n)
    print("built fe3", n, time.time() - t0, flush=True)
    dso = capi.DeviceArray.from_numpy(so); del so
    dq = capi.DeviceArray.from_numpy(pb.uniform(g, 3, 0, 1)); dqf = capi.DeviceArray(g); dsor = capi.DeviceArray((2,) + g); dres = capi.DeviceArray(g)
    K.setup_recip3(dso, dsor)
    for _ in range(3):
        K.relax3(dso, dqf, dq, dsor, 0); K.relax3(dso, dqf, dq, dsor, 1)
    capi.sync(); t0 = time.time(); reps = 10
    for _ in range(reps):
        K.relax3(dso, dqf, dq, dsor, 0); K.relax3(dso, dqf, dq, dsor, 1)
    capi.sync(); dt = (time.time() - t0) / (2 * reps)
    print(f"relax27 {n}^3: {dt*1e3:.3f} ms/sweep  {136.0*n**3/dt/1e9:.1f} GB/s algorithmic", flush=True)
    capi.sync(); t0 = time.time()
    for _ in range(reps): K.residual3(dso, dqf, dq, dres)
    capi.sync(); dt = (time.time() - t0) / reps
    print(f"residual27 {n}^3: {dt*1e3:.3f} ms  {136.0*n**3/dt/1e9:.1f} GB/s algorithmic", flush=True)
    for a in (dso, dq, dqf, dsor, dres): a.free()
