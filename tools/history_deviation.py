"""Per-iteration deviation of the GPU residual histories from the reference goldens (tests/golden/solves.json, made by
the reference's own Fortran): |h_gpu - h_ref| / h_ref for every cycle of every golden solve, and the first cycle at
which north_star's 1e-10 is exceeded without any absolute floor.  The C oracle's deviation from the same goldens is
printed beside it (two correctly rounded implementations of the same algorithm)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import cases
from cedar_amd import capi
from pyoracle import Oracle
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "solves.json")))
O = Oracle()
only = sys.argv[1:]
for name, (mk_op, mk_rhs, st) in cases.SOLVES.items():
    if only and name not in only:
        continue
    so, b = mk_op(), mk_rhs()
    want = np.array([float(gold[name]["res0_l2"])] + [float(v) for v in gold[name]["rel_l2"]])
    s = capi.Solver(so, **st)
    x = np.zeros_like(b)
    h = np.array(s.solve(b, x))
    s.close()
    ml = O.ml_create(so, **st)
    xo = np.zeros_like(b)
    ho = np.array(ml.solve(b, xo, maxiter=10))
    ml.close()
    m = min(len(h), len(want), len(ho))
    dg = np.abs(h[:m] - want[:m]) / want[:m]
    do = np.abs(ho[:m] - want[:m]) / want[:m]
    over = [i for i in range(m) if dg[i] > 1e-10]
    print("%-24s rel residual at the end %.2e | first cycle with |gpu-ref|/ref > 1e-10: %s" % (name, want[m - 1], over[0] if over else "none"))
    print("    gpu   : " + " ".join("%.1e" % v for v in dg))
    print("    oracle: " + " ".join("%.1e" % v for v in do), flush=True)
