import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
import problems as pb
from cedar_amd import capi
nx, ny = 130, 77
so = pb.aniso9(nx, ny); b = pb.rhs2(nx, ny)
for relax in ("point", "line-x", "line-y", "line-xy"):
    for flag in ("0", "1"):
        os.environ["CEDAR_AMD_YLINES_TRANSPOSED"] = flag
        s = capi.Solver(so, relax=relax, nrelax_pre=2, nrelax_post=1)
        x = np.zeros_like(b)
        h = s.solve(b, x)
        print(relax, flag, "level 0 x E ghost nonzero", np.count_nonzero(x[:, -1]), x[1:4, -1].tolist())
        for l in range(1, s.nlevels()):
            for what in ("x", "b", "res"):
                a = s.array(l, what)
                if a is None:
                    continue
                a = a[0]
                gh = [np.count_nonzero(a[:, 0]), np.count_nonzero(a[:, -1]), np.count_nonzero(a[0, :]), np.count_nonzero(a[-1, :])]
                if any(gh):
                    print("   level", l, what, a.shape, "nonzero ghosts W E S N", gh, "E sample", a[1:4, -1].tolist(), "N sample", a[-1, 1:4].tolist())
        s.close()
