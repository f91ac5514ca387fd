"""experiment: relax launch time vs grid size (plane-stride / row-stride resonance?)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
n = int(sys.argv[1])
so, b = capi.gallery("fe3", (n, n, n))
s = capi.Solver(so, share_operator=True)
x = capi.DeviceArray(b.shape)
s.time_relax(x, b, 4)
ms = s.time_relax(x, b, 20) / 20
print("n %%d sweep %%.3f ms  %%.0f GB/s algorithmic" %% (n, ms, 136.0 * n**3 / ms / 1e6))
''' % ROOT
for n in [512, 511, 510, 509, 508, 504, 500, 496, 512, 510, 504]:
    out = subprocess.run([sys.executable, "-c", code, str(n)], capture_output=True, text=True)
    print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
