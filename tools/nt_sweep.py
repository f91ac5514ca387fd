"""experiment: relax launch time with / without non-temporal operator loads (alternating processes)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
so, b = capi.gallery("fe3", (512, 512, 512))
s = capi.Solver(so, share_operator=True)
x = capi.DeviceArray(b.shape)
s.time_relax(x, b, 4)
print("relax launch %%.4f ms  vcycle %%.3f ms" %% (s.time_relax(x, b, 20) / 80, (s.time_vcycles(x, b, 2), s.time_vcycles(x, b, 6) / 6)[1]))
''' % ROOT
for nt in ["0", "1", "0", "1", "0", "1"]:
    env = dict(os.environ, CEDAR_AMD_NT=nt)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("nt", nt, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
