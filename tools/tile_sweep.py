"""experiment: relax launch time (HIP events, solver handle) vs row-tile shape; one process per shape"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = ["6,0", "3,3", "4,4", "3,2", "5,1", "4,3", "3,3", "4,4", "6,0"]
code = r'''
import os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
so, b = capi.gallery("fe3", (512, 512, 512))
s = capi.Solver(so, share_operator=True)
x = capi.DeviceArray(b.shape)
s.time_relax(x, b, 4)
print("relax launch %%.4f ms" %% (s.time_relax(x, b, 20) / 80))
''' % ROOT
for sh in shapes:
    env = dict(os.environ, CEDAR_AMD_TILE_RELAX=sh)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(sh, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
