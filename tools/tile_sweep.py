"""experiment: relax/residual time vs row-tile shape (env read once per process -> one process per shape)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = ["6,0", "3,3", "4,2", "5,1", "2,4", "4,3", "4,4", "3,2", "1,5", "0,6"]
for sh in shapes:
    env = dict(os.environ, CEDAR_AMD_TILE_RELAX=sh, CEDAR_AMD_TILE_RESID=sh)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_relax.py")], env=env, capture_output=True, text=True)
    print(sh, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
