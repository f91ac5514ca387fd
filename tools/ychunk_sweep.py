"""8192^2 line-xy V-cycle time vs the y-line chunk size (CEDAR_AMD_YCHUNK; 0 = whole colour at once)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import problems as pb
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
so = capi.DeviceArray.from_numpy(pb.aniso9(n, n))
b = capi.DeviceArray.from_numpy(pb.rhs2(n, n))
x = capi.DeviceArray(b.shape)
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
s = capi.Solver(so, relax="line-xy", share_operator=True)
for chunk in [0, 2048, 1024, 512, 256, 128, 0]:
    os.environ["CEDAR_AMD_YCHUNK"] = str(chunk)
    s.time_vcycles(x, b, 2)
    ms = s.time_vcycles(x, b, 5) / 5
    print(json.dumps({"n": n, "ychunk": chunk, "ms_per_vcycle": ms}), flush=True)
