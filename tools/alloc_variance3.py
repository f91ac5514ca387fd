"""is a slow operator allocation slow for one plain stream too?  relax sweep time vs a streaming sum of squares over the same allocation"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
keep = []
for rep in range(7):
    so2 = capi.DeviceArray(so.shape)
    so2.copy_from(so)
    s = capi.Solver(so2, share_operator=True)
    s.time_relax(x, b, 2)
    ms = s.time_relax(x, b, 6) / 6
    s.close()
    capi.lib.cedar_amd_l2norm(so2.ptr, n + 2, n + 2, 14 * (n + 2))
    capi.sync(); t0 = time.perf_counter()
    for _ in range(5):
        capi.lib.cedar_amd_l2norm(so2.ptr, n + 2, n + 2, 14 * (n + 2))
    capi.sync(); t1 = time.perf_counter()
    stream_ms = (t1 - t0) / 5 * 1e3
    print(json.dumps({"rep": rep, "so_ptr": hex(so2.ptr), "relax_ms_per_sweep": round(ms, 4), "stream_read_ms": round(stream_ms, 4),
                      "stream_TBps": round(14 * (n + 2) ** 3 * 8 / stream_ms / 1e9, 3)}), flush=True)
    if rep % 2 == 0:
        so2.free()
    else:
        keep.append(so2)
