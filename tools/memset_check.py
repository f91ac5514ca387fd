"""hipMemsetAsync under the HIP runtime torch brings along: does it zero what it should?"""
import sys, ctypes as C
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    print("torch", torch.__version__, torch.version.hip)
sys.path.insert(0, ".")
from cedar_amd import capi
lib = capi.lib
for shape in ((41, 67), (22, 35), (12, 19), (7, 11), (515, 515), (1, 5)):
    bad = []
    for rep in range(4):
        a = capi.DeviceArray.from_numpy(np.full(shape, 3.25))
        lib.cedar_amd_memset(a.ptr, 0, a.size * 8 - 1)  # odd byte count: takes the hipMemsetAsync branch; the last double keeps its top byte
        capi.sync()
        h = a.numpy()
        bad.append(int(np.count_nonzero(h)))
    print(shape, "nonzero after hipMemsetAsync(0) of all but the last byte (expect 1):", bad, flush=True)
