"""Times the brute-force 1-NN kernel (in-library HIP events) on ICP-sized sets; GRADSLAM_HIP_KNN_SPT=4|8."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, ".")
from gradslam_amd import ops, _C
rng = np.random.default_rng(0)
for ns, nt in ((19200, 26000), (78408, 80000), (2000, 3000)):
    src = torch.from_numpy(rng.standard_normal((ns, 3)).astype(np.float32)).cuda()
    tgt = torch.from_numpy(rng.standard_normal((nt, 3)).astype(np.float32)).cuda()
    ops.knn1(src, tgt)
    lib = _C.lib()
    lib.gs_profile_begin(64)
    for _ in range(10):
        ops.knn1(src, tgt)
    lib.gs_profile_end()
    ms, n, w = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    lib.gs_profile_read(0, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
    print(ns, nt, "us/launch %.1f" % (ms.value * 1e3 / n.value), "TFLOP/s(8 flop/pair) %.1f" % (8 * w.value / (ms.value * 1e-3) / 1e12))
