import sys, time, torch
sys.path.insert(0, ".")
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence
s = make_sequence(3, 480, 640, seed=0)
K = torch.from_numpy(s["intrinsics"][0]).cuda()
pts = []
for f in (0, 2):
    d = torch.from_numpy(s["depths"][f, ..., 0]).cuda()
    v, n, _, _ = ops.frame_maps(d, K)
    gv, gn = ops.global_maps(v, n, d, torch.from_numpy(s["poses"][0]).cuda())
    pts.append(ops.downsample_frame(gv, gn, None, d, 4)[:2])
(tgt, tn), (src, _) = pts
for ns in (src.shape[0], 300):
    ops.icp(src[:ns], tgt, tn, mode=1, numiters=20, return_idx=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.icp(src[:ns], tgt, tn, mode=1, numiters=20, return_idx=False)
    torch.cuda.synchronize()
    print("n_src %6d %8.1f us/solve %.2f us/kernel" % (ns, (time.perf_counter() - t0) / 50 * 1e6, (time.perf_counter() - t0) / 50 * 1e6 / 41))
