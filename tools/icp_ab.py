"""A/B of the two ICP engines of the grid path (persistent cooperative kernel vs one launch per
half-iteration): prints the solve time and a checksum of T; results must be bit-identical."""
import sys, time, hashlib
import torch
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
for mode in (1, 0):
    for ns in (src.shape[0], 4096, 300):
        T, idx = ops.icp(src[:ns], tgt, tn, mode=mode, numiters=20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            ops.icp(src[:ns], tgt, tn, mode=mode, numiters=20, return_idx=False)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 50 * 1e6
        h = hashlib.sha1(T.cpu().numpy().tobytes() + idx.cpu().numpy().tobytes()).hexdigest()[:12]
        print("mode %d n_src %6d  %8.1f us/solve  T sha %s  finite %s" % (mode, ns, us, h, bool(torch.isfinite(T).all())))
