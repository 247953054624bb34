"""Per-launch failure counters of the candidate lists of ordinary ICP queries (gs_localize_list_stats_i64) on the
benchmark workload:  python tools/list_stats_probe.py [B] [frames] [first frame counted (default 1)]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
F0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H, W = 480, 640
seqs = [make_sequence(L, H, W, seed=b) for b in range(B)]
st = lambda k: torch.from_numpy(np.stack([s[k] for s in seqs])).cuda()
poses = st("poses"); poses[:, 1:] = poses[:, :1]
frames = gs.RGBDImages(st("colors"), st("depths"), st("intrinsics"), poses)
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev = gs.Pointclouds(device="cuda"), None
tot = np.zeros((3, 64), np.int64)
worst = np.zeros(64, np.int64)
for f in range(L):
    live = frames[:, f]; pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
    if f >= F0:
        for b in range(B):
            fa, em, op = ops.localize_list_stats(torch.device("cuda", 0), b, H, W, 4, pc._buf["points"][b].shape[0])
            tot[0] += fa; tot[1] += em; tot[2] += op
            worst = np.maximum(worst, np.array(fa) + np.array(em) + np.array(op))
n = (L - F0) * B
print("frames %d..%d, map rows per sequence at the end: %s" % (F0, L - 1, [int(x.shape[0]) for x in pc.points_list]))
print("mean per solve over %d solves (19200 lattice slots): launch: failed lists / empty lists / points without a list / worst solve (sum)" % n)
for h in range(40):
    print("  launch %2d: %8.1f %8.1f %8.1f %6d" % (h, tot[0, h] / n, tot[1, h] / n, tot[2, h] / n, worst[h]))
