"""Debugging aid: per-block timeline of the two half-iteration launches of the tile engine's last iteration.
Needs a library built with GRADSLAM_HIP_BUILD_FLAGS=-DGS_ICP_TIMELINE.
    GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl.txt python tools/icp_tile_timeline.py [B] [H W]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gradslam_amd as gs
from gradslam_amd.datasets.synthetic import make_sequence
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640)
L = 4
seqs = [make_sequence(L, H, W, seed=b) for b in range(B)]
st = lambda k: torch.from_numpy(np.stack([s[k] for s in seqs])).cuda()
poses = st("poses"); poses[:, 1:] = poses[:, :1]
frames = gs.RGBDImages(st("colors"), st("depths"), st("intrinsics"), poses)
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev = gs.Pointclouds(device="cuda"), None
for f in range(L):
    live = frames[:, f]; pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
torch.cuda.synchronize()
path = os.environ["GRADSLAM_HIP_ICP_TIMELINE"]
print(open(path).readline().strip())
rows = np.loadtxt(path, dtype=np.uint64)
for half, name in ((0, "full"), (1, "look-ahead")):
    r = rows[rows[:, 0] == half]
    t = r[:, 2:8].astype(np.float64) / 100.0   # us
    t0 = t[:, 0].min()
    ph = np.diff(t, axis=1)
    mode = ((r[:, 9] >> 32) & 0xf).astype(int)
    npts = ((r[:, 9] >> 36) & 0xfff).astype(int)
    ncell = (r[:, 9] >> 48).astype(int)
    nbrute = (r[:, 9] & 0xffffffff).astype(int)
    print("%-10s blocks %d  start %.2f..%.2f  end %.2f..%.2f us  life mean %.2f max %.2f" % (
        name, len(r), (t[:, 0] - t0).min(), (t[:, 0] - t0).max(), (t[:, 5] - t0).min(), (t[:, 5] - t0).max(),
        (t[:, 5] - t[:, 0]).mean(), (t[:, 5] - t[:, 0]).max()))
    print("   phases mean / max (us): issue %.2f / %.2f  prologue %.2f / %.2f  search %.2f / %.2f  leftovers %.2f / %.2f  "
          "rows+sums %.2f / %.2f" % tuple(x for k in range(5) for x in (ph[:, k].mean(), ph[:, k].max())))
    print("   tiles by mode (0 empty, 1 slab, 2 global): %s   slab targets mean %.0f max %d   cells mean %.0f max %d   "
          "open after 2x2x2: total %d max %d   brute: %d" % (np.bincount(mode, minlength=3).tolist(), npts[mode == 1].mean() if (mode == 1).any() else 0,
                                                              npts.max(), ncell[mode == 1].mean() if (mode == 1).any() else 0, ncell.max(),
                                                              int(r[:, 8].sum()), int(r[:, 8].max()), int(nbrute.sum())))
