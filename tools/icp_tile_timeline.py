"""Debugging aid: per-block timeline of the two half-iteration launches of the tile engine's last iteration.
Needs a library built with GRADSLAM_HIP_BUILD_FLAGS=-DGS_ICP_TIMELINE.
    GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl.txt python tools/icp_tile_timeline.py [B] [H W] [frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gradslam_amd as gs
from gradslam_amd.datasets.synthetic import make_sequence
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640)
L = int(sys.argv[4]) if len(sys.argv) > 4 else 4
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
names = ("issue", "prologue", "lists", "scans", "leftovers", "rows+sums")
nl = int(rows[:, 0].max()) + 1
starts = np.array([rows[rows[:, 0] == l][:, 2].min() for l in range(nl)], dtype=np.float64) / 100.0
period = np.diff(starts)
print("launch periods (us) over the solve: " + " ".join("%.1f" % x for x in period) + "   mean %.2f" % period.mean())
detail = [int(x) for x in os.environ.get("TL_LAUNCHES", "0,1,2,3,%d,%d" % (nl - 2, nl - 1)).split(",")]
for l in detail:
    r = rows[rows[:, 0] == l]
    name = "it %d %s" % (l // 2, "full" if l % 2 == 0 else "look-ahead")
    t = r[:, 2:9].astype(np.float64) / 100.0   # us
    t0 = t[:, 0].min()
    ph = np.diff(t, axis=1)
    c = r[:, 9]
    nun, mode, ns, nh = (c & 0x3ff).astype(int), ((c >> 10) & 3).astype(int), ((c >> 12) & 0x3ff).astype(int), ((c >> 22) & 0x3ff).astype(int)
    npts, ncell = ((c >> 32) & 0xfff).astype(int), (c >> 44).astype(int)
    print("%-18s blocks %d  block start 0..%.2f  end %.2f..%.2f (pct 50/90/99: %s)  life mean %.2f max %.2f" % (
        name, len(r), (t[:, 0] - t0).max(), (t[:, 6] - t0).min(), (t[:, 6] - t0).max(),
        np.round(np.percentile(t[:, 6] - t0, [50, 90, 99]), 2).tolist(), (t[:, 6] - t[:, 0]).mean(), (t[:, 6] - t[:, 0]).max()))
    print("   phases mean / max (us): " + "  ".join("%s %.2f / %.2f" % (names[k], ph[:, k].mean(), ph[:, k].max()) for k in range(6)))
    m1 = mode == 1
    print("   tiles by mode (0 empty, 1 slab, 2 global): %s   slab targets mean %.0f max %d   cells mean %.0f max %d" % (
        np.bincount(mode, minlength=3).tolist(), npts[m1].mean() if m1.any() else 0, npts.max(), ncell[m1].mean() if m1.any() else 0, ncell.max()))
    print("   queries without a proof (scan pass): total %d  per tile mean %.1f max %d   open after the scan: total %d max %d   brute: %d" % (
        ns.sum(), ns.mean(), ns.max(), nh.sum(), nh.max(), nun.sum()))
    worst = np.argsort(-(t[:, 6] - t0))[:5]
    print("   slowest blocks (block, end, mode, scans, open, phases): " + "; ".join(
        "%d %.1f m%d s%d o%d [%s]" % (int(r[i, 1]), t[i, 6] - t0, mode[i], ns[i], nh[i], " ".join("%.1f" % x for x in ph[i])) for i in worst))
