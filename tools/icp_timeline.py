"""Debugging aid: per-block timeline of one ICP half-iteration launch (GRADSLAM_HIP_ICP_TIMELINE).
    GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl.txt [GRADSLAM_HIP_ICP_LANES=..] python tools/icp_timeline.py [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gradslam_amd as gs
from gradslam_amd.datasets.synthetic import make_sequence
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = 4
seqs = [make_sequence(L, 480, 640, seed=b) for b in range(B)]
st = lambda k: torch.from_numpy(np.stack([s[k] for s in seqs])).cuda()
poses = st("poses"); poses[:, 1:] = poses[:, :1]
frames = gs.RGBDImages(st("colors"), st("depths"), st("intrinsics"), poses)
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev = gs.Pointclouds(device="cuda"), None
for f in range(L):
    live = frames[:, f]; pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
torch.cuda.synchronize()
rows = np.loadtxt(os.environ["GRADSLAM_HIP_ICP_TIMELINE"], dtype=np.uint64)
print(open(os.environ["GRADSLAM_HIP_ICP_TIMELINE"]).readline().strip())
t0 = rows[:, 1].min()
start, end = (rows[:, 1] - t0).astype(np.float64) / 100.0, (rows[:, 2] - t0).astype(np.float64) / 100.0   # us
live = end > start
csum, cmax = rows[:, 3].astype(np.int64), rows[:, 4].astype(np.int64)
print("blocks %d  start: min %.2f max %.2f us   end: min %.2f max %.2f   life mean %.2f max %.2f" % (
    len(rows), start.min(), start.max(), end.min(), end.max(), (end - start).mean(), (end - start).max()))
pro = (rows[:, 5] - rows[:, 1]).astype(np.float64) / 100.0
sea = (rows[:, 6] - rows[:, 5]).astype(np.float64) / 100.0
unr = (rows[:, 7] - rows[:, 6]).astype(np.float64) / 100.0   # shell search of the queries stage 0 left open
rest = (rows[:, 2] - rows[:, 7]).astype(np.float64) / 100.0
nun = (rows[:, 8] & 0xffffffff).astype(np.int64)
beyond = (rows[:, 8] >> 32).astype(np.int64)
ok = rows[:, 5] > 0
print("phases (us, mean / max over blocks): prologue %.2f / %.2f   first search %.2f / %.2f   shells for open queries %.2f / %.2f   "
      "rest %.2f / %.2f" % (pro[ok].mean(), pro[ok].max(), sea[ok].mean(), sea[ok].max(), unr[ok].mean(), unr[ok].max(),
                            rest[ok].mean(), rest[ok].max()))
print("unresolved queries: total %d, per block max %d, blocks with any %d" % (nun.sum(), nun.max(), (nun > 0).sum()))
print("queries left open by the 2x2x2 stage: total %d, per block max %d" % (csum.sum(), csum.max()))
rot = 0
for s in range(B):
    m = ((np.arange(len(rows)) + rot) % B) == s
    print("seq %d: %d blocks  end max %.2f  open after stage 0: %d  slowest stage 0 %.1f  slowest shells %.1f" % (
        s, m.sum(), end[m].max(), csum[m].sum(), sea[m].max(), unr[m].max()))
for s in (0, 5):
    m = np.where(((np.arange(len(rows)) + rot) % B) == s)[0]
    print("seq %d per block (physical order = b/B): " % s + " ".join("%.0f" % sea[i] for i in m))
    print("seq %d open queries per block:          " % s + " ".join("%d" % csum[i] for i in m))
