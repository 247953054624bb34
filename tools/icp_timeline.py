"""Debugging aid: per-block timeline of the last iteration's two ICP half-iteration launches (library built with
-DGS_ICP_TIMELINE; GRADSLAM_HIP_ICP_TIMELINE=<path>: first half -> <path>, the look-ahead right behind it -> <path>.next).
    GRADSLAM_HIP_ICP_TIMELINE=/tmp/tl.txt [GRADSLAM_HIP_ICP_LANES=..] python tools/icp_timeline.py [B] [frames]
Prints, per launch: when its blocks start and end, the phases of a block (prologue / search or list check + left-overs /
rows), and the launch PERIOD = first block start of the first half to first block start of the look-ahead."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gradslam_amd as gs
from gradslam_amd.datasets.synthetic import make_sequence
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seqs = [make_sequence(L, 480, 640, seed=b) for b in range(B)]
st = lambda k: torch.from_numpy(np.stack([s[k] for s in seqs])).cuda()
poses = st("poses"); poses[:, 1:] = poses[:, :1]
frames = gs.RGBDImages(st("colors"), st("depths"), st("intrinsics"), poses)
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev = gs.Pointclouds(device="cuda"), None
for f in range(L):
    live = frames[:, f]; pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
torch.cuda.synchronize()
base = os.environ["GRADSLAM_HIP_ICP_TIMELINE"]
t_first = {}
for part, label in (("", "first half"), (".next", "look-ahead")):
    path = base + part
    rows = np.loadtxt(path, dtype=np.uint64)
    print("== %s: %s" % (label, open(path).readline().strip()))
    ok = rows[:, 5] > 0
    rows = rows[ok]
    t0 = rows[:, 1].min()
    t_first[part] = int(t0)
    start, end = (rows[:, 1] - t0).astype(np.float64) / 100.0, (rows[:, 2] - t0).astype(np.float64) / 100.0   # us
    csum = rows[:, 3].astype(np.int64)
    print("blocks %d  start: min %.2f max %.2f us   end: min %.2f max %.2f   life mean %.2f max %.2f" % (
        len(rows), start.min(), start.max(), end.min(), end.max(), (end - start).mean(), (end - start).max()))
    pro = (rows[:, 5] - rows[:, 1]).astype(np.float64) / 100.0
    sea = (rows[:, 6] - rows[:, 5]).astype(np.float64) / 100.0
    rest = (rows[:, 2] - rows[:, 6]).astype(np.float64) / 100.0
    nun = (rows[:, 8] & 0xffffffff).astype(np.int64)
    print("phases (us, mean / max over blocks): prologue %.2f / %.2f   search (list check) + left-overs %.2f / %.2f   rows %.2f / %.2f" % (
        pro.mean(), pro.max(), sea.mean(), sea.max(), rest.mean(), rest.max()))
    s1 = (rows[:, 9] - rows[:, 1]).astype(np.float64) / 100.0    # start -> row sums done
    s2 = (rows[:, 10] - rows[:, 9]).astype(np.float64) / 100.0   # -> 6x6 solve done (look-ahead half)
    s3 = (rows[:, 5] - rows[:, 10]).astype(np.float64) / 100.0   # -> scalar stage done, state published
    print("prologue (us, mean over blocks): loads + sums %.2f   solve %.2f   scalar stage + barrier %.2f" % (s1.mean(), s2.mean(), s3.mean()))
    print("left-over queries: total %d, blocks with any %d, per block max %d; brute-force queries %d" % (
        csum.sum(), (csum > 0).sum(), csum.max(), nun.sum()))
    order = np.argsort(end)[::-1][:5]
    print("slowest blocks (end us: prologue / search / rows, left-overs): " + "  ".join(
        "%.1f: %.1f / %.1f / %.1f, %d" % (end[i], pro[i], sea[i], rest[i], csum[i]) for i in order))
print("launch period (first block of the first half -> first block of the look-ahead): %.2f us" % (
    (t_first[".next"] - t_first[""]) / 100.0))
