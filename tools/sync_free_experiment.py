"""Experiment: how much would removing the per-frame host syncs (device->host count read-backs) gain?
Pass 1 records every count the host reads back; pass 2 replays the recorded values WITHOUT syncing
(the pipeline is deterministic, so the values are the true ones) and times the same frames."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import gradslam_amd as gs
from gradslam_amd import ops
from gradslam_amd.datasets.synthetic import make_sequence

L, W0 = 15, 3
seq = make_sequence(L, 480, 640, seed=0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
poses = seq["poses"].copy(); poses[1:] = poses[:1]
frames = gs.RGBDImages(T(seq["colors"][None]), T(seq["depths"][None]), T(seq["intrinsics"][None]), T(poses[None]))
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
rec, mode, pos = [], "record", [0]
orig = ops._count
def hooked(t):
    if mode == "record":
        v = orig(t); rec.append(v); return v
    v = rec[pos[0]]; pos[0] += 1; return v
ops._count = hooked
def run():
    pc, prev = gs.Pointclouds(device="cuda"), None
    for s in range(L):
        if s == W0:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        live = frames[:, s]
        pc, _ = slam.step(pc, live, prev, inplace=True); prev = live
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (L - W0) * 1e3, pc.points_list[0].shape[0]
print("with syncs   : %.3f ms/frame, map %d" % run())
mode = "replay"; pos[0] = 0
print("without syncs: %.3f ms/frame, map %d" % run())
