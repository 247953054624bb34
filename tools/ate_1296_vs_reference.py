"""ATE and surfel-count differences of the HIP PointFusion against the real reference's 8-frame run at 1296x968
(tests/golden/pf1296_s3.npz).   python tools/ate_1296_vs_reference.py"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gradslam_amd as gs
from gradslam_amd.datasets.synthetic import make_sequence
from gradslam_amd.metrics import ate_rmse as ate
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pf1296_s3.npz"))
L = g["poses"].shape[0]
s = make_sequence(L, 968, 1296, seed=3)
T = torch.from_numpy
poses = s["poses"].copy(); poses[1:] = poses[:1]
fr = gs.RGBDImages(T(s["colors"][None]).cuda(), T(s["depths"][None]).cuda(), T(s["intrinsics"][None]).cuda(), T(poses[None]).cuda())
slam = gs.slam.PointFusion(odom="gradicp", device="cuda")
pc, prev, rec, cnt = gs.Pointclouds(device="cuda"), None, [], []
for f in range(L):
    live = fr[:, f]; pc, p = slam.step(pc, live, prev, inplace=True); prev = live
    rec.append(p[0, 0].cpu().numpy()); cnt.append(pc.points_list[0].shape[0])
print("ATE", ate(np.stack(rec), g["poses"]), "count diffs", (np.asarray(cnt) - g["counts"]).tolist(), "counts", g["counts"].tolist())
