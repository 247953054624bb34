"""BASELINE configs[4] shape check: 1296x968 frames, long sequence, map growing to millions of surfels
on one MI355X.  Prints fps over windows of the sequence, the surfel count, capacity and ATE vs ground truth.
(Synthetic stand-in for ScanNet: the camera sweeps sideways so that new surface keeps entering the view.)"""
import sys, time, argparse
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import gradslam_amd as gs
from gradslam_amd.datasets.synthetic import gt_pose, tum_intrinsics


def frames_torch(s0, n, H, W, dev, hole_frac=0.05):
    """gradslam_amd.datasets.synthetic.make_sequence's scene, ray-cast with torch on the GPU (the numpy
    generator needs ~1 s per 1296x968 frame)."""
    K = tum_intrinsics(H, W)
    u, v = torch.meshgrid(torch.arange(W, dtype=torch.float64, device=dev), torch.arange(H, dtype=torch.float64, device=dev),
                          indexing="xy")
    rx, ry = (u - float(K[0, 2])) / float(K[0, 0]), (v - float(K[1, 2])) / float(K[1, 1])
    g = torch.Generator(device=dev); g.manual_seed(1234 + s0)
    depths, poses = [], []
    for s in range(s0, s0 + n):
        T = torch.from_numpy(gt_pose(s).astype(np.float64)).to(dev)
        R, t = T[:3, :3], T[:3, 3]
        d = torch.full((H, W), 2.0, dtype=torch.float64, device=dev)
        for _ in range(30):
            pw = torch.stack([rx * d, ry * d, d], -1) @ R.T + t
            zw = 2.0 + 0.3 * torch.sin(3.0 * pw[..., 0] + 0.4) * torch.cos(2.5 * pw[..., 1]) + 0.2 * pw[..., 0]
            d = d + (zw - pw[..., 2]) / R[2, 2]
        d = torch.where(torch.rand((H, W), generator=g, device=dev) < hole_frac, torch.zeros_like(d), d)
        depths.append(d.float()[..., None]); poses.append(T.float())
    colors = torch.rand((n, H, W, 3), generator=g, device=dev) * 255.0
    return colors, torch.stack(depths), torch.from_numpy(K[None]).to(dev), torch.stack(poses)

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--chunk", type=int, default=20)
ap.add_argument("--height", type=int, default=968)
ap.add_argument("--width", type=int, default=1296)
a = ap.parse_args()
dev = torch.device("cuda")
slam = gs.slam.PointFusion(odom="gradicp", device=dev)
pc, prev, poses_rec, poses_gt = gs.Pointclouds(device=dev), None, [], []
t_all = 0.0
for c0 in range(0, a.frames, a.chunk):
    n = min(a.chunk, a.frames - c0)
    colors, depths, K, gt = frames_torch(c0, n, a.height, a.width, dev)
    given = gt.clone()
    if c0 > 0:
        given[:] = 0   # only the very first pose is given; the rest is recovered by ICP
    frames = gs.RGBDImages(colors[None], depths[None], K[None], given[None])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(n):
        live = frames[:, s]
        pc, pose = slam.step(pc, live, prev, inplace=True)
        prev = live
        poses_rec.append(pose[0, 0])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    t_all += dt
    bound = pc._count_of(0)[0]
    cap = pc._buf["points"][0].shape[0]
    n_map = pc.points_list[0].shape[0]
    poses_gt.append(gt.cpu())
    print("frames %4d-%4d: %.2f ms/frame (%.0f fps)  surfels %d  (host bound was %d, capacity %d, %.0f MB store)"
          % (c0, c0 + n - 1, dt / n * 1e3, n / dt, n_map, bound, cap, cap * 40 / 1e6), flush=True)
rec = torch.stack(poses_rec).cpu()
gt = torch.cat(poses_gt)
print("total %d frames in %.2f s (%.0f fps); ATE vs ground truth %.3e m; finite poses %s; map finite %s"
      % (a.frames, t_all, a.frames / t_all, gs.metrics.ate_rmse(rec, gt), bool(torch.isfinite(rec).all()),
         bool(torch.isfinite(pc.points_list[0]).all())))
