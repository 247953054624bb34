"""Oracle drivers: the reference's ICPSLAM / PointFusion frame loop (slam/icpslam.py:99-264,
slam/pointfusion.py:107-112) for ONE sequence, composed from the oracle primitives.
TEST INFRASTRUCTURE ONLY (see oracle/gs_oracle.c header)."""
import math

import numpy as np

from . import oracle as o


class MapState:
    def __init__(self):
        self.points = np.zeros((0, 3), np.float32)
        self.normals = np.zeros((0, 3), np.float32)
        self.colors = np.zeros((0, 3), np.float32)
        self.ccounts = np.zeros((0, 1), np.float32)

    def __len__(self):
        return self.points.shape[0]


def localize(m, depth, K, prev_pose, v, n, rgb, dsratio, icp_kw):
    """slam/icpslam.py:238-247."""
    H, W = depth.shape
    gv, gn = o.global_maps(v, n, depth, prev_pose)
    src, _, _ = o.downsample_frame(gv, gn, rgb, depth, dsratio)
    pix = o.project_map(m.points, prev_pose, K, H, W)
    tgt, tgtn, _ = o.select_targets(pix, W, dsratio, m.points, m.normals)
    T, _ = o.icp(src, tgt, tgtn, init=None, compose=prev_pose, **icp_kw)
    return T


def run_sequence(colors, depths, K, poses=None, *, slam="pointfusion", odom="gradicp", dsratio=4,
                 numiters=20, damp=1e-8, dist_thresh=None, lambda_max=2.0, B=1.0, B2=1.0, nu=200.0,
                 dist_th=0.05, angle_th=20, sigma=0.6, renorm_all=True, per_frame=None):
    """colors (L,H,W,3), depths (L,H,W) or (L,H,W,1), K (4,4), poses (L,4,4) or None.
    Returns (MapState, recovered_poses (L,4,4))."""
    depths = np.asarray(depths, np.float32).reshape(depths.shape[0], depths.shape[1], depths.shape[2])
    L, H, W = depths.shape
    K = np.asarray(K, np.float32).reshape(4, 4)
    dot_th = math.cos((angle_th * math.pi) / 180)
    icp_kw = dict(mode=0 if odom == "icp" else 1, numiters=numiters, damp=damp, dist_thresh=dist_thresh,
                  lambda_max=lambda_max, B=B, B2=B2, nu=nu)
    m = MapState()
    out = np.zeros((L, 4, 4), np.float32)
    prev_pose = None
    for s in range(L):
        depth = depths[s]
        rgb = np.asarray(colors[s], np.float32)
        v, n, a, _ = o.frame_maps(depth, K, sigma)
        if s == 0 or odom == "gt":
            pose = np.eye(4, dtype=np.float32) if poses is None else np.asarray(poses[s], np.float32)
        else:
            pose = localize(m, depth, K, prev_pose, v, n, rgb, dsratio, icp_kw)
        gv, gn = o.global_maps(v, n, depth, pose)
        if slam == "pointfusion":
            if len(m):
                pix = o.project_map(m.points, pose, K, H, W)
                best, _ = o.associate(pix, m.points, m.normals, m.ccounts, gv, gn, dist_th, dot_th)
            else:
                best = np.full(H * W, -1, np.int32)
            m.points, m.normals, m.colors, m.ccounts = o.fuse_append(
                m.points, m.normals, m.colors, m.ccounts, best, gv, gn, rgb, a, depth, renorm_all)
        else:
            m.points, m.normals, m.colors, m.ccounts = o.append_valid(
                m.points if len(m) else None, m.normals, m.colors, None, gv, gn, rgb, None, depth)
        out[s] = pose
        prev_pose = pose
        if per_frame is not None:
            per_frame(s, m, pose)
    return m, out
