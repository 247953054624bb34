"""ICPSLAM driver with the reference's constructor, `forward`, `step`, `_localize`, `_map`
(slam/icpslam.py:18-264).  `_localize` keeps the reference's data flow (live frame back-projected
under the previous pose, map points active in the previous frame on the ds lattice as ICP target,
T_icp composed with the previous pose) but runs it as: gs_project_map_f32 ->
gs_select_targets_f32 -> gs_downsample_frame_f32 -> gs_icp_f32 (20 LM iterations + compose on
the device, no host sync inside)."""
import warnings
from typing import Optional, Union

import torch
import torch.nn as nn

from ..odometry.gradicp import GradICPOdometryProvider
from ..odometry.icp import ICPOdometryProvider
from ..structures.pointclouds import Pointclouds
from ..structures.rgbdimages import RGBDImages
from .fusionutils import update_map_aggregate

__all__ = ["ICPSLAM"]


class ICPSLAM(nn.Module):
    r"""Point-to-plane ICP odometry + aggregate mapping (every valid pixel is appended)."""

    def __init__(self, *, odom: str = "gradicp", dsratio: int = 4, numiters: int = 20, damp: float = 1e-8,
                 dist_thresh: Union[float, int, None] = None, lambda_max: Union[float, int] = 2.0,
                 B: Union[float, int] = 1.0, B2: Union[float, int] = 1.0, nu: Union[float, int] = 200.0,
                 device: Union[torch.device, str, None] = None):
        super().__init__()
        if odom not in ["gt", "icp", "gradicp"]:
            msg = "odometry method ({}) not supported for PointFusion. ".format(odom)
            msg += "Currently supported odometry modules for PointFusion are: 'gt', 'icp', 'gradicp'"
            raise ValueError(msg)
        odomprov = None
        if odom == "icp":
            odomprov = ICPOdometryProvider(numiters, damp, dist_thresh)
        elif odom == "gradicp":
            odomprov = GradICPOdometryProvider(numiters, damp, dist_thresh, lambda_max, B, B2, nu)
        self.odom = odom
        self.odomprov = odomprov
        self.dsratio = dsratio
        device = torch.device(device) if device is not None else torch.device("cpu")
        self.device = torch.Tensor().to(device).device

    def forward(self, frames: RGBDImages):
        r"""Returns (pointclouds: B global maps, poses (B, L, 4, 4))."""
        if not isinstance(frames, RGBDImages):
            raise TypeError("Expected frames to be of type gradslam.RGBDImages. Got {0}.".format(type(frames)))
        pointclouds = Pointclouds(device=self.device)
        batch_size, seq_len = frames.shape[:2]
        recovered_poses = torch.empty(batch_size, seq_len, 4, 4).to(self.device)
        prev_frame = None
        for s in range(seq_len):
            live_frame = frames[:, s].to(self.device)
            if s == 0 and live_frame.poses is None:
                live_frame.poses = (torch.eye(4, dtype=torch.float, device=self.device).view(1, 1, 4, 4)
                                    .repeat(batch_size, 1, 1, 1))
            pointclouds, live_frame.poses = self.step(pointclouds, live_frame, prev_frame, inplace=True)
            prev_frame = live_frame if self.odom != "gt" else None
            recovered_poses[:, s] = live_frame.poses[:, 0]
        return pointclouds, recovered_poses

    def step(self, pointclouds: Pointclouds, live_frame: RGBDImages, prev_frame: Optional[RGBDImages] = None,
             inplace: bool = False):
        if not isinstance(live_frame, RGBDImages):
            raise TypeError("Expected live_frame to be of type gradslam.RGBDImages. Got {0}.".format(
                type(live_frame)))
        live_frame.poses = self._localize(pointclouds, live_frame, prev_frame)
        pointclouds = self._map(pointclouds, live_frame, inplace)
        return pointclouds, live_frame.poses

    def _localize(self, pointclouds: Pointclouds, live_frame: RGBDImages, prev_frame: RGBDImages):
        if not isinstance(pointclouds, Pointclouds):
            raise TypeError("Expected pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(
                type(pointclouds)))
        if not isinstance(live_frame, RGBDImages):
            raise TypeError("Expected live_frame to be of type gradslam.RGBDImages. Got {0}.".format(
                type(live_frame)))
        if not isinstance(prev_frame, (RGBDImages, type(None))):
            raise TypeError("Expected prev_frame to be of type gradslam.RGBDImages or None. Got {0}.".format(
                type(prev_frame)))
        if prev_frame is not None:
            if self.odom == "gt":
                warnings.warn("`prev_frame` is not used when using `odom='gt'` (should be None)")
            elif not prev_frame.has_poses:
                raise ValueError("`prev_frame` should have poses, but did not.")
        if prev_frame is None and pointclouds.has_points and self.odom != "gt":
            warnings.warn("`prev_frame` was None despite `{}` odometry method. Using `live_frame` poses.".format(
                self.odom))
        if prev_frame is None or self.odom == "gt":
            if not live_frame.has_poses:
                raise ValueError("`live_frame` must have poses when `prev_frame` is None or `odom='gt'`.")
            return live_frame.poses

        if self.odom in ["icp", "gradicp"]:
            from .. import ops
            live_frame.poses = prev_frame.poses  # initial guess: the live frame sits at the previous pose
            fr = live_frame.to_channels_last()
            B, _, H, W = fr.shape
            K = prev_frame.intrinsics[:, 0].contiguous().float()
            prev_poses = prev_frame.poses[:, 0].contiguous().float()
            src_pts, tgt_pts, tgt_nrm = [], [], []
            # on the autograd tape whenever the reference's graph would be: the live depth, the previous pose or the
            # map (points / normals) requires grad -- for BOTH solvers (the hard-LM ICP back-propagates through its
            # accepted steps, gs_icp_backward_f32 mode 0)
            builtin = type(self.odomprov) in (ICPOdometryProvider, GradICPOdometryProvider)
            taped = torch.is_grad_enabled() and builtin and (
                fr.depth_image.requires_grad or prev_frame.poses.requires_grad or _map_requires_grad(pointclouds))
            if not taped and builtin:
                # fast path: no host read-back and no compaction of the source set.  The ICP source is the
                # frame's [::ds, ::ds] lattice of global vertices under the previous pose (NaN = no depth,
                # skipped by the solver), built from the LOCAL vertex map in one launch; the size of the
                # target set stays on the device.
                # All sequences of the batch go through ONE chain of launches (gs_localize_batch_f32).
                out = torch.empty((B, 1, 4, 4), dtype=torch.float32, device=fr.device)
                maps = []
                for b in range(B):
                    n_b, n_dev = pointclouds._count_of(b)
                    maps.append((pointclouds._buf["points"][b], pointclouds._buf["normals"][b], n_b, n_dev))
                if all(m[2] > 0 for m in maps):
                    ops.localize_batch(fr.vertex_map[:, 0], fr.depth_image[:, 0, ..., 0], K, prev_poses, maps,
                                       self.dsratio, mode=self.odomprov._mode, out=out.view(B, 4, 4),
                                       **self.odomprov._kwargs())
                    return out
            gvm = fr.global_vertex_map
            for b in range(B):
                # downsample_rgbdimages(live_frame): valid lattice pixels of the global vertex map
                if taped and gvm.requires_grad:
                    p = ops.DownsampleFramePointsFunction.apply(gvm[b, 0], fr.depth_image[b, 0, ..., 0].detach(),
                                                                self.dsratio)
                else:
                    p, _, _ = ops.downsample_frame(gvm[b, 0], None, None, fr.depth_image[b, 0, ..., 0], self.dsratio)
                src_pts.append(p)
                # find_active_map_points(pointclouds, prev_frame) + downsample_pointclouds, without tables
                P, N = pointclouds.points_list[b], pointclouds.normals_list[b]
                pix = ops.project_map(P, prev_poses[b], K[b], H, W)
                if taped and (P.requires_grad or N.requires_grad):
                    # the map is on the tape (differentiable mapping): gather the targets by row index so that the
                    # pose gradient also reaches the earlier frames through the map (same rows, same order)
                    rows = ops.active_table(pix, W)
                    idx = rows[(rows[:, 2] % self.dsratio == 0) & (rows[:, 3] % self.dsratio == 0), 1]
                    tp, tn = P[idx], N[idx]
                else:
                    tp, tn, _ = ops.select_targets(pix, W, self.dsratio, P, N)
                tgt_pts.append(tp)
                tgt_nrm.append(tn)
            if taped:
                # differentiable pose path: depth -> vertex -> global vertex -> ICP source -> (grad)ICP -> pose, and,
                # when the map is on the tape, earlier frames -> map -> ICP targets / normals -> pose.
                out = []
                for b in range(B):
                    T, _ = ops.grad_icp(src_pts[b], tgt_pts[b], tgt_nrm[b], None, mode=self.odomprov._mode,
                                        **self.odomprov._kwargs())
                    out.append(T)
                return _compose(torch.stack(out), prev_poses).unsqueeze(1)
            maps_pc = Pointclouds(points=tgt_pts, normals=tgt_nrm)
            frames_pc = Pointclouds(points=src_pts)
            if isinstance(self.odomprov, ICPOdometryProvider):
                # compose_transformations(T, prev_pose) fused into the last ICP kernel
                return self.odomprov.provide(maps_pc, frames_pc, compose_with=prev_poses)
            transform = self.odomprov.provide(maps_pc, frames_pc)  # user-supplied OdometryProvider plugin
            return _compose(transform.squeeze(1), prev_poses).unsqueeze(1)

    def _map(self, pointclouds: Pointclouds, live_frame: RGBDImages, inplace: bool = False):
        return update_map_aggregate(pointclouds, live_frame, inplace)


def _map_requires_grad(pointclouds):
    bufs = pointclouds._buf
    return any(bufs[k] is not None and any(t.requires_grad for t in bufs[k]) for k in ("points", "normals"))


def _compose(trans_01, trans_12):
    """kornia compose_transformations for plugin providers (4x4 plumbing on the device)."""
    out = torch.zeros_like(trans_01)
    out[..., :3, :3] = trans_01[..., :3, :3] @ trans_12[..., :3, :3]
    out[..., :3, 3:] = trans_01[..., :3, :3] @ trans_12[..., :3, 3:] + trans_01[..., :3, 3:]
    out[..., 3, 3] = 1.0
    return out
