"""ICPOdometryProvider / GradICPOdometryProvider with the reference's constructor arguments,
validation and return shape (odometry/icp.py:12-97, odometry/gradicp.py:12-122); every sequence
of the batch is one gs_icp_f32 call (whole LM loop on the device)."""
from typing import Union

import torch

from ..structures.pointclouds import Pointclouds
from .base import OdometryProvider

__all__ = ["ICPOdometryProvider", "GradICPOdometryProvider"]


def _check_pair(maps_pointclouds, frames_pointclouds, who):
    if not isinstance(maps_pointclouds, Pointclouds):
        raise TypeError("Expected maps_pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(
            type(maps_pointclouds)))
    if not isinstance(frames_pointclouds, Pointclouds):
        raise TypeError("Expected frames_pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(
            type(frames_pointclouds)))
    if maps_pointclouds.normals_list is None:
        raise ValueError("maps_pointclouds missing normals. Map normals must be provided if using " + who)
    if len(maps_pointclouds) != len(frames_pointclouds):
        raise ValueError("Batch size of maps_pointclouds and frames_pointclouds should be equal ({0} != {1})".format(
            len(maps_pointclouds), len(frames_pointclouds)))


class ICPOdometryProvider(OdometryProvider):
    r"""Point-to-plane ICP with the LM solver (reference: odometry/icp.py:12-97)."""

    def __init__(self, numiters: int = 20, damp: float = 1e-8, dist_thresh: Union[float, int, None] = None):
        self.numiters = numiters
        self.damp = damp
        self.dist_thresh = dist_thresh

    _mode = 0

    def _kwargs(self):
        return dict(numiters=self.numiters, damp=self.damp, dist_thresh=self.dist_thresh)

    def provide(self, maps_pointclouds: Pointclouds, frames_pointclouds: Pointclouds, *,
                compose_with: torch.Tensor = None) -> torch.Tensor:
        r"""Relative transform (B, 1, 4, 4) aligning `frames_pointclouds` to `maps_pointclouds`.
        `compose_with` (B, 4, 4), an extension over the reference, fuses the caller's
        compose_transformations(T, prev_pose) (slam/icpslam.py:245-247) into the last kernel."""
        _check_pair(maps_pointclouds, frames_pointclouds, type(self).__name__)
        from .. import ops
        transforms = []
        for b in range(len(maps_pointclouds)):
            src, tgt, tn = (frames_pointclouds.points_list[b], maps_pointclouds.points_list[b],
                            maps_pointclouds.normals_list[b])
            if torch.is_grad_enabled() and (src.requires_grad or tgt.requires_grad or tn.requires_grad):
                # differentiable like the reference's providers (hand-written HIP backward, both solvers)
                T, _ = ops.grad_icp(src, tgt, tn, None, mode=self._mode, **self._kwargs())
                if compose_with is not None:
                    from ..slam.icpslam import _compose
                    T = _compose(T, compose_with[b])
            else:
                T = ops.icp(src, tgt, tn, init=None, compose=None if compose_with is None else compose_with[b],
                            mode=self._mode, return_idx=False, **self._kwargs())
            transforms.append(T)
        return torch.stack(transforms).unsqueeze(1)


class GradICPOdometryProvider(ICPOdometryProvider):
    r"""Point-to-plane ICP with the gradLM solver (reference: odometry/gradicp.py:12-122)."""

    def __init__(self, numiters: int = 20, damp: float = 1e-8, dist_thresh: Union[float, int, None] = None,
                 lambda_max: Union[float, int] = 2.0, B: Union[float, int] = 1.0, B2: Union[float, int] = 1.0,
                 nu: Union[float, int] = 200.0):
        super().__init__(numiters, damp, dist_thresh)
        self.lambda_max = lambda_max
        self.B = B
        self.B2 = B2
        self.nu = nu

    _mode = 1

    def _kwargs(self):
        return dict(numiters=self.numiters, damp=self.damp, dist_thresh=self.dist_thresh,
                    lambda_max=self.lambda_max, B=self.B, B2=self.B2, nu=self.nu)
