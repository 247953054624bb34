"""PointFusion driver with the reference's constructor and `_map` override
(slam/pointfusion.py:16-112): ICPSLAM odometry + surfel fusion map update."""
import math
import warnings
from typing import Union

import torch

from ..structures.pointclouds import Pointclouds
from ..structures.rgbdimages import RGBDImages
from .fusionutils import update_map_fusion
from .icpslam import ICPSLAM

__all__ = ["PointFusion"]


class PointFusion(ICPSLAM):
    r"""Point-based Fusion (Keller et al.) on top of ICP/gradICP odometry."""

    def __init__(self, *, odom: str = "gradicp", dist_th: Union[float, int] = 0.05, angle_th: Union[float, int] = 20,
                 sigma: Union[float, int] = 0.6, dsratio: int = 4, numiters: int = 20, damp: float = 1e-8,
                 dist_thresh: Union[float, int, None] = None, lambda_max: Union[float, int] = 2.0,
                 B: Union[float, int] = 1.0, B2: Union[float, int] = 1.0, nu: Union[float, int] = 200.0,
                 device: Union[torch.device, str, None] = None):
        super().__init__(odom=odom, dsratio=dsratio, numiters=numiters, damp=damp, dist_thresh=dist_thresh,
                         lambda_max=lambda_max, B=B, B2=B2, nu=nu, device=device)
        if not (isinstance(dist_th, float) or isinstance(dist_th, int)):
            raise TypeError("Distance threshold must be of type float or int; but was of type {}.".format(
                type(dist_th)))
        if not (isinstance(angle_th, float) or isinstance(angle_th, int)):
            raise TypeError("Angle threshold must be of type float or int; but was of type {}.".format(
                type(angle_th)))
        if dist_th < 0:
            warnings.warn("Distance threshold ({}) should be non-negative.".format(dist_th))
        if not ((0 <= angle_th) and (angle_th <= 90)):
            warnings.warn("Angle threshold ({}) should be non-negative and <=90.".format(angle_th))
        self.dist_th = dist_th
        rad_th = (angle_th * math.pi) / 180
        self.dot_th = torch.cos(rad_th) if torch.is_tensor(rad_th) else math.cos(rad_th)
        self.sigma = sigma

    def step(self, pointclouds: Pointclouds, live_frame: RGBDImages, prev_frame=None, inplace: bool = False):
        # the plain SLAM loop (in place, nothing on the autograd tape, a map with surfels): one foreign call per frame
        # (slam/_fastpath.py: same kernels in the same order as _localize + _map below); anything else, and every
        # subclass that overrides _localize / _map, takes the generic path
        if inplace and type(self) is PointFusion and isinstance(live_frame, RGBDImages) and \
                isinstance(prev_frame, RGBDImages) and isinstance(pointclouds, Pointclouds):
            from ._fastpath import try_step
            res = try_step(self, pointclouds, live_frame, prev_frame)
            if res is not None:
                return res
        return super().step(pointclouds, live_frame, prev_frame, inplace)

    def _localize(self, pointclouds: Pointclouds, live_frame: RGBDImages, prev_frame: RGBDImages):
        if isinstance(live_frame, RGBDImages):
            # the fusion step needs the sample confidences exp(-|v|^2 / 2 sigma^2): have the kernel that
            # builds the vertex / normal maps of this frame emit them in the same pass
            live_frame._sigma_hint = float(self.sigma)
        return super()._localize(pointclouds, live_frame, prev_frame)

    def _map(self, pointclouds: Pointclouds, live_frame: RGBDImages, inplace: bool = False):
        return update_map_fusion(pointclouds, live_frame, self.dist_th, self.dot_th, self.sigma, inplace)
