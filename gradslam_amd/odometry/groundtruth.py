"""Ground-truth odometry provider (reference: odometry/groundtruth.py:11-78): the relative transform
between the poses of two single-frame RGBDImages batches, T = inv(T_1) . T_2, evaluated by
gs_relative_pose_f32."""
import torch

from ..structures.rgbdimages import RGBDImages
from .base import OdometryProvider

__all__ = ["GroundTruthOdometryProvider"]


class GroundTruthOdometryProvider(OdometryProvider):
    def provide(self, rgbdimages1: RGBDImages, rgbdimages2: RGBDImages) -> torch.Tensor:
        r"""Returns (B, 1, 4, 4): the pose of `rgbdimages2` relative to `rgbdimages1`."""
        for pos, frames in ((1, rgbdimages1), (2, rgbdimages2)):
            if not isinstance(frames, RGBDImages):
                raise TypeError("Expected input {0} (rgbdimages{0}) to be of type gradslam.RGBDImages. Got {1}.".format(
                    pos, type(frames)))
        for pos, frames in ((1, rgbdimages1), (2, rgbdimages2)):
            if frames.poses is None:
                raise ValueError("Input {0} (rgbdimages{0}) missing poses. Poses must be provided if using "
                                 "GroundTruthOdometryProvider".format(pos))
        for pos, frames in ((1, rgbdimages1), (2, rgbdimages2)):
            if frames.shape[1] != 1:
                raise ValueError("Sequence length of rgbdimages{0} must be 1, but was {1}.".format(pos, frames.shape[1]))
        if rgbdimages1.shape[0] != rgbdimages2.shape[0]:
            raise ValueError("Batch size of rgbdimages1 and rgbdimages2 should be equal ({0} != {1})".format(
                rgbdimages1.shape[0], rgbdimages2.shape[0]))
        from .. import ops
        B, L = rgbdimages1.shape[:2]
        return ops.relative_pose(rgbdimages1.poses.reshape(-1, 4, 4), rgbdimages2.poses.reshape(-1, 4, 4)).view(B, L, 4, 4)
