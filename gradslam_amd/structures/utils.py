"""pointclouds_from_rgbdimages with the reference's signature (structures/utils.py:7-57); the
masked raster-order gather is the HIP ordered-compaction kernel (gs_append_valid_f32)."""
import torch

from .pointclouds import Pointclouds
from .rgbdimages import RGBDImages

__all__ = ["pointclouds_from_rgbdimages"]


def pointclouds_from_rgbdimages(rgbdimages: RGBDImages, *, global_coordinates: bool = True,
                                filter_missing_depths: bool = True) -> Pointclouds:
    if not isinstance(rgbdimages, RGBDImages):
        raise TypeError("Expected rgbdimages to be of type gradslam.RGBDImages. Got {0}.".format(type(rgbdimages)))
    if not rgbdimages.shape[1] == 1:
        raise ValueError("Expected rgbdimages to have sequence length of 1. Got {0}.".format(rgbdimages.shape[1]))
    from .. import ops
    B, _, H, W = rgbdimages.shape
    rgbdimages = rgbdimages.to_channels_last()
    vertex_map = rgbdimages.global_vertex_map if global_coordinates else rgbdimages.vertex_map
    normal_map = rgbdimages.global_normal_map if global_coordinates else rgbdimages.normal_map
    if not filter_missing_depths:
        return Pointclouds(points=vertex_map.reshape(B, -1, 3).contiguous(),
                           normals=normal_map.reshape(B, -1, 3).contiguous(),
                           colors=rgbdimages.rgb_image.reshape(B, -1, 3).contiguous())
    points, normals, colors = [], [], []
    dev = rgbdimages.device
    for b in range(B):
        bufs = [torch.empty((H * W, 3), dtype=torch.float32, device=dev) for _ in range(3)]
        n = ops.append_valid_(bufs[0], bufs[1], bufs[2], None, 0, vertex_map[b, 0], normal_map[b, 0],
                              rgbdimages.rgb_image[b, 0], None, rgbdimages.depth_image[b, 0, ..., 0])
        points.append(bufs[0][:n]); normals.append(bufs[1][:n]); colors.append(bufs[2][:n])
    return Pointclouds(points=points, normals=normals, colors=colors)
