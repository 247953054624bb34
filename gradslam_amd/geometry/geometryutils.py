"""transform_pointcloud with the reference's signature (geometry/geometryutils.py:737-794),
evaluated by the HIP kernel gs_transform_points_f32."""
import torch

__all__ = ["transform_pointcloud"]


def transform_pointcloud(pointcloud: torch.Tensor, transform: torch.Tensor):
    if not torch.is_tensor(pointcloud):
        raise TypeError("pointcloud should be tensor, but was %r instead" % type(pointcloud))
    if not torch.is_tensor(transform):
        raise TypeError("transform should be tensor, but was %r instead" % type(transform))
    if not pointcloud.ndim == 2:
        raise ValueError("pointcloud should have ndim of 2, but had {} instead.".format(pointcloud.ndim))
    if not pointcloud.shape[1] == 3:
        raise ValueError("pointcloud.shape[1] should be 3 (x, y, z), but was {} instead.".format(pointcloud.shape[1]))
    if not transform.shape[-2:] == (4, 4):
        raise ValueError("transform should be of shape (4, 4), but was {} instead.".format(transform.shape))
    from .. import ops
    return ops.transform_points(pointcloud, transform)
