"""transform_pointcloud and relative_transformation with the reference's signatures
(geometry/geometryutils.py:737-794, :413-478), evaluated by the HIP kernels gs_transform_points_f32 and
gs_relative_pose_f32."""
import torch

__all__ = ["transform_pointcloud", "relative_transformation"]


def relative_transformation(trans_01: torch.Tensor, trans_02: torch.Tensor,
                            orthogonal_rotations: bool = False) -> torch.Tensor:
    r"""T_12 = inv(T_01) . T_02 for (N, 4, 4) or (4, 4) homogeneous transforms.  `orthogonal_rotations`
    is accepted for signature parity: the kernel always inverts the general matrix (in double), which
    also covers orthogonal rotations."""
    if not torch.is_tensor(trans_01):
        raise TypeError("Input trans_01 type is not a torch.Tensor. Got {}".format(type(trans_01)))
    if not torch.is_tensor(trans_02):
        raise TypeError("Input trans_02 type is not a torch.Tensor. Got {}".format(type(trans_02)))
    for t in (trans_01, trans_02):
        if t.dim() not in (2, 3) or tuple(t.shape[-2:]) != (4, 4):
            raise ValueError("Input must be a of the shape Nx4x4 or 4x4. Got {}".format(t.shape))
    if trans_01.dim() != trans_02.dim():
        raise ValueError("Input number of dims must match. Got {} and {}".format(trans_01.dim(), trans_02.dim()))
    if trans_01.shape != trans_02.shape:
        raise ValueError("Input shapes must match. Got {} and {}".format(trans_01.shape, trans_02.shape))
    from .. import ops
    return ops.relative_pose(trans_01.reshape(-1, 4, 4), trans_02.reshape(-1, 4, 4)).view(trans_02.shape)


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
