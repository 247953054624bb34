"""The projective helpers of the reference's geometry/projutils.py that the SLAM path touches.
`inverse_intrinsics` is evaluated inside the HIP back-projection kernel (gs_frame_maps_f32); the
function below is the API-level equivalent on (*, 4, 4) / (*, 3, 3) tensors (closed form,
geometry/projutils.py:405-450), kept as tensor plumbing because it is 4 scalars per camera."""
import torch

__all__ = ["homogenize_points", "unhomogenize_points", "project_points", "unproject_points", "inverse_intrinsics"]


def homogenize_points(pts: torch.Tensor):
    if not isinstance(pts, torch.Tensor):
        raise TypeError("Expected input type torch.Tensor. Got {} instead".format(type(pts)))
    if pts.dim() < 2:
        raise ValueError("Input tensor must have at least 2 dimensions. Got {} instad.".format(pts.dim()))
    return torch.nn.functional.pad(pts, (0, 1), "constant", 1.0)


def unhomogenize_points(pts: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    if not isinstance(pts, torch.Tensor):
        raise TypeError("Expected input type torch.Tensor. Got {} instead".format(type(pts)))
    if pts.dim() < 2:
        raise ValueError("Input tensor must have at least 2 dimensions. Got {} instad.".format(pts.dim()))
    w = pts[..., -1:]
    scale = torch.where(torch.abs(w) > eps, 1.0 / w, torch.ones_like(w))
    return scale * pts[..., :-1]


def _per_point_matrices(mat, pts_shape, k):
    """(the reference's broadcasting, projutils.py:214-229 / :378-398): a (k, k) matrix serves every point; a
    (*, k, k) stack is unsqueezed at dim -3 and broadcast against the points (*, P, .): returns the (m, k, k) stack
    and the number of consecutive points per matrix"""
    lead = tuple(pts_shape[:-1])   # one entry per point
    if mat.dim() == 2:
        n = 1
        for d in lead:
            n *= int(d)
        return mat.reshape(1, k, k), max(n, 1)
    full = torch.broadcast_to(mat.unsqueeze(-3), lead + (k, k))   # a view: no arithmetic
    per = int(lead[-1]) if len(lead) else 1
    return full[..., 0, :, :].reshape(-1, k, k).contiguous(), max(per, 1)


def project_points(cam_coords: torch.Tensor, proj_mat: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    r"""Camera-frame points (N, *, 3) or (*, 4) -> pixel coordinates (*, 2) (geometry/projutils.py:92-238; HIP kernel
    gs_project_points_f32)."""
    if not torch.is_tensor(cam_coords):
        raise TypeError("Expected input cam_coords to be of type torch.Tensor. Got {0} instead.".format(type(cam_coords)))
    if not torch.is_tensor(proj_mat):
        raise TypeError("Expected input proj_mat to be of type torch.Tensor. Got {0} instead.".format(type(proj_mat)))
    if cam_coords.dim() < 2:
        raise ValueError("Input cam_coords must have at least 2 dims. Got {0} instead.".format(cam_coords.dim()))
    if cam_coords.shape[-1] not in (3, 4):
        raise ValueError("Input cam_coords must have shape (*, 3), or (*, 4). Got {0} instead.".format(cam_coords.shape))
    if proj_mat.dim() < 2:
        raise ValueError("Input proj_mat must have at least 2 dims. Got {0} instead.".format(proj_mat.dim()))
    if proj_mat.shape[-1] != 4 or proj_mat.shape[-2] != 4:
        raise ValueError("Input proj_mat must have shape (*, 4, 4). Got {0} instead.".format(proj_mat.shape))
    if proj_mat.dim() > 2 and proj_mat.dim() != cam_coords.dim():
        raise ValueError("Input proj_mat must either have 2 dimensions, or have equal number of dimensions to cam_coords. "
                         "Got {0} instead.".format(proj_mat.dim()))
    if proj_mat.dim() > 2 and proj_mat.shape[0] != cam_coords.shape[0]:
        raise ValueError("Batch sizes of proj_mat and cam_coords do not match. Shapes: {0} and {1} respectively.".format(
            proj_mat.shape, cam_coords.shape))
    from .. import ops
    mats, per = _per_point_matrices(proj_mat, cam_coords.shape, 4)
    out = ops.project_points(cam_coords.reshape(-1, cam_coords.shape[-1]), mats, per)
    return out.view(tuple(cam_coords.shape[:-1]) + (2,)).to(cam_coords.dtype)


def unproject_points(pixel_coords: torch.Tensor, intrinsics_inv: torch.Tensor, depths: torch.Tensor) -> torch.Tensor:
    r"""Pixel coordinates (N, *, 2) or (*, 3) + depths -> camera-frame points (*, 3) (geometry/projutils.py:241-402; HIP
    kernel gs_unproject_points_f32)."""
    if not torch.is_tensor(pixel_coords):
        raise TypeError("Expected input pixel_coords to be of type torch.Tensor. Got {0} instead.".format(
            type(pixel_coords)))
    if not torch.is_tensor(intrinsics_inv):
        raise TypeError("Expected intrinsics_inv to be of type torch.Tensor. Got {0} instead.".format(
            type(intrinsics_inv)))
    if not torch.is_tensor(depths):
        raise TypeError("Expected depth to be of type torch.Tensor. Got {0} instead.".format(type(depths)))
    if pixel_coords.dim() < 2:
        raise ValueError("Input pixel_coords must have at least 2 dims. Got {0} instead.".format(pixel_coords.dim()))
    if pixel_coords.shape[-1] not in (2, 3):
        raise ValueError("Input pixel_coords must have shape (*, 2), or (*, 2). Got {0} instead.".format(
            pixel_coords.shape))
    if intrinsics_inv.dim() < 2:
        raise ValueError("Input intrinsics_inv must have at least 2 dims. Got {0} instead.".format(intrinsics_inv.dim()))
    if intrinsics_inv.shape[-1] != 3 or intrinsics_inv.shape[-2] != 3:
        raise ValueError("Input intrinsics_inv must have shape (*, 3, 3). Got {0} instead.".format(intrinsics_inv.shape))
    if intrinsics_inv.dim() > 2 and intrinsics_inv.dim() != pixel_coords.dim():
        raise ValueError("Input intrinsics_inv must either have 2 dimensions, or have equal number of dimensions to "
                         "pixel_coords. Got {0} instead.".format(intrinsics_inv.dim()))
    if intrinsics_inv.dim() > 2 and intrinsics_inv.shape[0] != pixel_coords.shape[0]:
        raise ValueError("Batch sizes of intrinsics_inv and pixel_coords do not match. Shapes: {0} and {1} "
                         "respectively.".format(intrinsics_inv.shape, pixel_coords.shape))
    if pixel_coords.shape[:-1] != depths.shape:
        raise ValueError("Input pixel_coords and depths must have the same shape for all dimensions except the last. "
                         " Got {0} and {1} respectively.".format(pixel_coords.shape, depths.shape))
    from .. import ops
    mats, per = _per_point_matrices(intrinsics_inv, pixel_coords.shape, 3)
    out = ops.unproject_points(pixel_coords.reshape(-1, pixel_coords.shape[-1]), mats, depths.reshape(-1), per)
    return out.view(tuple(pixel_coords.shape[:-1]) + (3,)).to(pixel_coords.dtype)


def inverse_intrinsics(K: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    if not torch.is_tensor(K):
        raise TypeError("Expected K to be of type torch.Tensor. Got {0} instead.".format(type(K)))
    if K.dim() < 2:
        raise ValueError("Input K must have at least 2 dims. Got {0} instead.".format(K.dim()))
    if not ((K.shape[-1] == 3 and K.shape[-2] == 3) or (K.shape[-1] == 4 and K.shape[-2] == 4)):
        raise ValueError("Input K must have shape (*, 4, 4) or (*, 3, 3). Got {0} instead.".format(K.shape))
    Kinv = torch.zeros_like(K)
    fx, fy, cx, cy = K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]
    Kinv[..., 0, 0] = 1.0 / (fx + eps)
    Kinv[..., 1, 1] = 1.0 / (fy + eps)
    Kinv[..., 0, 2] = -1.0 * cx / (fx + eps)
    Kinv[..., 1, 2] = -1.0 * cy / (fy + eps)
    Kinv[..., 2, 2] = 1
    Kinv[..., -1, -1] = 1
    return Kinv
