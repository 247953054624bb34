"""The projective helpers of the reference's geometry/projutils.py that the SLAM path touches.
`inverse_intrinsics` is evaluated inside the HIP back-projection kernel (gs_frame_maps_f32); the
function below is the API-level equivalent on (*, 4, 4) / (*, 3, 3) tensors (closed form,
geometry/projutils.py:405-450), kept as tensor plumbing because it is 4 scalars per camera."""
import torch

__all__ = ["homogenize_points", "unhomogenize_points", "inverse_intrinsics"]


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
