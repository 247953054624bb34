"""Host-side helpers of the dataset loaders with the reference's names and semantics
(datasets/datautils.py:20-260).  Small per-sequence bookkeeping on numpy arrays / tensors; the
per-pixel work of the loaders (resize, depth scaling, colour normalisation) is done on the device by
gs_ingest_depth_u16_f32 / gs_ingest_color_u8_f32."""
import warnings
from typing import List, Union

import numpy as np
import torch

__all__ = ["normalize_image", "channels_first", "scale_intrinsics", "pointquaternion_to_homogeneous",
           "poses_to_transforms"]


def normalize_image(rgb: Union[torch.Tensor, np.ndarray]):
    r"""[0, 255] -> [0, 1] (datautils.py:20-40)."""
    if torch.is_tensor(rgb):
        return rgb.float() / 255
    if isinstance(rgb, np.ndarray):
        return rgb.astype(float) / 255
    raise TypeError("Unsupported input rgb type: %r" % type(rgb))


def channels_first(rgb: Union[torch.Tensor, np.ndarray]):
    r"""(*, H, W, C) -> (*, C, H, W) (datautils.py:43-74)."""
    if not (isinstance(rgb, np.ndarray) or torch.is_tensor(rgb)):
        raise TypeError("Unsupported input rgb type {}".format(type(rgb)))
    if rgb.ndim < 3:
        raise ValueError("Input rgb must contain atleast 3 dims, but had {} dims.".format(rgb.ndim))
    if rgb.shape[-3] < rgb.shape[-1]:
        warnings.warn("Are you sure that the input is correct? Number of channels exceeds height of image: %r > %r"
                      % (rgb.shape[-1], rgb.shape[-3]))
    order = list(range(rgb.ndim - 3)) + [rgb.ndim - 1, rgb.ndim - 3, rgb.ndim - 2]
    if isinstance(rgb, np.ndarray):
        return np.ascontiguousarray(rgb.transpose(*order))
    return rgb.permute(*order).contiguous()


def scale_intrinsics(intrinsics: Union[np.ndarray, torch.Tensor], h_ratio: Union[float, int],
                     w_ratio: Union[float, int]):
    r"""Intrinsics of a frame resized by (h_ratio, w_ratio) (datautils.py:77-124): fx, cx scale with the
    width, fy, cy with the height."""
    if isinstance(intrinsics, np.ndarray):
        scaled = intrinsics.astype(np.float32).copy()
    elif torch.is_tensor(intrinsics):
        scaled = intrinsics.to(torch.float).clone()
    else:
        raise TypeError("Unsupported input intrinsics type {}".format(type(intrinsics)))
    if tuple(intrinsics.shape[-2:]) not in ((3, 3), (4, 4)):
        raise ValueError("intrinsics must have shape (*, 3, 3) or (*, 4, 4), but had shape {} instead".format(
            intrinsics.shape))
    if (intrinsics[..., -1, -1] != 1).any() or (intrinsics[..., 2, 2] != 1).any():
        warnings.warn("Incorrect intrinsics: intrinsics[..., -1, -1] and intrinsics[..., 2, 2] should be 1.")
    for (r, c), ratio in (((0, 0), w_ratio), ((1, 1), h_ratio), ((0, 2), w_ratio), ((1, 2), h_ratio)):
        scaled[..., r, c] *= ratio
    return scaled


def pointquaternion_to_homogeneous(pointquaternions: Union[np.ndarray, torch.Tensor], eps: float = 1e-12):
    r"""(tx, ty, tz, qx, qy, qz, qw) -> 4x4 [R | t] in float32 (datautils.py:127-205): q is scaled by
    1 / sqrt(|q|^2 / 2), the rotation is read off the outer product q q^T."""
    if not (isinstance(pointquaternions, np.ndarray) or torch.is_tensor(pointquaternions)):
        raise TypeError('"pointquaternions" must be of type "np.ndarray" or "torch.Tensor". Got {0}'.format(
            type(pointquaternions)))
    if not isinstance(eps, float):
        raise TypeError('"eps" must be of type "float". Got {0}.'.format(type(eps)))
    if pointquaternions.shape[-1] != 7:
        raise ValueError('"pointquaternions" must be of shape (*, 7). Got {0}.'.format(pointquaternions.shape))
    is_np = isinstance(pointquaternions, np.ndarray)
    pq = pointquaternions.astype(np.float32) if is_np else pointquaternions.float()
    t, q = pq[..., :3], pq[..., 3:7]
    half_norm = (0.5 * (q ** 2).sum(-1)[..., None]) ** 0.5
    q = q / (np.maximum(half_norm, eps) if is_np else torch.max(half_norm, torch.tensor(eps)))
    o = (q[..., :, None] * q[..., None, :])   # outer product, entries 2*qi*qj/|q|^2
    T = np.zeros((*pq.shape[:-1], 4, 4), np.float32) if is_np else torch.zeros(
        (*pq.shape[:-1], 4, 4), dtype=torch.float, device=pq.device)
    x, y, z, w = 0, 1, 2, 3
    for i in range(4):
        T[..., i, i] = 1.0
    T[..., 0, 0] -= o[..., y, y] + o[..., z, z]
    T[..., 1, 1] -= o[..., x, x] + o[..., z, z]
    T[..., 2, 2] -= o[..., x, x] + o[..., y, y]
    T[..., 0, 1] = o[..., x, y] - o[..., z, w]
    T[..., 0, 2] = o[..., x, z] + o[..., y, w]
    T[..., 1, 0] = o[..., x, y] + o[..., z, w]
    T[..., 1, 2] = o[..., y, z] - o[..., x, w]
    T[..., 2, 0] = o[..., x, z] - o[..., y, w]
    T[..., 2, 1] = o[..., y, z] + o[..., x, w]
    T[..., :3, 3] = t
    return T


def poses_to_transforms(poses: Union[np.ndarray, List[np.ndarray]]):
    r"""Frame-to-frame transforms inv(pose[i-1]) . pose[i], identity first (datautils.py:208-230)."""
    out = [np.eye(4)] + [np.linalg.inv(poses[i - 1]).dot(poses[i]) for i in range(1, len(poses))]
    return np.stack(out) if isinstance(poses, np.ndarray) else out
