"""Oracle for the backward pass of the frame maps (structures/rgbdimages.py:643-743 + slam/fusionutils.py:69-72):
float64 numpy reverse mode of depth -> (vertex, normal, alpha), pinned against the reference's own autograd
(tests/golden/depth_grad.npz: maps_depth_grad).  Test infrastructure only.

forward:  V[h,w] = ray[h,w] * depth * valid,  ray = (k00 w + k02, k11 h + k12, 1)
          dh[h,w] = V[h,w0+1] - V[h,w0], w0 = min(w, W-2);  dv[h,w] = V[h0+1,w] - V[h0,w], h0 = min(h, H-2)
          n = dh x dv;  N = n / where(|n| == 0, 1, |n|) * valid
          alpha = clamp(exp(-|V|^2 / (2 sigma^2)), 1e-7, 1.01)"""
import numpy as np


def frame_maps_backward(depth, K, sigma, v_bar, n_bar, a_bar, want_K=False):
    """depth_bar (H, W); with want_K also K_bar (4, 4): the adjoint of the intrinsics through inverse_intrinsics
    (k00 = 1 / (fx + eps), k02 = -cx / (fx + eps), same for y; geometry/projutils.py:437-449), pinned against
    tests/golden/intrinsics_grad.npz."""
    depth = np.asarray(depth, np.float64)
    H, W = depth.shape
    fx, fy, cx, cy = (float(K[0, 0]) + 1e-6), (float(K[1, 1]) + 1e-6), float(K[0, 2]), float(K[1, 2])
    # inverse_intrinsics (geometry/projutils.py:444-449) in float32 like the reference, then float64
    k00, k11 = np.float64(np.float32(1.0) / np.float32(fx)), np.float64(np.float32(1.0) / np.float32(fy))
    k02, k12 = np.float64(-np.float32(cx) / np.float32(fx)), np.float64(-np.float32(cy) / np.float32(fy))
    w, h = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    ray = np.stack([k00 * w + k02, k11 * h + k12, np.ones_like(w)], -1)
    valid = (depth > 0).astype(np.float64)
    V = ray * (depth * valid)[..., None]
    Vb = np.array(v_bar, np.float64, copy=True)
    # alpha
    s = (V * V).sum(-1)
    two = 2.0 * float(sigma) ** 2
    e = np.exp(-s / two)
    inside = (e >= 1e-7) & (e <= 1.01)
    Vb += (np.asarray(a_bar, np.float64) * np.where(inside, e, 0.0) * (-2.0 / two))[..., None] * V
    # normals
    w0 = np.minimum(np.arange(W), W - 2)
    h0 = np.minimum(np.arange(H), H - 2)
    dh = V[:, w0 + 1] - V[:, w0]
    dv = V[h0 + 1] - V[h0]
    n = np.cross(dh, dv)
    nrm = np.linalg.norm(n, axis=-1)
    den = np.where(nrm == 0, 1.0, nrm)
    Nb = np.asarray(n_bar, np.float64) * valid[..., None]
    u = n / den[..., None]
    nb = (Nb - np.where(nrm == 0, 0.0, (Nb * u).sum(-1))[..., None] * u) / den[..., None]
    dh_b = np.cross(dv, nb)      # d(dh x dv): dh_bar = dv x n_bar, dv_bar = n_bar x dh
    dv_b = np.cross(nb, dh)
    np.add.at(Vb, (slice(None), w0 + 1), dh_b)
    np.add.at(Vb, (slice(None), w0), -dh_b)
    np.add.at(Vb, (h0 + 1,), dv_b)
    np.add.at(Vb, (h0,), -dv_b)
    depth_bar = (Vb * ray).sum(-1) * valid
    if not want_K:
        return depth_bar
    dm = depth * valid
    kb00, kb02 = (Vb[..., 0] * w * dm).sum(), (Vb[..., 0] * dm).sum()
    kb11, kb12 = (Vb[..., 1] * h * dm).sum(), (Vb[..., 1] * dm).sum()
    K_bar = np.zeros((4, 4))
    K_bar[0, 0], K_bar[0, 2] = (-kb00 + kb02 * cx) / (fx * fx), -kb02 / fx
    K_bar[1, 1], K_bar[1, 2] = (-kb11 + kb12 * cy) / (fy * fy), -kb12 / fy
    return depth_bar, K_bar
