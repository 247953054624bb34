"""Oracle for the backward pass of fuse_with_map (slam/fusionutils.py:653-720): float64 numpy restatement of the
merge + append and of its reverse mode.  Test infrastructure only.  The whole differentiable mapping chain
(depth -> maps -> fuse) is pinned on the GPU against the reference's own autograd (tests/golden/fusion_grad.npz);
this file pins the fuse adjoint itself by finite differences of the float64 forward below.

Notation: old map rows x (points / normals / colours, (n, 3) each) and confidence counts cc (n,), frame maps f
gathered at the matched pixel `pix_of[n]` (-1 = unmatched: alpha = 0, f = 0), alpha (H*W,)."""
import numpy as np


def fuse_forward(old, cc, frame, alpha, pix_of, new_pix, merged=True):
    """old / frame: lists of three (., 3) float64 arrays; returns (fused attributes list, fused cc)."""
    n = cc.shape[0]
    m = pix_of >= 0
    a = np.where(m, alpha[np.where(m, pix_of, 0)], 0.0)
    cc2 = cc + a if merged else cc.copy()
    inv = 1.0 / np.where(cc2 == 0, 1.0, cc2)
    out = []
    for x, f in zip(old, frame):
        fx = np.where(m[:, None], f[np.where(m, pix_of, 0)], 0.0)
        out.append(np.concatenate([(cc[:, None] * x + a[:, None] * fx) * inv[:, None] if merged else x, f[new_pix]]))
    return out, np.concatenate([cc2, alpha[new_pix]])


def fuse_backward(old, cc, frame, alpha, pix_of, new_pix, bars, cc_bar_new, merged=True):
    """bars: adjoints of the three fused attributes ((n + k, 3) each), cc_bar_new (n + k,).
    Returns (old attribute adjoints list, cc adjoint, frame adjoints list ((P, 3) each), alpha adjoint (P,))."""
    n, P = cc.shape[0], alpha.shape[0]
    m = pix_of >= 0
    pm = np.where(m, pix_of, 0)
    f_bars = [np.zeros((P, 3)) for _ in range(3)]
    alpha_bar = np.zeros(P)
    for fb, b in zip(f_bars, bars):          # appended rows are copies of their pixel
        fb[new_pix] = b[n:]
    alpha_bar[new_pix] = cc_bar_new[n:]
    if not merged:
        return [b[:n].copy() for b in bars], cc_bar_new[:n].copy(), f_bars, alpha_bar
    a = np.where(m, alpha[pm], 0.0)
    cc2 = cc + a
    inv = 1.0 / np.where(cc2 == 0, 1.0, cc2)
    inv_bar, cc_bar, a_bar, old_bars = np.zeros(n), np.zeros(n), np.zeros(n), []
    for x, f, b, fb in zip(old, frame, bars, f_bars):
        fx = np.where(m[:, None], f[pm], 0.0)
        xb = b[:n]
        ub = xb * inv[:, None]
        inv_bar += (xb * (cc[:, None] * x + a[:, None] * fx)).sum(1)
        old_bars.append(cc[:, None] * ub)
        fb[pm[m]] = (a[:, None] * ub)[m]
        cc_bar += (x * ub).sum(1)
        a_bar += (fx * ub).sum(1)
    cc2_bar = cc_bar_new[:n] + np.where(cc2 != 0, -inv * inv * inv_bar, 0.0)
    alpha_bar[pm[m]] = (a_bar + cc2_bar)[m]
    return old_bars, cc_bar + cc2_bar, f_bars, alpha_bar
