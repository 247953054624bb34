"""Oracle for the backward pass of point_to_plane_gradICP (odometry/icputils.py:479-545): reverse-mode
differentiation of the gradLM loop in numpy float64, given the forward tape.  TEST INFRASTRUCTURE
ONLY.  Pinned against the reference's own autograd gradients (tests/golden/icp_grad.npz, generated
by oracle/make_golden.py); the HIP kernels (gs_icp_backward_f32) are compared against this.

Nearest-neighbour indices and the dist_thresh filter are constants of the differentiation, exactly
as in the reference (icputils.py:201-208 are index / boolean ops)."""
import ctypes as C

import numpy as np

from . import oracle as o


def icp_forward_tape(src, tgt, tgt_normals, init=None, numiters=20, damp=1e-8, dist_thresh=None, lambda_max=2.0,
                     B=1.0, B2=1.0, nu=200.0, mode=1):
    """gradICP (mode 1) / hard-LM ICP (mode 0) forward through the C oracle, returning (T, tape)."""
    src, tgt, tn = (np.ascontiguousarray(a, np.float32) for a in (src, tgt, tgt_normals))
    init = np.eye(4, dtype=np.float32) if init is None else np.ascontiguousarray(init, np.float32)
    ns = src.shape[0]
    prm = o.IcpParams(mode, numiters, damp, -1.0 if dist_thresh is None else dist_thresh, lambda_max, B, B2, nu)
    T = np.empty((4, 4), np.float32)
    idx = np.empty(ns, np.int64)
    trace = np.zeros((numiters, 12), np.float32)
    tape_src = np.zeros((numiters, ns, 3), np.float32)
    tape_idx = np.zeros((numiters, 2, ns), np.int32)
    tape_sys = np.zeros((numiters, 28), np.float32)
    f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    o.lib().gs_or_icp_tape(src.ctypes.data_as(f32p), C.c_int64(ns), tgt.ctypes.data_as(f32p), tn.ctypes.data_as(f32p),
                           C.c_int64(tgt.shape[0]), init.ctypes.data_as(f32p), None, C.byref(prm),
                           T.ctypes.data_as(f32p), idx.ctypes.data_as(C.POINTER(C.c_int64)), trace.ctypes.data_as(f32p),
                           tape_src.ctypes.data_as(f32p), tape_idx.ctypes.data_as(i32p), tape_sys.ctypes.data_as(f32p))
    tape = dict(src=tape_src, idx=tape_idx, sys=tape_sys, trace=trace, init=init,
                prm=dict(numiters=numiters, damp=damp, lambda_max=lambda_max, B=B, B2=B2, nu=nu, mode=mode))
    return T, tape


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], np.float64)


def se3_exp(xi):
    v, w = xi[:3], xi[3:]
    wh = hat(w)
    th = np.sqrt(w @ w)
    if np.float32(th) < np.float32(1e-6):
        R = np.eye(3) + wh
        V = R.copy()
    else:
        s, c = np.sin(th), np.cos(th)
        A, Bc, Cc = s / th, (1 - c) / th ** 2, (th - s) / th ** 3
        wh2 = wh @ wh
        R = np.eye(3) + A * wh + Bc * wh2
        V = np.eye(3) + Bc * wh + Cc * wh2
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def se3_exp_adjoint(xi, Tbar):
    """d <Tbar, Exp(xi)> / d xi  for geometry/se3utils.py:77-115."""
    v, w = xi[:3], xi[3:]
    Rb, tb = Tbar[:3, :3], Tbar[:3, 3]
    wh = hat(w)
    th = np.sqrt(w @ w)
    small = np.float32(th) < np.float32(1e-6)
    if small:
        V = np.eye(3) + wh
    else:
        s, c = np.sin(th), np.cos(th)
        A, Bc, Cc = s / th, (1 - c) / th ** 2, (th - s) / th ** 3
        wh2 = wh @ wh
        V = np.eye(3) + Bc * wh + Cc * wh2
    vb = V.T @ tb
    Vb = np.outer(tb, v)
    if small:
        whb = Rb + Vb
        thb = 0.0
    else:
        Ab = np.sum(Rb * wh)
        Bb = np.sum(Rb * wh2) + np.sum(Vb * wh)
        Cb = np.sum(Vb * wh2)
        W2b = Bc * Rb + Cc * Vb
        whb = A * Rb + Bc * Vb + W2b @ wh.T + wh.T @ W2b
        dA = (c * th - s) / th ** 2
        dB = (s * th - 2 * (1 - c)) / th ** 3
        dC = ((1 - c) * th - 3 * (th - s)) / th ** 4
        thb = Ab * dA + Bb * dB + Cb * dC
    wb = np.array([whb[2, 1] - whb[1, 2], whb[0, 2] - whb[2, 0], whb[1, 0] - whb[0, 1]])
    if not small:
        wb = wb + thb * w / th
    return np.concatenate([vb, wb])


def rows(s, d, n):
    A = np.concatenate([n, np.cross(s, n)], 1)
    b = np.sum(n * (d - s), 1)
    return A, b


def icp_backward(tape, tgt, tgt_normals, T_bar, src_in):
    """Returns (src_bar (Ns,3), tgt_bar (Nt,3), normals_bar (Nt,3), init_bar (4,4)) in float64."""
    tgt = np.asarray(tgt, np.float64)
    tn = np.asarray(tgt_normals, np.float64)
    p = tape["prm"]
    K = p["numiters"]
    lmax = np.float64(np.float32(p["lambda_max"]))
    lmin = np.float64(np.float32(1.0 / p["lambda_max"]))
    lrange = np.float64(np.float32(p["lambda_max"] - 1.0 / p["lambda_max"]))
    Bp, B2p, nu = float(p["B"]), float(p["B2"]), float(p["nu"])
    ns = tape["src"].shape[1]
    # replay the transforms
    Ts_list, Tr_list, Tk = [], [], [np.asarray(tape["init"], np.float64)]
    for k in range(K):
        xi = tape["trace"][k, 4:10].astype(np.float64)
        sig = float(tape["trace"][k, 3])
        if p.get("mode", 1) == 0:   # hard LM: the step is exp(xi) when accepted, the identity otherwise
            sig = 1.0 if np.float32(tape["trace"][k, 1]) < np.float32(tape["trace"][k, 0]) else 0.0
        Tr_list.append(se3_exp(xi))
        Ts_list.append(se3_exp(sig * xi))
        Tk.append(Ts_list[-1] @ Tk[-1])
    Tb = np.asarray(T_bar, np.float64).copy()   # adjoint of T_{k+1}
    sb_next = np.zeros((ns, 3))                 # adjoint of src_{k+1}
    lam_bar = 0.0                               # adjoint of damp_{k+1}
    tgt_bar, tn_bar = np.zeros_like(tgt), np.zeros_like(tn)
    for k in range(K - 1, -1, -1):
        s = tape["src"][k].astype(np.float64)
        idx, idx1 = tape["idx"][k, 0], tape["idx"][k, 1]
        keep, keep1 = idx >= 0, idx1 >= 0
        j, j1 = np.where(keep, idx, 0), np.where(keep1, idx1, 0)
        xi = tape["trace"][k, 4:10].astype(np.float64)
        err, new_err, sig = (float(tape["trace"][k, c]) for c in (0, 1, 3))
        lam = float(tape["sys"][k, 27])
        Ts, Tr = Ts_list[k], Tr_list[k]
        # T_{k+1} = Ts T_k ; src_{k+1} = Ts src_k
        Ts_bar = Tb @ Tk[k].T
        Ts_bar[:3, :3] += sb_next.T @ s
        Ts_bar[:3, 3] += sb_next.sum(0)
        Tb = Ts.T @ Tb
        sb = sb_next @ Ts[:3, :3]
        if p.get("mode", 1) == 0:
            sig = 1.0 if np.float32(new_err) < np.float32(err) else 0.0
        # Ts = Exp(sig * xi)
        ub = se3_exp_adjoint(sig * xi, Ts_bar)
        sig_bar = ub @ xi
        xi_bar = sig * ub
        # sigma, damping
        diff = np.float32(new_err) - np.float32(err)
        inside = -70.0 <= diff <= 70.0
        d = float(np.clip(diff, -70.0, 70.0))
        E, E2 = np.exp(-Bp * d), np.exp(-B2p * d)
        q = lmin + lrange / (1 + E)
        dq = lrange * Bp * E / (1 + E) ** 2
        dsig = (B2p * E2 / nu) * (1 + E2) ** (-1.0 / nu - 1.0)
        d_bar = sig_bar * dsig + lam_bar * lam * dq
        lam_bar = lam_bar * q
        if not inside or p.get("mode", 1) == 0:   # hard LM: accept test and damping schedule carry no gradient
            d_bar = 0.0
        e1_bar, e_bar = d_bar, -d_bar
        # look-ahead residual: s' = Tr s, b' = n'.(d' - s')
        s1 = s @ Tr[:3, :3].T + Tr[:3, 3]
        n1, d1 = tn[j1], tgt[j1]
        b1 = np.sum(n1 * (d1 - s1), 1) * keep1
        b1_bar = 2 * b1 * e1_bar
        s1_bar = -n1 * b1_bar[:, None]
        np.add.at(tgt_bar, j1, n1 * b1_bar[:, None])
        np.add.at(tn_bar, j1, (d1 - s1) * b1_bar[:, None])
        sb += s1_bar @ Tr[:3, :3]
        Tr_bar = np.zeros((4, 4))
        Tr_bar[:3, :3] = s1_bar.T @ s
        Tr_bar[:3, 3] = s1_bar.sum(0)
        xi_bar = xi_bar + se3_exp_adjoint(xi, Tr_bar)
        # xi = H^-1 g
        H = np.zeros((6, 6))
        q_ = 0
        for r in range(6):
            for c in range(r, 6):
                H[r, c] = H[c, r] = float(tape["sys"][k, q_])
                q_ += 1
        H = H + np.float64(np.float32(lam)) * np.eye(6)
        g_bar = np.linalg.solve(H.T, xi_bar)
        H_bar = -np.outer(g_bar, xi)
        lam_bar += np.trace(H_bar)
        Hs = H_bar + H_bar.T
        # rows
        n0, d0 = tn[j], tgt[j]
        A, b = rows(s, d0, n0)
        A, b = A * keep[:, None], b * keep
        a_bar = (A @ Hs.T + np.outer(b, g_bar)) * keep[:, None]
        b_bar = (A @ g_bar + 2 * b * e_bar) * keep
        an, ac = a_bar[:, :3], a_bar[:, 3:]
        sb += -n0 * b_bar[:, None] + np.cross(n0, ac)
        np.add.at(tn_bar, j, an + np.cross(ac, s) + (d0 - s) * b_bar[:, None])
        np.add.at(tgt_bar, j, n0 * b_bar[:, None])
        sb_next = sb
    init = np.asarray(tape["init"], np.float64)
    src_in = np.asarray(src_in, np.float64)
    init_bar = Tb.copy()   # T_0 = init
    init_bar[:3, :3] += sb_next.T @ src_in
    init_bar[:3, 3] += sb_next.sum(0)
    init_bar[3, :] = 0.0   # the bottom row is a constant of the parametrisation
    src_bar = sb_next @ init[:3, :3]
    return src_bar, tgt_bar, tn_bar, init_bar
