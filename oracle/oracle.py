"""numpy front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package gradslam_amd never imports this module.
Every function works on ONE sequence, mirrors one entry point of include/gradslam_hip.h and
takes/returns numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


class IcpParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("numiters", C.c_int), ("damp", C.c_float),
                ("dist_thresh", C.c_float), ("lambda_max", C.c_float), ("B", C.c_float),
                ("B2", C.c_float), ("nu", C.c_float)]


def build():
    subprocess.check_call(["make", "-C", _HERE, "--quiet"])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgs_oracle.so")
        src = os.path.join(_HERE, "gs_oracle.c")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
            build()
        _LIB = C.CDLL(so)
        for name in ("gs_or_downsample_frame", "gs_or_active_table", "gs_or_downsample_table",
                     "gs_or_select_targets", "gs_or_best_unique_rows", "gs_or_best_table",
                     "gs_or_fuse_append", "gs_or_append_valid"):
            getattr(_LIB, name).restype = C.c_int64
    return _LIB


def _f(a):
    return None if a is None else a.ctypes.data_as(f32p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def two_sigma_sq(sigma):
    return np.float32(2 * (float(sigma) ** 2))


# ------------------------------------------------------------------------------- K1
def frame_maps(depth, K, sigma=0.6):
    """depth (H,W) f32, K (4,4) -> vertex (H,W,3), normal (H,W,3), alpha (H,W), valid (H,W) bool"""
    depth = _c(depth, np.float32)
    K = _c(K, np.float32)
    H, W = depth.shape
    v = np.empty((H, W, 3), np.float32)
    n = np.empty((H, W, 3), np.float32)
    a = np.empty((H, W), np.float32)
    m = np.empty((H, W), np.uint8)
    lib().gs_or_frame_maps(_f(depth), _f(K), H, W, C.c_float(two_sigma_sq(sigma)), _f(v), _f(n), _f(a),
                           m.ctypes.data_as(u8p))
    return v, n, a, m.astype(bool)


def global_maps(vertex, normal, depth, pose):
    vertex = _c(vertex, np.float32)
    normal = _c(normal, np.float32)
    depth = _c(depth, np.float32)
    H, W = depth.shape
    gv = np.empty_like(vertex)
    gn = np.empty_like(normal)
    pose = None if pose is None else _c(pose, np.float32)
    lib().gs_or_global_maps(_f(vertex), _f(normal), _f(depth), _f(pose), H, W, _f(gv), _f(gn))
    return gv, gn


def alpha(points, sigma, eps=1e-7):
    points = _c(points, np.float32)
    out = np.empty(points.shape[:-1], np.float32)
    lib().gs_or_alpha(_f(points), C.c_int64(out.size), C.c_float(two_sigma_sq(sigma)), C.c_float(eps), _f(out))
    return out


# ------------------------------------------------------------------------------- K2
def downsample_frame(gvertex, gnormal, rgb, depth, ds):
    depth = _c(depth, np.float32)
    H, W = depth.shape
    cap = ((H + ds - 1) // ds) * ((W + ds - 1) // ds)
    pts = np.empty((cap, 3), np.float32)
    nrm = np.empty((cap, 3), np.float32)
    col = np.empty((cap, 3), np.float32)
    c = lib().gs_or_downsample_frame(_f(_c(gvertex, np.float32)), _f(_c(gnormal, np.float32)),
                                     _f(_c(rgb, np.float32)), _f(depth), H, W, ds, _f(pts), _f(nrm), _f(col))
    return pts[:c].copy(), nrm[:c].copy(), col[:c].copy()


def project_map(points, pose, K, H, W):
    points = _c(points, np.float32)
    pix = np.empty(points.shape[0], np.int32)
    lib().gs_or_project_map(_f(points), C.c_int64(points.shape[0]), _f(_c(pose, np.float32)),
                            _f(_c(K, np.float32)), H, W, pix.ctypes.data_as(i32p))
    return pix


def active_table(pix, W, b=0):
    pix = _c(pix, np.int32)
    rows = np.empty((pix.shape[0], 4), np.int64)
    c = lib().gs_or_active_table(pix.ctypes.data_as(i32p), C.c_int64(pix.shape[0]), W, C.c_int64(b),
                                 rows.ctypes.data_as(i64p))
    return rows[:c].copy()


def select_targets(pix, W, ds, points, normals, colors=None):
    pix = _c(pix, np.int32)
    N = pix.shape[0]
    points = _c(points, np.float32)
    normals = _c(normals, np.float32)
    colors = None if colors is None else _c(colors, np.float32)
    op = np.empty((N, 3), np.float32)
    on = np.empty((N, 3), np.float32)
    oc = np.empty((N, 3), np.float32)
    c = lib().gs_or_select_targets(pix.ctypes.data_as(i32p), C.c_int64(N), W, ds, _f(points), _f(normals),
                                   _f(colors), _f(op), _f(on), _f(oc) if colors is not None else None)
    return op[:c].copy(), on[:c].copy(), (oc[:c].copy() if colors is not None else None)


def downsample_table(rows, ds, points, normals, colors=None):
    rows = _c(rows, np.int64)
    R = rows.shape[0]
    points = _c(points, np.float32)
    normals = None if normals is None else _c(normals, np.float32)
    colors = None if colors is None else _c(colors, np.float32)
    op = np.empty((max(R, 1), 3), np.float32)
    on = np.empty((max(R, 1), 3), np.float32)
    oc = np.empty((max(R, 1), 3), np.float32)
    c = lib().gs_or_downsample_table(rows.ctypes.data_as(i64p), C.c_int64(R), ds, _f(points), _f(normals),
                                     _f(colors), _f(op), _f(on) if normals is not None else None,
                                     _f(oc) if colors is not None else None)
    return op[:c].copy(), (on[:c].copy() if normals is not None else None), (oc[:c].copy() if colors is not None else None)


# ------------------------------------------------------------------------------- K3/K4
def knn1(src, tgt):
    src = _c(src, np.float32)
    tgt = _c(tgt, np.float32)
    ns = src.shape[0]
    idx = np.empty(ns, np.int64)
    d2 = np.empty(ns, np.float32)
    lib().gs_or_knn1(_f(src), C.c_int64(ns), _f(tgt), C.c_int64(tgt.shape[0]), idx.ctypes.data_as(i64p), _f(d2))
    return idx, d2


def gauss_newton_rows(src, tgt, tgt_normals, dist_thresh=None):
    src = _c(src, np.float32)
    tgt = _c(tgt, np.float32)
    tn = _c(tgt_normals, np.float32)
    ns = src.shape[0]
    A = np.empty((ns, 6), np.float32)
    b = np.empty(ns, np.float32)
    idx = np.empty(ns, np.int64)
    keep = np.empty(ns, np.uint8)
    lib().gs_or_gauss_newton_rows(_f(src), C.c_int64(ns), _f(tgt), _f(tn), C.c_int64(tgt.shape[0]),
                                  C.c_float(-1.0 if dist_thresh is None else dist_thresh), _f(A), _f(b),
                                  idx.ctypes.data_as(i64p), keep.ctypes.data_as(u8p))
    return A, b, idx, keep.astype(bool)


def solve_normal_eq(A, b, damp=1e-8, keep=None):
    A = _c(A, np.float32)
    b = _c(b, np.float32).reshape(-1)
    x = np.empty(A.shape[1], np.float32)
    k = None if keep is None else _c(keep, np.uint8)
    lib().gs_or_solve_normal_eq(_f(A), _f(b), None if k is None else k.ctypes.data_as(u8p),
                                C.c_int64(A.shape[0]), A.shape[1], C.c_float(damp), _f(x))
    return x


def ingest_depth(raw_u16, H, W, scale_div):
    """datasets/tum.py:463-477 (_preprocess_depth): cv2.INTER_NEAREST on the float64 image, / scaling_factor,
    float32 last.  cv2 resizeNN: source index = min(floor(dst * (1 / (dst_size / src_size))), src_size - 1)."""
    raw = np.asarray(raw_u16)
    H0, W0 = raw.shape
    ys = np.arange(H) if H == H0 else np.minimum(np.floor(np.arange(H) * (1.0 / (H / H0))).astype(np.int64), H0 - 1)
    xs = np.arange(W) if W == W0 else np.minimum(np.floor(np.arange(W) * (1.0 / (W / W0))).astype(np.int64), W0 - 1)
    return (raw[np.ix_(ys, xs)].astype(np.float64) / float(scale_div)).astype(np.float32)


def _lin_coeffs(n_dst, n_src):
    scale = 1.0 / (n_dst / n_src)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    f = np.where(s < 0, np.float32(0), f)
    s = np.maximum(s, 0)
    f = np.where(s >= n_src - 1, np.float32(0), f).astype(np.float32)
    s = np.minimum(s, n_src - 1)
    return s, np.minimum(s + 1, n_src - 1), (np.float32(1) - f).astype(np.float32), f


def ingest_color(raw_u8, H, W, normalize=False):
    """datasets/tum.py:448-461 (_preprocess_color): cv2.INTER_LINEAR on the float64 image (OpenCV's
    resizeGeneric_ with float32 weights and float64 sums, x first, then y), optional / 255, float32 last."""
    raw = np.asarray(raw_u8).astype(np.float64)
    H0, W0 = raw.shape[:2]
    if (H, W) != (H0, W0):
        x0, x1, a0, a1 = _lin_coeffs(W, W0)
        y0, y1, b0, b1 = _lin_coeffs(H, H0)
        rows = raw[:, x0] * a0.astype(np.float64)[None, :, None] + raw[:, x1] * a1.astype(np.float64)[None, :, None]
        raw = rows[y0] * b0.astype(np.float64)[:, None, None] + rows[y1] * b1.astype(np.float64)[:, None, None]
    if normalize:
        raw = raw / 255
    return raw.astype(np.float32)


def relative_pose(T01, T02):
    T01, T02 = _c(T01, np.float32).reshape(-1, 4, 4), _c(T02, np.float32).reshape(-1, 4, 4)
    out = np.empty_like(T02)
    lib().gs_or_relative_pose(_f(T01), _f(T02), C.c_int64(T02.shape[0]), _f(out))
    return out


def se3_exp(xi):
    xi = _c(xi, np.float32).reshape(6)
    T = np.empty((4, 4), np.float32)
    lib().gs_or_se3_exp(_f(xi), _f(T))
    return T


def transform_points(pts, T):
    pts = _c(pts, np.float32)
    out = np.empty_like(pts)
    lib().gs_or_transform_points(_f(pts), C.c_int64(pts.shape[0]), _f(_c(T, np.float32)), _f(out))
    return out


def icp(src, tgt, tgt_normals, init=None, compose=None, mode=1, numiters=20, damp=1e-8, dist_thresh=None,
        lambda_max=2.0, B=1.0, B2=1.0, nu=200.0, return_trace=False):
    src = _c(src, np.float32)
    tgt = _c(tgt, np.float32)
    tn = _c(tgt_normals, np.float32)
    init = np.eye(4, dtype=np.float32) if init is None else _c(init, np.float32)
    comp = None if compose is None else _c(compose, np.float32)
    prm = IcpParams(mode, numiters, damp, -1.0 if dist_thresh is None else dist_thresh, lambda_max, B, B2, nu)
    T = np.empty((4, 4), np.float32)
    idx = np.empty(src.shape[0], np.int64)
    trace = np.zeros((numiters, 12), np.float32)
    lib().gs_or_icp(_f(src), C.c_int64(src.shape[0]), _f(tgt), _f(tn), C.c_int64(tgt.shape[0]), _f(init),
                    _f(comp), C.byref(prm), _f(T), idx.ctypes.data_as(i64p), _f(trace))
    return (T, idx, trace) if return_trace else (T, idx)


# ------------------------------------------------------------------------------- K5
def similar_rows(rows, points, normals, gvertex, gnormal, dist_th, dot_th):
    rows = _c(rows, np.int64)
    W = gvertex.shape[1]
    mask = np.empty(rows.shape[0], np.uint8)
    lib().gs_or_similar_rows(rows.ctypes.data_as(i64p), C.c_int64(rows.shape[0]), _f(_c(points, np.float32)),
                             _f(_c(normals, np.float32)), _f(_c(gvertex, np.float32)),
                             _f(_c(gnormal, np.float32)), W, C.c_float(dist_th), C.c_float(dot_th),
                             mask.ctypes.data_as(u8p))
    return mask.astype(bool)


def best_unique_rows(rows, points, ccounts, gvertex):
    rows = _c(rows, np.int64)
    H, W = gvertex.shape[:2]
    out = np.empty((max(rows.shape[0], 1), 4), np.int64)
    c = lib().gs_or_best_unique_rows(rows.ctypes.data_as(i64p), C.c_int64(rows.shape[0]),
                                     _f(_c(points, np.float32)), _f(_c(ccounts, np.float32)),
                                     _f(_c(gvertex, np.float32)), H, W, out.ctypes.data_as(i64p))
    return out[:c].copy()


def associate(pix, points, normals, ccounts, gvertex, gnormal, dist_th, dot_th):
    pix = _c(pix, np.int32)
    H, W = gvertex.shape[:2]
    best = np.empty(H * W, np.int32)
    sim = np.empty(pix.shape[0], np.uint8)
    lib().gs_or_associate(pix.ctypes.data_as(i32p), C.c_int64(pix.shape[0]), _f(_c(points, np.float32)),
                          _f(_c(normals, np.float32)), _f(_c(ccounts, np.float32)),
                          _f(_c(gvertex, np.float32)), _f(_c(gnormal, np.float32)), H, W, C.c_float(dist_th),
                          C.c_float(dot_th), best.ctypes.data_as(i32p), sim.ctypes.data_as(u8p))
    return best, sim.astype(bool)


def best_table(best_pix, H, W, b=0):
    best_pix = _c(best_pix, np.int32)
    out = np.empty((H * W, 4), np.int64)
    c = lib().gs_or_best_table(best_pix.ctypes.data_as(i32p), H, W, C.c_int64(b), out.ctypes.data_as(i64p))
    return out[:c].copy()


def rows_to_best_pix(rows, H, W):
    rows = _c(rows, np.int64)
    best = np.empty(H * W, np.int32)
    lib().gs_or_rows_to_best_pix(rows.ctypes.data_as(i64p), C.c_int64(rows.shape[0]), H, W,
                                 best.ctypes.data_as(i32p))
    return best


# ------------------------------------------------------------------------------- K6
def fuse_append(points, normals, colors, ccounts, best_pix, gvertex, gnormal, rgb, alpha_img, depth,
                renorm_all=True):
    """Returns new (points, normals, colors, ccounts) arrays (inputs are not modified)."""
    H, W = depth.shape
    N = points.shape[0]
    cap = N + H * W
    P = np.zeros((cap, 3), np.float32); P[:N] = points
    Nn = np.zeros((cap, 3), np.float32); Nn[:N] = normals
    Cc = np.zeros((cap, 3), np.float32); Cc[:N] = colors
    F = np.zeros((cap, 1), np.float32); F[:N] = np.asarray(ccounts, np.float32).reshape(N, 1)
    c = lib().gs_or_fuse_append(_f(P), _f(Nn), _f(Cc), _f(F), C.c_int64(N), C.c_int64(cap),
                                _c(best_pix, np.int32).ctypes.data_as(i32p), _f(_c(gvertex, np.float32)),
                                _f(_c(gnormal, np.float32)), _f(_c(rgb, np.float32)),
                                _f(_c(alpha_img, np.float32)), _f(_c(depth, np.float32)), H, W,
                                1 if renorm_all else 0)
    assert c >= 0
    return P[:c].copy(), Nn[:c].copy(), Cc[:c].copy(), F[:c].copy()


def append_valid(points, normals, colors, ccounts, gvertex, gnormal, rgb, alpha_img, depth):
    H, W = depth.shape
    N = 0 if points is None else points.shape[0]
    cap = N + H * W
    P = np.zeros((cap, 3), np.float32)
    Nn = np.zeros((cap, 3), np.float32)
    Cc = np.zeros((cap, 3), np.float32)
    F = np.zeros((cap, 1), np.float32)
    if N:
        P[:N] = points; Nn[:N] = normals; Cc[:N] = colors
        if ccounts is not None:
            F[:N] = np.asarray(ccounts, np.float32).reshape(N, 1)
    c = lib().gs_or_append_valid(_f(P), _f(Nn), _f(Cc), _f(F), C.c_int64(N), C.c_int64(cap),
                                 _f(_c(gvertex, np.float32)), _f(_c(gnormal, np.float32)),
                                 _f(_c(rgb, np.float32)), None if alpha_img is None else _f(_c(alpha_img, np.float32)),
                                 _f(_c(depth, np.float32)), H, W)
    return P[:c].copy(), Nn[:c].copy(), Cc[:c].copy(), F[:c].copy()
