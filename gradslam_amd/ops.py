"""Tensor-level wrappers of the C-ABI (one sequence per call).  Each function validates
devices, allocates outputs with torch, and enqueues the HIP kernels on torch's current stream.
Nothing here computes with torch: arithmetic happens in libgradslam_hip.so or not at all."""
import os

import torch

from . import _C
from ._C import Workspace, check, lib, ptr, require_device, stream

f32 = torch.float32

# True: the in-place SLAM drivers keep the surfel count of the map on the device between frames
# (the *_dc entry points) so that a frame never waits for a host read-back.  False: every count
# is read back as soon as it is produced (exact sizes on the host at all times).
DEVICE_COUNTS = os.environ.get("GRADSLAM_HIP_DEVICE_COUNTS", "1") != "0"


def _thresh(dist_thresh):
    """C-ABI encoding of dist_thresh: < 0 means None (no filter).  A negative USER threshold keeps no pair in the
    reference (squared distances are never below it, odometry/icputils.py:203-208); 0 has the same effect."""
    if dist_thresh is None:
        return -1.0
    return max(float(dist_thresh), 0.0)


def two_sigma_sq(sigma):
    if torch.is_tensor(sigma):
        sigma = float(sigma)
    return float(torch.tensor(2 * (float(sigma) ** 2), dtype=torch.float64).to(torch.float32))


def _c(t, dtype=f32):
    if t is None:
        return None
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


_IDENTITY4 = {}


def _identity4(dev):
    """read-only 4x4 identity of a device (the default initial transform of every ICP solve)"""
    t = _IDENTITY4.get(dev)
    if t is None:
        t = _IDENTITY4[dev] = torch.eye(4, dtype=f32, device=dev)
    return t


def _empty_rows(n, tail, dtype, dev):
    """torch.empty((n,) + tail) carved from an allocation whose row count is rounded up to a power of two: the
    per-frame temporaries sized by the (growing) surfel bound then change their allocation size only when the
    bound doubles, instead of asking the caching allocator for a slightly larger block every frame (each new
    size is a fresh hipMalloc while the host runs ahead of the device)."""
    cap = 1 << max(int(n) - 1, 0).bit_length() if n > 1024 else 1024
    return torch.empty((cap,) + tuple(tail), dtype=dtype, device=dev)[:n]


def _warn_detached(name, *tensors):
    """API functions without a backward kernel return detached tensors; say so instead of silently cutting the
    autograd graph (the reference's torch ops would have recorded it)."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        import warnings
        warnings.warn("gradslam_amd.%s has no backward kernel: the result is detached from the autograd graph"
                      % name, RuntimeWarning, stacklevel=3)


def _count(t):
    """Reads a device int64 counter back (one host sync)."""
    return int(t.item())


# ----------------------------------------------------------------------------------- K1
def frame_maps(depth, K, sigma=0.6, want_normal=True, want_alpha=True, want_valid=True, out=None):
    """depth (H,W) f32, K (4,4) f32 -> vertex (H,W,3), normal (H,W,3), alpha (H,W), valid (H,W) bool.
    out: optional (vertex, normal, alpha) contiguous float32 buffers to write into (None entries are
    allocated / skipped according to the want_* flags)."""
    depth, K = _c(depth), _c(K)
    dev = require_device(depth, K)
    H, W = depth.shape
    ov, on, oa = out if out is not None else (None, None, None)
    vertex = ov if ov is not None else torch.empty((H, W, 3), dtype=f32, device=dev)
    normal = on if on is not None else (torch.empty((H, W, 3), dtype=f32, device=dev) if want_normal else None)
    alpha = oa if oa is not None else (torch.empty((H, W), dtype=f32, device=dev) if want_alpha else None)
    valid = torch.empty((H, W), dtype=torch.uint8, device=dev) if want_valid else None
    require_device(vertex, normal, alpha)
    check(lib().gs_frame_maps_f32(ptr(depth), ptr(K), H, W, two_sigma_sq(sigma), ptr(vertex), ptr(normal),
                                  ptr(alpha), ptr(valid), stream(dev)), "gs_frame_maps_f32")
    return vertex, normal, alpha, (valid.view(torch.bool) if valid is not None else None)


def global_maps(vertex, normal, depth, pose, out=None):
    """out: optional (gvertex, gnormal) contiguous float32 buffers to write into."""
    vertex, normal, depth, pose = _c(vertex), _c(normal), _c(depth), _c(pose)
    dev = require_device(vertex, normal, depth, pose)
    H, W = depth.shape[:2]
    og, on = out if out is not None else (None, None)
    gv = og if og is not None else torch.empty_like(vertex)
    gn = on if on is not None else (torch.empty_like(normal) if normal is not None else None)
    require_device(gv, gn)
    check(lib().gs_global_maps_f32(ptr(vertex), ptr(normal), ptr(depth), ptr(pose), H, W, ptr(gv), ptr(gn),
                                   stream(dev)), "gs_global_maps_f32")
    return gv, gn


def _alpha_forward(points, sigma, eps):
    points = _c(points)
    dev = require_device(points)
    out = torch.empty(points.shape[:-1], dtype=f32, device=dev)
    check(lib().gs_alpha_f32(ptr(points), out.numel(), two_sigma_sq(sigma), float(eps), ptr(out), stream(dev)),
          "gs_alpha_f32")
    return out


class AlphaFunction(torch.autograd.Function):
    """get_alpha on an (n, 3) point array, differentiable w.r.t. the points and a tensor sigma
    (gs_alpha_backward_f32; the reference's own gradient check is tests/slam/test_fusionutils.py:56-75)."""

    @staticmethod
    def forward(ctx, points, sigma, eps):
        ctx.save_for_backward(points, sigma if torch.is_tensor(sigma) else None)
        ctx.sigma, ctx.eps = float(sigma), float(eps)
        return _alpha_forward(points, ctx.sigma, eps)

    @staticmethod
    def backward(ctx, a_bar):
        points, sigma_t = ctx.saved_tensors
        pts, a_bar = _c(points), _c(a_bar)
        dev = require_device(pts, a_bar)
        n = a_bar.numel()
        p_bar = torch.empty((n, 3), dtype=f32, device=dev)
        want_sigma = sigma_t is not None and ctx.needs_input_grad[1]
        terms = torch.empty(n, dtype=f32, device=dev) if want_sigma else None
        check(lib().gs_alpha_backward_f32(ptr(pts), n, two_sigma_sq(ctx.sigma), ctx.eps, ptr(a_bar), ptr(p_bar), ptr(terms),
                                          stream(dev)), "gs_alpha_backward_f32")
        s_bar = None
        if want_sigma:   # d alpha / d sigma = alpha |p|^2 / sigma^3, summed over the points in float64
            s_bar = (terms.double().sum() / (ctx.sigma ** 3)).to(sigma_t.dtype).reshape(sigma_t.shape).to(sigma_t.device)
        return p_bar.view(points.shape).to(points.dtype), s_bar, None


def alpha_of_points(points, sigma, eps=1e-7):
    """points (n, 3) -> alpha (n,); on the autograd tape when the points or a tensor sigma require grad."""
    if torch.is_grad_enabled() and (points.requires_grad or (torch.is_tensor(sigma) and sigma.requires_grad)):
        return AlphaFunction.apply(points, sigma, eps)
    return _alpha_forward(points, float(sigma), eps)


# ----------------------------------------------------------------------------------- K2
def downsample_frame(gvertex, gnormal, rgb, depth, ds, sync=True):
    """sync=False: no host read-back; returns the bound-sized buffers and the DEVICE count tensor
    (pts, nrm, col, count) for consumers that take device-side counts (icp(n_src_dev=...))."""
    gvertex, gnormal, rgb, depth = _c(gvertex), _c(gnormal), _c(rgb), _c(depth)
    dev = require_device(gvertex, gnormal, rgb, depth)
    H, W = depth.shape[:2]
    cap = ((H + ds - 1) // ds) * ((W + ds - 1) // ds)
    pts = torch.empty((cap, 3), dtype=f32, device=dev)
    nrm = torch.empty((cap, 3), dtype=f32, device=dev) if gnormal is not None else None
    col = torch.empty((cap, 3), dtype=f32, device=dev) if rgb is not None else None
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    check(lib().gs_downsample_frame_f32(ptr(gvertex), ptr(gnormal), ptr(rgb), ptr(depth), H, W, ds, ptr(pts),
                                        ptr(nrm), ptr(col), ptr(cnt), ptr(ws.scratch(0, H * W)), stream(dev)),
          "gs_downsample_frame_f32")
    if not sync:
        return pts, nrm, col, cnt
    c = _count(cnt)
    return pts[:c], (nrm[:c] if nrm is not None else None), (col[:c] if col is not None else None)


def lattice_source(vertex, depth, pose, ds):
    """ICP source of a frame without compaction: (ceil(H/ds)*ceil(W/ds), 3) global vertices of the lattice
    pixels, NaN where the pixel has no depth (ignored by icp's grid path)."""
    vertex, depth, pose = _c(vertex), _c(depth), _c(pose)
    dev = require_device(vertex, depth, pose)
    H, W = depth.shape[:2]
    out = torch.empty((-(-H // ds) * -(-W // ds), 3), dtype=f32, device=dev)
    check(lib().gs_lattice_source_f32(ptr(vertex), ptr(depth), ptr(pose), H, W, int(ds), ptr(out), stream(dev)),
          "gs_lattice_source_f32")
    return out


def project_map(points, pose, K, H, W, n_dev=None):
    """n_dev: device int64[1] holding the actual row count (points.shape[0] is then an upper bound)."""
    points, pose, K = _c(points), _c(pose), _c(K)
    dev = require_device(points, pose, K)
    n = points.shape[0]
    pix = _empty_rows(n, (), torch.int32, dev)
    if n_dev is not None:
        check(lib().gs_project_map_dc_f32(ptr(points), n, ptr(n_dev), ptr(pose), ptr(K), H, W, ptr(pix), stream(dev)),
              "gs_project_map_dc_f32")
        return pix
    check(lib().gs_project_map_f32(ptr(points), n, ptr(pose), ptr(K), H, W, ptr(pix), stream(dev)),
          "gs_project_map_f32")
    return pix


def active_table(pix, W, b=0):
    dev = require_device(pix)
    n = pix.shape[0]
    rows = torch.empty((n, 4), dtype=torch.int64, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    check(lib().gs_active_table_i64(ptr(pix), n, W, b, ptr(rows), ptr(cnt), ptr(ws.scratch(n, 0)), stream(dev)),
          "gs_active_table_i64")
    return rows[: _count(cnt)]


def select_targets(pix, W, ds, points, normals, colors=None, cap=None, sync=True, n_dev=None):
    """sync=False: no host read-back; returns (pts, nrm, col, count) with bound-sized buffers."""
    points, normals, colors = _c(points), _c(normals), _c(colors)
    dev = require_device(pix, points, normals, colors)
    n = pix.shape[0]
    cap = n if cap is None else int(cap)
    op = _empty_rows(cap, (3,), f32, dev)
    on = _empty_rows(cap, (3,), f32, dev) if normals is not None else None
    oc = _empty_rows(cap, (3,), f32, dev) if colors is not None else None
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    if n_dev is not None:
        check(lib().gs_select_targets_dc_f32(ptr(pix), n, ptr(n_dev), W, ds, ptr(points), ptr(normals), ptr(colors),
                                             ptr(op), ptr(on), ptr(oc), cap, ptr(cnt), ptr(ws.scratch(n, 0)),
                                             stream(dev)), "gs_select_targets_dc_f32")
    else:
        check(lib().gs_select_targets_f32(ptr(pix), n, W, ds, ptr(points), ptr(normals), ptr(colors), ptr(op),
                                          ptr(on), ptr(oc), cap, ptr(cnt), ptr(ws.scratch(n, 0)), stream(dev)),
              "gs_select_targets_f32")
    if not sync:
        return op, on, oc, cnt
    c = _count(cnt)
    if c > cap:
        raise _C.HipExtensionError("gs_select_targets_f32: %d targets exceed capacity %d" % (c, cap))
    return op[:c], (on[:c] if on is not None else None), (oc[:c] if oc is not None else None)


def downsample_table(rows, ds, points, normals=None, colors=None):
    rows = _c(rows, torch.int64)
    points, normals, colors = _c(points), _c(normals), _c(colors)
    dev = require_device(rows, points, normals, colors)
    r = rows.shape[0]
    op = torch.empty((r, 3), dtype=f32, device=dev)
    on = torch.empty((r, 3), dtype=f32, device=dev) if normals is not None else None
    oc = torch.empty((r, 3), dtype=f32, device=dev) if colors is not None else None
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    check(lib().gs_downsample_table_f32(ptr(rows), r, ds, ptr(points), ptr(normals), ptr(colors), ptr(op), ptr(on),
                                        ptr(oc), ptr(cnt), ptr(ws.scratch(r, 0)), stream(dev)),
          "gs_downsample_table_f32")
    c = _count(cnt)
    return op[:c], (on[:c] if on is not None else None), (oc[:c] if oc is not None else None)


# ----------------------------------------------------------------------------------- K3 / K4
def knn1(src, tgt):
    src, tgt = _c(src), _c(tgt)
    dev = require_device(src, tgt)
    ns = src.shape[0]
    idx = torch.empty(ns, dtype=torch.int64, device=dev)
    d2 = torch.empty(ns, dtype=f32, device=dev)
    best = torch.empty(ns, dtype=torch.int64, device=dev)
    check(lib().gs_knn1_f32(ptr(src), ns, ptr(tgt), tgt.shape[0], ptr(idx), ptr(d2), ptr(best), stream(dev)),
          "gs_knn1_f32")
    return idx, d2


def knn1_grid(src, tgt, return_unresolved=False):
    """Same as knn1 through the uniform-grid engine used inside the ICP loop (bit-identical)."""
    import ctypes
    src, tgt = _c(src), _c(tgt)
    dev = require_device(src, tgt)
    ns, nt = src.shape[0], tgt.shape[0]
    idx = torch.empty(ns, dtype=torch.int64, device=dev)
    d2 = torch.empty(ns, dtype=f32, device=dev)
    scratch = Workspace.get(dev).bytes("knn_grid", lib().gs_knn1_grid_scratch_bytes(ns, nt))
    unres = ctypes.c_int64(0)
    check(lib().gs_knn1_grid_f32(ptr(src), ns, ptr(tgt), nt, ptr(idx), ptr(d2), ptr(scratch),
                                 ctypes.byref(unres) if return_unresolved else None, stream(dev)), "gs_knn1_grid_f32")
    return (idx, d2, unres.value) if return_unresolved else (idx, d2)


def gauss_newton_rows(src, tgt, tgt_normals, dist_thresh=None):
    src, tgt, tn = _c(src), _c(tgt), _c(tgt_normals)
    dev = require_device(src, tgt, tn)
    ns = src.shape[0]
    A = torch.empty((ns, 6), dtype=f32, device=dev)
    b = torch.empty(ns, dtype=f32, device=dev)
    idx = torch.empty(ns, dtype=torch.int64, device=dev)
    keep = torch.empty(ns, dtype=torch.uint8, device=dev)
    best = torch.empty(ns, dtype=torch.int64, device=dev)
    check(lib().gs_gauss_newton_rows_f32(ptr(src), ns, ptr(tgt), ptr(tn), tgt.shape[0],
                                         _thresh(dist_thresh), ptr(A), ptr(b),
                                         ptr(idx), ptr(keep), ptr(best), stream(dev)), "gs_gauss_newton_rows_f32")
    return A, b, idx, keep.view(torch.bool)


def solve_normal_eq(A, b, damp=1e-8, keep=None):
    A, b = _c(A), _c(b).reshape(-1)
    keep = None if keep is None else _c(keep.view(torch.uint8) if keep.dtype == torch.bool else keep, torch.uint8)
    dev = require_device(A, b, keep)
    x = torch.empty(A.shape[1], dtype=f32, device=dev)
    check(lib().gs_solve_normal_eq_f32(ptr(A), ptr(b), ptr(keep), A.shape[0], A.shape[1], float(damp), ptr(x),
                                       stream(dev)), "gs_solve_normal_eq_f32")
    return x


def _se3_exp_forward(xi):
    xi = _c(xi).reshape(6)
    dev = require_device(xi)
    T = torch.empty((4, 4), dtype=f32, device=dev)
    check(lib().gs_se3_exp_f32(ptr(xi), ptr(T), stream(dev)), "gs_se3_exp_f32")
    return T


class Se3ExpFunction(torch.autograd.Function):
    """se3_exp, differentiable w.r.t. xi (gs_se3_exp_backward_f32: the adjoint the ICP backward uses)."""

    @staticmethod
    def forward(ctx, xi):
        ctx.save_for_backward(xi)
        return _se3_exp_forward(xi)

    @staticmethod
    def backward(ctx, T_bar):
        (xi,) = ctx.saved_tensors
        x, T_bar = _c(xi).reshape(6), _c(T_bar)
        dev = require_device(x, T_bar)
        out = torch.empty(6, dtype=f32, device=dev)
        check(lib().gs_se3_exp_backward_f32(ptr(x), ptr(T_bar), ptr(out), stream(dev)), "gs_se3_exp_backward_f32")
        return out.view(xi.shape).to(xi.dtype)


def se3_exp(xi):
    if torch.is_grad_enabled() and xi.requires_grad:
        return Se3ExpFunction.apply(xi)
    return _se3_exp_forward(xi)


def lie_small(op, vec, n_in, n_out):
    """so3_hat (op 0) / se3_hat (1) / so3_exp (2) of a 3- / 6-vector (gs_lie_small_f32)."""
    _warn_detached(("so3_hat", "se3_hat", "so3_exp")[op], vec)
    v = _c(vec).reshape(n_in)
    dev = require_device(v)
    out = torch.empty((n_out, n_out), dtype=f32, device=dev)
    check(lib().gs_lie_small_f32(int(op), ptr(v), ptr(out), stream(dev)), "gs_lie_small_f32")
    return out


def project_points(cam, proj, pts_per_mat):
    """cam (n, 3|4), proj (m, 4, 4) with n = m * pts_per_mat -> (n, 2) (gs_project_points_f32)."""
    _warn_detached("project_points", cam, proj)
    cam, proj = _c(cam), _c(proj)
    dev = require_device(cam, proj)
    n = cam.shape[0]
    out = torch.empty((n, 2), dtype=f32, device=dev)
    check(lib().gs_project_points_f32(ptr(cam), int(cam.shape[1]), n, ptr(proj), int(pts_per_mat), ptr(out), stream(dev)),
          "gs_project_points_f32")
    return out


def unproject_points(pix, kinv, depths, pts_per_mat):
    """pix (n, 2|3), kinv (m, 3, 3), depths (n,) -> (n, 3) (gs_unproject_points_f32)."""
    _warn_detached("unproject_points", pix, kinv, depths)
    pix, kinv, depths = _c(pix), _c(kinv), _c(depths)
    dev = require_device(pix, kinv, depths)
    n = pix.shape[0]
    out = torch.empty((n, 3), dtype=f32, device=dev)
    check(lib().gs_unproject_points_f32(ptr(pix), int(pix.shape[1]), n, ptr(kinv), int(pts_per_mat), ptr(depths), ptr(out),
                                        stream(dev)), "gs_unproject_points_f32")
    return out


def ingest_depth(raw_u16, H, W, scale_div):
    """(H0, W0) uint16 depth as decoded from the PNG -> (H, W) float32 metres."""
    raw = raw_u16.contiguous()
    assert raw.dtype in (torch.uint16, torch.int16) and raw.ndim == 2
    dev = require_device(raw)
    out = torch.empty((H, W), dtype=f32, device=dev)
    check(lib().gs_ingest_depth_u16_f32(ptr(raw), raw.shape[0], raw.shape[1], ptr(out), H, W, float(scale_div),
                                        stream(dev)), "gs_ingest_depth_u16_f32")
    return out


def ingest_color(raw_u8, H, W, normalize=False):
    """(H0, W0, 3) uint8 colour -> (H, W, 3) float32 in [0, 255] (or [0, 1] when normalize)."""
    raw = raw_u8.contiguous()
    assert raw.dtype == torch.uint8 and raw.ndim == 3 and raw.shape[2] == 3
    dev = require_device(raw)
    out = torch.empty((H, W, 3), dtype=f32, device=dev)
    check(lib().gs_ingest_color_u8_f32(ptr(raw), raw.shape[0], raw.shape[1], ptr(out), H, W, 1 if normalize else 0,
                                       stream(dev)), "gs_ingest_color_u8_f32")
    return out


def host_device_pointer(t):
    """Device-side address of a PINNED host tensor (gs_host_device_pointer), or None when the allocation is not mapped
    into the device's address space."""
    import ctypes
    if t.is_cuda or not t.is_pinned():
        return None
    out = ctypes.c_void_p()
    rc = lib().gs_host_device_pointer(ctypes.c_void_p(t.data_ptr()), ctypes.byref(out))
    return out.value if rc == 0 and out.value else None


def ingest_frames_native(depth_raw, color_raw, depth_out, color_out, scale_div, normalize=False, on_stream=None):
    """n native-size raw frames -> float32 images in ONE launch (gs_ingest_frames_native_f32): depth_raw (..., H, W) uint16
    -> depth_out (same pixels) float32 metres, color_raw (..., H, W, 3) uint8 -> color_out float32; either pair may be
    None.  The raw tensors live on the device of the outputs -- or in PINNED host memory mapped into its address space
    (the kernel then reads them over PCIe: the streaming ingest).  `on_stream`: a torch stream to launch on (default:
    the current one)."""
    ref = depth_raw if depth_raw is not None else color_raw
    dev = require_device(depth_out if depth_raw is not None else color_out)
    H, W = (depth_raw.shape[-2:] if depth_raw is not None else color_raw.shape[-3:-1])
    n = (depth_raw.numel() if depth_raw is not None else color_raw.numel() // 3) // (H * W)
    ptrs = []
    for raw, out, dt, c in ((depth_raw, depth_out, (torch.uint16, torch.int16), 1), (color_raw, color_out, (torch.uint8,), 3)):
        if raw is None:
            ptrs += [None, None]
            continue
        if raw.dtype not in dt or out.dtype != f32 or not raw.is_contiguous() or not out.is_contiguous() or \
                out.numel() != raw.numel() or raw.numel() != n * H * W * c or out.device != dev:
            raise _C.HipExtensionError("ingest_frames_native: raw / out buffers do not match")
        if raw.is_cuda:
            if raw.device != dev:
                raise _C.HipExtensionError("ingest_frames_native: raw frames on another device")
            rp = raw.data_ptr()
        else:
            rp = host_device_pointer(raw)
            if rp is None:
                raise _C.HipExtensionError("ingest_frames_native: host frames must be pinned and device-mapped "
                                           "(tensor.pin_memory()); there is no CPU path")
        ptrs += [_C.C.c_void_p(rp), ptr(out)]
    del ref
    st = stream(dev) if on_stream is None else _C.C.c_void_p(on_stream.cuda_stream)
    check(lib().gs_ingest_frames_native_f32(ptrs[0], ptrs[2], n, int(H), int(W), float(scale_div), 1 if normalize else 0,
                                            ptrs[1], ptrs[3], st), "gs_ingest_frames_native_f32")


def relative_pose(T01, T02):
    """(n, 4, 4) x (n, 4, 4) -> compose(inv(T01), T02) (relative_transformation of the reference)."""
    T01, T02 = _c(T01), _c(T02)
    dev = require_device(T01, T02)
    out = torch.empty_like(T02)
    check(lib().gs_relative_pose_f32(ptr(T01), ptr(T02), T02.numel() // 16, ptr(out), stream(dev)),
          "gs_relative_pose_f32")
    return out


def transform_points(pts, T):
    _warn_detached("transform_pointcloud", pts, T)
    pts, T = _c(pts), _c(T)
    dev = require_device(pts, T)
    out = torch.empty_like(pts)
    check(lib().gs_transform_points_f32(ptr(pts), pts.shape[0], ptr(T), ptr(out), stream(dev)),
          "gs_transform_points_f32")
    return out


def icp(src, tgt, tgt_normals, init=None, compose=None, mode=1, numiters=20, damp=1e-8, dist_thresh=None,
        lambda_max=2.0, B=1.0, B2=1.0, nu=200.0, return_idx=True, return_trace=False, n_src_dev=None,
        n_tgt_dev=None, out=None):
    """Whole (grad)LM point-to-plane ICP on the device; returns T (4,4) [, idx (Ns,)] [, trace].
    n_src_dev / n_tgt_dev: device int64 tensors holding the actual point counts (the row counts of
    src / tgt are then upper bounds): nothing is read back to the host (gs_icp_dc_f32).
    out: optional contiguous (4, 4) float32 tensor that receives T."""
    src, tgt, tn = _c(src), _c(tgt), _c(tgt_normals)
    dev = require_device(src, tgt, tn)
    init = _identity4(dev) if init is None else _c(init)
    compose = _c(compose)
    require_device(init, compose)
    ns, nt = src.shape[0], tgt.shape[0]
    prm = _C.IcpParams(int(mode), int(numiters), float(damp), _thresh(dist_thresh),
                       float(lambda_max), float(B), float(B2), float(nu))
    T = torch.empty((4, 4), dtype=f32, device=dev) if out is None else out
    assert T.shape == (4, 4) and T.dtype == f32 and T.is_contiguous() and T.device == dev
    idx = torch.empty(ns, dtype=torch.int64, device=dev) if return_idx else None
    ws = Workspace.get(dev)
    dc = n_src_dev is not None or n_tgt_dev is not None
    scratch = ws.bytes("icp", lib().gs_icp_scratch_bytes(ns, nt))
    if dc:
        require_device(n_src_dev, n_tgt_dev)
        check(lib().gs_icp_dc_f32(ptr(src), ns, ptr(n_src_dev), ptr(tgt), ptr(tn), nt, ptr(n_tgt_dev), ptr(init),
                                  ptr(compose), prm, ptr(T), ptr(idx), ptr(scratch), stream(dev)), "gs_icp_dc_f32")
    else:
        check(lib().gs_icp_f32(ptr(src), ns, ptr(tgt), ptr(tn), nt, ptr(init), ptr(compose), prm, ptr(T), ptr(idx),
                               ptr(scratch), stream(dev)), "gs_icp_f32")
    out = [T]
    if return_idx:
        out.append(idx)
    if return_trace:
        tr = torch.empty((numiters, 12), dtype=f32, device=dev)
        check(lib().gs_icp_trace_f32(ptr(scratch), numiters, ptr(tr), stream(dev)), "gs_icp_trace_f32")
        out.append(tr)
    return out[0] if len(out) == 1 else tuple(out)


def icp_map(src, map_points, map_normals, pix, W, ds, n_map_dev=None, init=None, compose=None, mode=1, numiters=20,
            damp=1e-8, dist_thresh=None, lambda_max=2.0, B=1.0, B2=1.0, nu=200.0, out=None):
    """(grad)LM ICP of `src` (NaN rows are skipped) against the map rows whose projection `pix` lies on the
    [::ds, ::ds] lattice, binned straight from the map (gs_icp_map_dc_f32): same T as select_targets + icp."""
    src, P, N = _c(src), _c(map_points), _c(map_normals)
    dev = require_device(src, P, N, pix)
    init = _identity4(dev) if init is None else _c(init)
    compose = _c(compose)
    require_device(init, compose, n_map_dev)
    ns, nm = src.shape[0], P.shape[0]
    prm = _C.IcpParams(int(mode), int(numiters), float(damp), _thresh(dist_thresh),
                       float(lambda_max), float(B), float(B2), float(nu))
    T = torch.empty((4, 4), dtype=f32, device=dev) if out is None else out
    assert T.shape == (4, 4) and T.dtype == f32 and T.is_contiguous() and T.device == dev
    scratch = Workspace.get(dev).bytes("icp", lib().gs_icp_scratch_bytes(ns, nm))
    check(lib().gs_icp_map_dc_f32(ptr(src), ns, None, ptr(P), ptr(N), ptr(pix), nm, ptr(n_map_dev), int(W), int(ds),
                                  ptr(init), ptr(compose), prm, ptr(T), ptr(scratch), stream(dev)), "gs_icp_map_dc_f32")
    return T


# ----------------------------------------------------------------------------------- K5
def similar_rows(rows, points, normals, gvertex, gnormal, dist_th, dot_th):
    rows = _c(rows, torch.int64)
    points, normals, gvertex, gnormal = _c(points), _c(normals), _c(gvertex), _c(gnormal)
    dev = require_device(rows, points, normals, gvertex, gnormal)
    W = gvertex.shape[1]
    mask = torch.empty(rows.shape[0], dtype=torch.uint8, device=dev)
    check(lib().gs_similar_rows_f32(ptr(rows), rows.shape[0], ptr(points), ptr(normals), ptr(gvertex), ptr(gnormal),
                                    W, float(dist_th), float(dot_th), ptr(mask), stream(dev)), "gs_similar_rows_f32")
    return mask.view(torch.bool)


def best_unique_rows(rows, points, ccounts, gvertex, b=0):
    rows = _c(rows, torch.int64)
    points, ccounts, gvertex = _c(points), _c(ccounts), _c(gvertex)
    dev = require_device(rows, points, ccounts, gvertex)
    H, W = gvertex.shape[:2]
    r = rows.shape[0]
    best = torch.empty(H * W, dtype=torch.int32, device=dev)
    out = torch.empty((H * W, 4), dtype=torch.int64, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    check(lib().gs_best_unique_rows_f32(ptr(rows), r, ptr(points), ptr(ccounts), ptr(gvertex), H, W, b, ptr(best),
                                        ptr(out), ptr(cnt), ptr(ws.scratch(r, H * W)), stream(dev)),
          "gs_best_unique_rows_f32")
    return out[: _count(cnt)], best


def associate(pix, points, normals, ccounts, gvertex, gnormal, dist_th, dot_th, want_similar=False, n_dev=None):
    points, normals, ccounts, gvertex, gnormal = _c(points), _c(normals), _c(ccounts), _c(gvertex), _c(gnormal)
    dev = require_device(pix, points, normals, ccounts, gvertex, gnormal)
    H, W = gvertex.shape[:2]
    n = pix.shape[0]
    best = torch.empty(H * W, dtype=torch.int32, device=dev)
    sim = torch.empty(n, dtype=torch.uint8, device=dev) if want_similar else None
    ws = Workspace.get(dev)
    if n_dev is not None:
        check(lib().gs_associate_dc_f32(ptr(pix), n, ptr(n_dev), ptr(points), ptr(normals), ptr(ccounts),
                                        ptr(gvertex), ptr(gnormal), H, W, float(dist_th), float(dot_th), ptr(best),
                                        ptr(sim), ptr(ws.scratch(n, H * W)), stream(dev)), "gs_associate_dc_f32")
    else:
        check(lib().gs_associate_f32(ptr(pix), n, ptr(points), ptr(normals), ptr(ccounts), ptr(gvertex),
                                     ptr(gnormal), H, W, float(dist_th), float(dot_th), ptr(best), ptr(sim),
                                     ptr(ws.scratch(n, H * W)), stream(dev)), "gs_associate_f32")
    return (best, sim.view(torch.bool)) if want_similar else best


def best_table(best_pix, H, W, b=0):
    dev = require_device(best_pix)
    out = torch.empty((H * W, 4), dtype=torch.int64, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    check(lib().gs_best_table_i64(ptr(best_pix), H, W, b, ptr(out), ptr(cnt), ptr(ws.scratch(0, H * W)),
                                  stream(dev)), "gs_best_table_i64")
    return out[: _count(cnt)]


def rows_to_best_pix(rows, H, W):
    rows = _c(rows, torch.int64)
    dev = require_device(rows)
    best = torch.empty(H * W, dtype=torch.int32, device=dev)
    check(lib().gs_rows_to_best_pix(ptr(rows), rows.shape[0], H, W, ptr(best), stream(dev)), "gs_rows_to_best_pix")
    return best


# ----------------------------------------------------------------------------------- K6
def fuse_append_(points, normals, colors, ccounts, n_map, best_pix, gvertex, gnormal, rgb, alpha, depth,
                 renorm_all=True, n_dev=None, sync=True):
    """In-place on capacity-backed buffers (rows >= n_map are free space).  Returns the new count
    (sync=False: as the device int64[1] tensor the kernels wrote, nothing is read back).
    n_dev: device-side count of the map (n_map is then its upper bound)."""
    gvertex, gnormal, rgb, alpha, depth = _c(gvertex), _c(gnormal), _c(rgb), _c(alpha), _c(depth)
    dev = require_device(points, normals, colors, ccounts, best_pix, gvertex, gnormal, rgb, alpha, depth)
    H, W = depth.shape[:2]
    cap = points.shape[0]
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    if n_dev is not None:
        check(lib().gs_fuse_append_dc_f32(ptr(points), ptr(normals), ptr(colors), ptr(ccounts), int(n_map),
                                          ptr(n_dev), cap, ptr(best_pix), ptr(gvertex), ptr(gnormal), ptr(rgb),
                                          ptr(alpha), ptr(depth), H, W, int(renorm_all), ptr(cnt),
                                          ptr(ws.scratch(n_map, H * W)), stream(dev)), "gs_fuse_append_dc_f32")
    else:
        check(lib().gs_fuse_append_f32(ptr(points), ptr(normals), ptr(colors), ptr(ccounts), int(n_map), cap,
                                       ptr(best_pix), ptr(gvertex), ptr(gnormal), ptr(rgb), ptr(alpha), ptr(depth),
                                       H, W, int(renorm_all), ptr(cnt), ptr(ws.scratch(n_map, H * W)),
                                       stream(dev)), "gs_fuse_append_f32")
    if not sync:
        return cnt
    c = _count(cnt)
    if c > cap:
        raise _C.HipExtensionError("gs_fuse_append_f32: surfel store overflow (%d > %d)" % (c, cap))
    return c


def update_map_fusion_(points, normals, colors, ccounts, n_map, vertex, normal, depth, rgb, alpha, pose, K, dist_th,
                       dot_th, renorm_all=True, n_dev=None, out=None):
    """One-call map update of a sequence (gs_update_map_fusion_dc_f32), in place on capacity-backed buffers.
    Returns (new count as a device int64[1] tensor, gvertex, gnormal, best_pix); nothing is read back."""
    vertex, normal, depth, rgb, alpha = _c(vertex), _c(normal), _c(depth), _c(rgb), _c(alpha)
    pose, K = _c(pose), _c(K)
    dev = require_device(points, normals, colors, ccounts, vertex, normal, depth, rgb, alpha, pose, K, n_dev)
    H, W = depth.shape[:2]
    cap = points.shape[0]
    gv, gn = out if out is not None else (torch.empty((H, W, 3), dtype=f32, device=dev),
                                          torch.empty((H, W, 3), dtype=f32, device=dev))
    best = torch.empty(H * W, dtype=torch.int32, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    scratch = Workspace.get(dev).bytes("map_update", lib().gs_update_map_scratch_bytes(int(n_map), H, W))
    check(lib().gs_update_map_fusion_dc_f32(ptr(points), ptr(normals), ptr(colors), ptr(ccounts), int(n_map), ptr(n_dev),
                                            cap, ptr(vertex), ptr(normal), ptr(depth), ptr(rgb), ptr(alpha), ptr(pose),
                                            ptr(K), H, W, float(dist_th), float(dot_th), 1 if renorm_all else 0, ptr(gv),
                                            ptr(gn), ptr(best), ptr(cnt), ptr(scratch), stream(dev)),
          "gs_update_map_fusion_dc_f32")
    return cnt, gv, gn, best


def append_valid_(points, normals, colors, ccounts, n_map, gvertex, gnormal, rgb, alpha, depth, n_dev=None,
                  sync=True):
    gvertex, gnormal, rgb, alpha, depth = _c(gvertex), _c(gnormal), _c(rgb), _c(alpha), _c(depth)
    dev = require_device(points, normals, colors, ccounts, gvertex, gnormal, rgb, alpha, depth)
    H, W = depth.shape[:2]
    cap = points.shape[0]
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    if n_dev is not None:
        check(lib().gs_append_valid_dc_f32(ptr(points), ptr(normals), ptr(colors), ptr(ccounts), int(n_map),
                                           ptr(n_dev), cap, ptr(gvertex), ptr(gnormal), ptr(rgb), ptr(alpha),
                                           ptr(depth), H, W, ptr(cnt), ptr(ws.scratch(n_map, H * W)), stream(dev)),
              "gs_append_valid_dc_f32")
    else:
        check(lib().gs_append_valid_f32(ptr(points), ptr(normals), ptr(colors), ptr(ccounts), int(n_map), cap,
                                        ptr(gvertex), ptr(gnormal), ptr(rgb), ptr(alpha), ptr(depth), H, W, ptr(cnt),
                                        ptr(ws.scratch(n_map, H * W)), stream(dev)), "gs_append_valid_f32")
    if not sync:
        return cnt
    c = _count(cnt)
    if c > cap:
        raise _C.HipExtensionError("gs_append_valid_f32: surfel store overflow (%d > %d)" % (c, cap))
    return c


# ----------------------------------------------------------------------------------- batched frame loop
def frame_maps_batch(depth, K, sigma=None, out=None):
    """K1 for every frame of a (B, L, H, W) depth stack in one launch (its (H, W) images must be contiguous; the
    sequence / frame strides are free, so a slice frames[:, s] of a longer stack is read in place): K (B, 4, 4) ->
    vertex (B, L, H, W, 3), normal (B, L, H, W, 3), alpha (B, L, H, W) or None (sigma None)."""
    K = _c(K)
    if depth.dtype != f32:
        depth = depth.to(f32)
    Bn, L, H, W = depth.shape
    if depth.stride(3) != 1 or depth.stride(2) != W or (L > 1 and depth.stride(1) < H * W) or \
            (Bn > 1 and depth.stride(0) < H * W):
        depth = depth.contiguous()
    dev = require_device(K)
    if depth.device != dev:
        raise _C.HipExtensionError("gradslam_amd kernels run on the GPU only; all tensors on one device")
    ov, on, oa = out if out is not None else (None, None, None)
    vertex = ov if ov is not None else torch.empty((Bn, L, H, W, 3), dtype=f32, device=dev)
    normal = on if on is not None else torch.empty((Bn, L, H, W, 3), dtype=f32, device=dev)
    alpha = oa if oa is not None else (torch.empty((Bn, L, H, W), dtype=f32, device=dev) if sigma is not None else None)
    require_device(vertex, normal, alpha)
    tss = two_sigma_sq(0.6 if sigma is None else sigma)
    sb, sl = (depth.stride(0) if Bn > 1 else L * H * W), (depth.stride(1) if L > 1 else H * W)
    step = max(32768 // L, 1) if Bn * L > 32768 and L <= 32768 else Bn   # gridDim.z limit: whole sequences per launch
    for b0 in range(0, Bn, step):
        m = min(step, Bn - b0)
        check(lib().gs_frame_maps_batch_f32(depth.data_ptr() + 4 * b0 * sb, sb, sl, ptr(K[b0:]), m * L, L, H, W, tss,
                                            ptr(vertex[b0:]), ptr(normal[b0:]),
                                            None if alpha is None else ptr(alpha[b0:]), stream(dev)),
              "gs_frame_maps_batch_f32")
    return vertex, normal, alpha


def _batch_view(t, inner_ndim):
    """(B, ...) float32 tensor whose per-sequence slices t[b] are contiguous -> (tensor kept alive, base pointer,
    batch stride in bytes).  A slice frames[:, s] of a (B, L, ...) stack qualifies as is (its batch stride is L
    frames): no copy; anything else is made contiguous first."""
    if t.dtype != f32:
        t = t.to(f32)
    inner = t.shape[t.ndim - inner_ndim:]
    want, acc = [], 1
    for d in reversed(inner):
        want.append(acc)
        acc *= int(d)
    if tuple(t.stride()[t.ndim - inner_ndim:]) != tuple(reversed(want)) or (t.shape[0] > 1 and t.stride(0) < acc):
        t = t.contiguous()
    return t, t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else acc) * 4


def _map_view(bufs, cap, n_bound, n_dev):
    P, N, Cc, F = bufs
    return _C.MapView(P.data_ptr(), N.data_ptr(), 0 if Cc is None else Cc.data_ptr(), 0 if F is None else F.data_ptr(),
                      int(cap), int(n_bound), 0 if n_dev is None else n_dev.data_ptr())


def localize_batch(vertex, depth, K, prev_poses, maps, ds, mode=1, numiters=20, damp=1e-8, dist_thresh=None,
                   lambda_max=2.0, B=1.0, B2=1.0, nu=200.0, out=None):
    """ICPSLAM._localize for all sequences of a batch in one chain of launches (gs_localize_batch_f32).
    vertex (B, H, W, 3) LOCAL vertex maps of the live frames, depth (B, H, W), K / prev_poses (B, 4, 4);
    maps: per sequence (points, normals, n_bound, n_dev) with capacity-backed buffers (n_dev None: n_bound exact).
    Returns the recovered poses (B, 4, 4) = T_icp @ prev_pose."""
    K, prev_poses = _c(K), _c(prev_poses)
    Bn, H, W = depth.shape
    vertex, v0, vs = _batch_view(vertex, 3)
    depth, d0, dstr = _batch_view(depth, 2)
    dev = require_device(K, prev_poses)
    if vertex.device != dev or depth.device != dev or not vertex.is_cuda:
        raise _C.HipExtensionError("gradslam_amd kernels run on the GPU only; all tensors on one device")
    T = torch.empty((Bn, 4, 4), dtype=f32, device=dev) if out is None else out
    assert T.shape == (Bn, 4, 4) and T.dtype == f32 and T.is_contiguous() and T.device == dev
    prm = _C.IcpParams(int(mode), int(numiters), float(damp), _thresh(dist_thresh), float(lambda_max), float(B),
                       float(B2), float(nu))
    ws = Workspace.get(dev)
    seqs = (_C.LocalizeSeq * Bn)()
    L = lib()
    k0, p0, t0 = K.data_ptr(), prev_poses.data_ptr(), T.data_ptr()
    for b in range(Bn):
        P, N, n_bound, n_dev = maps[b]
        require_device(P, N, n_dev)
        if P.dtype != f32 or N.dtype != f32 or N.shape[0] < P.shape[0]:
            # (raw pointers go to the library: a float64 / float16 map would be read as float32 garbage)
            raise _C.HipExtensionError("localize_batch needs float32 map points / normals on buffers of equal capacity "
                                       "(got %s / %s, %d / %d rows)" % (P.dtype, N.dtype, P.shape[0], N.shape[0]))
        cap = P.shape[0]
        scratch = ws.bytes("localize%d" % b, L.gs_localize_scratch_bytes(H, W, int(ds), cap))
        q = seqs[b]
        q.vertex, q.depth, q.K16, q.prev_pose16 = v0 + b * vs, d0 + b * dstr, k0 + b * 64, p0 + b * 64
        q.map = _map_view((P, N, None, None), cap, n_bound, n_dev)
        q.out_pose16, q.scratch = t0 + b * 64, scratch.data_ptr()
    check(L.gs_localize_batch_f32(seqs, Bn, H, W, int(ds), prm, stream(dev)), "gs_localize_batch_f32")
    return T


def localize_far_stats(device, b, H, W, ds, capacity):
    """Diagnostics of the last localisation of sequence b on `device` (gs_localize_far_stats_i64): (source points the
    first search found far from every target, how many of them ended the solve with a proven candidate list, source
    points of the second list-building pass, how many of the first have a list that fits)."""
    L = lib()
    scratch = Workspace.get(device).bytes("localize%d" % b, L.gs_localize_scratch_bytes(H, W, int(ds), int(capacity)))
    import ctypes
    out = (ctypes.c_int64 * 4)()
    check(L.gs_localize_far_stats_i64(scratch.data_ptr(), H, W, int(ds), int(capacity), out, stream(device)),
          "gs_localize_far_stats_i64")
    return tuple(int(v) for v in out)


def localize_list_stats(device, b, H, W, ds, capacity):
    """Diagnostics of the candidate lists of ordinary source points in the last localisation of sequence b on `device`
    (gs_localize_list_stats_i64): (failed[64], empty_list[64], without_list[64]) per launch of the solve (launch = 2 x
    iteration + half).  All zero when the solve kept no lists."""
    L = lib()
    scratch = Workspace.get(device).bytes("localize%d" % b, L.gs_localize_scratch_bytes(H, W, int(ds), int(capacity)))
    import ctypes
    out = (ctypes.c_int64 * 192)()
    check(L.gs_localize_list_stats_i64(scratch.data_ptr(), H, W, int(ds), int(capacity), out, stream(device)),
          "gs_localize_list_stats_i64")
    v = [int(x) for x in out]
    return v[:64], v[64:128], v[128:]


def update_map_fusion_batch_(maps, vertex, normal, depth, rgb, alpha, poses, K, dist_th, dot_th, renorm_all=True,
                             out=None):
    """update_map_fusion of all sequences of a batch, in place on their capacity-backed buffers
    (gs_update_map_fusion_batch_f32: 4 launches for the whole batch).  maps: per sequence
    (points, normals, colors, ccounts, n_bound, n_dev).  vertex / normal / rgb (B, H, W, 3), depth / alpha (B, H, W),
    poses / K (B, 4, 4).  Returns (counts int64 (B,) on the device, gvertex, gnormal, best_pix (B, H*W))."""
    poses, K = _c(poses), _c(K)
    dev = require_device(poses, K)
    Bn, H, W = depth.shape
    views = [_batch_view(t, nd) for t, nd in ((vertex, 3), (normal, 3), (depth, 2), (rgb, 3), (alpha, 2))]
    if any(v[0].device != dev for v in views):
        raise _C.HipExtensionError("gradslam_amd kernels run on the GPU only; all tensors on one device")
    gv, gn = out if out is not None else (torch.empty((Bn, H, W, 3), dtype=f32, device=dev),
                                          torch.empty((Bn, H, W, 3), dtype=f32, device=dev))
    gviews = [_batch_view(t, 3) for t in (gv, gn)]
    assert gviews[0][0] is gv and gviews[1][0] is gn, "output global maps must have contiguous (H, W, 3) slices"
    best = torch.empty((Bn, H * W), dtype=torch.int32, device=dev)
    cnt = torch.empty(Bn, dtype=torch.int64, device=dev)
    ws = Workspace.get(dev)
    seqs = (_C.UpdateSeq * Bn)()
    L = lib()
    P1 = H * W * 4
    base = [t.data_ptr() for t in (poses, K, best, cnt)]
    for b in range(Bn):
        P, N, Cc, F, n_bound, n_dev = maps[b]
        require_device(P, N, Cc, F, n_dev)
        cap = P.shape[0]
        scratch = ws.bytes("map_update%d" % b, L.gs_update_map_scratch_bytes(cap, H, W))
        u = seqs[b]
        u.map = _map_view((P, N, Cc, F), cap, n_bound, n_dev)
        u.vertex, u.normal, u.depth, u.rgb, u.alpha = (v[1] + b * v[2] for v in views)
        u.pose16, u.K16 = base[0] + b * 64, base[1] + b * 64
        u.gvertex, u.gnormal = gviews[0][1] + b * gviews[0][2], gviews[1][1] + b * gviews[1][2]
        u.best_pix, u.new_count_out, u.scratch = base[2] + b * P1, base[3] + b * 8, scratch.data_ptr()
    check(L.gs_update_map_fusion_batch_f32(seqs, Bn, H, W, float(dist_th), float(dot_th), 1 if renorm_all else 0,
                                           stream(dev)), "gs_update_map_fusion_batch_f32")
    return cnt, gv, gn, best


# ----------------------------------------------------------------------------------- K7 (autograd)
def icp_with_tape(src, tgt, tgt_normals, init=None, numiters=20, damp=1e-8, dist_thresh=None, lambda_max=2.0,
                  B=1.0, B2=1.0, nu=200.0, mode=1):
    """(grad)ICP forward that also records the tape gs_icp_backward_f32 needs.  Returns (T, idx, tape, prm)."""
    src, tgt, tn = _c(src), _c(tgt), _c(tgt_normals)
    dev = require_device(src, tgt, tn)
    init = torch.eye(4, dtype=f32, device=dev) if init is None else _c(init)
    require_device(init)
    ns, nt = src.shape[0], tgt.shape[0]
    prm = _C.IcpParams(int(mode), int(numiters), float(damp), _thresh(dist_thresh),
                       float(lambda_max), float(B), float(B2), float(nu))
    T = torch.empty((4, 4), dtype=f32, device=dev)
    idx = torch.empty(ns, dtype=torch.int64, device=dev)
    scratch = Workspace.get(dev).bytes("icp", lib().gs_icp_scratch_bytes(ns, nt))
    tape = torch.empty(lib().gs_icp_tape_bytes(ns, int(numiters)), dtype=torch.uint8, device=dev)
    check(lib().gs_icp_tape_f32(ptr(src), ns, ptr(tgt), ptr(tn), nt, ptr(init), None, prm, ptr(T), ptr(idx),
                                ptr(scratch), ptr(tape), stream(dev)), "gs_icp_tape_f32")
    return T, idx, tape, prm


def icp_backward(tape, prm, src, tgt, tgt_normals, init, T_bar, need=(True, True, True, True)):
    """dL/dT -> (dL/dsrc, dL/dtgt, dL/dnormals, dL/dinit); entries not needed are None."""
    src, tgt, tn, init, T_bar = _c(src), _c(tgt), _c(tgt_normals), _c(init), _c(T_bar)
    dev = require_device(tape, src, tgt, tn, init, T_bar)
    ns, nt = src.shape[0], tgt.shape[0]
    gs = torch.empty_like(src) if need[0] else None
    gt = torch.empty_like(tgt) if need[1] else None
    gn = torch.empty_like(tn) if need[2] else None
    gi = torch.empty((4, 4), dtype=f32, device=dev) if need[3] else None
    scratch = Workspace.get(dev).bytes("icp_bwd", lib().gs_icp_backward_scratch_bytes(ns, nt))
    check(lib().gs_icp_backward_f32(ptr(tape), ptr(src), ns, ptr(tgt), ptr(tn), nt, ptr(init), prm, ptr(T_bar),
                                    ptr(gs), ptr(gt), ptr(gn), ptr(gi), ptr(scratch), stream(dev)),
          "gs_icp_backward_f32")
    return gs, gt, gn, gi


class GradICPFunction(torch.autograd.Function):
    """point_to_plane_gradICP as a differentiable op: forward = gs_icp_tape_f32, backward =
    gs_icp_backward_f32 (hand-written reverse mode; no PyTorch ops on the tape)."""

    @staticmethod
    def forward(ctx, src, tgt, tgt_normals, init, numiters, damp, dist_thresh, lambda_max, B, B2, nu, mode=1):
        T, idx, tape, prm = icp_with_tape(src, tgt, tgt_normals, init, numiters, damp, dist_thresh, lambda_max, B, B2,
                                          nu, mode)
        ctx.save_for_backward(src, tgt, tgt_normals, init, tape)
        ctx.prm = prm
        ctx.mark_non_differentiable(idx)
        return T, idx

    @staticmethod
    def backward(ctx, T_bar, _idx_bar):
        src, tgt, tn, init, tape = ctx.saved_tensors
        need = tuple(ctx.needs_input_grad[:4])
        gs, gt, gn, gi = icp_backward(tape, ctx.prm, src, tgt, tn, init, T_bar.contiguous(), need)
        return (gs, gt, gn, gi) + (None,) * 8


def grad_icp(src, tgt, tgt_normals, init=None, numiters=20, damp=1e-8, dist_thresh=None, lambda_max=2.0, B=1.0,
             B2=1.0, nu=200.0, mode=1):
    """Differentiable gradICP (mode 1) / hard-LM ICP (mode 0): (T (4,4), idx (Ns,)).  Uses the plain forward when
    nothing requires grad."""
    dev = src.device
    init = torch.eye(4, dtype=f32, device=dev) if init is None else init
    if torch.is_grad_enabled() and any(t.requires_grad for t in (src, tgt, tgt_normals, init)):
        return GradICPFunction.apply(src, tgt, tgt_normals, init, numiters, damp, dist_thresh, lambda_max, B, B2, nu,
                                     mode)
    return icp(src, tgt, tgt_normals, init=init, mode=mode, numiters=numiters, damp=damp, dist_thresh=dist_thresh,
               lambda_max=lambda_max, B=B, B2=B2, nu=nu)


class FuseAppendFunction(torch.autograd.Function):
    """fuse_with_map of one sequence as a differentiable op (out of place): old map rows + frame maps -> fused map.
    forward = gs_fuse_append_f32 on a copy, backward = gs_fuse_append_backward_f32.  Correspondences (best_pix)
    and depth (validity only) are constants."""

    @staticmethod
    def forward(ctx, points, normals, colors, ccounts, gvertex, gnormal, rgb, alpha, depth, best_pix, renorm_all):
        n0 = points.shape[0]
        H, W = depth.shape[:2]
        dev = gvertex.device
        bufs = [torch.empty((n0 + H * W, c), dtype=f32, device=dev) for c in (3, 3, 3, 1)]
        for dst, src in zip(bufs, (points, normals, colors, ccounts)):
            dst[:n0] = src
        n1 = fuse_append_(*bufs, n0, best_pix, gvertex, gnormal, rgb, alpha, depth, renorm_all)
        ctx.save_for_backward(points, normals, colors, ccounts, gvertex, gnormal, rgb, alpha, depth, best_pix)
        ctx.renorm_all, ctx.n1 = int(renorm_all), n1
        return tuple(b[:n1] for b in bufs)

    @staticmethod
    def backward(ctx, P_bar, N_bar, C_bar, F_bar):
        points, normals, colors, ccounts, gvertex, gnormal, rgb, alpha, depth, best_pix = ctx.saved_tensors
        dev = gvertex.device
        n0, n1 = points.shape[0], ctx.n1
        H, W = depth.shape[:2]
        bars = [_c(t) if t is not None else torch.zeros((n1, c), dtype=f32, device=dev)
                for t, c in zip((P_bar, N_bar, C_bar, F_bar), (3, 3, 3, 1))]
        old = [torch.empty((n0, c), dtype=f32, device=dev) for c in (3, 3, 3, 1)]
        gv_b, gn_b, rgb_b = (torch.empty((H, W, 3), dtype=f32, device=dev) for _ in range(3))
        a_b = torch.empty((H, W), dtype=f32, device=dev)
        ws = Workspace.get(dev)
        check(lib().gs_fuse_append_backward_f32(ptr(_c(points)), ptr(_c(normals)), ptr(_c(colors)), ptr(_c(ccounts)), n0,
                                                ptr(best_pix), ptr(_c(gvertex)), ptr(_c(gnormal)), ptr(_c(rgb)),
                                                ptr(_c(alpha)), ptr(_c(depth)), H, W, int(ctx.renorm_all),
                                                ptr(bars[0]), ptr(bars[1]), ptr(bars[2]), ptr(bars[3]), n1, ptr(old[0]),
                                                ptr(old[1]), ptr(old[2]), ptr(old[3]), ptr(gv_b), ptr(gn_b), ptr(rgb_b),
                                                ptr(a_b), ptr(ws.scratch(n0, H * W)), stream(dev)),
              "gs_fuse_append_backward_f32")
        return old[0], old[1], old[2], old[3], gv_b, gn_b, rgb_b, a_b, None, None, None


class FrameMapsFunction(torch.autograd.Function):
    """depth (H,W) -> (vertex, normal, alpha); differentiable w.r.t. depth (gs_frame_maps_backward_f32)."""

    @staticmethod
    def forward(ctx, depth, K, sigma):
        v, n, a, _ = frame_maps(depth, K, sigma, want_valid=False)
        ctx.save_for_backward(depth, K)
        ctx.sigma = sigma
        return v, n, a

    @staticmethod
    def backward(ctx, v_bar, n_bar, a_bar):
        depth, K = ctx.saved_tensors
        depth_c, K_c = _c(depth), _c(K)
        dev = require_device(depth_c, K_c)
        H, W = depth_c.shape
        v_bar, n_bar, a_bar = _c(v_bar), _c(n_bar), _c(a_bar)
        d_bar = torch.empty((H, W), dtype=f32, device=dev) if ctx.needs_input_grad[0] else None
        scratch = torch.empty((H, W, 6), dtype=f32, device=dev) if n_bar is not None else None
        K_bar, k_scratch = None, None
        if ctx.needs_input_grad[1]:   # gradient w.r.t. the intrinsics (fx, fy, cx, cy entries)
            K_bar = torch.empty((4, 4), dtype=f32, device=dev)
            k_scratch = Workspace.get(dev).bytes("kbar", lib().gs_frame_maps_backward_kbar_scratch_bytes(H, W))
        check(lib().gs_frame_maps_backward_f32(ptr(depth_c), ptr(K_c), H, W, two_sigma_sq(ctx.sigma), ptr(v_bar),
                                               ptr(n_bar), ptr(a_bar), ptr(d_bar), ptr(scratch), ptr(K_bar),
                                               ptr(k_scratch), stream(dev)), "gs_frame_maps_backward_f32")
        return d_bar, K_bar, None


class GlobalMapsFunction(torch.autograd.Function):
    """(vertex, normal) + pose -> (gvertex, gnormal); differentiable w.r.t. the local maps and the pose."""

    @staticmethod
    def forward(ctx, vertex, normal, depth, pose):
        gv, gn = global_maps(vertex, normal, depth, pose)
        ctx.save_for_backward(depth, pose, vertex, normal)
        return gv, gn

    @staticmethod
    def backward(ctx, gv_bar, gn_bar):
        depth, pose, vertex, normal = ctx.saved_tensors
        depth_c, pose_c = _c(depth), _c(pose)
        dev = require_device(depth_c, pose_c)
        H, W = depth_c.shape[:2]
        gv_bar, gn_bar = _c(gv_bar), _c(gn_bar)
        v_bar = torch.empty((H, W, 3), dtype=f32, device=dev) if gv_bar is not None else None
        n_bar = torch.empty((H, W, 3), dtype=f32, device=dev) if gn_bar is not None else None
        check(lib().gs_global_maps_backward_f32(ptr(gv_bar), ptr(gn_bar), ptr(depth_c), ptr(pose_c), H, W, ptr(v_bar),
                                                ptr(n_bar), stream(dev)), "gs_global_maps_backward_f32")
        pose_bar = None
        if ctx.needs_input_grad[3]:
            pose_bar = torch.empty((4, 4), dtype=f32, device=dev)
            scratch = Workspace.get(dev).bytes("pose_bar", lib().gs_global_maps_pose_backward_scratch_bytes(H, W))
            check(lib().gs_global_maps_pose_backward_f32(ptr(_c(vertex)), ptr(_c(normal)), ptr(depth_c), ptr(gv_bar),
                                                         ptr(gn_bar), H, W, ptr(pose_bar), ptr(scratch), stream(dev)),
                  "gs_global_maps_pose_backward_f32")
        return v_bar, n_bar, None, pose_bar


class DownsampleFramePointsFunction(torch.autograd.Function):
    """global vertex map -> compact lattice points; backward scatters the adjoints to their pixels."""

    @staticmethod
    def forward(ctx, gvertex, depth, ds):
        pts, _, _ = downsample_frame(gvertex, None, None, depth, ds)
        ctx.save_for_backward(depth)
        ctx.ds = ds
        return pts

    @staticmethod
    def backward(ctx, pts_bar):
        (depth,) = ctx.saved_tensors
        depth_c, pts_bar = _c(depth), _c(pts_bar)
        dev = require_device(depth_c, pts_bar)
        H, W = depth_c.shape[:2]
        gv_bar = torch.empty((H, W, 3), dtype=f32, device=dev)
        ws = Workspace.get(dev)
        check(lib().gs_downsample_frame_backward_f32(ptr(pts_bar), ptr(depth_c), H, W, ctx.ds, ptr(gv_bar),
                                                     ptr(ws.scratch(0, H * W)), stream(dev)),
              "gs_downsample_frame_backward_f32")
        return gv_bar, None, None
