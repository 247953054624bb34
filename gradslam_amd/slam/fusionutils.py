"""Function-level mirror of the reference's slam/fusionutils.py (same names, arguments, table
layouts, ordering contracts and error messages); the bodies are HIP kernels.

    get_alpha                          -> gs_alpha_f32 / fused in gs_frame_maps_f32   (fusionutils.py:16-73)
    are_points_close / are_normals_similar -> gs_similar_rows_f32                      (:76-195)
    find_active_map_points             -> gs_project_map_f32 + gs_active_table_i64    (:198-287)
    find_similar_map_points            -> gs_similar_rows_f32                         (:290-411)
    find_best_unique_correspondences   -> gs_best_unique_rows_f32                     (:414-546)
    find_correspondences               -> gs_project_map_f32 + gs_associate_f32       (:549-577)
    fuse_with_map                      -> gs_rows_to_best_pix + gs_fuse_append_f32    (:580-722)
    update_map_aggregate               -> gs_append_valid_f32                         (:725-758)
    update_map_fusion                  -> project + associate + fuse, no tables       (:761-789)

`pc2im_bnhw` tables are int64 (rows, 4) = [b, n, h, w] exactly as in the reference.
"""
import warnings
from typing import Union

import torch

from ..structures.pointclouds import Pointclouds
from ..structures.rgbdimages import RGBDImages

__all__ = ["update_map_fusion", "update_map_aggregate"]

# parity mode (default): rewrite every map row as (cc*x)*(1/cc) each frame exactly like the
# reference's full-map weighted average (fusionutils.py:678-699).  False leaves unmatched
# surfels untouched (saves 72 B/surfel/frame of HBM traffic, results differ by <= 1 ulp/frame).
RENORMALIZE_UNMATCHED = True


# --------------------------------------------------------------------------- small helpers
def get_alpha(points: torch.Tensor, sigma: Union[torch.Tensor, float, int], dim: int = -1, keepdim: bool = False,
              eps: float = 1e-7) -> torch.Tensor:
    if not torch.is_tensor(points):
        raise TypeError("Expected input points to be of type torch.Tensor. Got {0} instead.".format(type(points)))
    if not (torch.is_tensor(sigma) or isinstance(sigma, float) or isinstance(sigma, int)):
        raise TypeError("Expected input sigma to be of type torch.Tensor or float or int. Got {0} instead.".format(
            type(sigma)))
    if not isinstance(eps, float):
        raise TypeError("Expected input eps to be of type float. Got {0} instead.".format(type(eps)))
    if points.shape[dim] != 3:
        raise ValueError("Expected length of dim-th ({0}th) dimension to be 3. Got {1} instead.".format(
            dim, points.shape[dim]))
    if torch.is_tensor(sigma) and sigma.ndim != 0:
        raise ValueError("Expected sigma.ndim to be 0 (scalar). Got {0}.".format(sigma.ndim))
    from .. import ops
    moved = points.movedim(dim, -1)
    alpha = ops.alpha_of_points(moved.reshape(-1, 3), sigma, eps).view(moved.shape[:-1])
    return alpha.unsqueeze(dim) if keepdim else alpha


def _pairwise_checks(tensor1, tensor2, th, th_name, dim):
    if not torch.is_tensor(tensor1):
        raise TypeError("Expected input tensor1 to be of type torch.Tensor. Got {0} instead.".format(type(tensor1)))
    if not torch.is_tensor(tensor2):
        raise TypeError("Expected input tensor2 to be of type torch.Tensor. Got {0} instead.".format(type(tensor2)))
    if not (isinstance(th, float) or isinstance(th, int)):
        raise TypeError("Expected input {0} to be of type float or int. Got {1} instead.".format(th_name, type(th)))
    if tensor1.shape != tensor2.shape:
        raise ValueError("tensor1 and tensor2 should have the same shape, but had shapes {0} and {1} respectively."
                         .format(tensor1.shape, tensor2.shape))
    if tensor1.shape[dim] != 3:
        raise ValueError("Expected length of input tensors' dim-th ({0}th) dimension to be 3. Got {1} instead."
                         .format(dim, tensor1.shape[dim]))


def _pairwise(tensor1, tensor2, dim, dist_th, dot_th, use_points):
    """Runs gs_similar_rows_f32 on row i of tensor1 against row i of tensor2 with the unused half
    of the test neutralised."""
    from .. import ops
    a = tensor1.movedim(dim, -1)
    out_shape = a.shape[:-1]
    a = a.reshape(-1, 3).contiguous().float()
    b = tensor2.movedim(dim, -1).reshape(-1, 3).contiguous().float()
    n = a.shape[0]
    ar = torch.arange(n, device=a.device)
    rows = torch.stack([torch.zeros_like(ar), ar, torch.zeros_like(ar), ar], 1)
    zero = torch.zeros_like(a)
    if use_points:   # |t1 - t2| < dist_th; normals (1,0,0).(1,0,0) = 1 > -inf
        ex = torch.zeros_like(a)
        ex[:, 0] = 1
        mask = ops.similar_rows(rows, b, ex, a.view(1, n, 3), ex.view(1, n, 3), dist_th, float("-inf"))
    else:            # t1 . t2 > dot_th; points 0 - 0 = 0 < +inf
        mask = ops.similar_rows(rows, zero, b, zero.view(1, n, 3), a.view(1, n, 3), float("inf"), dot_th)
    return mask.view(out_shape)


def are_points_close(tensor1: torch.Tensor, tensor2: torch.Tensor, dist_th: Union[float, int], dim: int = -1):
    _pairwise_checks(tensor1, tensor2, dist_th, "dist_th", dim)
    return _pairwise(tensor1, tensor2, dim, float(dist_th), 0.0, True)


def are_normals_similar(tensor1: torch.Tensor, tensor2: torch.Tensor, dot_th: Union[float, int], dim: int = -1):
    _pairwise_checks(tensor1, tensor2, dot_th, "dot_th", dim)
    return _pairwise(tensor1, tensor2, dim, 0.0, float(dot_th), False)


def _check_pc(pointclouds):
    if not isinstance(pointclouds, Pointclouds):
        raise TypeError("Expected pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(type(pointclouds)))


def _check_rgbd(rgbdimages):
    if not isinstance(rgbdimages, RGBDImages):
        raise TypeError("Expected rgbdimages to be of type gradslam.RGBDImages. Got {0}.".format(type(rgbdimages)))


def _check_table(pc2im_bnhw):
    if not torch.is_tensor(pc2im_bnhw):
        raise TypeError("Expected input pc2im_bnhw to be of type torch.Tensor. Got {0} instead.".format(
            type(pc2im_bnhw)))
    if pc2im_bnhw.dtype != torch.int64:
        raise TypeError("Expected input pc2im_bnhw to have dtype of torch.int64 (torch.long), not {0}.".format(
            pc2im_bnhw.dtype))


def _check_table_shape(pc2im_bnhw):
    if pc2im_bnhw.ndim != 2:
        raise ValueError("Expected pc2im_bnhw.ndim of 2. Got {0}.".format(pc2im_bnhw.ndim))
    if pc2im_bnhw.shape[1] != 4:
        raise ValueError("Expected pc2im_bnhw.shape[1] to be 4. Got {0}.".format(pc2im_bnhw.shape[1]))


def _check_seq1(rgbdimages):
    if rgbdimages.shape[1] != 1:
        raise ValueError("Expected rgbdimages to have sequence length of 1. Got {0}.".format(rgbdimages.shape[1]))


def _check_batch(pointclouds, rgbdimages):
    if len(rgbdimages) != len(pointclouds):
        raise ValueError("Expected equal batch sizes for pointclouds and rgbdimages. Got {0} and {1} respectively."
                         .format(len(pointclouds), len(rgbdimages)))


def _rows_of(pc2im_bnhw, b, B):
    """(rows of sequence b, their positions in the table)"""
    if B == 1:
        return pc2im_bnhw, None
    sel = (pc2im_bnhw[:, 0] == b).nonzero().flatten()
    return pc2im_bnhw[sel], sel


def _frame(rgbdimages):
    """channels-last float32 views the kernels consume: K (B,4,4), poses (B,4,4), depth (B,H,W)."""
    fr = rgbdimages.to_channels_last()
    return fr, fr.intrinsics[:, 0].contiguous().float(), fr.poses[:, 0].contiguous().float()


# --------------------------------------------------------------------------- tables
def find_active_map_points(pointclouds: Pointclouds, rgbdimages: RGBDImages) -> torch.Tensor:
    _check_pc(pointclouds)
    _check_rgbd(rgbdimages)
    _check_seq1(rgbdimages)
    device = pointclouds.device
    if not pointclouds.has_points:
        return torch.empty((0, 4), dtype=torch.int64, device=device)
    _check_batch(pointclouds, rgbdimages)
    from .. import ops
    fr, K, poses = _frame(rgbdimages)
    _, _, H, W = fr.shape
    tables = []
    for b in range(len(pointclouds)):
        pix = ops.project_map(pointclouds.points_list[b], poses[b], K[b], H, W)
        tables.append(ops.active_table(pix, W, b))
    pc2im_bnhw = tables[0] if len(tables) == 1 else torch.cat(tables, 0)
    if pc2im_bnhw.shape[0] == 0:
        warnings.warn("No active map points were found")
    return pc2im_bnhw


def find_similar_map_points(pointclouds: Pointclouds, rgbdimages: RGBDImages, pc2im_bnhw: torch.Tensor,
                            dist_th: Union[float, int], dot_th: Union[float, int]):
    _check_pc(pointclouds)
    _check_rgbd(rgbdimages)
    _check_table(pc2im_bnhw)
    _check_seq1(rgbdimages)
    _check_table_shape(pc2im_bnhw)
    device = pointclouds.device
    if not pointclouds.has_points or pc2im_bnhw.shape[0] == 0:
        return torch.empty((0, 4), dtype=torch.int64, device=device), torch.empty(0, dtype=torch.bool, device=device)
    _check_batch(pointclouds, rgbdimages)
    if not pointclouds.has_normals:
        raise ValueError("Pointclouds must have normals for finding similar map points, but did not.")
    from .. import ops
    fr = rgbdimages.to_channels_last()
    B = len(pointclouds)
    mask = torch.zeros(pc2im_bnhw.shape[0], dtype=torch.bool, device=device)
    for b in range(B):
        rows, sel = _rows_of(pc2im_bnhw, b, B)
        if rows.shape[0] == 0:
            continue
        m = ops.similar_rows(rows, pointclouds.points_list[b], pointclouds.normals_list[b],
                             fr.global_vertex_map[b, 0], fr.global_normal_map[b, 0], dist_th, dot_th)
        if sel is None:
            mask = m
        else:
            mask[sel] = m
    pc2im_bnhw_similar = pc2im_bnhw[mask]
    if len(pc2im_bnhw_similar) == 0:
        warnings.warn("No similar map points were found (despite total {0} active points across the batch)".format(
            pc2im_bnhw.shape[0]), RuntimeWarning)
    return pc2im_bnhw_similar, mask


def find_best_unique_correspondences(pointclouds: Pointclouds, rgbdimages: RGBDImages,
                                     pc2im_bnhw: torch.Tensor) -> torch.Tensor:
    _check_pc(pointclouds)
    _check_table(pc2im_bnhw)
    _check_seq1(rgbdimages)
    _check_table_shape(pc2im_bnhw)
    device = pointclouds.device
    if not pointclouds.has_points or pc2im_bnhw.shape[0] == 0:
        return torch.empty((0, 4), dtype=torch.int64, device=device)
    _check_batch(pointclouds, rgbdimages)
    if not pointclouds.has_features:
        raise ValueError("Pointclouds must have features for finding best unique correspondences, but did not.")
    from .. import ops
    fr = rgbdimages.to_channels_last()
    B = len(pointclouds)
    out = []
    for b in range(B):
        rows, _ = _rows_of(pc2im_bnhw, b, B)
        if rows.shape[0] == 0:
            continue
        uq, _ = ops.best_unique_rows(rows, pointclouds.points_list[b], pointclouds.features_list[b][:, :1],
                                     fr.global_vertex_map[b, 0], b)
        out.append(uq)
    return out[0] if len(out) == 1 else torch.cat(out, 0)


def _best_pix_per_sequence(pointclouds, rgbdimages, dist_th, dot_th):
    """Fused find_correspondences: per sequence the (H*W,) int32 winner table (no pc2im rows)."""
    from .. import ops
    fr, K, poses = _frame(rgbdimages)
    _, _, H, W = fr.shape
    best = []
    for b in range(len(rgbdimages)):
        n_b, n_dev = pointclouds._count_of(b) if len(pointclouds) else (0, None)
        if pointclouds._buf["points"] is None or n_b == 0:
            best.append(torch.full((H * W,), -1, dtype=torch.int32, device=fr.device))
            continue
        # rows [0, n_b) of the capacity-backed store; n_dev (if any) holds the exact count on the device
        P, N, F = (pointclouds._buf[k][b][:n_b] for k in ("points", "normals", "features"))
        pix = ops.project_map(P, poses[b], K[b], H, W, n_dev=n_dev)
        best.append(ops.associate(pix, P, N, F[:, :1], fr.global_vertex_map[b, 0], fr.global_normal_map[b, 0],
                                  dist_th, dot_th, n_dev=n_dev))
    return best


def find_correspondences(pointclouds: Pointclouds, rgbdimages: RGBDImages, dist_th: Union[float, int],
                         dot_th: Union[float, int]) -> torch.Tensor:
    _check_pc(pointclouds)
    _check_rgbd(rgbdimages)
    _check_seq1(rgbdimages)
    if not pointclouds.has_points:
        return torch.empty((0, 4), dtype=torch.int64, device=pointclouds.device)
    _check_batch(pointclouds, rgbdimages)
    if not pointclouds.has_normals:
        raise ValueError("Pointclouds must have normals for finding similar map points, but did not.")
    if not pointclouds.has_features:
        raise ValueError("Pointclouds must have features for finding best unique correspondences, but did not.")
    from .. import ops
    _, _, H, W = rgbdimages.shape
    best = _best_pix_per_sequence(pointclouds, rgbdimages, dist_th, dot_th)
    tables = [ops.best_table(bp, H, W, b) for b, bp in enumerate(best)]
    return tables[0] if len(tables) == 1 else torch.cat(tables, 0)


# --------------------------------------------------------------------------- map updates
def _fuse(pointclouds, rgbdimages, best_pix, sigma, inplace):
    from .. import ops
    fr = rgbdimages.to_channels_last()
    B, _, H, W = fr.shape
    if pointclouds.device != fr.device:
        raise ValueError("Device of pointclouds to append and to be appended must match: ({0} != {1})".format(
            fr.device, pointclouds.device))
    alpha = fr._alpha_map(sigma)
    gv, gn = fr.global_vertex_map, fr.global_normal_map
    rgb, depth = fr.rgb_image.contiguous().float(), fr.depth_image.contiguous().float()
    if len(pointclouds) == 0:
        if not inplace:  # nothing to merge into: build the result in a fresh store
            pointclouds, inplace = Pointclouds(device=pointclouds.device), True
        pointclouds._init_empty_batch(B, 1)
    out = pointclouds if inplace else Pointclouds(device=pointclouds.device)
    if not inplace:
        out._init_empty_batch(B, pointclouds._buf["features"][0].shape[-1])
    # fusionutils.py:659 skips the merge only when the table of the WHOLE batch is empty: with B > 1 a sequence
    # without matches is still renormalised when another one has some (mode 2 of gs_fuse_append_f32)
    renorm = RENORMALIZE_UNMATCHED
    if renorm and B > 1 and bool(torch.stack([(bp >= 0).any() for bp in best_pix]).any()):   # (one read-back)
        renorm = 2
    for b in range(B):
        if inplace and ops.DEVICE_COUNTS:
            # the count stays on the device: no read-back, the host only tracks an upper bound.  Reserve FIRST:
            # the bound may tighten between two look-ups, and the capacity must cover the bound that is passed on
            P, N, C, F = pointclouds._reserve(b, H * W, pointclouds.RESERVE_FRAMES)
            n0, n_dev = pointclouds._count_of(b)
            if P.dtype != torch.float32 or F.shape[-1] != 1:
                raise ValueError("map fusion needs float32 surfels with one feature column (the confidence count)")
            cnt = ops.fuse_append_(P, N, C, F, n0, best_pix[b], gv[b, 0], gn[b, 0], rgb[b, 0], alpha[b, 0, ..., 0],
                                   depth[b, 0, ..., 0], renorm, n_dev=n_dev, sync=False)
            pointclouds._set_count_dev(b, cnt, H * W)
            continue
        n0 = pointclouds._n[b]
        P, N, C, F = pointclouds._reserve(b, H * W)
        if P.dtype != torch.float32 or F.shape[-1] != 1:
            raise ValueError("map fusion needs float32 surfels with one feature column (the confidence count)")
        n1 = ops.fuse_append_(P, N, C, F, n0, best_pix[b], gv[b, 0], gn[b, 0], rgb[b, 0], alpha[b, 0, ..., 0],
                              depth[b, 0, ..., 0], renorm)
        if inplace:
            pointclouds._set_count(b, n1)
        else:
            # the reference merges into its argument before cloning (fusionutils.py:696-719): the
            # input keeps the merged rows (count unchanged), the appended rows exist only in the copy
            for k, src in zip(("points", "normals", "colors", "features"), (P, N, C, F)):
                out._buf[k][b] = src[:n1].clone()
            out._set_count(b, n1)
    return out


def fuse_with_map(pointclouds: Pointclouds, rgbdimages: RGBDImages, pc2im_bnhw: torch.Tensor,
                  sigma: Union[torch.Tensor, float, int], inplace: bool = False) -> Pointclouds:
    _check_pc(pointclouds)
    _check_rgbd(rgbdimages)
    _check_table(pc2im_bnhw)
    _check_table_shape(pc2im_bnhw)
    if pointclouds.has_points:
        if not pointclouds.has_normals:
            raise ValueError("Pointclouds must have normals for map fusion, but did not.")
        if not pointclouds.has_colors:
            raise ValueError("Pointclouds must have colors for map fusion, but did not.")
        if not pointclouds.has_features:
            raise ValueError("Pointclouds must have features (ccounts) for map fusion, but did not.")
    from .. import ops
    B, _, H, W = rgbdimages.shape
    best = []
    for b in range(B):
        rows, _ = _rows_of(pc2im_bnhw.to(rgbdimages.device), b, B)
        best.append(ops.rows_to_best_pix(rows, H, W))
    return _fuse(pointclouds, rgbdimages, best, sigma, inplace)


def update_map_aggregate(pointclouds: Pointclouds, rgbdimages: RGBDImages, inplace: bool = False) -> Pointclouds:
    _check_pc(pointclouds)
    _check_rgbd(rgbdimages)
    _check_seq1(rgbdimages)
    from .. import ops
    if _wants_map_grad(pointclouds, rgbdimages):
        return _aggregate_differentiable(pointclouds, rgbdimages, inplace)
    fr = rgbdimages.to_channels_last()
    B, _, H, W = fr.shape
    if not inplace:
        pointclouds = pointclouds.clone()
    if len(pointclouds) == 0:
        pointclouds._init_empty_batch(B, 0)
    gv, gn = fr.global_vertex_map, fr.global_normal_map
    rgb, depth = fr.rgb_image.contiguous().float(), fr.depth_image.contiguous().float()
    if pointclouds.has_features:
        raise ValueError("pointclouds to append and to be appended must either both have or not have features: "
                         "(False != True)")
    for b in range(B):
        if inplace and ops.DEVICE_COUNTS:
            P, N, C, _ = pointclouds._reserve(b, H * W, pointclouds.RESERVE_FRAMES)   # before _count_of: see _fuse
            n0, n_dev = pointclouds._count_of(b)
            cnt = ops.append_valid_(P, N, C, None, n0, gv[b, 0], gn[b, 0], rgb[b, 0], None, depth[b, 0, ..., 0],
                                    n_dev=n_dev, sync=False)
            pointclouds._set_count_dev(b, cnt, H * W)
            continue
        P, N, C, _ = pointclouds._reserve(b, H * W)
        n1 = ops.append_valid_(P, N, C, None, pointclouds._n[b], gv[b, 0], gn[b, 0], rgb[b, 0], None,
                               depth[b, 0, ..., 0])
        pointclouds._set_count(b, n1)
    return pointclouds


def _aggregate_differentiable(pointclouds, rgbdimages, inplace):
    """update_map_aggregate on the autograd tape: FuseAppendFunction with an empty correspondence table (old rows
    pass through untouched, every valid pixel is appended); the confidence column it carries is dropped."""
    from .. import ops
    fr = rgbdimages.to_channels_last()
    B, _, H, W = fr.shape
    gv, gn = fr.global_vertex_map, fr.global_normal_map
    rgb, depth = fr.rgb_image.float(), fr.depth_image.float()
    if len(pointclouds) and pointclouds.has_features:
        raise ValueError("pointclouds to append and to be appended must either both have or not have features: "
                         "(False != True)")
    new = {k: [] for k in ("points", "normals", "colors")}
    none = torch.full((H * W,), -1, dtype=torch.int32, device=fr.device)
    zeros_a = torch.zeros((H, W), dtype=torch.float32, device=fr.device)
    for b in range(B):
        if len(pointclouds) == 0:
            old = [torch.empty((0, 3), dtype=torch.float32, device=fr.device) for _ in range(3)]
        else:
            n_b = pointclouds._n[b]
            old = [pointclouds._buf[k][b][:n_b] for k in ("points", "normals", "colors")]
        cc = torch.zeros((old[0].shape[0], 1), dtype=torch.float32, device=fr.device)
        fused = ops.FuseAppendFunction.apply(old[0], old[1], old[2], cc, gv[b, 0], gn[b, 0], rgb[b, 0], zeros_a,
                                             depth[b, 0, ..., 0].detach(), none, False)
        for k, t in zip(new, fused[:3]):
            new[k].append(t)
    out = pointclouds if inplace else Pointclouds(device=pointclouds.device)
    if len(out) == 0:
        out._init_empty_batch(B, 0)
    for k in new:
        out._buf[k] = new[k]
    out._n = [t.shape[0] for t in new["points"]]
    out._invalidate()
    return out


def update_map_fusion(pointclouds: Pointclouds, rgbdimages: RGBDImages, dist_th: Union[float, int],
                      dot_th: Union[float, int], sigma: Union[torch.Tensor, float, int],
                      inplace: bool = False) -> Pointclouds:
    _check_pc(pointclouds)
    _check_rgbd(rgbdimages)
    _check_seq1(rgbdimages)
    # a map whose counts are device-side has had frames fused into it: no read-back just for this check
    if pointclouds._dcount or pointclouds.has_points:
        _check_batch(pointclouds, rgbdimages)
    from .. import ops
    if pointclouds._dcount or pointclouds.has_points:
        # the checks find_correspondences (fusionutils.py:568-575) and fuse_with_map (:641-653) would run
        if not pointclouds.has_normals:
            raise ValueError("Pointclouds must have normals for finding similar map points, but did not.")
        if not pointclouds.has_features:
            raise ValueError("Pointclouds must have features for finding best unique correspondences, but did not.")
        if not pointclouds.has_colors:
            raise ValueError("Pointclouds must have colors for map fusion, but did not.")
    if _wants_map_grad(pointclouds, rgbdimages):
        return _fuse_differentiable(pointclouds, rgbdimages, dist_th, dot_th, sigma, inplace)
    if inplace and ops.DEVICE_COUNTS and _one_call_update_ok(pointclouds, rgbdimages):
        return _update_map_one_call(pointclouds, rgbdimages, dist_th, dot_th, sigma)
    best = _best_pix_per_sequence(pointclouds, rgbdimages, dist_th, dot_th)
    return _fuse(pointclouds, rgbdimages, best, sigma, inplace)


def _wants_map_grad(pointclouds, rgbdimages):
    """the fused map has to stay on the autograd tape: a frame input or the map itself requires grad"""
    if not torch.is_grad_enabled() or rgbdimages.device.type != "cuda":
        return False
    if rgbdimages.depth_image.requires_grad or rgbdimages.rgb_image.requires_grad:
        return True
    if rgbdimages.poses is not None and rgbdimages.poses.requires_grad:
        return True
    bufs = pointclouds._buf
    return any(bufs[k] is not None and any(t.requires_grad for t in bufs[k]) for k in bufs)


def _fuse_differentiable(pointclouds, rgbdimages, dist_th, dot_th, sigma, inplace):
    """update_map_fusion on the autograd tape (FuseAppendFunction per sequence, out of place as autograd requires):
    depth -> vertex / normal / alpha -> global maps -> merge + append; correspondences are constants.  With
    inplace=True the new tensors replace the buffers of the given object (as the reference's padded setters do)."""
    from .. import ops
    fr, K, poses = _frame(rgbdimages)
    B, _, H, W = fr.shape
    alpha = fr._alpha_map(sigma)
    gv, gn = fr.global_vertex_map, fr.global_normal_map
    rgb, depth = fr.rgb_image.float(), fr.depth_image.float()
    out = pointclouds if inplace else Pointclouds(device=pointclouds.device)
    if len(pointclouds) == 0 and inplace:
        pointclouds._init_empty_batch(B, 1)
    new = {k: [] for k in ("points", "normals", "colors", "features")}
    olds, bests = [], []
    for b in range(B):
        if len(pointclouds) == 0:
            old = [torch.empty((0, c), dtype=torch.float32, device=fr.device) for c in (3, 3, 3, 1)]
        else:
            n_b = pointclouds._n[b]
            old = [pointclouds._buf[k][b][:n_b] for k in ("points", "normals", "colors", "features")]
        with torch.no_grad():
            if old[0].shape[0]:
                pix = ops.project_map(old[0], poses[b], K[b], H, W)
                best = ops.associate(pix, old[0], old[1], old[3][:, :1], gv[b, 0], gn[b, 0], dist_th, dot_th)
            else:
                best = torch.full((H * W,), -1, dtype=torch.int32, device=fr.device)
        olds.append(old)
        bests.append(best)
    # the merge is skipped only when NO sequence of the batch has a match (fusionutils.py:659): mode 2, as in _fuse
    renorm = RENORMALIZE_UNMATCHED
    if renorm and B > 1 and bool(torch.stack([(bp >= 0).any() for bp in bests]).any()):
        renorm = 2
    for b in range(B):
        old, best = olds[b], bests[b]
        fused = ops.FuseAppendFunction.apply(old[0], old[1], old[2], old[3], gv[b, 0], gn[b, 0], rgb[b, 0],
                                             alpha[b, 0, ..., 0], depth[b, 0, ..., 0].detach(), best, renorm)
        for k, t in zip(new, fused):
            new[k].append(t)
    if not inplace and len(pointclouds) == 0:
        out._init_empty_batch(B, 1)
    elif not inplace:
        out._init_empty_batch(B, 1)
    for k in new:
        out._buf[k] = new[k]
    out._n = [t.shape[0] for t in new["points"]]
    out._invalidate()
    return out


def _one_call_update_ok(pointclouds, rgbdimages):
    """the fused entry point covers the SLAM loop's case: float32 surfels with one feature column (or an empty
    map), poses present, nothing on the autograd tape"""
    if rgbdimages.poses is None or pointclouds.device != rgbdimages.device or not rgbdimages.device.type == "cuda":
        return False
    if torch.is_grad_enabled() and rgbdimages.depth_image.requires_grad:
        return False
    if len(pointclouds) == 0:
        return True
    if len(pointclouds) != len(rgbdimages):
        return False
    bufs = pointclouds._buf
    return all(bufs[k] is not None for k in ("points", "normals", "colors", "features")) and \
        bufs["points"][0].dtype == torch.float32 and bufs["features"][0].shape[-1] == 1


def _update_map_one_call(pointclouds, rgbdimages, dist_th, dot_th, sigma):
    """update_map_fusion through gs_update_map_fusion_batch_f32: global maps, association and fuse of EVERY sequence
    of the batch in 6 launches (workgroup -> sequence), surfel counts on the device.  Also fills the frame's
    global-map cache."""
    from .. import ops
    fr, K, poses = _frame(rgbdimages)
    B, _, H, W = fr.shape
    alpha = fr._alpha_map(sigma)
    vm, nm = fr.vertex_map, fr.normal_map
    rgb, depth = fr.rgb_image, fr.depth_image   # a frame slice of a (B, L, ...) stack is consumed in place (no copy)
    if len(pointclouds) == 0:
        pointclouds._init_empty_batch(B, 1)
    gv = torch.empty((B, 1, H, W, 3), dtype=torch.float32, device=fr.device)
    gn = torch.empty_like(gv)
    maps = []
    for b in range(B):
        P, N, C, F = pointclouds._reserve(b, H * W, pointclouds.RESERVE_FRAMES)   # before _count_of: see _fuse
        n0, n_dev = pointclouds._count_of(b)
        maps.append((P, N, C, F, n0, n_dev))
    cnt, _, _, _ = ops.update_map_fusion_batch_(maps, vm[:, 0], nm[:, 0], depth[:, 0, ..., 0], rgb[:, 0],
                                                alpha[:, 0, ..., 0], poses, K, dist_th, dot_th, RENORMALIZE_UNMATCHED,
                                                out=(gv[:, 0], gn[:, 0]))
    pointclouds._set_counts_dev(cnt, H * W)
    fr._global_vertex_map, fr._global_normal_map = gv, gn
    return pointclouds
