"""Function-level mirror of the reference's odometry/icputils.py: same names, argument meaning,
shapes and error behaviour; the bodies are calls into libgradslam_hip.so.

    solve_linear_system      -> gs_solve_normal_eq_f32          (icputils.py:22-90)
    gauss_newton_solve       -> gs_gauss_newton_rows_f32        (icputils.py:93-232)
    point_to_plane_ICP       -> gs_icp_f32 mode 0               (icputils.py:235-367)
    point_to_plane_gradICP   -> gs_icp_f32 mode 1               (icputils.py:370-545)
    downsample_pointclouds   -> gs_downsample_table_f32         (icputils.py:548-620)
    downsample_rgbdimages    -> gs_downsample_frame_f32         (icputils.py:623-669)
"""
from typing import Optional, Union

import torch

from ..structures.pointclouds import Pointclouds
from ..structures.rgbdimages import RGBDImages

__all__ = ["solve_linear_system", "gauss_newton_solve", "point_to_plane_ICP", "point_to_plane_gradICP",
           "downsample_pointclouds", "downsample_rgbdimages"]


def _tensor_check(x, name):
    if not torch.is_tensor(x):
        raise TypeError("Expected {0} to be of type torch.Tensor. Got {1}.".format(name, type(x)))


def solve_linear_system(A: torch.Tensor, b: torch.Tensor, damp: Union[float, torch.Tensor] = 1e-8):
    _tensor_check(A, "A")
    _tensor_check(b, "b")
    if not (isinstance(damp, float) or torch.is_tensor(damp)):
        raise TypeError("Expected damp to be of type float or torch.Tensor. Got {0}.".format(type(damp)))
    if torch.is_tensor(damp) and damp.ndim != 0:
        raise ValueError("Expected torch.Tensor damp to have ndim=0 (scalar). Got {0}.".format(damp.ndim))
    if A.ndim != 2:
        raise ValueError("A should have ndim=2, but had ndim={}".format(A.ndim))
    if b.ndim != 2:
        raise ValueError("b should have ndim=2, but had ndim={}".format(b.ndim))
    if b.shape[1] != 1:
        raise ValueError("b.shape[1] should 1, but was {0}".format(b.shape[1]))
    if A.shape[0] != b.shape[0]:
        raise ValueError("A.shape[0] and b.shape[0] should be equal ({0} != {1})".format(A.shape[0], b.shape[0]))
    from .. import ops
    return ops.solve_normal_eq(A, b, float(damp)).view(-1, 1)


def _gn_checks(src_pc, tgt_pc, tgt_normals, dist_thresh):
    _tensor_check(src_pc, "src_pc")
    _tensor_check(tgt_pc, "tgt_pc")
    _tensor_check(tgt_normals, "tgt_normals")
    if not (isinstance(dist_thresh, (float, int)) or dist_thresh is None):
        raise TypeError("Expected dist_thresh to be of type float or int. Got {0}.".format(type(dist_thresh)))
    for t, name in ((src_pc, "src_pc"), (tgt_pc, "tgt_pc"), (tgt_normals, "tgt_normals")):
        if t.ndim != 3:
            raise ValueError("{0} should have ndim=3, but had ndim={1}".format(name, t.ndim))
    for t, name in ((src_pc, "src_pc"), (tgt_pc, "tgt_pc"), (tgt_normals, "tgt_normals")):
        if t.shape[0] != 1:
            raise ValueError("{0}.shape[0] should be 1, but was {1} instead".format(name, t.shape[0]))
    if tgt_pc.shape[1] != tgt_normals.shape[1]:
        raise ValueError("tgt_pc.shape[1] and tgt_normals.shape[1] must be equal. Got {0}!={1}".format(
            tgt_pc.shape[1], tgt_normals.shape[1]))
    for t, name in ((src_pc, "src_pc"), (tgt_pc, "tgt_pc"), (tgt_normals, "tgt_normals")):
        if t.shape[2] != 3:
            raise ValueError("{0}.shape[2] should be 3, but was {1} instead".format(name, t.shape[2]))


def gauss_newton_solve(src_pc: torch.Tensor, tgt_pc: torch.Tensor, tgt_normals: torch.Tensor,
                       dist_thresh: Union[float, int, None] = None):
    r"""Returns (A (Nsf, 6), b (Nsf, 1), chamfer_indices (Nsf,)) like the reference."""
    _gn_checks(src_pc, tgt_pc, tgt_normals, dist_thresh)
    from .. import ops
    A, b, idx, keep = ops.gauss_newton_rows(src_pc[0], tgt_pc[0], tgt_normals[0], dist_thresh)
    if dist_thresh is not None:
        A, b, idx = A[keep], b[keep], idx[keep]
    return A, b.view(-1, 1), idx


def _icp_checks(src_pc, tgt_pc, tgt_normals, initial_transform, numiters):
    _tensor_check(src_pc, "src_pc")
    _tensor_check(tgt_pc, "tgt_pc")
    _tensor_check(tgt_normals, "tgt_normals")
    if not (torch.is_tensor(initial_transform) or initial_transform is None):
        raise TypeError("Expected initial_transform to be of type torch.Tensor. Got {0}.".format(
            type(initial_transform)))
    if not isinstance(numiters, int):
        raise TypeError("Expected numiters to be of type int. Got {0}.".format(type(numiters)))


def _init_checks(initial_transform):
    # the reference dereferences initial_transform before substituting the identity for None
    # (icputils.py:298): None raises AttributeError there too
    if initial_transform.ndim != 2:
        raise ValueError("Expected initial_transform.ndim to be 2. Got {0}.".format(initial_transform.ndim))
    if not (initial_transform.shape[0] == 4 and initial_transform.shape[1] == 4):
        raise ValueError("Expected initial_transform.shape to be (4, 4). Got {0}.".format(initial_transform.shape))


def point_to_plane_ICP(src_pc: torch.Tensor, tgt_pc: torch.Tensor, tgt_normals: torch.Tensor,
                       initial_transform: Optional[torch.Tensor] = None, numiters: int = 20, damp: float = 1e-8,
                       dist_thresh: Union[float, int, None] = None):
    _icp_checks(src_pc, tgt_pc, tgt_normals, initial_transform, numiters)
    _init_checks(initial_transform)
    from .. import ops
    # differentiable (hand-written HIP backward through the accepted LM steps) when any input requires grad
    T, idx = ops.grad_icp(src_pc[0], tgt_pc[0], tgt_normals[0], init=initial_transform, numiters=numiters, damp=damp,
                          dist_thresh=dist_thresh, mode=0)
    return T, idx


def point_to_plane_gradICP(src_pc: torch.Tensor, tgt_pc: torch.Tensor, tgt_normals: torch.Tensor,
                           initial_transform: Optional[torch.Tensor] = None, numiters: int = 20, damp: float = 1e-8,
                           dist_thresh: Union[float, int, None] = None, lambda_max: Union[float, int] = 2.0,
                           B: Union[float, int] = 1.0, B2: Union[float, int] = 1.0, nu: Union[float, int] = 200.0):
    _icp_checks(src_pc, tgt_pc, tgt_normals, initial_transform, numiters)
    if not isinstance(lambda_max, (float, int)):
        raise TypeError("Expected lambda_max to be of type float or int; got {0}".format(type(lambda_max)))
    if not isinstance(B, (float, int)):
        raise TypeError("Expected B to be of type float or int; got {0}".format(type(B)))
    if not isinstance(B2, (float, int)):
        raise TypeError("Expected B2 to be of type float or int; got {0}".format(type(B2)))
    if not isinstance(nu, (float, int)):
        raise TypeError("Expected nu to be of type float or int; got {0}".format(type(nu)))
    _init_checks(initial_transform)
    from .. import ops
    # differentiable (hand-written HIP backward) when any input requires grad
    T, idx = ops.grad_icp(src_pc[0], tgt_pc[0], tgt_normals[0], init=initial_transform, numiters=numiters,
                          damp=damp, dist_thresh=dist_thresh, lambda_max=lambda_max, B=B, B2=B2, nu=nu)
    return T, idx


def downsample_pointclouds(pointclouds: Pointclouds, pc2im_bnhw: torch.Tensor, ds_ratio: int) -> Pointclouds:
    if not isinstance(pointclouds, Pointclouds):
        raise TypeError("Expected pointclouds to be of type gradslam.Pointclouds. Got {0}.".format(type(pointclouds)))
    if not torch.is_tensor(pc2im_bnhw):
        raise TypeError("Expected pc2im_bnhw to be of type torch.Tensor. Got {0}.".format(type(pc2im_bnhw)))
    if not isinstance(ds_ratio, int):
        raise TypeError("Expected ds_ratio to be of type int. Got {0}.".format(type(ds_ratio)))
    if pc2im_bnhw.ndim != 2:
        raise ValueError("Expected pc2im_bnhw to have ndim=2. Got {0}.".format(pc2im_bnhw.ndim))
    if pc2im_bnhw.shape[1] != 4:
        raise ValueError("pc2im_bnhw.shape[1] must be 4, but was {0}.".format(pc2im_bnhw.shape[1]))
    from .. import ops
    pts, nrm, col = [], [], []
    for b in range(len(pointclouds)):
        rows = pc2im_bnhw[pc2im_bnhw[:, 0] == b] if len(pointclouds) > 1 else pc2im_bnhw
        p, n, c = ops.downsample_table(rows, ds_ratio, pointclouds.points_list[b],
                                       None if pointclouds.normals_list is None else pointclouds.normals_list[b],
                                       None if pointclouds.colors_list is None else pointclouds.colors_list[b])
        pts.append(p); nrm.append(n); col.append(c)
    return Pointclouds(points=pts, normals=None if pointclouds.normals_list is None else nrm,
                       colors=None if pointclouds.colors_list is None else col)


def downsample_rgbdimages(rgbdimages: RGBDImages, ds_ratio: int) -> Pointclouds:
    if not isinstance(rgbdimages, RGBDImages):
        raise TypeError("Expected rgbdimages to be of type gradslam.RGBDImages. Got {0}.".format(type(rgbdimages)))
    if not isinstance(ds_ratio, int):
        raise TypeError("Expected ds_ratio to be of type int. Got {0}.".format(type(ds_ratio)))
    if rgbdimages.shape[1] != 1:
        raise ValueError("Sequence length of rgbdimages must be 1, but was {0}.".format(rgbdimages.shape[1]))
    from .. import ops
    fr = rgbdimages.to_channels_last()
    pts, nrm, col = [], [], []
    for b in range(len(fr)):
        gvm = fr.global_vertex_map[b, 0]
        p, n, c = ops.downsample_frame(gvm, fr.global_normal_map[b, 0], fr.rgb_image[b, 0],
                                       fr.depth_image[b, 0, ..., 0], ds_ratio)
        if torch.is_grad_enabled() and gvm.requires_grad:  # keep the points on the autograd tape
            p = ops.DownsampleFramePointsFunction.apply(gvm, fr.depth_image[b, 0, ..., 0].detach(), ds_ratio)
        pts.append(p); nrm.append(n); col.append(c)
    return Pointclouds(points=pts, normals=nrm, colors=col)
