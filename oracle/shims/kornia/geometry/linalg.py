"""Test-infrastructure shim for the two kornia functions the reference's hot path
imports (slam/fusionutils.py:5, slam/icpslam.py:6). kornia is not installed in this image
and is not vendored under /root/reference; this restates the documented rigid-transform
algebra: inverse [R^T, -R^T t] and composition [R1 R2, R1 t2 + t1]."""
import torch


def inverse_transformation(trans_12):
    rmat_12 = trans_12[..., :3, 0:3]
    tvec_12 = trans_12[..., :3, 3:4]
    rmat_21 = torch.transpose(rmat_12, -1, -2)
    tvec_21 = torch.matmul(-rmat_21, tvec_12)
    trans_21 = torch.zeros_like(trans_12)
    trans_21[..., :3, 0:3] += rmat_21
    trans_21[..., :3, -1:] += tvec_21
    trans_21[..., -1, -1:] += 1.0
    return trans_21


def compose_transformations(trans_01, trans_12):
    rmat_01 = trans_01[..., :3, :3]
    rmat_12 = trans_12[..., :3, :3]
    tvec_01 = trans_01[..., :3, -1:]
    tvec_12 = trans_12[..., :3, -1:]
    rmat_02 = torch.matmul(rmat_01, rmat_12)
    tvec_02 = torch.matmul(rmat_01, tvec_12) + tvec_01
    trans_02 = torch.zeros_like(trans_01)
    trans_02[..., :3, 0:3] += rmat_02
    trans_02[..., :3, -1:] += tvec_02
    trans_02[..., -1, -1:] += 1.0
    return trans_02
