"""Pose / map parity metrics (the reference's gradslam.metrics package is empty)."""
import torch

__all__ = ["ate_rmse"]


def ate_rmse(poses_a: torch.Tensor, poses_b: torch.Tensor) -> float:
    """Absolute trajectory error: RMSE over frames of the translation difference of two
    (..., L, 4, 4) pose stacks expressed in the same world frame."""
    d = poses_a[..., :3, 3].double() - poses_b[..., :3, 3].double()
    return float(torch.sqrt((d * d).sum(-1).mean()))
