"""se3_exp with the reference's signature (geometry/se3utils.py:77-115), evaluated by the HIP
kernel gs_se3_exp_f32 (double-precision Rodrigues, rounded once)."""
import torch

__all__ = ["se3_exp"]


def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    assert torch.is_tensor(xi), "Input must be of type torch.tensor."
    from .. import ops
    return ops.se3_exp(xi.reshape(-1))
