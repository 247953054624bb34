"""The reference's geometry/se3utils.py (so3_hat, se3_hat, so3_exp :11-74, se3_exp :77-115) with the same
signatures, evaluated by HIP kernels: gs_lie_small_f32 (the hats are scatters of the input's components; so3_exp is
the rotation block of the double-precision Rodrigues formula, rounded once) and gs_se3_exp_f32."""
import torch

__all__ = ["so3_hat", "se3_hat", "so3_exp", "se3_exp"]


def so3_hat(omega: torch.Tensor) -> torch.Tensor:
    assert torch.is_tensor(omega), "Input must be of type torch.tensor."
    from .. import ops
    return ops.lie_small(0, omega, 3, 3).to(omega.dtype)


def se3_hat(xi: torch.Tensor) -> torch.Tensor:
    assert torch.is_tensor(xi), "Input must be of type torch.tensor."
    from .. import ops
    return ops.lie_small(1, xi, 6, 4).to(xi.dtype)


def so3_exp(omega: torch.Tensor) -> torch.Tensor:
    assert torch.is_tensor(omega), "Input must be of type torch.Tensor."
    from .. import ops
    return ops.lie_small(2, omega, 3, 3).to(omega.dtype)


def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    assert torch.is_tensor(xi), "Input must be of type torch.tensor."
    from .. import ops
    return ops.se3_exp(xi.reshape(-1))
