from .icp import GradICPOdometryProvider  # noqa: F401

__all__ = ["GradICPOdometryProvider"]
