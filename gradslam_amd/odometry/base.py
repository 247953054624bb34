"""OdometryProvider: the reference's only plugin interface (odometry/base.py:6-19)."""
from abc import ABC, abstractmethod

__all__ = ["OdometryProvider"]


class OdometryProvider(ABC):
    r"""Base class of odometry providers: subclasses override `provide(maps_pointclouds,
    frames_pointclouds) -> (B, 1, 4, 4)`."""

    def __init__(self, *params):
        pass

    @abstractmethod
    def provide(self, *args, **kwargs):
        raise NotImplementedError
