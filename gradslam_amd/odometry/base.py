"""Plugin interface of the odometry stage (the reference's only plugin ABC, odometry/base.py:6-19).

An odometry provider turns a batch of map clouds and a batch of live-frame clouds into the rigid
transforms that align each frame cloud with its map cloud.  `ICPSLAM` / `PointFusion` call
`provide(maps_pointclouds, frames_pointclouds)` and expect a `(B, 1, 4, 4)` tensor on the device
of `maps_pointclouds`; custom providers may be assigned to `slam.odomprov`."""
import abc

__all__ = ["OdometryProvider"]


class OdometryProvider(abc.ABC):
    def __init__(self, *params):
        # providers keep their own hyper-parameters; nothing to initialise here
        super().__init__()

    @abc.abstractmethod
    def provide(self, *args, **kwargs):
        """Returns the (B, 1, 4, 4) transforms for the given (maps, frames) pointcloud batches."""
        raise NotImplementedError("OdometryProvider subclasses must implement provide()")

    def __call__(self, *args, **kwargs):
        return self.provide(*args, **kwargs)
