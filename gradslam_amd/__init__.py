"""gradslam_amd — MI355X-native dense-SLAM hot path behind gradslam's Python API.

    from gradslam_amd import RGBDImages, Pointclouds
    from gradslam_amd.slam import PointFusion, ICPSLAM

Same classes, functions, argument meaning and error behaviour as gradslam v0.1.0 for the path
depth -> vertex/normal maps -> (grad)ICP odometry -> PointFusion map update; the bodies are
hand-written HIP kernels for gfx950 in gradslam_amd/csrc (C-ABI: include/gradslam_hip.h).
The HIP library is loaded on first use and there is no CPU / PyTorch fallback."""
from .version import __version__  # noqa: F401
from .geometry import *  # noqa: F401,F403
from . import odometry, slam, metrics  # noqa: F401
from .structures import *  # noqa: F401,F403
