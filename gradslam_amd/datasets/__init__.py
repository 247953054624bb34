from . import datautils, tumutils  # noqa: F401
from .icl import ICL  # noqa: F401
from .scannet import Scannet  # noqa: F401
from .tum import TUM  # noqa: F401
