from .base import *  # noqa: F401,F403
from .gradicp import *  # noqa: F401,F403
from .icp import *  # noqa: F401,F403
from .groundtruth import *  # noqa: F401,F403
from . import icputils  # noqa: F401
