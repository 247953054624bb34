from .pointclouds import *  # noqa: F401,F403
from .rgbdimages import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
from . import structutils  # noqa: F401,E402
