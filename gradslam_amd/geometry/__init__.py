from .projutils import *  # noqa: F401,F403
from .geometryutils import *  # noqa: F401,F403
from .se3utils import *  # noqa: F401,F403
