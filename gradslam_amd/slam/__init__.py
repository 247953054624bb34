from .icpslam import *  # noqa: F401,F403
from .fusionutils import *  # noqa: F401,F403
from .pointfusion import *  # noqa: F401,F403
