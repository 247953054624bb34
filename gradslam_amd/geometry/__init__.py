from .projutils import *  # noqa: F401,F403
