from .distributed import *  # noqa: F401,F403
