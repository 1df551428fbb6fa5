from . import bigimage  # noqa: F401
