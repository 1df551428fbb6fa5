from . import infer_tool  # noqa: F401
