from . import mls, mlstaps  # noqa: F401
