from . import biquads, linearfilter  # noqa: F401
