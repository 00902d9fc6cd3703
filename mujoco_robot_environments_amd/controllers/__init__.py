from .osc import OSC  # noqa: F401
from .min_max import MinMax  # noqa: F401
