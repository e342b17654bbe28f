"""Covariance kernels behind the reference's plugin API (ref: gptools/kernel/__init__.py:23-29)."""
from .core import *            # noqa: F401,F403
from .squared_exponential import *   # noqa: F401,F403
from .matern import *          # noqa: F401,F403
from .noise import *           # noqa: F401,F403
from .rational_quadratic import *   # noqa: F401,F403
