"""gptools_amd -- MI355X-native implementation of gptools' covariance-build + Cholesky
log-marginal-likelihood hot path behind the reference's GaussianProcess / Kernel API.

Public names mirror the reference's top-level namespace (ref: gptools/__init__.py:23-30) for the
parts of the library that sit on that path.
"""
__version__ = "0.1.0"

from .error_handling import *      # noqa: F401,F403
from .utils import *               # noqa: F401,F403
from .kernel import *              # noqa: F401,F403
from .mean import *                # noqa: F401,F403
from .gaussian_process import *    # noqa: F401,F403
