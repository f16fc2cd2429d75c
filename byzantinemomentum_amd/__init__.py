"""byzantinemomentum_amd — MI355X-native Byzantine-robust gradient aggregation hot path.

HIP kernels (csrc/, built into libbm_gar.so, C ABI in include/bm_gar.h) behind the plugin
surface of LPD-EPFL/ByzantineMomentum's `aggregators/` package.  See DESIGN.md.
"""

from . import _lib  # noqa: F401
from . import gars  # noqa: F401
from . import stats  # noqa: F401
from . import layout  # noqa: F401
from . import graphs  # noqa: F401
from .gars import (median, trmean, phocas, meamed, krum, bulyan, brute, aksel, average, cge)  # noqa: F401
from .stats import compute_avg_dev_max  # noqa: F401

__version__ = "0.1.0"
