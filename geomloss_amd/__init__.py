"""geomloss_amd — MI355X-native Sinkhorn / kernel losses behind geomloss's ``SamplesLoss`` API.

Drop-in for the hot path of jeanfeydy/geomloss 0.3.1::

    from geomloss_amd import SamplesLoss          # instead of: from geomloss import SamplesLoss
    L = SamplesLoss("sinkhorn", p=2, blur=0.05, backend="online")(x, y)

The soft-min (log-sum-exp) and kernel reductions over the implicit N x M cost matrix run as
hand-written gfx950 HIP kernels (``geomloss_amd/csrc``, C-ABI in ``include/glhip.h``).
"""

__version__ = "0.1.0"

from .samples_loss import SamplesLoss
from . import hip


# `from geomloss import ImagesBarycenter, sinkhorn_divergence` (reference __init__.py:5-7): the image / volume path.
# `geomloss_amd.sinkhorn_divergence` is the point-cloud solver MODULE (the mirror of _legacy/sinkhorn_divergence.py); it
# is also callable, and calling it runs the image-OT function of that name (sinkhorn_images.sinkhorn_divergence).
from .wasserstein_barycenter_images import ImagesBarycenter  # noqa: E402
from . import sinkhorn_divergence  # noqa: E402
from . import ot  # noqa: E402  (`from geomloss import ot`: ot.solve_sample on the same kernels)

__all__ = ["SamplesLoss", "ImagesBarycenter", "sinkhorn_divergence", "hip", "ot"]
