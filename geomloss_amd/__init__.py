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


def _out_of_scope(name, ref):
    def raiser(*args, **kwargs):
        raise NotImplementedError(
            f"geomloss_amd.{name}: the grid / image path of geomloss ({ref}) is outside the scope of this package "
            "(SURVEY.md §2 rows 6b-8); only the point-cloud `SamplesLoss` hot path is implemented."
        )
    raiser.__name__ = name
    return raiser


# kept importable, like `from geomloss import ImagesBarycenter, sinkhorn_divergence` (reference __init__.py:5-7).
# `geomloss_amd.sinkhorn_divergence` is the solver module (the mirror of _legacy/sinkhorn_divergence.py); calling it
# like the reference's image-OT function of the same name raises the same out-of-scope error.
ImagesBarycenter = _out_of_scope("ImagesBarycenter", "_legacy/wasserstein_barycenter_images.py")
from . import sinkhorn_divergence  # noqa: E402

__all__ = ["SamplesLoss", "ImagesBarycenter", "sinkhorn_divergence", "hip"]
