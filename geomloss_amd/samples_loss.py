"""``SamplesLoss``: the user-facing loss layer, drop-in for ``geomloss.SamplesLoss``.

Mirror of the reference's ``_legacy/samples_loss.py``: same constructor arguments, same call forms
``L(x, y)`` / ``L(α, x, β, y)`` / ``L(l_x, α, x, l_y, β, y)``, same shape checks and error messages,
same ``backend="auto"`` heuristic and the same output shapes.  The ``online`` and ``multiscale``
backends run on the HIP kernels of this package (MI355X); ``tensorized`` is dense PyTorch.
"""

import warnings
from functools import lru_cache, partial

import torch
from torch.nn import Module

from . import hip
from .kernel_samples import kernel_multiscale, kernel_online, kernel_tensorized
from .sinkhorn_samples import sinkhorn_multiscale, sinkhorn_online, sinkhorn_tensorized

_BACKENDS = ("tensorized", "online", "multiscale")
_KERNEL_DRIVERS = dict(zip(_BACKENDS, (kernel_tensorized, kernel_online, kernel_multiscale)))

# routines[loss][backend]; "hausdorff" is wired to the kernel drivers without a kernel name,
# as in the reference (samples_loss.py:22-26), and fails there with KeyError(None).
routines = {
    "sinkhorn": dict(zip(_BACKENDS, (sinkhorn_tensorized, sinkhorn_online, sinkhorn_multiscale))),
    "hausdorff": dict(_KERNEL_DRIVERS),
}
for _name in ("energy", "gaussian", "laplacian"):
    routines[_name] = {bk: partial(fn, name=_name) for bk, fn in _KERNEL_DRIVERS.items()}


@lru_cache(maxsize=256)
def _uniform_weight(N, dtype):
    """``(torch.ones(N, dtype=dtype) / N)[0]`` as a Python float: the quotient rounded once in ``dtype`` (IEEE division for float32 /
    float64, computed in float32 and rounded for the half types — on the CPU as on the GPU), exactly representable in it."""
    return (torch.ones((), dtype=dtype) / N).item()


def _squeeze_trailing_unit(w, other, ndim_flat, msg_both, msg_w, msg_other):
    """Validates weights given as (..., N) or (..., N, 1) and flattens the trailing unit axis."""
    if w.dim() not in (ndim_flat, ndim_flat + 1):
        raise ValueError(msg_both)
    if w.dim() == ndim_flat + 1:
        if w.shape[-1] > 1:
            raise ValueError(msg_w)
        if other.shape[-1] > 1:
            raise ValueError(msg_other)
        return True
    return False


def _check_labels(l, n, letter, size_letter, cloud):
    if l is None:
        return None
    if l.dim() not in (1, 2):
        raise ValueError(
            f"Without batches, the vector of labels '{letter}' should be encoded as an ({size_letter},) or ({size_letter},1) tensor."
        )
    if l.dim() == 2:
        if l.shape[1] > 1:
            raise ValueError(
                f"Without batches, the vector of labels '{letter}' should be encoded as ({size_letter},) or ({size_letter},1) tensors."
            )
        l = l.view(-1)
    if len(l) != n:
        raise ValueError(
            f"The vector of labels '{letter}' should have the same length as the point cloud '{cloud}'."
        )
    return l


class SamplesLoss(Module):
    """Geometric loss between two (batches of) weighted point clouds.

    Args:
        loss: "sinkhorn" | "hausdorff" | "energy" | "gaussian" | "laplacian".
        p: exponent of the ground cost |x-y|^p / p of the Sinkhorn divergence (1 or 2).
        blur: finest length scale: kernel width, or eps = blur^p for Sinkhorn.
        reach: typical scale of the soft marginal constraints (None = balanced OT); rho = reach^p.
        diameter: upper bound on the cloud diameter (None = bounding-box diagonal, one host sync).
        scaling: ratio between successive blur scales of the epsilon-scaling descent.
        truncate: kernel truncation radius, in units of blur, for the multiscale backend (default 5).
        cost: dense cost function for "tensorized"; for "online"/"multiscale" one of the two built-in
            formula strings (or a (formula, function) pair for "multiscale").
        kernel: dense kernel function for the kernel losses with "tensorized".
        cluster_scale: voxel size of the coarse level of "multiscale" (None = about 2000 voxels).
        debias: Sinkhorn divergence (True) or raw entropic OT cost (False).
        potentials: return the dual potentials (F, G) instead of the loss value.
        verbose: print the multiscale schedule.
        backend: "auto" | "tensorized" | "online" | "multiscale".
    """

    def __init__(
        self, loss="sinkhorn", p=2, blur=0.05, reach=None, diameter=None, scaling=0.5, truncate=5,
        cost=None, kernel=None, cluster_scale=None, debias=True, potentials=False, verbose=False,
        backend="auto",
    ):
        super().__init__()
        self.loss = loss
        self.backend = backend
        self.p = p
        self.blur = blur
        self.reach = reach
        self.truncate = truncate
        self.diameter = diameter
        self.scaling = scaling
        self.cost = cost
        self.kernel = kernel
        self.cluster_scale = cluster_scale
        self.debias = debias
        self.potentials = potentials
        self.verbose = verbose

    # ------------------------------------------------------------------ dispatch

    def _choose_backend(self, l_x, l_y, B, N, M, D, x=None):
        backend = self.backend
        if l_x is not None or l_y is not None:
            if backend not in ("auto", "multiscale"):
                raise ValueError(
                    'Explicit cluster labels are only supported with the "auto" and "multiscale" backends.'
                )
            return "multiscale"
        if backend != "auto":
            return backend
        if M * N <= 5000**2:
            # The reference picks the dense path here ("fastest for small clouds" with KeOps' launch costs).  On a GPU, for
            # what the HIP kernels cover natively — D <= 3, built-in cost and kernel, fp32 / bf16 / fp16 points — the
            # matrix-free path is faster at every size (N = 2000: 0.4 vs 1.9 ms per loss) and needs no N x M memory.
            if (x is not None and x.is_cuda and D <= 3 and self.cost is None and self.kernel is None
                    and x.dtype != torch.float64 and hip.library_available()):
                return "online"
            return "tensorized"  # quadratic memory, fastest for small clouds on the CPU
        if D <= 3 and self.loss == "sinkhorn" and M * N > 10000**2 and self.p == 2:
            return "multiscale"  # kernel truncation pays off in low dimension
        return "online"

    def forward(self, *args):
        """Computes the loss between sampled measures; see the class docstring for the call forms."""
        l_x, α, x, l_y, β, y = self.process_args(*args)
        B, N, M, D, l_x, α, l_y, β = self.check_shapes(l_x, α, x, l_y, β, y)

        backend = self._choose_backend(l_x, l_y, B, N, M, D, x)

        if backend == "multiscale":  # single measures only
            if B == 1:
                α, x, β, y = α.squeeze(0), x.squeeze(0), β.squeeze(0), y.squeeze(0)
            elif B > 1:
                warnings.warn(
                    "The 'multiscale' backend do not support batchsize > 1. "
                    + "Using 'tensorized' instead: beware of memory overflows!"
                )
                backend = "tensorized"

        if B == 0 and backend in ("tensorized", "online"):  # these two work on batches
            α, x, β, y = α.unsqueeze(0), x.unsqueeze(0), β.unsqueeze(0), y.unsqueeze(0)

        values = routines[self.loss][backend](
            α, x, β, y,
            p=self.p, blur=self.blur, reach=self.reach, diameter=self.diameter, scaling=self.scaling,
            truncate=self.truncate, cost=self.cost, kernel=self.kernel, cluster_scale=self.cluster_scale,
            debias=self.debias, potentials=self.potentials, labels_x=l_x, labels_y=l_y, verbose=self.verbose,
            # (set by ShardedSamplesLoss on its copy: `diameter` is the all-reduced bounding box, a true bound of this shard)
            diameter_bounds_the_data=getattr(self, "_diameter_bounds_the_data", False),
        )

        if self.potentials:
            F, G = values
            return F.view_as(α), G.view_as(β)
        if backend == "multiscale":  # scalar result
            return values if B == 0 else values.view(-1)
        return values[0] if B == 0 else values  # (B,) vector of results

    # ------------------------------------------------------------------ argument handling

    def process_args(self, *args):
        if len(args) == 6:
            return args
        if len(args) == 4:
            α, x, β, y = args
            return None, α, x, None, β, y
        if len(args) == 2:
            x, y = args
            return None, self.generate_weights(x), x, None, self.generate_weights(y), y
        raise ValueError(
            "A SamplesLoss accepts two (x, y), four (α, x, β, y) or six (l_x, α, x, l_y, β, y)  arguments."
        )

    def generate_weights(self, x):
        # Uniform weights 1/N (``samples_loss.py:325-335``), created on the device of x: the reference builds them on the CPU
        # and copies (`torch.ones(N).type_as(x)`), a pageable host-to-device copy that stalls the HIP queue for ~90 ms every
        # few calls (measured: a 3.5-ms batched loss spiking to 90-190 ms, round 3, HISTORY.md).
        # One fill launch: the value is `1 / N` rounded as that division rounds it in the dtype of x (_uniform_weight).
        if x.dim() == 2:
            N = x.shape[0]
            return torch.full((N,), _uniform_weight(N, x.dtype), dtype=x.dtype, device=x.device)
        if x.dim() == 3:
            B, N, _ = x.shape
            return torch.full((B, N), _uniform_weight(N, x.dtype), dtype=x.dtype, device=x.device)
        raise ValueError("Input samples 'x' and 'y' should be encoded as (N,D) or (B,N,D) (batch) tensors.")

    def check_shapes(self, l_x, α, x, l_y, β, y):
        if α.dim() != β.dim():
            raise ValueError("Input weights 'α' and 'β' should have the same number of dimensions.")
        if x.dim() != y.dim():
            raise ValueError("Input samples 'x' and 'y' should have the same number of dimensions.")
        if x.shape[-1] != y.shape[-1]:
            raise ValueError("Input samples 'x' and 'y' should have the same last dimension.")

        if x.dim() == 2:  # single pair of measures: B = 0 flags "no batch axis"
            B = 0
            N, D = x.shape
            M, _ = y.shape
            if _squeeze_trailing_unit(
                α, β, 1,
                "Without batches, input weights 'α' and 'β' should be encoded as (N,) or (N,1) tensors.",
                "Without batches, input weights 'α' should be encoded as (N,) or (N,1) tensors.",
                "Without batches, input weights 'β' should be encoded as (M,) or (M,1) tensors.",
            ):
                α, β = α.view(-1), β.view(-1)
            l_x = _check_labels(l_x, N, "l_x", "N", "x")
            l_y = _check_labels(l_y, M, "l_y", "M", "y")
            N2, M2 = α.shape[0], β.shape[0]

        elif x.dim() == 3:  # batch of measures
            B, N, D = x.shape
            B2, M, _ = y.shape
            if B != B2:
                raise ValueError("Samples 'x' and 'y' should have the same batchsize.")
            if _squeeze_trailing_unit(
                α, β, 2,
                "With batches, input weights 'α' and 'β' should be encoded as (B,N) or (B,N,1) tensors.",
                "With batches, input weights 'α' should be encoded as (B,N) or (B,N,1) tensors.",
                "With batches, input weights 'β' should be encoded as (B,M) or (B,M,1) tensors.",
            ):
                α, β = α.squeeze(-1), β.squeeze(-1)
            if l_x is not None or l_y is not None:
                raise NotImplementedError('The "multiscale" backend has not been implemented with batches.')
            B2, N2 = α.shape
            B3, M2 = β.shape
            if B != B2:
                raise ValueError("Samples 'x' and weights 'α' should have the same batchsize.")
            if B != B3:
                raise ValueError("Samples 'y' and weights 'β' should have the same batchsize.")

        else:
            raise ValueError("Input samples 'x' and 'y' should be encoded as (N,D) or (B,N,D) (batch) tensors.")

        if N != N2:
            raise ValueError("Weights 'α' and samples 'x' should have compatible shapes.")
        if M != M2:
            raise ValueError("Weights 'β' and samples 'y' should have compatible shapes.")

        return B, N, M, D, l_x, α, l_y, β
