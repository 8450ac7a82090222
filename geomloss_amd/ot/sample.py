"""``ot.solve_sample``: optimal transport between point clouds on the HIP kernels.

Mirror of the reference's ``ot/_implementations/sample.py`` (``softmin_sample :91-180``, ``solve_sample :190-395``,
``OTResultSample :447-639``) and of the parts of ``ot/_ot_result.py`` (``LinearOperator :7-160``, ``OTResult :164-454``)
that a sample-based result needs: same arguments, same checks and error messages, same attributes.

What differs, by construction:

* The reference evaluates the cost matrix lazily with KeOps (or densely without it); here ``C(x_i, y_j) = |x_i - y_j|^2``
  only ever exists inside ``glhip_softmin_fwd`` / ``glhip_sinkhorn_iter4``.  ``plan`` / ``density`` still materialise an
  (N, M) array because that is what the attribute *is*; ``lazy_plan``, ``density_operator``, ``plan_operator`` and the
  marginals are matrix-free (log-domain soft-min reductions) and work at any size.
* The kernels compute in fp32 on the GPU.  NumPy arrays, CPU tensors and float64 inputs are accepted like in the
  reference and every result is cast back to the caller's library / dtype / device, but the arithmetic is fp32 on
  ``cuda:0`` (or the device of the inputs).  There is no CPU path: without a GPU or without the HIP extension the
  solver raises.
* Gradients flow through the first cloud of each cost (``X_a`` for f, ``X_b`` for g), as in the legacy API; the
  reference's dense cost matrices also carry the (envelope-theorem-redundant) dependence on the second cloud.
"""

from functools import cached_property
from typing import NamedTuple

import numpy as np
import torch

from .. import hip
from .sinkhorn_ot import SampleCost, annealing_parameters, max_diameter, sinkhorn_cost, sinkhorn_loop


class ArrayProperties(NamedTuple):
    B: int
    N: int
    M: int
    dtype: object
    device: object
    library: str


# ---------------------------------------------------------------------------------------------------------------------
#  argument checks (``_arguments.py``) and conversions (``_backends``, ``_input_validation/converters.py``)
# ---------------------------------------------------------------------------------------------------------------------

def _library(x):
    if isinstance(x, (np.ndarray, np.generic)):
        return "numpy"
    if isinstance(x, torch.Tensor):
        return "torch"
    raise ValueError(f"Expected a NumPy array, a PyTorch tensor or a KeOps LazyTensor, but found {x} of type {type(x)}.")


def _convert(v):
    """Lists and tuples become float64 NumPy arrays (``converters.py:33-35``)."""
    return np.array(v, dtype=np.float64) if isinstance(v, (list, tuple)) else v


def check_regularization(*, reg, unbalanced, unbalanced_type, method, tol, max_iter):
    if reg < 0:
        raise ValueError(f"Parameter 'reg' should be >= 0. Received {reg}.")
    elif reg == 0:
        raise NotImplementedError("Currently, we require that reg > 0.")
    if unbalanced is not None and unbalanced <= 0:
        raise ValueError("Parameter 'unbalanced' should be None (= +infty) " f"or > 0. Received {unbalanced}.")
    if unbalanced_type != "KL":
        raise NotImplementedError("Currently, we only support unbalanced OT with " "a 'KL' penalty on the marginal constraints.")
    if method != "auto":
        raise NotImplementedError("Currently, we only support a single method.")
    if max_iter is None:
        raise ValueError("The 'max_iter' parameter should be a positive integer.")
    if tol is not None:
        raise NotImplementedError("Currently, we do not support rigorous stopping criteria.")


def _single(values, what, fix):
    values = list(set(values))
    if len(values) > 1:
        raise ValueError(f"The input arrays {what}: received a collection of {values}, which is ambiguous. {fix}")
    return values[0]


def check_library_dtype_device(*args):
    library = _single([_library(a) for a in args], "do not come from the same tensor library",
                      "To fix this error, please cast all arrays using a single library.")
    dtype = _single([a.dtype for a in args], "do not have the same numerical dtype",
                    "To fix this error, please cast all arrays to the same numerical dtype.")
    device = _single([a.device if library == "torch" else "cpu" for a in args], "are not stored on the same device",
                     "To fix this error, please move all arrays to the same RAM or GPU device.")
    return library, dtype, device


def check_marginal(m, *, ones_like, marginal_size, name):
    if m is None:
        m = (torch.ones_like(ones_like) if isinstance(ones_like, torch.Tensor) else np.ones_like(ones_like)) / marginal_size
    if m.shape != ones_like.shape:
        raise ValueError(f"The marginal '{name}' should be of shape {ones_like.shape}. "
                         f"Instead, received an array of shape {m.shape}.")
    if (m < 0).any():
        raise ValueError(f"The marginal '{name}' contains negative values. " f"We require that {name} >= 0.")
    return m


def check_marginal_masses(sum_a, sum_b, rtol=1e-3):
    if abs(sum_a - sum_b) / (sum_a + sum_b) > rtol:
        raise ValueError(
            "The two arrays of marginal weights 'a' and 'b' do not sum up to the same value."
            "As a consequence, the balanced OT problem is not feasible. "
            "To fix this error, you may either normalize the two marginals ",
            "to make sure that their weights sum up to compatible values "
            "(= 1 for probability distributions), or use UNbalanced optimal "
            "transport with the 'unbalanced' keyword argument.",
        )


def _compute_device(device, library):
    if library == "torch" and device.type == "cuda":
        return device
    if not torch.cuda.is_available():
        raise RuntimeError("geomloss_amd.ot.solve_sample runs on the HIP kernels only and found no GPU "
                           "(torch.cuda.is_available() is False).  There is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device())


def _to_gpu(v, dev):
    t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
    return t.to(device=dev, dtype=torch.float32)


def stable_log(a):
    """log(a) with log(0) = -100000 (``_backends/torch.py:24-28``)."""
    return torch.where(a > 0, a.log(), torch.full_like(a, -100000.0))


# ---------------------------------------------------------------------------------------------------------------------
#  the soft-min of the reference API
# ---------------------------------------------------------------------------------------------------------------------

def softmin_sample(eps, log_weights, costs, potentials):
    """``f_i = -eps log sum_j exp(log_b_j + (g_j - |x_i - y_j|^2) / eps)`` (``sample.py:91-180``) on the GPU.

    ``costs`` is a :class:`SampleCost`-like pair ``(x, y)`` of clouds instead of a LazyTensor.  ``eps = inf`` is the weighted
    mean of ``C - g`` (three moments of the column cloud), ``eps = 0`` the hard C-transform ``min_j (C_ij - g_j)``
    (``glhip_cmin_fwd``; not reachable from ``solve_sample``, whose argument check refuses ``reg = 0`` like the reference's).
    """
    x, y = (costs.x, costs.yd) if isinstance(costs, SampleCost) else costs
    eps = float(eps)
    assert eps >= 0, "We only support non-negative temperatures (eps >= 0)."
    if eps == float("inf"):
        from .sinkhorn_ot import _mean_cost
        b = log_weights.exp()
        return 2.0 * _mean_cost(x, y, b) - (potentials * b).sum() / b.sum()
    if eps == 0:     # hard C-transform: min_j (|x_i - y_j|^2 - g_j) = 2 min_j (|x_i - y_j|^2 / 2 - g_j / 2)   (glhip_cmin_fwd)
        return 2.0 * hip.cmin(x, y, 0.5 * potentials, p=2)
    return 2.0 * hip.softmin(eps / 2, x, y.detach(), (log_weights + potentials / eps).detach())


# ---------------------------------------------------------------------------------------------------------------------
#  solver
# ---------------------------------------------------------------------------------------------------------------------

def solve_sample(X_a, X_b, a=None, b=None, cost="sqeuclidean", debias=False, reg=None, unbalanced=None,
                 unbalanced_type="KL", method="auto", max_iter=None, tol=None, blur=None, reach=None):
    """Solves an optimal transport problem between two point clouds (``sample.py:190-395``).

    ``X_a`` (N,D), ``X_b`` (M,D), optional weights ``a`` (N,), ``b`` (M,); ``reg`` = entropic temperature for the cost
    ``|x-y|^2`` (or ``blur`` with ``reg = 2 blur^2``), ``unbalanced`` = marginal penalty (or ``reach``), ``max_iter``
    annealing steps.  Returns an :class:`OTResultSample`."""
    X_a, X_b, a, b = (_convert(v) for v in (X_a, X_b, a, b))
    if cost != "sqeuclidean":
        raise NotImplementedError()      # sample.py:84-85: the only cost the reference implements
    p = 2
    if blur is not None:
        if reg is not None:
            raise ValueError("Parameters 'reg' and 'blur' are redundant. Please specify only one of them.")
        reg = p * (blur**p)
    if reach is not None:
        if unbalanced is not None:
            raise ValueError("Parameters 'unbalanced' and 'reach' are redundant. Please specify only one of them.")
        unbalanced = p * (reach**p)
    check_regularization(reg=reg, unbalanced=unbalanced, unbalanced_type=unbalanced_type, method=method, tol=tol,
                         max_iter=max_iter)

    if len(X_a.shape) != 2:
        raise ValueError(f"Expected X_a to be a (N, D) array. Received {X_a.shape}.")
    if len(X_b.shape) != 2:
        raise ValueError(f"Expected X_b to be a (M, D) array. Received {X_b.shape}.")
    N, D = X_a.shape
    M, D_ = X_b.shape
    if D != D_:
        raise ValueError(f"Expected X_a and X_b to have the same number of coordinates per sample. "
                         f"Received D={D} for X_a and D={D_} for X_b.")
    a = check_marginal(a, ones_like=X_a[:, 0], marginal_size=N, name="a")
    b = check_marginal(b, ones_like=X_b[:, 0], marginal_size=M, name="b")
    if unbalanced is None:
        check_marginal_masses(float(a.sum()), float(b.sum()))
    library, dtype, device = check_library_dtype_device(X_a, X_b, a, b)
    props = ArrayProperties(B=0, N=N, M=M, dtype=dtype, device=device, library=library)

    hip.load_library()                                   # raises if the extension is not built
    dev = _compute_device(device, library)
    x, y, a32, b32 = (_to_gpu(v, dev) for v in (X_a, X_b, a, b))
    descent = annealing_parameters(maxmin_cost=max_diameter(x.detach(), y.detach()) ** p, eps=reg, rho=unbalanced,
                                   n_iter=max_iter)
    costs = SampleCost(x, y)
    potentials = sinkhorn_loop(cost=costs, log_a=stable_log(a32.detach()), log_b=stable_log(b32.detach()), descent=descent,
                               debias=debias, last_extrapolation=True)
    return OTResultSample(X_a=x, X_b=y, a=a32, b=b32, cost=cost, reg=reg, reg_type="KL", unbalanced=unbalanced,
                          unbalanced_type=unbalanced_type, debias=debias, potentials=potentials, array_properties=props)


def solve_sample_batch(*args, **kwargs):
    raise NotImplementedError("This function is not implemented yet.")       # sample.py:404-429


# ---------------------------------------------------------------------------------------------------------------------
#  results
# ---------------------------------------------------------------------------------------------------------------------

class LinearOperator:
    """Matrix-free linear map with a transpose (``_ot_result.py:7-160``): ``op @ v``, ``op.T``, ``op.shape``."""

    def __init__(self, *, matmat, rmatmat, input_shape, output_shape):
        self.matmat, self.rmatmat = matmat, rmatmat
        self.input_shape, self.output_shape = tuple(input_shape), tuple(output_shape)

    def __matmul__(self, x):
        vector = tuple(x.shape) == self.input_shape
        if vector:
            x = x[..., None]
        elif tuple(x.shape[:-1]) != self.input_shape:
            raise ValueError(f"Expected an input of shape {self.input_shape} or {self.input_shape + ('V',)}, got {tuple(x.shape)}.")
        out = self.matmat(x)
        return out[..., 0] if vector else out

    @property
    def shape(self):
        return self.output_shape + self.input_shape

    def transpose(self):
        return LinearOperator(matmat=self.rmatmat, rmatmat=self.matmat, input_shape=self.output_shape,
                              output_shape=self.input_shape)

    @property
    def T(self):
        return self.transpose()

    def rescale(self, *, input_scaling, output_scaling):
        """diag(output_scaling) @ self @ diag(input_scaling)."""
        a, b = output_scaling, input_scaling
        return LinearOperator(matmat=lambda s: a[..., None] * self.matmat(b[..., None] * s),
                              rmatmat=lambda s: b[..., None] * self.rmatmat(a[..., None] * s),
                              input_shape=self.input_shape, output_shape=self.output_shape)


class OTResultSample:
    """Result of :func:`solve_sample` (``sample.py:447-639`` + ``_ot_result.py:164-454``): lazily computed, cached attributes
    ``value``, ``potential_a/b/aa/bb``, ``plan``, ``density``, ``lazy_plan``, ``lazy_density``, ``density_operator``,
    ``plan_operator``, ``marginal_a/b``, ``a_to_b``, ``b_to_a``, ``citation``.  Arrays come back in the library, dtype and
    device of the solver's inputs; operators act on fp32 GPU tensors."""

    def __init__(self, *, X_a, X_b, a, b, cost, reg, reg_type, unbalanced, unbalanced_type, debias, potentials,
                 array_properties):
        self._X_a, self._X_b, self._a, self._b, self._cost = X_a, X_b, a, b, cost
        self._reg, self._reg_type, self._unbalanced, self._unbalanced_type = reg, reg_type, unbalanced, unbalanced_type
        self._debias, self._potentials, self._array_properties = debias, potentials, array_properties
        ap = array_properties
        self._shapes = {"a": (ap.N,), "b": (ap.M,), "C": (ap.N, ap.M), "B": ()}

    def cast(self, x, shape):
        """To the caller's library / dtype / device, with the documented shape (``_backends/__init__.py:47-63``)."""
        ap = self._array_properties
        x = x.reshape(self._shapes[shape])
        if ap.library == "numpy":
            return x.detach().cpu().numpy().astype(ap.dtype)
        return x.to(dtype=ap.dtype, device=ap.device)

    # dual potentials ------------------------------------------------------------------------------------------------
    @cached_property
    def potential_a(self):
        return self.cast(self._potentials.f_ba, "a")

    @cached_property
    def potential_b(self):
        return self.cast(self._potentials.g_ab, "b")

    @cached_property
    def potential_aa(self):
        if self._potentials.f_aa is None:
            raise ValueError("The self-interaction potential `f_aa` is not defined. "
                             "To fix this issue, run your OT solver with `debias = True`.")
        return self.cast(self._potentials.f_aa, "a")

    @cached_property
    def potential_bb(self):
        if self._potentials.g_bb is None:
            raise ValueError("The self-interaction potential `g_bb` is not defined. "
                             "To fix this issue, run your OT solver with `debias = True`.")
        return self.cast(self._potentials.g_bb, "b")

    # transport plan -------------------------------------------------------------------------------------------------
    def _check_kl(self):
        if self._reg_type != "KL":
            raise NotImplementedError("Currently, we only support 'KL' " "as regularization for the OT problem.")
        assert self._reg > 0

    @cached_property
    def density(self):
        """exp((f_i + g_j - C_ij) / eps) as a dense (N, M) array (``sample.py:517-560``)."""
        self._check_kl()
        f, g = self._potentials.f_ba.detach().double(), self._potentials.g_ab.detach().double()
        x, y = self._X_a.detach().double(), self._X_b.detach().double()
        C = (x * x).sum(1)[:, None] - 2.0 * (x @ y.t()) + (y * y).sum(1)[None, :]      # sample.py:55-60 (dense branch)
        return self.cast(((f[:, None] + g[None, :] - C) / self._reg).exp(), "C")

    @cached_property
    def plan(self):
        """a_i b_j density_ij (``sample.py:603-617``)."""
        ap = self._array_properties
        d = self.density
        a, b = self.cast(self._a, "a"), self.cast(self._b, "b")
        return d * a[:, None] * b[None, :]

    def _apply_density(self, rows, cols, f_rows, g_cols, s):
        """sum_j exp((f_i + g_j - C_ij)/eps) s_jv for s (M, V), through soft-min reductions in the log domain: positive and
        negative parts of every column of s are reduced separately, so no exponential of a raw potential is ever formed."""
        eps = self._reg
        s = s.to(device=rows.device, dtype=torch.float32)
        out = torch.zeros((rows.shape[0], s.shape[1]), dtype=torch.float32, device=rows.device)
        for v in range(s.shape[1]):
            for sign in (1.0, -1.0):
                part = (sign * s[:, v]).clamp_min(0)
                if not bool((part > 0).any()):
                    continue
                h = stable_log(part) + g_cols / eps
                sm = 2.0 * hip.softmin(eps / 2, rows.detach(), cols.detach(), h.detach())    # -eps log sum_j exp(h_j - C_ij/eps)
                out[:, v] += sign * ((f_rows - sm) / eps).exp()
        return out

    @cached_property
    def density_operator(self):
        """The density as a matrix-free :class:`LinearOperator` (``sample.py:590-601``; no dense matrix is formed)."""
        self._check_kl()
        x, y, f, g = self._X_a, self._X_b, self._potentials.f_ba.detach(), self._potentials.g_ab.detach()
        return LinearOperator(matmat=lambda s: self._apply_density(x, y, f, g, s),
                              rmatmat=lambda s: self._apply_density(y, x, g, f, s),
                              input_shape=self._shapes["b"], output_shape=self._shapes["a"])

    @cached_property
    def lazy_density(self):
        """Symbolic density: the reference returns a KeOps LazyTensor (``sample.py:562-588``), we return the operator."""
        return self.density_operator

    @cached_property
    def plan_operator(self):
        return self.density_operator.rescale(input_scaling=self._b.detach(), output_scaling=self._a.detach())

    @cached_property
    def lazy_plan(self):
        return self.plan_operator

    # values ---------------------------------------------------------------------------------------------------------
    @cached_property
    def value(self):
        if self._reg_type != "KL":
            raise NotImplementedError("Currently, we only support 'KL' " "as regularization for the OT problem.")
        if self._unbalanced_type != "KL":
            raise NotImplementedError("Currently, we only support 'KL' " "as regularization for the marginal constraints.")
        v = sinkhorn_cost(a=self._a, b=self._b, potentials=self._potentials, eps=self._reg, rho=self._unbalanced,
                          debias=self._debias)
        return self.cast(v, "B")

    @cached_property
    def marginal_a(self):
        """a_i (density @ b)_i (``_ot_result.py:388-396``)."""
        return self.cast(self._a.detach() * (self.density_operator @ self._b.detach()), "a")

    @cached_property
    def marginal_b(self):
        return self.cast(self._b.detach() * (self.density_operator.T @ self._a.detach()), "b")

    @property
    def a_to_b(self):
        return None          # _ot_result.py:412-417

    @property
    def b_to_a(self):
        return None

    @property
    def citation(self):
        return ("@inproceedings{feydy2019interpolating, title={Interpolating between optimal transport and MMD using "
                "Sinkhorn divergences}, author={Feydy, Jean and S{\\'e}journ{\\'e}, Thibault and Vialard, Fran{\\c{c}}ois-Xavier "
                "and Amari, Shun-ichi and Trouve, Alain and Peyr{\\'e}, Gabriel}, booktitle={AISTATS}, year={2019}}")
