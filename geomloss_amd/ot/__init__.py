"""``geomloss_amd.ot`` — the sample-based solver of the reference's ``geomloss.ot`` API on the HIP kernels
(SURVEY §8f N3).  Only the path that reaches the point-cloud reductions is built: ``solve_sample`` and its result type.
The matrix / grid solvers and barycenters of ``geomloss.ot`` are outside the hot-path scope and raise."""

from .sample import LinearOperator, OTResultSample, solve_sample, solve_sample_batch, softmin_sample
from .sinkhorn_ot import annealing_parameters, max_diameter, sinkhorn_cost, sinkhorn_loop

OTResult = OTResultSample


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f"geomloss_amd.ot.{name} is not part of this package: only the point-cloud solver "
                                  "`solve_sample` runs on the HIP kernels (use the reference's geomloss.ot for the rest).")
    fn.__name__ = name
    return fn


solve, solve_batch, barycenter = _out_of_scope("solve"), _out_of_scope("solve_batch"), _out_of_scope("barycenter")
solve_grid, barycenter_grid, barycenter_sample = (_out_of_scope("solve_grid"), _out_of_scope("barycenter_grid"),
                                                  _out_of_scope("barycenter_sample"))

__all__ = sorted(["solve", "solve_batch", "barycenter", "solve_sample", "solve_sample_batch", "barycenter_sample",
                  "OTResultSample", "solve_grid", "barycenter_grid", "OTResult", "LinearOperator"])
