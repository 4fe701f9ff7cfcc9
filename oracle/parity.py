"""Test infrastructure (like everything under oracle/): the parity metric of the GPU tests, smoke() and bench.py's
checker leg -- ELEMENT-WISE relative error.

north_star: "within 1e-10 relative fp64 on emergent intensity".  `rel(a, b) < tol` holds exactly when every element
satisfies |a - b| < tol (|b| + floor) with floor = the smallest non-zero |b| of the comparison (so an element the
reference holds at 1e-6 of the maximum is still checked at 1e-10 of ITS value, and an element the reference holds at
exactly 0 -- the never-visited last site, voronoi_utils.jl:266 -- must come out below 1e-10 of the smallest
intensity there is).  The returned number prints with the max-norm ratio |a - b|.max() / |b|.max() beside it."""
import numpy as np


class Err(float):
    maxnorm = 0.0

    def __repr__(self):
        return f"{float(self):.3e} (element-wise; max-norm {self.maxnorm:.3e})"

    __str__ = __repr__


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ab = np.abs(b)
    nz = ab[ab > 0]
    floor = float(nz.min()) if nz.size else 1.0
    diff = np.abs(a - b)
    diff = np.where(np.isnan(diff), np.inf, diff)                      # a NaN on either side is a failure
    e = Err(float((diff / (ab + floor)).max()) if diff.size else 0.0)
    e.maxnorm = float(diff.max() / max(float(ab.max()), 1e-300)) if diff.size else 0.0
    return e
