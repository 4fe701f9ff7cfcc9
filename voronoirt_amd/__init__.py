"""voronoirt_amd -- MI355X-native formal solver behind VoronoiRT's driver surface.

Only what the hot path needs: the C-ABI library (csrc/ -> libvrt_hip.so), the host-side mirror
of the reference interface (api.py), seeded synthetic grids (synth.py) and the one-process-per-GPU
sharding of the angle x wavelength loop (distributed.py).  Importing this package does not
load the shared library; the first call does, and fails loudly if it is missing.
"""
from .api import (Delaunay_downII, Delaunay_upII, FormalPlan, J_lambda_voronoi, VoronoiSites,  # noqa: F401
                  direction, quadrature_directions, read_cell, read_quadrature, voro, QUADRATURE_DIR,
                  short_characteristics_batch, short_characteristics_down, short_characteristics_up,
                  RegularSolver, LineCase, Lambda_voronoi, Lambda_voronoi_host, J_lambda_voronoi_line, MultiDevicePlan)
from ._lib import VrtError  # noqa: F401

__all__ = ["Delaunay_upII", "Delaunay_downII", "FormalPlan", "J_lambda_voronoi", "VoronoiSites",
           "direction", "quadrature_directions", "read_cell", "read_quadrature", "voro", "VrtError",
           "QUADRATURE_DIR", "short_characteristics_up", "short_characteristics_down",
           "short_characteristics_batch", "RegularSolver", "LineCase", "Lambda_voronoi", "Lambda_voronoi_host",
           "J_lambda_voronoi_line", "MultiDevicePlan"]
