"""ctypes binding of the CPU oracle (oracle/vrt_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (voronoirt_amd) never imports this module.

All arrays keep the reference's Julia (column-major, 1-based) conventions, see the header of
vrt_oracle.c.  numpy arrays passed in are converted to contiguous float64/int64 buffers that
have exactly Julia's memory layout:
  positions  -> shape (n, 3) C-order  == Julia (3, n) column-major
  neighbours -> shape (D+1, n) C-order == Julia (n, D+1) column-major
  lines      -> shape (n, D, 3) C-order == Julia (3, D, n) column-major
  S, J       -> shape (n, nlam) C-order == Julia (nlam, n) column-major
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvrt_oracle.so")

_c_i64 = ctypes.c_int64
_c_dbl = ctypes.c_double
_p_i64 = ctypes.POINTER(ctypes.c_int64)
_p_dbl = ctypes.POINTER(ctypes.c_double)
_p_i32 = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile) if the .so is missing or stale."""
    src = os.path.join(_HERE, "vrt_oracle.c")
    src2 = os.path.join(_HERE, "vrt_oracle_physics.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src),
                                                                                   os.path.getmtime(src2)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvrt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_read_neighbours.argtypes = [ctypes.c_char_p, _c_i64, _c_i64, _p_i64, _p_i64]
        L.orc_read_neighbours.restype = ctypes.c_int
        L.orc_sort_by_layer.argtypes = [_p_i64, _c_i64, _c_i64, _p_i64]
        L.orc_sort_by_layer.restype = _c_i64
        L.orc_sortperm_stable.argtypes = [_p_i64, _c_i64, _p_i64]
        L.orc_sortperm_stable.restype = None
        L.orc_reduce_layers.argtypes = [_p_i64, _c_i64, _p_i64]
        L.orc_reduce_layers.restype = _c_i64
        L.orc_delaunay_lines.argtypes = [_p_dbl, _p_i64, _c_i64, _c_i64, _c_dbl, _c_dbl, _c_dbl,
                                         _c_dbl, _p_dbl]
        L.orc_delaunay_lines.restype = None
        L.orc_smallest_angle.argtypes = [_c_i64, _p_i64, _c_i64, _c_i64, _p_dbl, _p_dbl, _p_dbl,
                                         _p_i64]
        L.orc_smallest_angle.restype = ctypes.c_int
        L.orc_linear_weights.argtypes = [_c_dbl, _p_dbl, _p_dbl, _p_dbl]
        L.orc_linear_weights.restype = None
        L.orc_delaunay.argtypes = [ctypes.c_int, _p_dbl, _p_dbl, _p_dbl, _p_dbl, _p_dbl, _p_i64,
                                   _c_i64, _c_i64, _p_dbl, _p_i64, _c_i64, _p_i64, _c_i64, _p_dbl]
        L.orc_delaunay.restype = ctypes.c_int
        L.orc_upwind_table.argtypes = [_p_dbl, _p_dbl, _p_i64, _c_i64, _c_i64, _p_dbl, _p_i64,
                                       _p_dbl, _p_dbl, _p_dbl, _p_i32]
        L.orc_upwind_table.restype = None
        L.orc_direction.argtypes = [_c_dbl, _c_dbl, _p_dbl]
        L.orc_direction.restype = None
        L.orc_J_voronoi.argtypes = [_c_i64, _p_dbl, _p_dbl, _p_dbl, _c_i64, _p_dbl, _p_dbl,
                                    ctypes.c_int, _p_dbl, _p_dbl, _p_dbl, _p_i64, _c_i64, _c_i64,
                                    _p_dbl, _p_i64, _c_i64, _p_i64, _p_i64, _c_i64, _p_i64,
                                    _c_i64, ctypes.c_int, _p_dbl]
        L.orc_J_voronoi.restype = ctypes.c_int
        L.orc_short_characteristics.argtypes = [ctypes.c_int, _p_dbl, _p_dbl, _p_dbl, _p_dbl, _p_dbl,
                                                _p_dbl, _p_dbl, _c_i64, _c_i64, _c_i64, _c_i64, _p_dbl,
                                                ctypes.POINTER(ctypes.c_int)]
        L.orc_short_characteristics.restype = ctypes.c_int
        L.orc_humlicek_w4.argtypes = [_c_dbl, _c_dbl, _p_dbl, _p_dbl]
        L.orc_humlicek_w4.restype = None
        L.orc_voigt_profile.argtypes = [_c_dbl, _c_dbl, _c_dbl]
        L.orc_voigt_profile.restype = _c_dbl
        L.orc_line_opacity.argtypes = [_p_dbl, _c_i64, _c_i64, _p_dbl, _c_dbl, _c_dbl, _p_dbl, _p_dbl, _p_dbl,
                                       _p_dbl, _p_dbl, _p_dbl]
        L.orc_line_opacity.restype = None
        L.orc_calculate_R.argtypes = [_c_i64, _c_i64, _p_dbl, _p_i64, _p_dbl, _p_dbl, _c_dbl, _c_dbl, _p_dbl, _p_dbl,
                                      _c_dbl, _p_dbl, _p_dbl, _p_dbl, _p_dbl, _c_dbl, _c_dbl, _c_dbl, _p_dbl]
        L.orc_calculate_R.restype = None
        L.orc_revised_populations.argtypes = [_c_i64, _p_dbl, _p_dbl, _p_dbl, _p_dbl]
        L.orc_revised_populations.restype = None
        L.orc_line_terms.argtypes = [_c_i64, _p_dbl, _p_dbl, _p_dbl, _c_dbl, _c_dbl, _c_dbl, _p_dbl, _p_dbl]
        L.orc_line_terms.restype = None
        L.orc_max_threads.argtypes = []
        L.orc_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_p_dbl) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_p_i64) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


@dataclass
class OracleSites:
    """The reference's `VoronoiSites` (src/voronoi_utils.jl:7-28), ray-tracing fields only."""
    positions: np.ndarray      # (n, 3) rows = site, cols (z, x, y)
    neighbours: np.ndarray     # (D+1, n) int64, row 0 = count; 1-based ids, <=0 walls
    delaunay_lines: np.ndarray  # (n, D, 3)
    layers_up: np.ndarray      # reduced offsets, 1-based (length L_up+1)
    layers_down: np.ndarray
    perm_up: np.ndarray        # 1-based
    perm_down: np.ndarray
    bounds: tuple              # (z_min, z_max, x_min, x_max, y_min, y_max)
    n: int

    @property
    def D(self):
        return self.neighbours.shape[0] - 1


def read_neighbours(fname: str, n_sites: int, max_guess: int = 70) -> np.ndarray:
    """File-parsing half of read_cell (voronoi_utils.jl:42-70); returns the trimmed matrix."""
    M = np.zeros((max_guess + 1, n_sites), dtype=np.int64)
    mx = _c_i64(0)
    rc = lib().orc_read_neighbours(fname.encode(), n_sites, max_guess, _i(M), ctypes.byref(mx))
    if rc:
        raise RuntimeError(f"orc_read_neighbours failed rc={rc}")
    return np.ascontiguousarray(M[: mx.value + 1])


def sort_by_layer(neighbours: np.ndarray, boundary: int) -> np.ndarray:
    nbr = _i64(neighbours)
    n = nbr.shape[1]
    layers = np.zeros(n, dtype=np.int64)
    L = lib().orc_sort_by_layer(_i(nbr), n, boundary, _i(layers))
    if L < 0:
        raise RuntimeError("unreachable site in layering (reference would loop forever)")
    return layers


def sortperm(layers: np.ndarray) -> np.ndarray:
    layers = _i64(layers)
    perm = np.zeros(layers.size, dtype=np.int64)
    lib().orc_sortperm_stable(_i(layers), layers.size, _i(perm))
    return perm


def reduce_layers(sorted_layers: np.ndarray) -> np.ndarray:
    s = _i64(sorted_layers)
    out = np.zeros(int(s.max()) + 1, dtype=np.int64)
    ln = lib().orc_reduce_layers(_i(s), s.size, _i(out))
    return out[:ln]


def delaunay_lines(positions, neighbours, x_min, x_max, y_min, y_max) -> np.ndarray:
    pos = _f64(positions)
    nbr = _i64(neighbours)
    n = pos.shape[0]
    D = nbr.shape[0] - 1
    lines = np.full((n, D, 3), np.nan)
    lib().orc_delaunay_lines(_d(pos), _i(nbr), n, D, x_min, x_max, y_min, y_max, _d(lines))
    return lines


def make_sites(positions, neighbours, bounds) -> OracleSites:
    """read_cell after parsing (voronoi_utils.jl:70-84): layers, stable perm, reduced offsets,
    Delaunay lines."""
    pos = _f64(positions)
    nbr = _i64(neighbours)
    n = pos.shape[0]
    z_min, z_max, x_min, x_max, y_min, y_max = bounds
    lu = sort_by_layer(nbr, -5)
    pu = sortperm(lu)
    ru = reduce_layers(lu[pu - 1])
    ld = sort_by_layer(nbr, -6)
    pd = sortperm(ld)
    rd = reduce_layers(ld[pd - 1])
    lines = delaunay_lines(pos, nbr, x_min, x_max, y_min, y_max)
    return OracleSites(pos, nbr, lines, ru, rd, pu, pd, tuple(bounds), n)


def read_cell(fname: str, n_sites: int, positions, bounds) -> OracleSites:
    """read_cell (voronoi_utils.jl:36-85)."""
    return make_sites(positions, read_neighbours(fname, n_sites), bounds)


def direction(theta_deg: float, phi_deg: float) -> np.ndarray:
    k = np.zeros(3)
    lib().orc_direction(theta_deg, phi_deg, _d(k))
    return k


def linear_weights(dtau: float):
    a, b, e = _c_dbl(), _c_dbl(), _c_dbl()
    lib().orc_linear_weights(dtau, ctypes.byref(a), ctypes.byref(b), ctypes.byref(e))
    return a.value, b.value, e.value


def smallest_angle(i0: int, sites: OracleSites, k):
    """smallest_angle for 0-based site i0; returns (dots[2], idx[2] 1-based)."""
    k = _f64(k)
    dots = np.zeros(2)
    idx = np.zeros(2, dtype=np.int64)
    rc = lib().orc_smallest_angle(i0, _i(sites.neighbours), sites.n, sites.D,
                                  _d(sites.delaunay_lines), _d(k), _d(dots), _i(idx))
    return rc, dots, idx


def upwind_table(sites: OracleSites, k):
    k = _f64(k)
    n = sites.n
    up = np.zeros((n, 2), dtype=np.int64)
    dots = np.zeros((n, 2))
    w = np.zeros((n, 2))
    r = np.zeros((n, 2))
    status = np.zeros(n, dtype=np.int32)
    lib().orc_upwind_table(_d(k), _d(sites.positions), _i(sites.neighbours), n, sites.D,
                           _d(sites.delaunay_lines), _i(up), _d(dots), _d(w), _d(r),
                           status.ctypes.data_as(_p_i32))
    return up, dots, w, r, status


def _solve(direction_sign, k, S, I0, alpha, sites: OracleSites, n_sweeps):
    k = _f64(k)
    S = _f64(S)
    I0 = _f64(I0)
    alpha = _f64(alpha)
    layers, perm = (sites.layers_up, sites.perm_up) if direction_sign > 0 else \
        (sites.layers_down, sites.perm_down)
    if I0.size != layers[1] - 1:
        raise ValueError(f"I_0 has length {I0.size}, boundary layer has {layers[1] - 1} sites")
    out = np.zeros(sites.n)
    rc = lib().orc_delaunay(direction_sign, _d(k), _d(S), _d(I0), _d(alpha), _d(sites.positions),
                            _i(sites.neighbours), sites.n, sites.D, _d(sites.delaunay_lines),
                            _i(layers), layers.size, _i(perm), n_sweeps, _d(out))
    if rc:
        raise RuntimeError("site without upwind neighbour")
    return out


def Delaunay_upII(k, S, I_0, alpha, sites: OracleSites, n_sweeps: int = 3):
    """src/irregular_ray_tracing.jl:15-82"""
    return _solve(+1, k, S, I_0, alpha, sites, n_sweeps)


def Delaunay_downII(k, S, I_0, alpha, sites: OracleSites, n_sweeps: int = 3):
    """src/irregular_ray_tracing.jl:96-163"""
    return _solve(-1, k, S, I_0, alpha, sites, n_sweeps)


def J_voronoi(weights, theta, phi, S, alpha, sites: OracleSites, I0_up=None, I0_down=None,
              n_sweeps: int = 3, nthreads: int = 1, alpha_mode: int | None = None):
    """J_λ_voronoi (src/lambda_iteration.jl:60-113 / src/lambda_continuum.jl:27-56).
    S: (n, nlam).  alpha: (n,), (n, nlam) or (n_angles, n, nlam)."""
    weights = _f64(weights)
    theta = _f64(theta)
    phi = _f64(phi)
    S = _f64(S)
    if S.ndim == 1:
        S = S.reshape(-1, 1)
    n, nlam = S.shape
    alpha = _f64(alpha)
    if alpha_mode is None:
        alpha_mode = {1: 0, 2: 1, 3: 2}[alpha.ndim]
    if alpha_mode == 1 and alpha.ndim == 1:
        alpha = alpha.reshape(n, 1)
    I0u = _f64(I0_up) if I0_up is not None else None
    I0d = _f64(I0_down) if I0_down is not None else None
    J = np.zeros((n, nlam))
    rc = lib().orc_J_voronoi(weights.size, _d(weights), _d(theta), _d(phi), nlam, _d(S), _d(alpha),
                             alpha_mode, _d(I0u), _d(I0d), _d(sites.positions),
                             _i(sites.neighbours), n, sites.D, _d(sites.delaunay_lines),
                             _i(sites.layers_up), sites.layers_up.size, _i(sites.perm_up),
                             _i(sites.layers_down), sites.layers_down.size, _i(sites.perm_down),
                             n_sweeps, nthreads, _d(J))
    if rc:
        raise RuntimeError("site without upwind neighbour")
    return J


def max_threads() -> int:
    return lib().orc_max_threads()


def _short_characteristics(up, k, S_0, I_0, alpha, z, x, y, n_sweeps, return_planes=False):
    """Regular-grid solver.  Arrays in numpy (ny, nx, nz) C-order == Julia (nz, nx, ny); I_0 in
    (ny, nx) C-order == Julia (nx, ny)."""
    k = _f64(k)
    S_0 = _f64(S_0)
    alpha = _f64(alpha)
    I_0 = _f64(I_0)
    z, x, y = _f64(z), _f64(x), _f64(y)
    ny, nx, nz = S_0.shape
    assert (nz, nx, ny) == (z.size, x.size, y.size) and I_0.shape == (ny, nx) and alpha.shape == S_0.shape
    out = np.zeros_like(S_0)
    kinds = np.zeros(nz, dtype=np.int32)
    rc = lib().orc_short_characteristics(1 if up else 0, _d(k), _d(S_0), _d(I_0), _d(alpha), _d(z),
                                         _d(x), _d(y), nz, nx, ny, n_sweeps, _d(out),
                                         kinds.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if rc:
        raise RuntimeError("orc_short_characteristics failed")
    return (out, kinds) if return_planes else out


def short_characteristics_up(k, S_0, I_0, alpha, z, x, y, n_sweeps=3, return_planes=False):
    """src/characteristics.jl:19-95"""
    return _short_characteristics(True, k, S_0, I_0, alpha, z, x, y, n_sweeps, return_planes)


def short_characteristics_down(k, S_0, I_0, alpha, z, x, y, n_sweeps=3, return_planes=False):
    """src/characteristics.jl:110-180"""
    return _short_characteristics(False, k, S_0, I_0, alpha, z, x, y, n_sweeps, return_planes)


# ---- physics either side of the formal solve (vrt_oracle_physics.c) --------------------------------
def humlicek_w4(x: float, y: float) -> complex:
    """Humlíček (1982) w4 approximation of the Faddeeva function w(x + i y), y >= 0."""
    re, im = _c_dbl(), _c_dbl()
    lib().orc_humlicek_w4(float(x), float(y), ctypes.byref(re), ctypes.byref(im))
    return complex(re.value, im.value)


def voigt_profile(a: float, v: float, dD: float) -> float:
    """Transparency.jl's voigt_profile(a, v, ΔλD) = H(a, v) / (sqrt(π) ΔλD)  (src/line.jl:133)."""
    return float(lib().orc_voigt_profile(float(a), float(v), float(dD)))


def line_opacity(k, lam, lambda0, c0, velocity, doppler, gamma, line_strength, alpha_cont):
    """α_tot (n, nlam) for one direction k -- lambda_iteration.jl:72-80, :89, :93-96."""
    k, lam = _f64(k), _f64(lam)
    velocity, doppler, gamma = _f64(velocity), _f64(doppler), _f64(gamma)
    line_strength, alpha_cont = _f64(line_strength), _f64(alpha_cont)
    n = doppler.size
    out = np.zeros((n, lam.size))
    lib().orc_line_opacity(_d(k), n, lam.size, _d(lam), float(lambda0), float(c0), _d(velocity), _d(doppler),
                           _d(gamma), _d(line_strength), _d(alpha_cont), _d(out))
    return out


def calculate_R(lam, blocks, J, planck2, lambda0, c0, doppler, gamma, sigma_bb_const, sigma_bf1, sigma_bf2,
                temperature, lte, hc_over_kB, pref_ij, pref_ji):
    """calculate_R (src/rates.jl:154-201); J (n, nlam), lte (3, n) C-order == Julia (n, 3); returns
    R as (n, 3, 3) C-order with R[i, c, r] == Julia R[r+1, c+1, i+1]."""
    lam, J, planck2 = _f64(lam), _f64(J), _f64(planck2)
    blocks = _i64(blocks)
    n = J.shape[0]
    R = np.zeros((n, 3, 3))
    lib().orc_calculate_R(n, lam.size, _d(lam), _i(blocks), _d(J), _d(planck2), float(lambda0), float(c0),
                          _d(_f64(doppler)), _d(_f64(gamma)), float(sigma_bb_const), _d(_f64(sigma_bf1)),
                          _d(_f64(sigma_bf2)), _d(_f64(temperature)), _d(_f64(lte)), float(hc_over_kB),
                          float(pref_ij), float(pref_ji), _d(R))
    return R


def revised_populations(R, C, atom_density):
    """get_revised_populations (src/populations.jl:191-221); returns (3, n) C-order == Julia (n, 3)."""
    R, C, atom_density = _f64(R), _f64(C), _f64(atom_density)
    n = atom_density.size
    out = np.zeros((3, n))
    lib().orc_revised_populations(n, _d(R), _d(C), _d(atom_density), _d(out))
    return out


def line_terms(gamma_static, gamma_unsold, populations, strength_const, Bij, Bji):
    """(γ_constant of the populations (3, n), src/broadening.jl:63-82 as called at lambda_iteration.jl:72-75;
    αline_λ's population factor, src/line.jl:219-225)"""
    gs, gu, pops = _f64(gamma_static), _f64(gamma_unsold), _f64(populations)
    n = gs.size
    gamma, strength = np.zeros(n), np.zeros(n)
    lib().orc_line_terms(n, _d(gs), _d(gu), _d(pops), strength_const, Bij, Bji, _d(gamma), _d(strength))
    return gamma, strength
