"""Host-side mirror of the reference's interface for the formal-solve path, over the C ABI.

Same names and argument meaning as the Julia originals so that tests read like the reference's
drivers:

  read_quadrature      src/functions.jl:33-63
  VoronoiSites         src/voronoi_utils.jl:7-28      (ray-tracing fields; grid lives on the GPU)
  read_cell            src/voronoi_utils.jl:36-85
  Delaunay_upII        src/irregular_ray_tracing.jl:15-82
  Delaunay_downII      src/irregular_ray_tracing.jl:96-163
  J_lambda_voronoi     src/lambda_iteration.jl:60-113 / src/lambda_continuum.jl:27-56 (J_λ_voronoi)
  Lambda_voronoi       src/lambda_iteration.jl:205-300 (Λ_voronoi; device-resident loop)

Arrays are numpy with the reference's memory layout (see voronoirt_amd/synth.py): positions
(n, 3) [z, x, y], neighbours (D+1, n) with 1-based ids, S / alpha / J (n, nlam) wavelength
fastest.  Nothing here computes intensities on the CPU: every call goes through libvrt_hip.so
and raises `VrtError` when no HIP device is present.
"""
from __future__ import annotations

import ctypes
import os
import re
import weakref

import numpy as np

from . import _lib
from ._lib import VrtError, check

QUADRATURE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quadratures")


def _d(a):
    return a.ctypes.data_as(_lib.p_dbl) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_lib.p_i64) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def read_quadrature(fname: str):
    """weights, θ (deg), ϕ (deg), n_points -- src/functions.jl:33-63.

    The reference derives the point count from the digits after the first 'n' of the PATH
    (functions.jl:36-48), which breaks on directories containing an 'n'; here the rule is applied
    to the file's base name, and a count that disagrees with the file is an error (the reference
    would raise BoundsError or silently keep zero weights)."""
    path = fname
    if not os.path.exists(path):
        cand = os.path.join(QUADRATURE_DIR, os.path.basename(fname))
        if os.path.exists(cand):
            path = cand
    base = os.path.basename(path)
    m = re.search(r"n(\d+)", base)
    if not m:
        raise ValueError(f"cannot derive the number of quadrature points from {base!r}")
    n_points = int(m.group(1))
    rows = [ln.split() for ln in open(path) if ln.strip()]
    if len(rows) != n_points:
        raise ValueError(f"{base}: name says {n_points} points, file has {len(rows)}")
    arr = np.array([[float(v) for v in r[:3]] for r in rows])
    return arr[:, 0].copy(), arr[:, 1].copy(), arr[:, 2].copy(), n_points


def direction(theta_deg: float, phi_deg: float) -> np.ndarray:
    """k = [cos θ, cos ϕ sin θ, sin ϕ sin θ] -- src/lambda_iteration.jl:87"""
    k = np.zeros(3)
    _lib.load().vrt_direction(float(theta_deg), float(phi_deg), _d(k))
    return k


class VoronoiSites:
    """The reference's `VoronoiSites` (src/voronoi_utils.jl:7-28) backed by a device-resident
    grid handle (`vrt_grid`).  `device=-1` builds a host-only handle (layers / permutations /
    schedule introspection, no compute)."""

    def __init__(self, positions, neighbours, bounds, device: int = 0, _from_file: str | None = None):
        L = _lib.load()
        self.positions = _f64(positions)
        if self.positions.ndim != 2 or self.positions.shape[1] != 3:
            raise ValueError("positions must have shape (n, 3) with columns (z, x, y)")
        self.n = self.positions.shape[0]
        self.bounds = tuple(float(b) for b in bounds)
        self.z_min, self.z_max, self.x_min, self.x_max, self.y_min, self.y_max = self.bounds
        self.device = device
        b = np.array(self.bounds, dtype=np.float64)
        h = ctypes.c_void_p()
        if _from_file is not None:
            check(L.vrt_grid_create_from_file(_from_file.encode(), self.n, _d(self.positions),
                                              _d(b), device, ctypes.byref(h)))
            self.neighbours = None
        else:
            self.neighbours = np.ascontiguousarray(neighbours, dtype=np.int64)
            if self.neighbours.ndim != 2 or self.neighbours.shape[1] != self.n:
                raise ValueError("neighbours must have shape (D+1, n)")
            check(L.vrt_grid_create(self.n, _d(self.positions), _i(self.neighbours),
                                    self.neighbours.shape[0], _d(b), device, ctypes.byref(h)))
        self._h = h
        self.max_neighbours = int(L.vrt_grid_max_neighbours(h))
        self.layers_up = self._layers(+1)
        self.layers_down = self._layers(-1)
        self.perm_up = self._perm(+1)
        self.perm_down = self._perm(-1)
        self._plans = {}
        self._live_plans = weakref.WeakSet()    # every FormalPlan built on this grid (closed with it)
        self._options = {}                      # tuning options set on this grid (set_option)

    # -- introspection ------------------------------------------------------------------------
    def _layers(self, d):
        L = _lib.load()
        out = np.zeros(int(L.vrt_grid_num_layer_offsets(self._h, d)), dtype=np.int64)
        check(L.vrt_grid_get_layers(self._h, d, _i(out)))
        return out

    def _perm(self, d):
        out = np.zeros(self.n, dtype=np.int64)
        check(_lib.load().vrt_grid_get_perm(self._h, d, _i(out)))
        return out

    def set_option(self, name: str, value) -> None:
        """Tuning option (`vrt_grid_set_option` / `vrt_plan_set_option`, e.g. VRT_PATH = auto | levels | tiles |
        steps | patches) for every plan of this grid: the single-solve plans cached inside the handle, the
        live FormalPlans and those created later.  Environment variables of the same names are read once, at
        plan creation."""
        check(_lib.load().vrt_grid_set_option(self._h, name.encode(), str(value).encode()))
        self._options[name] = str(value)
        for p in list(self._live_plans):
            try:
                p.set_option(name, value)
            except VrtError:
                pass          # an option that shapes what plan creation builds: it holds for the plans created from now on

    def storage_order(self, d: int) -> np.ndarray:
        """1-based site id at every storage position of direction d (> 0 up, < 0 down): the site
        order of the library's native layouts (VRT_ALPHA_ANGLE_NATIVE)."""
        out = np.zeros(self.n, dtype=np.int64)
        check(_lib.load().vrt_grid_get_storage_order(self._h, d, _i(out)))
        return out

    @property
    def Delaunay_lines(self) -> np.ndarray:
        """(n, D, 3) == Julia (3, D, n); computed on the device at construction."""
        out = np.zeros((self.n, self.max_neighbours, 3))
        check(_lib.load().vrt_grid_get_delaunay_lines(self._h, _d(out)))
        return out

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # a plan holds a pointer to the grid: close every plan still alive before the grid goes,
            # also those the caller created (a later plan.close() is then a no-op)
            for p in list(self._live_plans):
                p.close()
            self._plans = {}
            _lib.load().vrt_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def voro(positions, bounds, neighbours_file: str | None = None, max_guess: int = 70) -> np.ndarray:
    """The reference's `voro` step (src/functions.jl:13-23: run the voro++ wrapper on the sites file)
    in-process: Voronoi neighbours of `positions` (n, 3) [z, x, y] in `bounds` = (z_min, z_max,
    x_min, x_max, y_min, y_max), periodic in x and y, walls -5 / -6 in z.  Returns the (D+1, n)
    matrix `read_cell` builds; with `neighbours_file` also writes the voro++ "%i %n" text file.
    Host code (no GPU needed)."""
    L = _lib.load()
    pos = _f64(positions)
    n = pos.shape[0]
    b = np.array([float(v) for v in bounds], dtype=np.float64)
    M = np.zeros((max_guess + 1, n), dtype=np.int64)
    mx = ctypes.c_int64()
    check(L.vrt_tessellate(n, _d(pos), _d(b), max_guess + 1, _i(M), ctypes.byref(mx)))
    if neighbours_file is not None:
        check(L.vrt_write_neighbours_file(neighbours_file.encode(), n, _i(M), max_guess + 1))
    return np.ascontiguousarray(M[: mx.value + 1])


def read_cell(fname: str, n_sites: int, positions, bounds, device: int = 0) -> VoronoiSites:
    """read_cell (src/voronoi_utils.jl:36-85): parse the voro++ "%i %n" neighbour file, layer the
    grid from both walls, sort, and compute the Delaunay lines.  `bounds` =
    (z_min, z_max, x_min, x_max, y_min, y_max)."""
    pos = _f64(positions)
    if pos.shape[0] != n_sites:
        raise ValueError("positions does not have n_sites rows")
    return VoronoiSites(pos, None, bounds, device=device, _from_file=fname)


class FormalPlan:
    """Upwind tables + sweep schedule for a set of directions on one grid (`vrt_plan`)."""

    def __init__(self, sites: VoronoiSites, k, n_sweeps: int = 3, dirs=None):
        L = _lib.load()
        self.sites = sites
        self.k = _f64(np.atleast_2d(k))
        if self.k.shape[1] != 3:
            raise ValueError("k must have shape (n_angles, 3)")
        self.n_angles = self.k.shape[0]
        self.n_sweeps = int(n_sweeps)
        h = ctypes.c_void_p()
        if dirs is None:
            check(L.vrt_plan_create(sites.handle, self.n_angles, _d(self.k), self.n_sweeps,
                                    ctypes.byref(h)))
        else:
            d = np.ascontiguousarray(dirs, dtype=np.int32)
            check(L.vrt_plan_create_ex(sites.handle, self.n_angles, _d(self.k),
                                       d.ctypes.data_as(_lib.p_int), self.n_sweeps, ctypes.byref(h)))
        self._h = h
        sites._live_plans.add(self)
        for name, value in sites._options.items():      # options set on the grid follow into its plans
            try:
                self.set_option(name, value)
            except VrtError:
                pass                                     # a creation-only option: the environment presets those

    @property
    def num_levels(self) -> int:
        return int(_lib.load().vrt_plan_num_levels(self._h))

    @property
    def num_nodes(self) -> int:
        return int(_lib.load().vrt_plan_num_nodes(self._h))

    def upwind(self, angle: int):
        """(up (n,2) 1-based ids, dots (n,2), weights (n,2), path lengths (n,2))"""
        n = self.sites.n
        up = np.zeros((n, 2), dtype=np.int64)
        dots = np.zeros((n, 2))
        w = np.zeros((n, 2))
        r = np.zeros((n, 2))
        check(_lib.load().vrt_plan_get_upwind(self._h, angle, _i(up), _d(dots), _d(w), _d(r)))
        return up, dots, w, r

    def execute(self, S, alpha, weights=None, I0_up=None, I0_down=None, want_J=True,
                want_I=False, alpha_mode=None):
        """Host arrays in, host arrays out.  S (n, nlam); alpha (n,), (n, nlam) or
        (n_angles, n, nlam).  Returns (J or None, I or None) with I of shape
        (n_angles, n, nlam)."""
        S = _f64(S)
        if S.ndim == 1:
            S = S.reshape(-1, 1)
        n, nlam = S.shape
        if n != self.sites.n:
            raise ValueError("S has the wrong number of sites")
        alpha = _f64(alpha)
        if alpha_mode is None:
            alpha_mode = {1: _lib.ALPHA_SITE, 2: _lib.ALPHA_SITE_LAM,
                          3: _lib.ALPHA_ANGLE_SITE_LAM}[alpha.ndim]
        want = {_lib.ALPHA_SITE: n, _lib.ALPHA_SITE_LAM: n * nlam,
                _lib.ALPHA_ANGLE_SITE_LAM: self.n_angles * n * nlam}[alpha_mode]
        if alpha.size != want:
            raise ValueError("alpha has the wrong size for its mode")
        n1u = int(self.sites.layers_up[1] - 1)
        n1d = int(self.sites.layers_down[1] - 1)
        if I0_up is not None:
            I0_up = _f64(I0_up).reshape(-1, nlam)
            if I0_up.shape[0] != n1u:
                raise ValueError(f"I0_up has {I0_up.shape[0]} rows, bottom layer has {n1u} sites")
        if I0_down is not None:
            I0_down = _f64(I0_down).reshape(-1, nlam)
            if I0_down.shape[0] != n1d:
                raise ValueError(f"I0_down has {I0_down.shape[0]} rows, top layer has {n1d} sites")
        w = _f64(weights) if weights is not None else np.ones(self.n_angles)
        if w.size != self.n_angles:
            raise ValueError("weights has the wrong length")
        J = np.zeros((n, nlam)) if want_J else None
        Iout = np.zeros((self.n_angles, n, nlam)) if want_I else None
        check(_lib.load().vrt_plan_execute(self._h, nlam, nlam, _d(S), _d(alpha), alpha_mode,
                                           _d(I0_up), _d(I0_down), _d(w), _d(J), _d(Iout)))
        return J, Iout

    def execute_dev(self, nlam: int, ld: int, dS: int, dalpha: int, alpha_mode: int, weights,
                    dJ: int = 0, dI0_up: int = 0, dI0_down: int = 0, dI_out: int = 0,
                    stream: int = 0, f32: bool = False) -> None:
        """Device pointers (ints, e.g. torch.Tensor.data_ptr()) and a hipStream_t handle
        (torch.cuda.current_stream().cuda_stream).  Asynchronous on `stream`.  f32=True: the
        buffers hold float32 values (fp32 value path, arithmetic stays fp64)."""
        w = _f64(weights)
        fn = _lib.load().vrt_plan_execute_dev_f32 if f32 else _lib.load().vrt_plan_execute_dev
        check(fn(self._h, nlam, ld, dS, dalpha, alpha_mode, dI0_up or None, dI0_down or None, _d(w),
                 dJ or None, dI_out or None, stream or None))

    # ---- S and J in sweep order (include/voronoirt.h: vrt_plan_execute_native_dev) ----
    def native_plane_count(self, nlam: int) -> int:
        """values (float64, or float32 with the f32 entry points) of ONE direction's sweep-order plane set of S or J."""
        return int(_lib.load().vrt_plan_native_plane_count(self._h, nlam))

    def to_native_dev(self, nlam: int, ld: int, d_in: int, d_up: int = 0, d_down: int = 0, stream: int = 0, f32: bool = False) -> None:
        """Device (n, ld) array -> the sweep-order plane sets of both directions (either may be 0)."""
        fn = _lib.load().vrt_plan_to_native_dev_f32 if f32 else _lib.load().vrt_plan_to_native_dev
        check(fn(self._h, nlam, ld, d_in, d_up or None, d_down or None, stream or None))

    def from_native_dev(self, d: int, nlam: int, ld: int, d_native: int, d_out: int, stream: int = 0, f32: bool = False) -> None:
        """The sweep-order plane set of direction d (> 0: up) -> a device (n, ld) array."""
        fn = _lib.load().vrt_plan_from_native_dev_f32 if f32 else _lib.load().vrt_plan_from_native_dev
        check(fn(self._h, int(d), nlam, ld, d_native, d_out, stream or None))

    def J_from_native_dev(self, nlam: int, ld: int, dJ_up: int, dJ_down: int, dJ: int, stream: int = 0, f32: bool = False) -> None:
        """J = J_up + J_down in the caller's (n, ld) layout."""
        fn = _lib.load().vrt_plan_j_from_native_dev_f32 if f32 else _lib.load().vrt_plan_j_from_native_dev
        check(fn(self._h, nlam, ld, dJ_up or None, dJ_down or None, dJ, stream or None))

    def execute_native_dev(self, nlam: int, dS_up: int, dS_down: int, dalpha: int, alpha_mode: int, weights,
                           dJ_up: int = 0, dJ_down: int = 0, dI0_up: int = 0, dI0_down: int = 0, stream: int = 0,
                           f32: bool = False) -> None:
        """`execute_dev` with S read from and J reduced into sweep-order plane sets, in place (no layout change)."""
        w = _f64(weights)
        fn = _lib.load().vrt_plan_execute_native_dev_f32 if f32 else _lib.load().vrt_plan_execute_native_dev
        check(fn(self._h, nlam, dS_up or None, dS_down or None, dalpha, alpha_mode, dI0_up or None, dI0_down or None, _d(w),
                 dJ_up or None, dJ_down or None, stream or None))

    def check(self) -> None:
        """Raises if a chained launch of an earlier ASYNCHRONOUS execute gave up (call after synchronising)."""
        check(_lib.load().vrt_plan_check(self._h))

    def native_alpha_count(self, nlam: int) -> int:
        """Number of float64 values of the native per-angle alpha buffer (ALPHA_ANGLE_NATIVE)."""
        return int(_lib.load().vrt_plan_native_alpha_count(self._h, nlam))

    @property
    def native_pair_block(self) -> int:
        """Wavelength pairs of a site kept side by side in the native layout (include/voronoirt.h)."""
        return int(_lib.load().vrt_plan_native_pair_block(self._h))

    @property
    def native_pair_block_f32(self) -> int:
        """The same for the float native buffer (fp32 value path)."""
        return int(_lib.load().vrt_plan_native_pair_block_f32(self._h))

    def native_to_site_major(self, native, nlam: int, n_angles: int):
        """Host helper (tests, debugging): a native per-angle buffer (numpy float64 or float32) ->
        (n_angles, n, nlam) with rows in STORAGE order of each angle's direction."""
        native = np.asarray(native)
        n = self.sites.n
        B = self.native_pair_block_f32 if native.dtype == np.float32 else self.native_pair_block
        npair = (nlam + 1) // 2
        per = native.reshape(n_angles, npair * n * 2)
        out = np.empty((n_angles, n, 2 * npair), dtype=native.dtype)
        q0 = 0
        widths = [B] * (npair // B) + [1 << b for b in range(B.bit_length() - 2, -1, -1) if (npair % B) & (1 << b)]
        for w in widths:
            blk = per[:, q0 * n * 2:(q0 + w) * n * 2].reshape(n_angles, n, 2 * w)
            out[:, :, 2 * q0:2 * (q0 + w)] = blk
            q0 += w
        return out[:, :, :nlam]

    def alpha_to_native_dev(self, nlam: int, ld: int, dalpha: int, dalpha_native: int, stream: int = 0,
                            f32: bool = False) -> None:
        """Device (n_angles, n, ld) per-angle alpha -> the native layout, once per change of alpha
        (f32: float32 buffers, for execute_dev(..., f32=True))."""
        fn = _lib.load().vrt_plan_alpha_to_native_dev_f32 if f32 else _lib.load().vrt_plan_alpha_to_native_dev
        check(fn(self._h, nlam, ld, dalpha, dalpha_native, stream or None))

    def line_opacity_dev(self, lam, lambda0: float, c0: float, d_velocity: int, d_doppler: int, d_gamma: int,
                         d_line_strength: int, d_alpha_cont: int, d_alpha_native: int, stream: int = 0,
                         f32: bool = False) -> None:
        """Fused opacity prologue (`vrt_line_opacity_dev[_f32]`): α_tot of every angle of this plan from
        per-site line parameters (device pointers, float64), written in the native layout (f32: stored as
        float32 for execute_dev(..., f32=True))."""
        lam = _f64(lam)
        fn = _lib.load().vrt_line_opacity_dev_f32 if f32 else _lib.load().vrt_line_opacity_dev
        check(fn(self._h, lam.size, _d(lam), float(lambda0), float(c0), d_velocity, d_doppler, d_gamma, d_line_strength,
                 d_alpha_cont, d_alpha_native, stream or None))

    def set_option(self, name: str, value) -> None:
        """Tuning option of this plan (`vrt_plan_set_option`); results never depend on it."""
        if getattr(self, "_h", None):
            check(_lib.load().vrt_plan_set_option(self._h, name.encode(), str(value).encode()))

    def last_sweep_timing(self):
        ms = ctypes.c_double()
        launches = ctypes.c_int64()
        check(_lib.load().vrt_plan_last_sweep_timing(self._h, ctypes.byref(ms),
                                                     ctypes.byref(launches)))
        return ms.value, launches.value

    @property
    def last_launches(self) -> int:
        """Kernel launches of the last execute's sweep (1: the chained patch launch); waits for it to finish and
        raises if a chained launch gave up waiting for a dependency."""
        return int(self.last_sweep_timing()[1])

    @property
    def last_path(self) -> str:
        """Device path of the last execute: "levels", "tiles" or "steps" ("" before the first)."""
        return {0: "", 1: "levels", 2: "tiles", 3: "steps", 4: "patches"}[int(_lib.load().vrt_plan_last_path(self._h))]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.load().vrt_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDevicePlan:
    """Several GPUs of a node from ONE process (`vrt_multi_*`): a grid + plan per device and an in-process RCCL
    communicator; `execute` is `FormalPlan.execute` for the node (host arrays in, J out), sharded by
    wavelength blocks when nλ >= devices, by angles (one RCCL all-reduce of J) otherwise.  Listing a device
    twice rehearses the sharding on a one-GPU box (no RCCL, partial sums added by a kernel)."""

    def __init__(self, positions, neighbours, bounds, k, dirs=None, n_sweeps: int = 3, devices=(0,)):
        L = _lib.load()
        pos = _f64(positions)
        nbr = np.ascontiguousarray(neighbours, dtype=np.int64)
        b = np.array([float(v) for v in bounds], dtype=np.float64)
        self.k = _f64(np.atleast_2d(k))
        self.n, self.n_angles = pos.shape[0], self.k.shape[0]
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        d = np.ascontiguousarray(dirs, dtype=np.int32) if dirs is not None else None
        h = ctypes.c_void_p()
        check(L.vrt_multi_create(dev.size, dev.ctypes.data_as(_lib.p_int), self.n, _d(pos), _i(nbr), nbr.shape[0], _d(b),
                                 self.n_angles, _d(self.k), d.ctypes.data_as(_lib.p_int) if d is not None else None,
                                 int(n_sweeps), ctypes.byref(h)))
        self._h = h

    def set_shard(self, mode: str) -> None:
        check(_lib.load().vrt_multi_set_shard(self._h, mode.encode()))

    @property
    def last_shard(self) -> str:
        return {0: "", 1: "lambda", 2: "angle"}[int(_lib.load().vrt_multi_last_shard(self._h))]

    @property
    def uses_rccl(self) -> bool:
        return bool(_lib.load().vrt_multi_uses_rccl(self._h))

    def execute(self, S, alpha, weights, I0_up=None, I0_down=None, alpha_mode=None) -> np.ndarray:
        S = _f64(S)
        if S.ndim == 1:
            S = S.reshape(-1, 1)
        n, nlam = S.shape
        alpha = _f64(alpha)
        if alpha_mode is None:
            alpha_mode = {1: _lib.ALPHA_SITE, 2: _lib.ALPHA_SITE_LAM, 3: _lib.ALPHA_ANGLE_SITE_LAM}[alpha.ndim]
        if I0_up is not None:
            I0_up = _f64(I0_up).reshape(-1, nlam)
        if I0_down is not None:
            I0_down = _f64(I0_down).reshape(-1, nlam)
        w = _f64(weights)
        J = np.zeros((n, nlam))
        check(_lib.load().vrt_multi_execute(self._h, nlam, nlam, _d(S), _d(alpha), alpha_mode, _d(I0_up), _d(I0_down),
                                            _d(w), _d(J)))
        return J

    def execute_line(self, S, populations, case, weights, perm_up, n1: int) -> np.ndarray:
        """`vrt_multi_execute_line`: J_λ_voronoi of the line case from host arrays, wavelength blocks over the devices,
        every device making the per-angle α_tot of its own wavelengths (`case`: a LineCase; I_0 = B_0 of the bottom
        layer, lambda_iteration.jl:99-101)."""
        S = _f64(S)
        n, nlam = S.shape
        pops = np.asarray(populations)
        gamma = _f64(case.gamma(pops))
        strength = _f64(case.strength_const * (pops[0] * case.Bij - pops[1] * case.Bji))
        lam, vel, dop, ac = _f64(case.lam), _f64(case.velocity), _f64(case.doppler), _f64(case.alpha_cont)
        I0 = _f64(np.asarray(case.B0)[np.asarray(perm_up)[:n1] - 1])
        J = np.zeros((n, nlam))
        check(_lib.load().vrt_multi_execute_line(self._h, nlam, nlam, _d(lam), float(case.lambda0), float(case.c0), _d(vel),
                                                 _d(dop), _d(gamma), _d(strength), _d(ac), _d(S), _d(I0), None,
                                                 _d(_f64(weights)), _d(J)))
        return J

    def lambda_iteration(self, eps_conv: float, maxiter: int, case, weights):
        """Λ_voronoi (src/lambda_iteration.jl:205-300) across the devices (`vrt_multi_lambda_*`): wavelength blocks per
        device, one all-reduce of the rate-integral shares per iteration.  Returns (J, S_new, populations (3, n), history)."""
        L = _lib.load()
        lc, keep = case.c_struct()
        h = ctypes.c_void_p()
        check(L.vrt_multi_lambda_create(self._h, ctypes.byref(lc), _d(_f64(weights)), ctypes.byref(h)))
        n, nlam = self.n, int(keep["lam"].size)
        history, diff, i = [], 1.0, 0
        try:
            while diff > eps_conv and i < maxiter:
                d = ctypes.c_double()
                check(L.vrt_multi_lambda_iterate(h, ctypes.byref(d)))
                diff = d.value
                history.append(diff)
                i += 1
                if diff != diff:
                    import warnings
                    warnings.warn(f"lambda_iteration: NaN DIFF! at iteration {i} -- stopping, results are not converged")
            J, S, pops = np.zeros((n, nlam)), np.zeros((n, nlam)), np.zeros((3, n))
            check(L.vrt_multi_lambda_get(h, _d(J), _d(S), _d(pops), None, None))
            if i == 0:
                S[:] = keep["B0"]
                pops[:] = keep["lte"]
            return J, S, pops, history
        finally:
            L.vrt_multi_lambda_destroy(h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.load().vrt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _single(fn_name, k, S, I_0, alpha, sites: VoronoiSites, n_sweeps: int):
    L = _lib.load()
    k = _f64(k)
    S = _f64(S)
    I_0 = _f64(I_0)
    alpha = _f64(alpha)
    if S.shape != (sites.n,) or alpha.shape != (sites.n,):
        raise ValueError("S and alpha must be vectors with one entry per site")
    out = np.zeros(sites.n)
    check(getattr(L, fn_name)(sites.handle, _d(k), _d(S), _d(I_0), I_0.size, _d(alpha),
                              int(n_sweeps), _d(out)))
    return out


def Delaunay_upII(k, S, I_0, alpha, sites: VoronoiSites, n_sweeps: int = 3) -> np.ndarray:
    """Intensity at every site for rays travelling up (src/irregular_ray_tracing.jl:15-82).
    I_0 is the boundary intensity of perm_up[1 : layers_up[2]-1]."""
    return _single("vrt_delaunay_up", k, S, I_0, alpha, sites, n_sweeps)


def Delaunay_downII(k, S, I_0, alpha, sites: VoronoiSites, n_sweeps: int = 3) -> np.ndarray:
    """Intensity at every site for rays travelling down (src/irregular_ray_tracing.jl:96-163)."""
    return _single("vrt_delaunay_down", k, S, I_0, alpha, sites, n_sweeps)


def quadrature_directions(theta, phi) -> np.ndarray:
    return np.stack([direction(t, p) for t, p in zip(theta, phi)])


def J_lambda_voronoi(S_lambda, alpha, sites: VoronoiSites, quadrature: str, I0_up=None,
                     I0_down=None, n_sweeps: int = 3) -> np.ndarray:
    """J_λ_voronoi: mean intensity J = Σ_angles w · I over a quadrature file
    (src/lambda_iteration.jl:60-113 for nλ > 1, src/lambda_continuum.jl:27-56 for the continuum).
    The angle × wavelength loop the reference threads over λ runs as one batched device solve.
    The opacity / boundary-intensity physics stays with the caller: `alpha` is α_tot (per site,
    per (site, λ) or per (angle, site, λ)), `I0_up` is B_λ(T) of the bottom layer
    (lambda_iteration.jl:99-101), `I0_down` defaults to zeros (:105-106)."""
    weights, theta, phi, _ = read_quadrature(quadrature)
    key = (os.path.basename(quadrature), int(n_sweeps))
    plan = sites._plans.get(key)
    if plan is None:
        # the reference branches on θ in degrees (lambda_iteration.jl:98,104), not on sign(k_z)
        dirs = [1 if t > 90 else (-1 if t < 90 else 0) for t in theta]
        plan = FormalPlan(sites, quadrature_directions(theta, phi), n_sweeps, dirs=dirs)
        sites._plans[key] = plan
    J, _ = plan.execute(S_lambda, alpha, weights=weights, I0_up=I0_up, I0_down=I0_down)
    return J


def build_schedule(sites: VoronoiSites, dir: int, up, n_sweeps: int = 3):
    """Host-side dependency schedule of one direction (introspection; works on a device=-1
    handle).  `up` is an (n, 2) array of 1-based upwind ids.  Returns (site ids 1-based sorted by
    level, zero-read flags, level offsets)."""
    L = _lib.load()
    up = np.ascontiguousarray(up, dtype=np.int64)
    h = ctypes.c_void_p()
    check(L.vrt_schedule_build(sites.handle, int(dir), _i(up), int(n_sweeps), ctypes.byref(h)))
    try:
        nn = int(L.vrt_schedule_num_nodes(h))
        nl = int(L.vrt_schedule_num_levels(h))
        site = np.zeros(nn, dtype=np.int64)
        z = np.zeros(nn, dtype=np.int32)
        off = np.zeros(nl + 1, dtype=np.int64)
        check(L.vrt_schedule_get(h, _i(site), z.ctypes.data_as(_lib.p_i32), _i(off)))
    finally:
        L.vrt_schedule_destroy(h)
    return site, z, off


def build_layer_schedule(sites: VoronoiSites, dir: int, up, n_sweeps: int = 3):
    """Layer-local schedule of the LDS layer-tile kernel (introspection, host only).  Returns
    (vis (n,) uint32: four packed 8-bit in-layer visit levels per site, nlev per layer (index =
    1-based layer), number of visits)."""
    L = _lib.load()
    up = np.ascontiguousarray(up, dtype=np.int64)
    vis = np.zeros(sites.n, dtype=np.uint32)
    layers = sites.layers_up if dir > 0 else sites.layers_down
    nlev = np.zeros(layers.size, dtype=np.int32)
    nv = ctypes.c_int64()
    check(L.vrt_layer_schedule(sites.handle, int(dir), _i(up), int(n_sweeps),
                               vis.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                               nlev.ctypes.data_as(_lib.p_i32), ctypes.byref(nv)))
    return vis, nlev, nv.value


def build_patch_schedule(sites: VoronoiSites, dir: int, up, n_sweeps: int = 3, own_target: int = 768,
                         entry_cap: int = 1024) -> dict:
    """Patch schedule of the fused layer kernel (introspection, host only): layers cut into ranges of
    consecutive storage positions, each with its in-layer dependency cone (own sites + halo).
    Returns a dict of numpy arrays (see vrt_patch_schedule_get) plus the counts."""
    L = _lib.load()
    up = np.ascontiguousarray(up, dtype=np.int64)
    h = ctypes.c_void_p()
    cnt = np.zeros(6, dtype=np.int64)
    check(L.vrt_patch_schedule_build(sites.handle, int(dir), _i(up), int(n_sweeps), int(own_target), int(entry_cap),
                                     ctypes.byref(h), _i(cnt)))
    try:
        P, E = int(cnt[0]), int(cnt[1])
        out = {"layer_patch_off": np.zeros(int(cnt[5]), dtype=np.int32), "patch_own_lo": np.zeros(P, dtype=np.int32),
               "patch_own_cnt": np.zeros(P, dtype=np.int32), "patch_nlev": np.zeros(P, dtype=np.int32),
               "patch_ent_off": np.zeros(P + 1, dtype=np.int64), "entry_pos": np.zeros(E, dtype=np.int32),
               "entry_vis": np.zeros(E, dtype=np.uint32), "entry_loc": np.zeros(E, dtype=np.uint32)}
        u32 = ctypes.POINTER(ctypes.c_uint32)
        check(L.vrt_patch_schedule_get(h, out["layer_patch_off"].ctypes.data_as(_lib.p_i32),
                                       out["patch_own_lo"].ctypes.data_as(_lib.p_i32),
                                       out["patch_own_cnt"].ctypes.data_as(_lib.p_i32),
                                       out["patch_nlev"].ctypes.data_as(_lib.p_i32), _i(out["patch_ent_off"]),
                                       out["entry_pos"].ctypes.data_as(_lib.p_i32),
                                       out["entry_vis"].ctypes.data_as(u32), out["entry_loc"].ctypes.data_as(u32)))
        out["dep_off"] = np.zeros(P + 1, dtype=np.int64)
        check(L.vrt_patch_schedule_get_deps(h, _i(out["dep_off"]), None))
        out["dep_list"] = np.zeros(int(out["dep_off"][-1]), dtype=np.int32)
        check(L.vrt_patch_schedule_get_deps(h, None, out["dep_list"].ctypes.data_as(_lib.p_i32)))
        # the angle's layer schedule, as the builder derives it on the way (= build_layer_schedule's outputs)
        out["layer_vis"] = np.zeros(sites.n, dtype=np.uint32)
        out["layer_nlev"] = np.zeros((sites.layers_up if dir > 0 else sites.layers_down).size, dtype=np.int32)
        nv = ctypes.c_int64()
        check(L.vrt_patch_schedule_get_layers(h, out["layer_vis"].ctypes.data_as(u32),
                                              out["layer_nlev"].ctypes.data_as(_lib.p_i32), ctypes.byref(nv)))
        out["layer_visits"] = nv.value
    finally:
        L.vrt_patch_schedule_destroy(h)
    out.update(patches=P, entries=E, visits=int(cnt[2]), live_visits=int(cnt[3]), max_entries=int(cnt[4]))
    return out


def layer_sorted_slots(sites: VoronoiSites, dir: int, vis):
    """Thread assignment of the layer-step level kernel for a layer schedule `vis` (host only):
    (store, self) = 1-based site id per storage position, 0-based storage position per sorted
    index (inside each layer sorted stably by visit pattern)."""
    vis = np.ascontiguousarray(vis, dtype=np.uint32)
    store = np.zeros(sites.n, dtype=np.int64)
    self_ = np.zeros(sites.n, dtype=np.int64)
    check(_lib.load().vrt_layer_sorted_slots(sites.handle, int(dir), vis.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                             _i(store), _i(self_)))
    return store, self_


# ---- regular grid ---------------------------------------------------------------
def short_characteristics_batch(k, up, S_0, I_0, alpha, z, x, y, n_sweeps: int = 3, device: int = 0):
    """Batched regular-grid formal solve.  k (n_solve, 3); up (n_solve,) bools; S_0 / alpha either
    one array (ny, nx, nz) shared by every solve or (n_solve, ny, nx, nz); I_0 (n_solve, ny, nx).
    numpy C-order (ny, nx, nz) is Julia's (nz, nx, ny).  Returns I (n_solve, ny, nx, nz)."""
    k = _f64(np.atleast_2d(k))
    ns = k.shape[0]
    upv = np.ascontiguousarray(np.asarray(up, dtype=bool).reshape(ns), dtype=np.int32)
    z, x, y = _f64(z), _f64(x), _f64(y)
    nz, nx, ny = z.size, x.size, y.size
    S_0, alpha, I_0 = _f64(S_0), _f64(alpha), _f64(I_0)
    vol = nz * nx * ny

    def stride(a, name):
        if a.shape == (ny, nx, nz):
            return 0
        if a.shape == (ns, ny, nx, nz):
            return vol
        raise ValueError(f"{name} has shape {a.shape}, expected {(ny, nx, nz)} or {(ns, ny, nx, nz)}")
    sS, sA = stride(S_0, "S_0"), stride(alpha, "alpha")
    I_0 = I_0.reshape(ns, ny, nx)
    out = np.zeros((ns, ny, nx, nz))
    check(_lib.load().vrt_short_characteristics(nz, nx, ny, _d(z), _d(x), _d(y), ns, _d(k),
                                                upv.ctypes.data_as(_lib.p_int), _d(S_0), sS, _d(alpha),
                                                sA, _d(I_0), int(n_sweeps), int(device), _d(out)))
    return out


def short_characteristics_up(k, S_0, I_0, alpha, z, x, y, n_sweeps: int = 3, device: int = 0):
    """Intensity on the regular grid for rays travelling up (src/characteristics.jl:19-95);
    the reference's `atmos` argument is replaced by its three axes."""
    return short_characteristics_batch([k], [True], S_0, I_0, alpha, z, x, y, n_sweeps, device)[0]


def short_characteristics_down(k, S_0, I_0, alpha, z, x, y, n_sweeps: int = 3, device: int = 0):
    """Intensity on the regular grid for rays travelling down (src/characteristics.jl:110-180)."""
    return short_characteristics_batch([k], [False], S_0, I_0, alpha, z, x, y, n_sweeps, device)[0]


class RegularSolver:
    """Device-resident regular-grid short characteristics (`vrt_regular_*`): the handle owns the
    grid axes and workspaces; `execute_dev` takes device pointers (torch `data_ptr()`) to S, alpha
    (Julia (nz, nx, ny) order, shared or per solve), I_0 (nx, ny, n_solve) and the output
    (nz, nx, ny, n_solve), and is asynchronous on `stream`."""

    def __init__(self, z, x, y, device: int = 0):
        z, x, y = _f64(z), _f64(x), _f64(y)
        self.nz, self.nx, self.ny = z.size, x.size, y.size
        self._h = ctypes.c_void_p()
        check(_lib.load().vrt_regular_create(self.nz, self.nx, self.ny, _d(z), _d(x), _d(y), int(device),
                                             ctypes.byref(self._h)))

    def execute_dev(self, k, up, dS: int, S_stride: int, dalpha: int, alpha_stride: int, dI0: int,
                    dI_out: int, n_sweeps: int = 3, stream: int = 0, field_period: int = 0):
        k = _f64(np.atleast_2d(k))
        ns = k.shape[0]
        upv = np.ascontiguousarray(np.asarray(up, dtype=bool).reshape(ns), dtype=np.int32)
        check(_lib.load().vrt_regular_execute_dev(self._h, ns, _d(k), upv.ctypes.data_as(_lib.p_int), dS,
                                                  int(S_stride), dalpha, int(alpha_stride), int(field_period),
                                                  dI0, int(n_sweeps), dI_out, stream or None))

    def last_solve_ms(self) -> float:
        out = ctypes.c_double()
        check(_lib.load().vrt_regular_last_solve_ms(self._h, ctypes.byref(out)))
        return out.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.load().vrt_regular_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lambda_update_dev(sites: VoronoiSites, nlam: int, ld: int, dJ: int, dB: int, deps: int, dS_old: int,
                      dS_new: int, stream: int = 0) -> float:
    """Device-resident Λ-iteration epilogue: S_new = (1 - ε) J + ε B (src/lambda_iteration.jl:261-263)
    and the convergence measure max |1 - S_old/S_new| of `criterion` (:325-349), which is returned.
    Arguments are device pointers (torch data_ptr())."""
    out = ctypes.c_double()
    check(_lib.load().vrt_lambda_update_dev(sites.handle, nlam, ld, dJ, dB, deps, dS_old, dS_new,
                                            ctypes.byref(out), stream or None))
    return out.value


def lambda_update_native_dev(sites: VoronoiSites, nlam: int, dJ_up: int, dJ_down: int, dB_up: int, deps: int, dS_up: int,
                             dS_down: int, stream: int = 0) -> float:
    """`lambda_update_dev` on sweep-order plane sets (`vrt_lambda_update_native_dev`): J = J_up + J_down, B in the up
    order, the old S read from and the new S written to dS_up (and its down-order copy to dS_down)."""
    out = ctypes.c_double()
    check(_lib.load().vrt_lambda_update_native_dev(sites.handle, nlam, dJ_up or None, dJ_down or None, dB_up, deps, dS_up,
                                                   dS_down, ctypes.byref(out), stream or None))
    return out.value


def rates_populations_native_dev(sites: VoronoiSites, lam, blocks, dJ_up: int, dJ_down: int, planck2, lambda0: float, c0: float,
                                 d_doppler: int, d_gamma: int, sigma_bb_const: float, sigma_bf1, sigma_bf2,
                                 d_temperature: int, d_lte: int, hc_over_kB: float, pref_ij: float, pref_ji: float,
                                 d_C: int, d_atom_density: int, d_R: int, d_populations: int, stream: int = 0) -> None:
    """`rates_populations_dev` with J read from the sweep-order plane sets of both directions."""
    lam, planck2 = _f64(lam), _f64(planck2)
    blocks = np.ascontiguousarray(blocks, dtype=np.int64)
    s1, s2 = _f64(sigma_bf1), _f64(sigma_bf2)
    check(_lib.load().vrt_rates_populations_native_dev(sites.handle, lam.size, _d(lam), _i(blocks), dJ_up or None, dJ_down or None,
                                                       _d(planck2), float(lambda0), float(c0), d_doppler, d_gamma,
                                                       float(sigma_bb_const), _d(s1), _d(s2), d_temperature, d_lte,
                                                       float(hc_over_kB), float(pref_ij), float(pref_ji), d_C, d_atom_density,
                                                       d_R, d_populations, stream or None))


def rates_populations_dev(sites: VoronoiSites, lam, blocks, ld: int, dJ: int, planck2, lambda0: float, c0: float,
                          d_doppler: int, d_gamma: int, sigma_bb_const: float, sigma_bf1, sigma_bf2,
                          d_temperature: int, d_lte: int, hc_over_kB: float, pref_ij: float, pref_ji: float,
                          d_C: int, d_atom_density: int, d_R: int, d_populations: int, stream: int = 0) -> None:
    """Device-resident rates + populations epilogue (`vrt_rates_populations_dev`): calculate_R
    (src/rates.jl:154-201) and get_revised_populations (src/populations.jl:191-221) from J in place."""
    lam, planck2 = _f64(lam), _f64(planck2)
    blocks = np.ascontiguousarray(blocks, dtype=np.int64)
    s1, s2 = _f64(sigma_bf1), _f64(sigma_bf2)
    check(_lib.load().vrt_rates_populations_dev(sites.handle, lam.size, ld, _d(lam), _i(blocks), dJ, _d(planck2),
                                                float(lambda0), float(c0), d_doppler, d_gamma, float(sigma_bb_const),
                                                _d(s1), _d(s2), d_temperature, d_lte, float(hc_over_kB),
                                                float(pref_ij), float(pref_ji), d_C, d_atom_density, d_R,
                                                d_populations, stream or None))


def line_terms_dev(sites: VoronoiSites, d_gamma_static: int, d_gamma_unsold: int, d_populations: int,
                   strength_const: float, Bij: float, Bji: float, d_gamma: int = 0, d_line_strength: int = 0,
                   stream: int = 0) -> None:
    """γ of the current populations (γ_constant, src/broadening.jl:63-82, as J_λ_voronoi evaluates it every
    iteration, lambda_iteration.jl:72-75) and the λ-independent factor of αline_λ (src/line.jl:219-225), on the
    device (`vrt_line_terms_dev`; device pointers)."""
    check(_lib.load().vrt_line_terms_dev(sites.handle, d_gamma_static or None, d_gamma_unsold or None, d_populations,
                                         float(strength_const), float(Bij), float(Bji), d_gamma or None,
                                         d_line_strength or None, stream or None))


# ---- the Λ-iteration driver, device-resident (src/lambda_iteration.jl:205-300, Λ_voronoi) -------------
class LineCase:
    """The per-site inputs Λ_voronoi derives before its loop (LTE populations, α_cont, B_0, ε, C; through
    Transparency.jl, which is outside this path) plus the line's constants, as plain numbers in ONE unit
    system.  Arrays: `lam` (nλ,) all wavelengths (bound-bound block first, then the two bound-free
    blocks; `blocks` = their six [lo, hi) offsets), `velocity` (n, 3) [z, x, y], `doppler`, `alpha_cont`,
    `eps`, `temperature`, `atom_density` (n,), `B0` (n, nλ), `lte` (3, n), `C` (n, 3, 3), `planck2` (nλ,),
    `sigma_bf1`, `sigma_bf2` (one per wavelength of their block).  γ_constant (src/broadening.jl:63-82) is
    `gamma_static` + `gamma_unsold` (n_1 + n_2): the natural + Stark widths, fixed per site, and the van der
    Waals width per unit neutral-hydrogen density, which follows the populations every iteration."""
    FIELDS = ("lam", "blocks", "lambda0", "c0", "velocity", "doppler", "gamma_static", "gamma_unsold", "alpha_cont", "eps",
              "temperature", "atom_density", "B0", "lte", "C", "planck2", "sigma_bf1", "sigma_bf2", "strength_const", "Bij",
              "Bji", "sigma_bb_const", "hc_over_kB", "pref_ij", "pref_ji")

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, kw.pop(k))
        if kw:
            raise TypeError(f"unexpected fields {sorted(kw)}")

    def gamma(self, populations) -> np.ndarray:
        """γ_constant for populations (3, n)"""
        pops = np.asarray(populations)
        return np.asarray(self.gamma_static) + np.asarray(self.gamma_unsold) * (pops[0] + pops[1])

    def c_struct(self):
        """(vrt_line_case, the arrays it points into) for the host-pointer entry points"""
        keep = {k: _f64(getattr(self, k)) for k in ("lam", "velocity", "doppler", "gamma_static", "gamma_unsold", "alpha_cont",
                                                   "eps", "temperature", "atom_density", "B0", "lte", "C", "planck2",
                                                   "sigma_bf1", "sigma_bf2")}
        lc = _lib.LineCaseStruct()
        lc.nlam = keep["lam"].size
        lc.lambda_ = _d(keep["lam"])
        for q, v in enumerate(np.asarray(self.blocks, dtype=np.int64).reshape(6)):
            lc.blocks[q] = int(v)
        lc.lambda0, lc.c0 = float(self.lambda0), float(self.c0)
        for cname, k in (("velocity", "velocity"), ("doppler_width", "doppler"), ("gamma_static", "gamma_static"),
                         ("gamma_unsold", "gamma_unsold"), ("alpha_cont", "alpha_cont"), ("eps", "eps"),
                         ("temperature", "temperature"), ("atom_density", "atom_density"), ("B0", "B0"),
                         ("lte_populations", "lte"), ("C", "C"), ("planck2", "planck2"), ("sigma_bf1", "sigma_bf1"),
                         ("sigma_bf2", "sigma_bf2")):
            setattr(lc, cname, _d(keep[k]))
        for k in ("strength_const", "Bij", "Bji", "sigma_bb_const", "hc_over_kB", "pref_ij", "pref_ji"):
            setattr(lc, k, float(getattr(self, k)))
        return lc, keep


def _quadrature_plan(sites: VoronoiSites, quadrature: str, n_sweeps: int):
    w, th, ph, _ = read_quadrature(quadrature)
    key = (os.path.basename(quadrature), int(n_sweeps))
    plan = sites._plans.get(key)
    if plan is None:
        dirs = [1 if t > 90 else (-1 if t < 90 else 0) for t in th]
        plan = FormalPlan(sites, quadrature_directions(th, ph), n_sweeps, dirs=dirs)
        sites._plans[key] = plan
    return plan, w


def J_lambda_voronoi_line(S_lambda, populations, sites: VoronoiSites, case: LineCase, quadrature: str,
                          n_sweeps: int = 3) -> np.ndarray:
    """J_λ_voronoi, line method (src/lambda_iteration.jl:60-113), from HOST arrays through ONE call
    (`vrt_plan_execute_line`): γ and the line strength of `populations` (3, n), α_tot of every angle made on the
    device, I_0 = B_0 of the bottom layer for the up rays (:99-101), zeros for the down rays (:105-106)."""
    plan, w = _quadrature_plan(sites, quadrature, n_sweeps)
    S = _f64(S_lambda)
    n, nlam = S.shape
    pops = np.asarray(populations)
    gamma = _f64(case.gamma(pops))
    strength = _f64(case.strength_const * (pops[0] * case.Bij - pops[1] * case.Bji))
    lam, vel, dop, ac = _f64(case.lam), _f64(case.velocity), _f64(case.doppler), _f64(case.alpha_cont)
    n1 = int(sites.layers_up[1] - 1)
    I0 = _f64(np.asarray(case.B0)[sites.perm_up[:n1] - 1])
    J = np.zeros((n, nlam))
    check(_lib.load().vrt_plan_execute_line(plan._h, nlam, nlam, _d(lam), float(case.lambda0), float(case.c0), _d(vel),
                                            _d(dop), _d(gamma), _d(strength), _d(ac), _d(S), _d(I0), None, _d(_f64(w)),
                                            _d(J)))
    return J


def _Lambda_voronoi_native(eps_conv: float, maxiter: int, sites: VoronoiSites, case: LineCase, quadrature: str, n_sweeps: int):
    import torch
    w, th, ph, nq = read_quadrature(quadrature)
    dev = torch.device("cuda", sites.device)
    n, nlam = sites.n, int(np.asarray(case.lam).size)
    plan = FormalPlan(sites, quadrature_directions(th, ph), n_sweeps, dirs=[1 if t > 90 else (-1 if t < 90 else 0) for t in th])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    d_vel, d_dop, d_gs, d_gu, d_ac, d_eps, d_T = (t(getattr(case, k)) for k in
                                                  ("velocity", "doppler", "gamma_static", "gamma_unsold", "alpha_cont", "eps",
                                                   "temperature"))
    d_B, d_lte, d_C, d_atom = t(case.B0), t(case.lte), t(case.C), t(case.atom_density)
    pops = d_lte.clone()
    st = torch.cuda.current_stream().cuda_stream
    cnt = plan.native_plane_count(nlam)
    S_up, S_dn, B_up, J_up, J_dn = (torch.zeros(cnt, dtype=torch.float64, device=dev) for _ in range(5))
    plan.to_native_dev(nlam, nlam, d_B.data_ptr(), S_up.data_ptr(), S_dn.data_ptr(), stream=st)      # S_new = B_0
    plan.to_native_dev(nlam, nlam, d_B.data_ptr(), B_up.data_ptr(), 0, stream=st)
    native = torch.empty(plan.native_alpha_count(nlam), dtype=torch.float64, device=dev)
    d_R = torch.empty((n, 3, 3), dtype=torch.float64, device=dev)
    d_gam, strength = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
    n1 = int(sites.layers_up[1] - 1)
    I0_up = d_B[torch.as_tensor(sites.perm_up[:n1] - 1, device=dev)].contiguous()
    history = []
    diff, i = 1.0, 0
    try:
        while diff > eps_conv and i < maxiter:
            line_terms_dev(sites, d_gs.data_ptr(), d_gu.data_ptr(), pops.data_ptr(), case.strength_const, case.Bij, case.Bji,
                           d_gam.data_ptr(), strength.data_ptr(), stream=st)
            plan.line_opacity_dev(case.lam, case.lambda0, case.c0, d_vel.data_ptr(), d_dop.data_ptr(), d_gam.data_ptr(),
                                  strength.data_ptr(), d_ac.data_ptr(), native.data_ptr(), stream=st)
            plan.execute_native_dev(nlam, S_up.data_ptr(), S_dn.data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w,
                                    dJ_up=J_up.data_ptr(), dJ_down=J_dn.data_ptr(), dI0_up=I0_up.data_ptr(), stream=st)
            diff = lambda_update_native_dev(sites, nlam, J_up.data_ptr(), J_dn.data_ptr(), B_up.data_ptr(), d_eps.data_ptr(),
                                            S_up.data_ptr(), S_dn.data_ptr(), stream=st)
            new_pops = torch.empty_like(pops)
            rates_populations_native_dev(sites, case.lam, case.blocks, J_up.data_ptr(), J_dn.data_ptr(), case.planck2, case.lambda0,
                                         case.c0, d_dop.data_ptr(), d_gam.data_ptr(), case.sigma_bb_const, case.sigma_bf1,
                                         case.sigma_bf2, d_T.data_ptr(), d_lte.data_ptr(), case.hc_over_kB, case.pref_ij,
                                         case.pref_ji, d_C.data_ptr(), d_atom.data_ptr(), d_R.data_ptr(), new_pops.data_ptr(), stream=st)
            pops = new_pops
            history.append(diff)
            i += 1
            if diff != diff:
                import warnings
                warnings.warn(f"Lambda_voronoi: NaN DIFF! at iteration {i} -- stopping, results are not converged")
        J, S = torch.zeros((n, nlam), dtype=torch.float64, device=dev), torch.zeros((n, nlam), dtype=torch.float64, device=dev)
        plan.J_from_native_dev(nlam, nlam, J_up.data_ptr(), J_dn.data_ptr(), J.data_ptr(), stream=st)
        plan.from_native_dev(1, nlam, nlam, S_up.data_ptr(), S.data_ptr(), stream=st)
        torch.cuda.synchronize()
        plan.check()
        return J.cpu().numpy(), S.cpu().numpy(), pops.cpu().numpy(), history
    finally:
        plan.close()


def Lambda_voronoi_host(eps_conv: float, maxiter: int, sites: VoronoiSites, case: LineCase, quadrature: str,
                        n_sweeps: int = 3):
    """Λ_voronoi (src/lambda_iteration.jl:205-300) for a host WITHOUT device arrays: the library owns the device
    state (`vrt_lambda_create` / `_iterate` / `_get`), one call per iteration, only the criterion's scalar
    comes back inside the loop.  Returns (J, S_new, populations (3, n), history)."""
    L = _lib.load()
    plan, w = _quadrature_plan(sites, quadrature, n_sweeps)
    lc, keep = case.c_struct()
    h = ctypes.c_void_p()
    check(L.vrt_lambda_create(plan._h, ctypes.byref(lc), _d(_f64(w)), ctypes.byref(h)))
    n, nlam = sites.n, int(keep["lam"].size)
    history, diff, i = [], 1.0, 0                              # criterion(S_new = B, S_old = 0) = 1
    try:
        while diff > eps_conv and i < maxiter:
            d = ctypes.c_double()
            check(L.vrt_lambda_iterate(h, ctypes.byref(d)))
            diff = d.value
            history.append(diff)
            i += 1
            if diff != diff:
                import warnings
                warnings.warn(f"Lambda_voronoi_host: NaN DIFF! at iteration {i} -- stopping, results are not converged")
        J, S, pops = np.zeros((n, nlam)), np.zeros((n, nlam)), np.zeros((3, n))
        check(L.vrt_lambda_get(h, _d(J), _d(S), _d(pops), None, None))
        if i == 0:
            S[:] = keep["B0"]
            pops[:] = keep["lte"]
        return J, S, pops, history
    finally:
        L.vrt_lambda_destroy(h)


def Lambda_voronoi(eps_conv: float, maxiter: int, sites: VoronoiSites, case: LineCase, quadrature: str,
                   n_sweeps: int = 3, native: bool = False):
    """Λ_voronoi (src/lambda_iteration.jl:205-300) with everything between two convergence checks on
    the device, over the device-pointer entry points: per iteration `vrt_line_terms_dev` (γ and the line
    strength of the current populations, :72-75), `vrt_line_opacity_dev` (α_tot of every angle, :72-96),
    `vrt_plan_execute_dev` (J_λ, :84-111), `vrt_lambda_update_dev` (S_new and the criterion's scalar, :261-263,
    :325-349) and `vrt_rates_populations_dev` (:269, :274); only that scalar crosses PCIe inside the loop.
    Starts in LTE with S = B_0 like the reference.
    native=True: S and J stay in the sweep's own per-direction plane sets between the steps
    (`vrt_plan_execute_native_dev`, `vrt_lambda_update_native_dev`, `vrt_rates_populations_native_dev`): no layout
    change inside the loop, the same results bit for bit.
    Returns (J, S_new, populations (3, n), history of the criterion's differences) as numpy arrays."""
    import torch
    if native:
        return _Lambda_voronoi_native(eps_conv, maxiter, sites, case, quadrature, n_sweeps)
    w, th, ph, nq = read_quadrature(quadrature)
    dev = torch.device("cuda", sites.device)
    n, nlam = sites.n, int(np.asarray(case.lam).size)
    plan = FormalPlan(sites, quadrature_directions(th, ph), n_sweeps, dirs=[1 if t > 90 else (-1 if t < 90 else 0) for t in th])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    d_vel, d_dop, d_gs, d_gu, d_ac, d_eps, d_T = (t(getattr(case, k)) for k in
                                                  ("velocity", "doppler", "gamma_static", "gamma_unsold", "alpha_cont", "eps",
                                                   "temperature"))
    d_B, d_lte, d_C, d_atom = t(case.B0), t(case.lte), t(case.C), t(case.atom_density)
    pops = d_lte.clone()                                       # populations = copy(LTE_pops)
    S_new, S_old, J = d_B.clone(), torch.zeros_like(d_B), torch.zeros_like(d_B)
    native = torch.empty(plan.native_alpha_count(nlam), dtype=torch.float64, device=dev)
    d_R = torch.empty((n, 3, 3), dtype=torch.float64, device=dev)
    d_gam, strength = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev)
    n1 = int(sites.layers_up[1] - 1)
    bottom = torch.as_tensor(sites.perm_up[:n1] - 1, device=dev)
    I0_up = d_B[bottom].contiguous()                           # B_λ(λ_l, T) of the bottom layer, :99-101
    st = torch.cuda.current_stream().cuda_stream
    history = []
    diff, i = 1.0, 0                                           # criterion(S_new = B, S_old = 0) = |1 - 0/B| = 1
    try:
        while diff > eps_conv and i < maxiter:                 # criterion, :325-349
            S_old.copy_(S_new)
            # γ_constant of the current populations (:72-75) and αline_λ's population factor (src/line.jl:219-225)
            line_terms_dev(sites, d_gs.data_ptr(), d_gu.data_ptr(), pops.data_ptr(), case.strength_const, case.Bij, case.Bji,
                           d_gam.data_ptr(), strength.data_ptr(), stream=st)
            plan.line_opacity_dev(case.lam, case.lambda0, case.c0, d_vel.data_ptr(), d_dop.data_ptr(), d_gam.data_ptr(),
                                  strength.data_ptr(), d_ac.data_ptr(), native.data_ptr(), stream=st)
            plan.execute_dev(nlam, nlam, S_old.data_ptr(), native.data_ptr(), _lib.ALPHA_ANGLE_NATIVE, w,
                             dJ=J.data_ptr(), dI0_up=I0_up.data_ptr(), stream=st)
            diff = lambda_update_dev(sites, nlam, nlam, J.data_ptr(), d_B.data_ptr(), d_eps.data_ptr(), S_old.data_ptr(),
                                     S_new.data_ptr(), stream=st)
            new_pops = torch.empty_like(pops)
            rates_populations_dev(sites, case.lam, case.blocks, nlam, J.data_ptr(), case.planck2, case.lambda0, case.c0,
                                  d_dop.data_ptr(), d_gam.data_ptr(), case.sigma_bb_const, case.sigma_bf1, case.sigma_bf2,
                                  d_T.data_ptr(), d_lte.data_ptr(), case.hc_over_kB, case.pref_ij, case.pref_ji,
                                  d_C.data_ptr(), d_atom.data_ptr(), d_R.data_ptr(), new_pops.data_ptr(), stream=st)
            pops = new_pops
            history.append(diff)
            i += 1
            if diff != diff:                                   # the reference prints "NaN DIFF!" (:336-338); its
                import warnings                                # NaN diff then ends the loop (NaN > ϵ is false)
                warnings.warn(f"Lambda_voronoi: NaN DIFF! at iteration {i} -- stopping, results are not converged")
        torch.cuda.synchronize()
        return J.cpu().numpy(), S_new.cpu().numpy(), pops.cpu().numpy(), history
    finally:
        plan.close()
