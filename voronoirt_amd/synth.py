"""Seeded synthetic grids and fields.

The reference builds its Voronoi grids by rejection-sampling sites from a Bifrost snapshot
(src/sample_grids.jl:223-230, src/compare_line.jl:64-106) and running voro++
(rt_preprocessing/output_sites.cc:35-49).  Neither the snapshot nor voro++ exists in the build
image, so the grids used by the tests and the benchmark are generated here, in the same
formats the reference's `read_cell` consumes (src/voronoi_utils.jl:36-63):

  G1  `bcc_grid`      jittered body-centred-cubic lattice, any size, numpy only.  The BCC
                      Voronoi cell (truncated octahedron) is a simple polytope, so its 14-face
                      topology is exact under small jitter.  Site ids are randomly permuted
                      (the reference's sites come out of a sampler in random order) and every
                      neighbour row is shuffled (voro++ face order is arbitrary).
  G2  `voronoi_grid`  a true periodic-xy / walled-z Voronoi tessellation from
                      scipy.spatial.Delaunay with periodic and mirror image points, for the
                      small committed parity fixtures; `voronoi_neighbours` does the same for
                      GIVEN sites (in-process stand-in for the voro++ preprocessing step).

Array conventions follow the reference (Julia, 1-based ids):
  positions  (n, 3) float64 C-order, columns (z, x, y)   == Julia positions[3, n]
  neighbours (D+1, n) int64 C-order, row 0 = count       == Julia neighbours[n, D+1]
             ids 1-based; -5 = bottom wall (z_min), -6 = top wall (z_max)
  bounds     (z_min, z_max, x_min, x_max, y_min, y_max)   (src/io.jl:122-124)
"""
from __future__ import annotations

import numpy as np

BOTTOM_WALL = -5   # src/voronoi_utils.jl:97
TOP_WALL = -6      # src/voronoi_utils.jl:141

# Bifrost-shaped box: x,y in [0, 6 Mm] periodic, z from -0.5 Mm upwards
BOX_XY = 6.0e6
Z_MIN = -0.5e6

# a, c of the BCC lattice for the BASELINE configs (2*a*a*c sites)
BCC_CONFIGS = {
    "C2": (37, 90),     # 246 420 sites
    "C3": (59, 143),    # 995 566 sites
    "C4": (59, 143),
    "C5": (94, 227),    # 4 011 544 sites
}


def _pack_rows(cols: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """cols: (n, K) int64 candidate neighbour ids, 0 = empty slot.  Shuffles each row, moves
    the non-empty entries to the front and returns the (D+1, n) reference matrix."""
    n, K = cols.shape
    keys = rng.random((n, K))
    keys[cols == 0] = 2.0                       # empties sort last
    order = np.argsort(keys, axis=1, kind="stable")
    packed = np.take_along_axis(cols, order, axis=1)
    counts = (packed != 0).sum(axis=1)
    D = int(counts.max())
    out = np.zeros((D + 1, n), dtype=np.int64)
    out[0] = counts
    out[1:] = packed[:, :D].T
    return out


def bcc_grid(a: int, c: int, seed: int, jitter: float = 0.05, box_xy: float = BOX_XY,
             z_min: float = Z_MIN, permute_ids: bool = True):
    """G1: jittered BCC lattice of a x a x c cubic cells (2*a*a*c sites).

    Returns (positions (n,3) [z,x,y], neighbours (D+1,n), bounds)."""
    if a < 3 or c < 2:
        raise ValueError("bcc_grid needs a >= 3 and c >= 2")
    rng = np.random.default_rng(seed)
    h = box_xy / a
    ncell = a * a * c
    n = 2 * ncell
    ii, jj, kk = np.meshgrid(np.arange(a), np.arange(a), np.arange(c), indexing="ij")
    ii = ii.ravel()
    jj = jj.ravel()
    kk = kk.ravel()

    def cid(i, j, k, centre):
        """lattice id (0-based) of corner/centre site (i, j, k) with x,y wrap; -1 outside z."""
        ok = (k >= 0) & (k < c)
        v = ((np.mod(i, a) * a + np.mod(j, a)) * c + np.clip(k, 0, c - 1)) + (ncell if centre else 0)
        return np.where(ok, v, -1)

    cand_corner = []   # for corner sites: 8 centres + 6 corners
    for di in (-1, 0):
        for dj in (-1, 0):
            for dk in (-1, 0):
                cand_corner.append(cid(ii + di, jj + dj, kk + dk, True))
    for (di, dj, dk) in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        cand_corner.append(cid(ii + di, jj + dj, kk + dk, False))
    cand_centre = []   # for centre sites: 8 corners + 6 centres
    for di in (0, 1):
        for dj in (0, 1):
            for dk in (0, 1):
                cand_centre.append(cid(ii + di, jj + dj, kk + dk, False))
    for (di, dj, dk) in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        cand_centre.append(cid(ii + di, jj + dj, kk + dk, True))
    cand = np.concatenate([np.stack(cand_corner, 1), np.stack(cand_centre, 1)], axis=0)  # (n,14)

    # which wall replaces the missing neighbours: below -> bottom, above -> top
    kz = np.concatenate([kk, kk])
    missing = cand < 0
    wall = np.zeros(n, dtype=np.int64)
    has_missing = missing.any(axis=1)
    wall[has_missing & (kz == 0)] = BOTTOM_WALL
    wall[has_missing & (kz == c - 1)] = TOP_WALL
    if c == 1:
        raise ValueError("c must be >= 2")

    # lattice positions: corners at (k + 1/4) h, centres at (k + 3/4) h above z_min
    x = np.concatenate([ii * h, (ii + 0.5) * h])
    y = np.concatenate([jj * h, (jj + 0.5) * h])
    z = np.concatenate([z_min + (kk + 0.25) * h, z_min + (kk + 0.75) * h])
    jit = (rng.random((n, 3)) - 0.5) * (2.0 * jitter * h)
    pos_lat = np.stack([z, x, y], axis=1) + jit
    z_max = z_min + c * h
    # keep jittered sites strictly inside the periodic box
    pos_lat[:, 1] = np.mod(pos_lat[:, 1], box_xy)
    pos_lat[:, 2] = np.mod(pos_lat[:, 2], box_xy)

    # random relabelling of the sites: new id of lattice site s is newid[s] (1-based)
    if permute_ids:
        newid = rng.permutation(n).astype(np.int64) + 1
    else:
        newid = np.arange(1, n + 1, dtype=np.int64)
    cols = np.zeros((n, 15), dtype=np.int64)
    cols[:, :14] = np.where(cand >= 0, newid[np.clip(cand, 0, n - 1)], 0)
    cols[:, 14] = wall
    nbr_lat = _pack_rows(cols, rng)                   # rows still in lattice order
    inv = np.empty(n, dtype=np.int64)
    inv[newid - 1] = np.arange(n)
    positions = np.ascontiguousarray(pos_lat[inv])
    neighbours = np.ascontiguousarray(nbr_lat[:, inv])
    bounds = (z_min, z_max, 0.0, box_xy, 0.0, box_xy)
    return positions, neighbours, bounds


def voronoi_neighbours(positions: np.ndarray, bounds, margin: float = 0.3, shuffle_seed: int | None = 0):
    """Neighbour matrix of the Voronoi tessellation of GIVEN sites, periodic in x and y, walls at
    z_min / z_max: an in-process replacement (scipy/Qhull) for the reference's preprocessing
    step, which forks the voro++ program `output_sites` and parses its text output
    (src/functions.jl:13-23, rt_preprocessing/output_sites.cc:35-49, src/voronoi_utils.jl:42-63).
    positions (n, 3) columns (z, x, y) inside `bounds` = (z_min, z_max, x_min, x_max, y_min, y_max).
    Returns the (D+1, n) matrix `read_cell` would build (row 0 = count, 1-based ids, -5 / -6 walls).

    Periodicity is imposed with image copies of the sites within `margin` (fraction of the box)
    of an x/y edge; the walls with mirror images across z_min / z_max: for a point set that is
    mirror-symmetric about a plane the Voronoi cells never cross the plane, so the cell of a
    site in the augmented set is exactly the wall-cut cell voro++ reports, and a Delaunay edge
    to any bottom (top) mirror image is a face on the bottom (top) wall.  voro++ lists the faces
    of a cell in construction order, which is not reproducible here: rows are shuffled with
    `shuffle_seed` (None keeps ascending ids).  `margin` (fraction of the box whose sites get
    periodic / mirror images) must exceed a few cell diameters: with few sites or a strongly
    stratified density the sparse cells are larger than the default 0.3 and some of their neighbours
    across the seam are missed (a valid neighbour graph for the solver tests, but not the exact
    tessellation: `vrt.voro` is, see tests/test_host.py, which compares at margin > 1)."""
    from scipy.spatial import Delaunay

    base = np.ascontiguousarray(positions, dtype=np.float64)
    n = base.shape[0]
    rng = np.random.default_rng(shuffle_seed) if shuffle_seed is not None else None
    z_min, z_max, x_min, x_max, y_min, y_max = bounds
    Lz, Lx, Ly = z_max - z_min, x_max - x_min, y_max - y_min
    x, y = base[:, 1], base[:, 2]
    pts = [base]
    owner = [np.arange(n)]
    kind = [np.zeros(n, dtype=np.int64)]        # 0 site/periodic image, -5/-6 mirror image
    # periodic images (x,y), only for sites near an edge
    for sx in (-1, 0, 1):
        for sy in (-1, 0, 1):
            if sx == 0 and sy == 0:
                continue
            sel = np.ones(n, dtype=bool)
            if sx == 1:
                sel &= (x - x_min) < margin * Lx
            if sx == -1:
                sel &= (x_max - x) < margin * Lx
            if sy == 1:
                sel &= (y - y_min) < margin * Ly
            if sy == -1:
                sel &= (y_max - y) < margin * Ly
            idx = np.nonzero(sel)[0]
            p = base[idx].copy()
            p[:, 1] += sx * Lx
            p[:, 2] += sy * Ly
            pts.append(p)
            owner.append(idx)
            kind.append(np.zeros(idx.size, dtype=np.int64))
    allp = np.concatenate(pts)
    allo = np.concatenate(owner)
    allk = np.concatenate(kind)
    # mirror images across the walls of everything (sites and periodic images) near a wall
    for wallz, tag in ((z_min, BOTTOM_WALL), (z_max, TOP_WALL)):
        near = np.abs(allp[:, 0] - wallz) < margin * Lz
        idx = np.nonzero(near & (allk == 0))[0]
        p = allp[idx].copy()
        p[:, 0] = 2.0 * wallz - p[:, 0]
        allp = np.concatenate([allp, p])
        allo = np.concatenate([allo, allo[idx]])
        allk = np.concatenate([allk, np.full(idx.size, tag, dtype=np.int64)])

    tri = Delaunay(allp)
    indptr, indices = tri.vertex_neighbor_vertices
    rows = []
    D = 0
    for i in range(n):
        nb = indices[indptr[i]:indptr[i + 1]]
        k = allk[nb]
        ids = allo[nb[k == 0]] + 1
        ids = ids[ids != i + 1]
        # a site can touch two images of the same neighbour only in tiny boxes; keep unique ids
        ids = np.unique(ids)
        ent = list(ids)
        if (k == BOTTOM_WALL).any():
            ent.append(BOTTOM_WALL)
        if (k == TOP_WALL).any():
            ent.append(TOP_WALL)
        ent = np.array(ent, dtype=np.int64)
        if rng is not None:
            rng.shuffle(ent)
        rows.append(ent)
        D = max(D, ent.size)
    neighbours = np.zeros((D + 1, n), dtype=np.int64)
    for i, ent in enumerate(rows):
        neighbours[0, i] = ent.size
        neighbours[1:ent.size + 1, i] = ent
    return neighbours


def voronoi_grid(n: int, seed: int, bounds=(0.0, 1.0, 0.0, 1.0, 0.0, 1.0), scale_height=None,
                 margin: float = 0.3):
    """G2: random sites + their true Voronoi neighbour lists (`voronoi_neighbours`).

    Sites are uniform in x,y and either uniform in z or (scale_height = H) distributed with
    density proportional to exp(-(z - z_min)/H), mimicking the reference's density-weighted
    samplers (src/sample_grids.jl:223-230)."""
    rng = np.random.default_rng(seed)
    z_min, z_max, x_min, x_max, y_min, y_max = bounds
    Lz, Lx, Ly = z_max - z_min, x_max - x_min, y_max - y_min
    x = x_min + rng.random(n) * Lx
    y = y_min + rng.random(n) * Ly
    u = rng.random(n)
    if scale_height is None:
        z = z_min + u * Lz
    else:
        H = scale_height
        z = z_min - H * np.log(1.0 - u * (1.0 - np.exp(-Lz / H)))
    base = np.stack([z, x, y], axis=1)
    neighbours = voronoi_neighbours(base, bounds, margin=margin, shuffle_seed=int(rng.integers(1 << 31)))
    return np.ascontiguousarray(base), neighbours, tuple(bounds)


def regular_lattice_grid(nx: int, ny: int, nz: int, seed: int = 0, jitter: float = 0.0):
    """Unit-cube simple-cubic lattice with 6-neighbour lists (tiny analytic test cases)."""
    rng = np.random.default_rng(seed)
    n = nx * ny * nz
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()

    def sid(i, j, k):
        return (np.mod(i, nx) * ny + np.mod(j, ny)) * nz + k + 1

    cols = np.zeros((n, 6), dtype=np.int64)
    cols[:, 0] = sid(ii + 1, jj, kk)
    cols[:, 1] = sid(ii - 1, jj, kk)
    cols[:, 2] = sid(ii, jj + 1, kk)
    cols[:, 3] = sid(ii, jj - 1, kk)
    cols[:, 4] = np.where(kk + 1 < nz, sid(ii, jj, np.minimum(kk + 1, nz - 1)), TOP_WALL)
    cols[:, 5] = np.where(kk - 1 >= 0, sid(ii, jj, np.maximum(kk - 1, 0)), BOTTOM_WALL)
    neighbours = _pack_rows(cols, rng)
    pos = np.stack([(kk + 0.5) / nz, (ii + 0.5) / nx, (jj + 0.5) / ny], axis=1)
    if jitter:
        pos = pos + (rng.random((n, 3)) - 0.5) * 2.0 * jitter / max(nx, ny, nz)
    return np.ascontiguousarray(pos), neighbours, (0.0, 1.0, 0.0, 1.0, 0.0, 1.0)


# ---------------------------------------------------------------------------------------------
# text formats shared with the reference
# ---------------------------------------------------------------------------------------------
def write_sites_file(path: str, positions: np.ndarray) -> None:
    """voro++ input: "id\\tx\\ty\\tz", 1-based id (src/io.jl:16-20; the reference passes
    positions[2,:], positions[3,:], positions[1,:] because rows are (z, x, y))."""
    with open(path, "w") as f:
        for i, (z, x, y) in enumerate(positions, start=1):
            f.write(f"{i}\t{float(x)!r}\t{float(y)!r}\t{float(z)!r}\n")


def write_neighbours_file(path: str, neighbours: np.ndarray, seed: int | None = 0) -> None:
    """voro++ "%i %n" output (rt_preprocessing/output_sites.cc:49): "id n1 n2 ... nk" per cell,
    lines in arbitrary (block) order -- shuffled here when seed is not None."""
    n = neighbours.shape[1]
    order = np.arange(n)
    if seed is not None:
        np.random.default_rng(seed).shuffle(order)
    with open(path, "w") as f:
        for i in order:
            c = int(neighbours[0, i])
            f.write(" ".join([str(i + 1)] + [str(int(v)) for v in neighbours[1:c + 1, i]]) + "\n")


# ---------------------------------------------------------------------------------------------
# fields
# ---------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def counter_uniform(seed: int, stream: int, index: np.ndarray) -> np.ndarray:
    """U[0,1) from a counter-based generator keyed by (seed, stream, index)."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream))
        v = _splitmix64(index.astype(np.uint64) ^ key)
    return (v >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synthetic_fields(positions: np.ndarray, bounds, nlam: int, seed: int, n_angles: int = 0,
                     alpha0: float = 1.0e-2, scale_height: float = 0.7e6,
                     site_slice: slice | None = None):
    """S(site, λ) > 0 and α(site, λ) per SURVEY 8d.

    α = α0 · exp(-(z - z_min)/H) · (1 + 0.1 u) · ψ(λ) with a Gaussian "line core" ψ, so the
    neighbour-to-neighbour Δτ spans ~1e-6 .. 1e3 over the box and all three branches of
    `linear_weights` (src/functions.jl:484-500) are exercised.  S = 1 + 0.5 sin(2π z/Lz) + 0.1 u.
    With n_angles > 0 an angle-dependent α (n_angles, n, nlam) is returned (line case, where the
    Doppler-shifted profile makes α depend on the direction, src/lambda_iteration.jl:89-96).
    Returns (S (n, nlam), alpha)."""
    z = positions[:, 0]
    n = z.size
    z_min, z_max = bounds[0], bounds[1]
    Lz = z_max - z_min
    site = np.arange(n, dtype=np.uint64)
    lam = np.arange(nlam, dtype=np.uint64)
    idx = site[:, None] * np.uint64(nlam) + lam[None, :]
    uS = counter_uniform(seed, 1, idx)
    S = 1.0 + 0.5 * np.sin(2.0 * np.pi * (z - z_min) / Lz)[:, None] + 0.1 * uS
    centre = 0.5 * (nlam - 1)
    sigma = max(nlam / 6.0, 1.0)
    lamf = np.arange(nlam, dtype=np.float64)
    strat = alpha0 * np.exp(-(z - z_min) / scale_height)
    if n_angles <= 0:
        uA = counter_uniform(seed, 2, idx)
        psi = 1.0 + 9.0 * np.exp(-((lamf - centre) / sigma) ** 2)
        alpha = strat[:, None] * (1.0 + 0.1 * uA) * psi[None, :]
        return S, alpha
    alpha = np.empty((n_angles, n, nlam))
    for a in range(n_angles):
        uA = counter_uniform(seed, 100 + a, idx)
        shift = 0.15 * sigma * np.cos(2.0 * np.pi * a / n_angles)   # direction-dependent core
        psi = 1.0 + 9.0 * np.exp(-((lamf - centre - shift) / sigma) ** 2)
        alpha[a] = strat[:, None] * (1.0 + 0.1 * uA) * psi[None, :]
    return S, alpha
