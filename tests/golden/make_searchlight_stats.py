#!/usr/bin/env python3
"""Reduces the reference's committed Voronoi searchlight images
(data/searchlight_data/I_160_45_voronoi.npy, I_20_15_voronoi.npy: 510 x 510 nearest-neighbour
rasters of Delaunay_upII / Delaunay_downII on 51^3 unseeded random sites, alpha = S = 0, a disk of
radius 0.1 lit on the boundary; src/compare_searchlight.jl:10-152) to a few statistics.  The site
positions are not reproducible (unseeded rand), so only beam position and width can be compared:
they pin the DIRECTION CONVENTION of the Voronoi solver (k points from a site towards its
upwind side) and its numerical diffusion against reference output.  Run in the build container
(reads /root/reference); the JSON it writes is the committed fixture."""
import json
import os
import sys

import numpy as np

REF = "/root/reference/data/searchlight_data"
HERE = os.path.dirname(os.path.abspath(__file__))


RADIAL_EDGES = [0.0, 0.03, 0.06, 0.09, 0.12, 0.15, 0.18, 0.21]


def radial_profile(a, c0, c1):
    """Mean intensity in annuli around the beam centroid (periodic distances, box = 1)."""
    n = a.shape[0]
    g = (np.arange(n) + 0.5) / n
    d0 = np.abs(g - c0) % 1.0
    d0 = np.minimum(d0, 1.0 - d0)
    d1 = np.abs(g - c1) % 1.0
    d1 = np.minimum(d1, 1.0 - d1)
    r = np.sqrt(d0[:, None] ** 2 + d1[None, :] ** 2)
    return [float(a[(r >= lo) & (r < hi)].mean()) for lo, hi in zip(RADIAL_EDGES[:-1], RADIAL_EDGES[1:])]


def stats(a):
    n = a.shape[0]
    ang = 2 * np.pi * (np.arange(n) + 0.5) / n
    w0, w1 = a.sum(axis=1), a.sum(axis=0)
    c0 = float(np.angle((w0 * np.exp(1j * ang)).sum()) / (2 * np.pi) % 1)
    c1 = float(np.angle((w1 * np.exp(1j * ang)).sum()) / (2 * np.pi) % 1)
    # circular spread: 1 - |mean resultant|
    r0 = float(abs((w0 * np.exp(1j * ang)).sum()) / w0.sum())
    r1 = float(abs((w1 * np.exp(1j * ang)).sum()) / w1.sum())
    return {"centroid": [c0, c1], "resultant": [r0, r1], "mean": float(a.mean()), "max": float(a.max()),
            "frac_above_0.05": float((a > 0.05).mean()),
            "radial_edges": RADIAL_EDGES, "radial_profile": radial_profile(a, c0, c1)}


out = {}
for name in ("I_160_45_voronoi", "I_20_15_voronoi"):
    out[name] = stats(np.load(os.path.join(REF, name + ".npy")))
json.dump(out, open(os.path.join(HERE, "voronoi_searchlight_reference_stats.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
