#!/usr/bin/env python3
"""Known answers for the Voigt profile: Re w(v + i a) of the Faddeeva function from
scipy.special.wofz (exact to 1e-13) on a grid of (a, v) that covers the four regions of
Humlíček's w4 and the ranges the line runs in (a = 1e-4 ... 10, |v| up to 600 = qwing of
sample_λ_line, src/line.jl:44).  scipy exists in the build container only; the .npz written here is
the committed fixture."""
import os

import numpy as np
from scipy.special import wofz

HERE = os.path.dirname(os.path.abspath(__file__))
a = np.array([1e-4, 1e-3, 1e-2, 0.05, 0.2, 0.5, 1.0, 2.0, 5.0, 10.0])
v = np.concatenate([np.linspace(0, 6, 49), np.array([7.5, 10, 14.9, 15.1, 20, 50, 100, 300, 600.0])])
v = np.concatenate([-v[:0:-3], v])
A, V = np.meshgrid(a, v, indexing="ij")
H = wofz(V + 1j * A).real
np.savez(os.path.join(HERE, "voigt_wofz.npz"), a=A.ravel(), v=V.ravel(), H=H.ravel())
print(A.size, "points; H range", H.min(), H.max())
