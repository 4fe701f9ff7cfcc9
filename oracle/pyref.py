"""Second, independent restatement of the reference algorithm in pure-Python loops (small cases
only).  TEST INFRASTRUCTURE: it cross-checks oracle/vrt_oracle.c against transcription slips; it
is written line by line from the Julia source with 1-based indexing emulated by a leading dummy
element, and shares no code with the C oracle or the product.

Follows: src/voronoi_utils.jl:93-130,138-174 (_sort_by_layer_*), :186-245 (calc_Delaunay_lines),
:253-269 (reduce_layers), :360-396 (smallest_angle); src/functions.jl:392-395 (trapezoidal),
:484-500 (linear_weights); src/irregular_ray_tracing.jl:15-82,96-163 (Delaunay_upII/downII).
"""
import math


def sort_by_layer(neighbours, n_sites, boundary):
    """neighbours[i][j], 1-based i, j; neighbours[i][1] = count (voronoi_utils.jl:93-130)."""
    layers = [0] * (n_sites + 1)
    for i in range(1, n_sites + 1):
        n_nb = neighbours[i][1]
        for j in range(1, n_nb + 1):
            if neighbours[i][j + 1] == boundary:
                layers[i] = 1
    lower_layer = 1
    while True:
        for i in range(1, n_sites + 1):
            if layers[i] == 0:
                n_nb = neighbours[i][1]
                for j in range(1, n_nb + 1):
                    nb = neighbours[i][j + 1]
                    if nb > 0 and layers[nb] == lower_layer:
                        layers[i] = lower_layer + 1
                        break
        if not any(v == 0 for v in layers[1:]):
            break
        lower_layer += 1
    return layers


def sortperm(layers, n_sites):
    """Julia sortperm is stable; returns 1-based perm with a dummy at index 0."""
    return [0] + sorted(range(1, n_sites + 1), key=lambda i: layers[i])


def reduce_layers(sorted_layers):
    """sorted_layers: 1-based list (dummy at 0) -- voronoi_utils.jl:253-269."""
    n = len(sorted_layers) - 1
    reduced = [0] * (max(sorted_layers[1:]) + 2)
    reduced[1] = 1
    layer = 2
    for i in range(1, n + 1):
        if sorted_layers[i] == layer:
            reduced[layer] = i
            layer += 1
    reduced[len(reduced) - 1] = n
    return reduced


def calc_delaunay_lines(positions, neighbours, n_sites, x_min, x_max, y_min, y_max):
    """positions[i] = [_, z, x, y] (1-based components) -- voronoi_utils.jl:186-245."""
    lines = {}
    for i in range(1, n_sites + 1):
        position = positions[i]
        x_r_r = x_max - position[2]
        x_r_l = position[2] - x_min
        y_r_r = y_max - position[3]
        y_r_l = position[3] - y_min
        n_nb = neighbours[i][1]
        for j in range(1, n_nb + 1):
            nb = neighbours[i][j + 1]
            if nb > 0:
                p_n = list(positions[nb])
                x_i_r = abs(x_max - p_n[2])
                x_i_l = abs(p_n[2] - x_min)
                if x_r_r + x_i_l < position[2] - p_n[2]:
                    p_n[2] = x_max + p_n[2] - x_min
                elif x_r_l + x_i_r < p_n[2] - position[2]:
                    p_n[2] = x_min + x_max - p_n[2]
                y_i_r = abs(y_max - p_n[3])
                y_i_l = abs(p_n[3] - y_min)
                if y_r_r + y_i_l < position[3] - p_n[3]:
                    p_n[3] = y_max + p_n[3] - y_min
                elif y_r_l + y_i_r < p_n[3] - position[3]:
                    p_n[3] = y_min + y_max - p_n[3]
                p_d = [0.0, p_n[1] - position[1], p_n[2] - position[2], p_n[3] - position[3]]
                nrm = math.sqrt((p_d[1] * p_d[1] + p_d[2] * p_d[2]) + p_d[3] * p_d[3])
                lines[(j, i)] = [0.0, p_d[1] / nrm, p_d[2] / nrm, p_d[3] / nrm]
    return lines


def smallest_angle(n, nbs, k, lines):
    """voronoi_utils.jl:360-396; nbs is the 1-based neighbour vector of site n."""
    dots = [0.0, -1.0, -1.0]
    indices = [0, None, None]
    for i in range(1, len(nbs)):
        nb = nbs[i]
        if nb > 0:
            norm_dir = lines[(i, n)]
            dot_product = (k[1] * norm_dir[1] + k[2] * norm_dir[2]) + k[3] * norm_dir[3]
            if dot_product > dots[2]:
                if dot_product > dots[1]:
                    dots[1] = dot_product
                    indices[1] = nb
                else:
                    dots[2] = dot_product
                    indices[2] = nb
    if dots[2] <= 0:
        dots[2] = 0.0
        indices[2] = indices[1]
    return dots, indices


def linear_weights(dtau):
    """functions.jl:484-500"""
    if dtau < 5e-4:
        e = 1 - dtau + 0.5 * dtau ** 2
        a = dtau * (1 / 2 - dtau / 3)
        b = dtau * (1 / 2 - dtau / 6)
    elif dtau > 50:
        e = 0.0
        a = 1 / dtau
        b = 1.0 - a
    else:
        e = math.exp(-dtau)
        a = (1 - e) / dtau - e
        b = 1 - a - e
    return a, b, e


def delaunay(up, k, S, I_0, alpha, positions, neighbours, lines, layers, perm, n_sweeps):
    """Delaunay_upII (up=True, irregular_ray_tracing.jl:15-82) / Delaunay_downII (:96-163).
    All sequences 1-based with a dummy first element; returns I likewise."""
    p = 7.0
    n = len(S) - 1
    I = [0.0] * (n + 1)
    max_layer = len(layers) - 1
    lower_idx = layers[2] - 1
    for t in range(1, lower_idx + 1):
        I[perm[t]] = I_0[t]
    for layer in range(2, max_layer):
        lower_idx = layers[layer]
        upper_idx = layers[layer + 1]
        for _sweep in range(1, n_sweeps + 1):
            order = range(lower_idx, upper_idx) if up else range(upper_idx - 1, lower_idx - 1, -1)
            for i in order:
                idx = perm[i]
                position = positions[idx]
                n_nb = neighbours[idx][1]
                nbs = [0] + [neighbours[idx][j] for j in range(2, n_nb + 2)]
                dots, upwind = smallest_angle(idx, nbs, k, lines)
                s = dots[1] ** p + dots[2] ** p
                w = [0.0, dots[1] ** p / s, dots[2] ** p / s]
                I[idx] = 0.0
                for rn in (1, 2):
                    u = upwind[rn]
                    up_pos = positions[u]
                    r = math.sqrt(((position[1] - up_pos[1]) ** 2 + (position[2] - up_pos[2]) ** 2)
                                  + (position[3] - up_pos[3]) ** 2)
                    dtau = r * (alpha[idx] + alpha[u]) / 2
                    a, b, e = linear_weights(dtau)
                    I[idx] += (e * I[u] + a * S[u] + b * S[idx]) * w[rn]
    return I
