"""Builds libvrt_hip.so (the C-ABI product library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvrt_hip.so")
SOURCES = ["vrt_api.cpp", "vrt_grid.cpp", "vrt_schedule.cpp", "vrt_patch.cpp", "vrt_lambda.cpp", "vrt_multi.cpp", "vrt_tessellate.cpp", "vrt_kernels.hip", "vrt_tables.hip", "vrt_layers.hip", "vrt_patch.hip", "vrt_regular.hip", "vrt_physics.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("vrt_internal.h", "vrt_device.h", "vrt_layout_kernels.h", "vrt_tile_kernels.h", "vrt_step_kernels.h")] + [os.path.join(ROOT, "include", "voronoirt.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """Compile every HIP/C++ source of the product into voronoirt_amd/libvrt_hip.so.

    diag=True builds voronoirt_amd/libvrt_hip_diag.so with -DVRT_DIAG instead: the timing diagnostics
    that switch pieces of the memory traffic off (VRT_DEBUG_FLAGS, VRT_DEBUG_SKIP_LEVELS,
    VRT_TILE_DEBUG; wrong results) exist only there.  tools/flags_sweep.sh selects it with
    VRT_LIB_PATH; nothing else loads it."""
    if diag:
        return _build(os.path.join(HERE, "libvrt_hip_diag.so"), ["-DVRT_DIAG"], verbose)
    if not force and not is_stale():
        return LIB
    return _build(LIB, [], verbose)


def _build(out: str, extra, verbose: bool) -> str:
    cmd = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        # build-wide floating-point contract: no FMA contraction on host or device, so neighbour
        # choices match the CPU oracle bit for bit
        "-ffp-contract=off", "-fno-fast-math",
        "-Wall", "-Wno-unused-result",
        "-I", os.path.join(ROOT, "include"), "-I", CSRC,
        "-o", out,
    ] + list(extra) + [os.path.join(CSRC, s) for s in SOURCES] + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
