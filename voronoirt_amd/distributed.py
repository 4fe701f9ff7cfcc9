"""One-process-per-GPU sharding of the angle x wavelength loop of J_λ_voronoi.

Every (angle, λ) formal solve is independent given S, α and I_0 (src/lambda_iteration.jl:84-111);
the only coupling is J_λ[l, :] = Σ_angles w · I.  Sites of one solve do not shard (the upwind
dependency runs through all layers), so the grid and the per-angle tables are replicated on
every rank.  Two partitions of the work units:

  "lambda"  each rank owns a contiguous block of wavelengths and runs ALL angles for it, so it
            owns whole rows J[l, :] and no data-path collective is needed (an optional
            all-gather replicates J for a caller that wants it everywhere);
  "angle"   each rank runs a subset of the angles for ALL wavelengths and the partial J's are
            summed with one all-reduce (RCCL over xGMI on GPUs; gloo in the CPU tests) --
            the scheme BASELINE.json's north star names, needed when nλ < world size.

torch.distributed is plumbing here: the collectives operate on tensors whose storage the HIP
library wrote through raw device pointers.
"""
from __future__ import annotations

import os

import numpy as np


def partition(n_units: int, world: int, rank: int):
    """Contiguous block partition: the first n_units % world ranks get one extra unit
    (51 λ over 8 ranks -> 7,7,7,6,6,6,6,6).  Returns (start, stop)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    return start, stop


def angle_assignment(theta_deg, world: int):
    """Balanced assignment of angles to ranks for the "angle" mode: up and down rays have
    different schedules and costs, so they are dealt round-robin separately.  Returns a list of
    index arrays, one per rank; θ = 90 directions (skipped by the solver) are left out."""
    theta = np.asarray(theta_deg, dtype=np.float64)
    ups = [i for i in range(theta.size) if theta[i] > 90]
    downs = [i for i in range(theta.size) if theta[i] < 90]
    out = [[] for _ in range(world)]
    for j, i in enumerate(ups):
        out[j % world].append(i)
    for j, i in enumerate(downs):
        out[(world - 1 - j) % world].append(i)
    return [np.array(sorted(o), dtype=np.int64) for o in out]


def init_process_group(backend: str | None = None, force: bool = False):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).
    backend None -> "nccl" (= RCCL on ROCm) when a GPU is visible, else "gloo".  force: a group also for a world
    of one rank (the collectives then execute, on one member)."""
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1 and not force:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def allreduce_J(J_partial):
    """Sum the angle-sharded partial mean intensities in place (the J all-reduce of the
    "angle" mode).  No-op for a single process."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(J_partial, op=dist.ReduceOp.SUM)
    return J_partial


def allgather_J_lambda(J_block, nlam_total: int):
    """Replicate a λ-sharded J: every rank contributes its (n, nlam_local) block; returns the
    (n, nlam_total) array in wavelength order.  Blocks follow `partition`."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return J_block
    world = dist.get_world_size()
    n = J_block.shape[0]
    sizes = [partition(nlam_total, world, r) for r in range(world)]
    width = max(b - a for a, b in sizes)
    padded = torch.zeros((n, width), dtype=J_block.dtype, device=J_block.device)
    padded[:, : J_block.shape[1]] = J_block
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded.contiguous())
    return torch.cat([parts[r][:, : sizes[r][1] - sizes[r][0]] for r in range(world)], dim=1)


def allgather_J_blocks(J_block, nlam_total: int, out=None):
    """The J all-gather of the "lambda-strong" mode as ONE collective into a preallocated
    (world, n, width) buffer (width = the largest wavelength block; narrower blocks are padded): no list of
    tensors, no concatenation inside the step.  Returns (buffer, [(start, stop)] per rank);
    `assemble_J_blocks` builds the (n, nlam_total) array when a caller wants it."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    n = J_block.shape[0]
    sizes = [partition(nlam_total, world, r) for r in range(world)]
    width = max(b - a for a, b in sizes)
    if out is None or tuple(out.shape) != (world, n, width) or out.dtype != J_block.dtype or out.device != J_block.device:
        out = torch.zeros((world, n, width), dtype=J_block.dtype, device=J_block.device)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        out[0, :, : J_block.shape[1]] = J_block
        return out, sizes
    if J_block.shape[1] == width and J_block.is_contiguous():
        mine = J_block
    else:
        mine = torch.zeros((n, width), dtype=J_block.dtype, device=J_block.device)
        mine[:, : J_block.shape[1]] = J_block
    dist.all_gather_into_tensor(out.view(world * n, width), mine)     # concatenation along dim 0 = the (world, n, width) buffer
    return out, sizes


def assemble_J_blocks(buf, sizes):
    import torch
    return torch.cat([buf[r][:, : b - a] for r, (a, b) in enumerate(sizes)], dim=1)
