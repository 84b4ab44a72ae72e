"""Particle sharding across the GPUs of one node and the one collective of the path.

Refinement shards embarrassingly (no data-path collective; rows are gathered on the host).
Reconstruction keeps private half-map accumulators per rank and sums them once:
one all-reduce (sum, float32) over RCCL/xGMI — the in-memory form of dump files +
local_merge3d + merge3d's summation (src/pyp/refine/frealign/frealign.py:1838-1903, :2075-2093).
"""
import math

import numpy as np


def split_ranges(frames, cores):
    """1-based inclusive particle ranges exactly as PYP fans them out
    (src/pyp/system/local_run.py:507-514): increment = ceil(frames/cores); ranges
    [first, min(first+increment, frames)] stepping increment+1."""
    if frames < 1 or cores < 1:
        raise ValueError("ERROR: frames and cores must be positive")
    inc = math.ceil(frames / cores)
    return [(first, min(first + inc, frames)) for first in range(1, frames + 1, inc + 1)]


def shard_bounds(n, world, rank):
    """Balanced contiguous 0-based [lo, hi) shard of n particles for `rank` of `world`."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local_rows, n_total, world, rank, group=None):
    """All ranks contribute their refined rows; every rank returns the full (n_total, 32) table in
    particle order.  Host-side (gloo or nccl object collectives are avoided: fixed-size tensors)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return np.asarray(local_rows)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    sizes = [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]
    mx = max(sizes)
    buf = torch.zeros((mx, local_rows.shape[1]), dtype=torch.float64, device=dev)
    buf[: len(local_rows)] = torch.as_tensor(np.asarray(local_rows), dtype=torch.float64, device=dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return np.concatenate([o[:s].cpu().numpy() for o, s in zip(outs, sizes)], axis=0)


_COMM = {}


def library_comm(device, group=None):
    """The library's own RCCL communicator for this process group (made once): rank 0 draws the id, torch.distributed only carries
    the 128 bytes; the data path itself is ppm_accum_reduce (no torch types at the boundary, include/ppm.h)."""
    import torch
    import torch.distributed as dist
    from . import host
    key = (id(group), int(device))
    if key in _COMM:
        return _COMM[key]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on = torch.device("cuda", int(device)) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.zeros(128, dtype=torch.uint8, device=on)
    if rank == 0:
        t.copy_(torch.frombuffer(bytearray(host.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(t, src=0, group=group)
    _COMM[key] = host.make_comm(world, rank, bytes(t.cpu().numpy().tobytes()), device=int(device))
    return _COMM[key]


def reduce_accumulator_handle(acc, device, group=None, root=-1):
    """Sum an `host.Accumulator` over the ranks through the C ABI (ppm_accum_reduce over the library's communicator); counters
    included.  One process per GPU with distinct devices (RCCL refuses two ranks on one device)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return acc.counts()
    acc.reduce(library_comm(device, group), root)
    return acc.counts()


def reduce_accumulators(acc_tensor, counts, group=None):
    """Sum the half-map accumulators (a flat float32 tensor of ppm_accum_floats(box) elements, on the
    GPU for RCCL) and the two particle counters over all ranks, in place."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return acc_tensor, counts
    dist.all_reduce(acc_tensor, op=dist.ReduceOp.SUM, group=group)
    c = torch.as_tensor(list(counts), dtype=torch.int64, device=acc_tensor.device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    return acc_tensor, [int(x) for x in c.cpu()]


def shard_units(unit_ids, world, rank):
    """Constrained refinement / sub-tomogram alignment: the units of a rank (particles, tilts, table rows) as an inclusive
    (first, last) range over the SORTED unit ids, balanced by count; units never interact, so there is no collective - every
    rank writes its own `<param>_<first>_<last>` output exactly like one job of create_csp_split_commands
    (src/pyp/system/local_run.py:416-463).  Returns None when the rank has nothing to do."""
    ids = np.sort(np.unique(np.asarray(unit_ids).astype(np.int64)))
    lo, hi = shard_bounds(len(ids), world, rank)
    if hi <= lo:
        return None
    return int(ids[lo]), int(ids[hi - 1])


def merge_unit_results(rows_list, particles_list, tilts_list, rows_ref, particles_ref, tilts_ref, kind_particles=True):
    """Union of per-rank constrained-refinement results (each a full copy in which only the rank's units changed), the way
    Parameters.merge overlays range outputs on the original (src/pyp/inout/metadata/cistem_star_file.py:655-692)."""
    rows, parts, tl = rows_ref.copy(), particles_ref.copy(), tilts_ref.copy()
    for r, p, t in zip(rows_list, particles_list, tilts_list):
        ch = np.any(r != rows_ref, axis=1)
        rows[ch] = r[ch]
        if kind_particles:
            chp = np.any(p != particles_ref, axis=1); parts[chp] = p[chp]
        else:
            cht = np.any(t != tilts_ref, axis=1); tl[cht] = t[cht]
    return rows, parts, tl
