"""Multi-GPU layout of the front end: independent sequences / contiguous time-slice ranges
are dealt one per rank (one process per GPU); there is no collective on the data path.  The only exchange
is the final gather of fixed-capacity per-slice keypoint records to rank 0 (RCCL over xGMI on GPUs, gloo in
the CPU tests)."""
import torch
import torch.distributed as dist


def slice_range(n_slices, rank, world, halo=1):
    """Contiguous range of time-slices for `rank` plus `halo` slices of overlap on the left (frame-to-frame
    matching needs the previous slice's descriptors).  Returns (first_owned, last_owned_exclusive, first_loaded)."""
    base, rem = divmod(n_slices, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi, max(lo - halo, 0)


def sequence_for_rank(n_sequences, rank, world):
    """C5: sequence s goes to rank s % world."""
    return [s for s in range(n_sequences) if s % world == rank]


def gather_records(tensors, dst=0):
    """Gather same-shaped per-rank tensors (keypoints, descriptors, counts) to `dst`.
    Returns a list (one entry per input tensor) of per-rank tensor lists on dst, None elsewhere."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    out = []
    for t in tensors:
        if world == 1:
            out.append([t])
            continue
        lst = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, lst, dst=dst)
        out.append(lst)
    return out if rank == dst else None
