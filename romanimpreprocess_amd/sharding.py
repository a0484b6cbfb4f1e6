"""Work partitioning across the GPUs of a node (one process per GPU, ``torch.distributed``).

Ramps are independent ((exposure, SCA) items; the reference runs them as a SLURM array with no exchange,
``runs/summer2025run/OpenUniverse_to_L1L2.job:4-7``), so the data path needs no collective.  The only
traffic is the broadcast of the work-item list from rank 0 (RCCL over xGMI on the GPU box, gloo in the CPU
tests) and, for the many-realisations harness, one all-reduce of the per-pixel moment planes.
"""

import torch
import torch.distributed as dist


def scatter_items(items, device="cpu"):
    """Rank 0 owns ``items`` (list of ints); every rank returns its round-robin share ``items[rank::world]``.

    Works without an initialised process group (single process: returns everything)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(items)
    rank, world = dist.get_rank(), dist.get_world_size()
    n = torch.tensor([len(items) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=0)
    buf = (torch.tensor(list(items), dtype=torch.int32, device=device) if rank == 0
           else torch.empty(int(n.item()), dtype=torch.int32, device=device))
    dist.broadcast(buf, src=0)
    return buf[rank::world].tolist()


def allreduce_sum_(tensors):
    """In-place SUM over ranks of a list of tensors (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tensors


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
