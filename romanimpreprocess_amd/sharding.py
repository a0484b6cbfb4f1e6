"""Work partitioning across the GPUs of a node (one process per GPU, ``torch.distributed``).

Ramps are independent ((exposure, SCA) items; the reference runs them as a SLURM array with no exchange,
``runs/summer2025run/OpenUniverse_to_L1L2.job:4-7``), so the data path needs no collective.  The only
traffic is the broadcast of the work-item list from rank 0 (RCCL over xGMI on the GPU box, gloo in the CPU
tests).  The many-realisations harness is the one place with a real exchange step: its per-pixel statistics (sums in
realisation order, medians over realisations) need every realisation of a pixel on one GPU, so the stacks go from
"my realisations, all rows" to "all realisations, my rows" with one all-to-all per stack (``seeds_to_rows``), and the
finished planes are collected on rank 0 (``gather_rows``).
"""

import torch
import torch.distributed as dist


def scatter_items(items, device="cpu"):
    """Rank 0 owns ``items`` (list of ints); every rank returns its round-robin share ``items[rank::world]``.

    Works without an initialised process group (single process: returns everything)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(items)
    rank, world = dist.get_rank(), dist.get_world_size()
    if dist.get_backend() == "gloo":
        device = "cpu"
    n = torch.tensor([len(items) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=0)
    buf = (torch.tensor(list(items), dtype=torch.int32, device=device) if rank == 0
           else torch.empty(int(n.item()), dtype=torch.int32, device=device))
    dist.broadcast(buf, src=0)
    return buf[rank::world].tolist()


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _all_to_all(recv, send, out_split, in_split):
    """``all_to_all_single`` on 1-D tensors.  RCCL moves device tensors directly (xGMI); the gloo backend (CPU tests,
    rehearsals of several ranks on one GPU) only takes host tensors, so device tensors are staged through the host."""
    if dist.get_backend() == "gloo" and send.is_cuda:
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu(), out_split, in_split)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, out_split, in_split)


def row_bounds(ny, world):
    """Row range of rank r = [bounds[r], bounds[r+1])."""
    return [ny * r // world for r in range(world + 1)]


def seeds_to_rows(stack, nseeds):
    """``stack`` (S_local, ny, nx): this rank's realisations, round robin (global index j = rank + k * world).
    Returns ``(rows_stack, y0)``: (nseeds, nrows, nx) with every realisation in order for this rank's row range,
    which starts at row ``y0``.  One ``all_to_all_single`` (RCCL / gloo); a single process gets its stack back."""
    rank, world = _world()
    if world == 1:
        assert stack.shape[0] == nseeds
        return stack, 0
    ny, nx = stack.shape[1:]
    b = row_bounds(ny, world)
    counts = [len(range(r, nseeds, world)) for r in range(world)]
    assert stack.shape[0] == counts[rank], (stack.shape, counts)
    nrows = b[rank + 1] - b[rank]
    send = torch.cat([stack[:, b[q]:b[q + 1]].reshape(-1) for q in range(world)])
    in_split = [counts[rank] * (b[q + 1] - b[q]) * nx for q in range(world)]
    out_split = [counts[s] * nrows * nx for s in range(world)]
    recv = torch.empty(sum(out_split), dtype=stack.dtype, device=stack.device)
    _all_to_all(recv, send, out_split, in_split)
    del send
    out = torch.empty((nseeds, nrows, nx), dtype=stack.dtype, device=stack.device)
    off = 0
    for s in range(world):
        out[s::world] = recv[off:off + out_split[s]].view(counts[s], nrows, nx)
        off += out_split[s]
    return out, b[rank]


def gather_rows(planes, ny):
    """``planes`` (P, nrows, nx) for this rank's row range -> (P, ny, nx) on rank 0 (None elsewhere)."""
    rank, world = _world()
    if world == 1:
        return planes
    P, _, nx = planes.shape
    b = row_bounds(ny, world)
    in_split = [planes.numel()] + [0] * (world - 1)
    out_split = [P * (b[s + 1] - b[s]) * nx if rank == 0 else 0 for s in range(world)]
    recv = torch.empty(sum(out_split), dtype=planes.dtype, device=planes.device)
    _all_to_all(recv, planes.reshape(-1).contiguous(), out_split, in_split)
    if rank != 0:
        return None
    out = torch.empty((P, ny, nx), dtype=planes.dtype, device=planes.device)
    off = 0
    for s in range(world):
        out[:, b[s]:b[s + 1]] = recv[off:off + out_split[s]].view(P, b[s + 1] - b[s], nx)
        off += out_split[s]
    return out


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
