"""Multi-GPU sharding of the batch axis: one process per GPU, torch.distributed.

The path shards over independent instances, so there is NO data-path collective: rank r of G owns the
contiguous block [r*Bsz/G, (r+1)*Bsz/G) of the instance-minor arrays (all K initial states of one
system stay on one rank, so the M_V max-reduction is local).  The only exchange is the final
gather of the per-instance cost curves J_T / M_V -- backend "nccl" is RCCL over xGMI on MI355X,
"gloo" on CPU for the tests -- plus an optional all-reduce of per-column (min, max, sum, sum^2) that
reproduces the statistics the reference plots (/root/reference/utils.py:895-898) without gathering.
"""
import numpy as np


def shard_bounds(Bsz, rank, world):
    """Contiguous block partition; the first Bsz % world ranks get one extra instance."""
    base, extra = divmod(int(Bsz), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch, rank, world, keys=("A", "B", "x0", "A_true_inst", "B_true_inst")):
    """Slice the per-instance arrays (last axis = instance) of a batch dict; shared data is kept."""
    Bsz = batch["A"].shape[-1]
    lo, hi = shard_bounds(Bsz, rank, world)
    out = dict(batch)
    for k in keys:
        if k in batch and batch[k] is not None:
            out[k] = np.ascontiguousarray(batch[k][..., lo:hi])
    out["Bsz"] = hi - lo
    out["shard"] = (lo, hi)
    return out


def all_gather_costs(local, Bsz, group=None):
    """Gather the per-instance results of every rank into one (Bsz,) tensor on every rank.

    `local` is a 1-D torch tensor (device tensor under nccl/RCCL, CPU tensor under gloo) holding this
    rank's shard in shard order.  Shards may be ragged (Bsz % world != 0): they are padded to the
    largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(Bsz, r, world)[1] - shard_bounds(Bsz, r, world)[0] for r in range(world)]
    assert local.numel() == sizes[rank], "local shard size does not match the partition"
    m = max(sizes)
    if all(sz == m for sz in sizes):
        out = torch.empty(world * m, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros(m, dtype=local.dtype, device=local.device)
    pad[: local.numel()] = local
    out = torch.empty(world * m, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)])


def column_stats(local, group=None):
    """(min, max, mean, std) per column of the global (n_sys, cols) table from this rank's rows.

    Reproduces np.min/np.max/np.mean/np.std(axis=0) of /root/reference/utils.py:895-898 with one
    all-reduce of 3*cols + 1 numbers (max-reduce of (max, -min), sum-reduce of (sum, sum^2, count))."""
    import torch
    import torch.distributed as dist
    local = local.double()
    mm = torch.cat([local.max(dim=0).values, (-local).max(dim=0).values])
    ss = torch.cat([local.sum(dim=0), (local * local).sum(dim=0),
                    torch.tensor([float(local.shape[0])], dtype=torch.float64, device=local.device)])
    dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(ss, op=dist.ReduceOp.SUM, group=group)
    c = local.shape[1]
    n = ss[-1]
    mean = ss[:c] / n
    var = torch.clamp(ss[c:2 * c] / n - mean * mean, min=0.0)
    return -mm[c:], mm[:c], mean, torch.sqrt(var)
