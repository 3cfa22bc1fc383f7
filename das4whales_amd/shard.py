"""Channel-block sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Row-independent stages of the path -- dsp.bp_filt, detect.compute_cross_correlogram, picks --
shard by contiguous channel block with no communication; the filtered t-x matrix (or, much
cheaper, the correlogram / picks) is reassembled with ONE all-gather.  The f-k filter is a global
2-D transform: applying it per channel shard is a different filter, so it is not
offered here per shard.
"""
import torch
import torch.distributed as dist


def channel_block(nx, world_size, rank):
    """Contiguous balanced partition of nx channels: ranks < nx % world get one extra row."""
    base, extra = divmod(int(nx), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_rows(local, nx_total, group=None):
    """Reassemble a row-sharded [rows_local, ...] tensor into [nx_total, ...] on every rank with a
    single all-gather (dist.all_gather_into_tensor).  Uneven shards are padded to the largest block
    for the collective and stripped afterwards."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    blocks = [channel_block(nx_total, world, r) for r in range(world)]
    if local.shape[0] != blocks[rank][1] - blocks[rank][0]:
        raise ValueError("local block has %d rows, expected %d" % (local.shape[0], blocks[rank][1] - blocks[rank][0]))
    rmax = max(b[1] - b[0] for b in blocks)
    tail = tuple(local.shape[1:])
    local = local.contiguous()
    if nx_total % world == 0:
        out = torch.empty((nx_total,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    padded = torch.zeros((rmax,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    buf = torch.empty((world * rmax,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * rmax: r * rmax + (b[1] - b[0])] for r, b in enumerate(blocks)], dim=0)


def map_channel_blocks(fn, x_full, gather=True, group=None):
    """Apply a row-independent operator to this rank's channel block of `x_full` ([nx, ns], present
    on every rank or memory-mapped) and optionally all-gather the result."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    a, b = channel_block(x_full.shape[0], world, rank)
    y = fn(x_full[a:b])
    return all_gather_rows(y, x_full.shape[0], group=group) if gather else y
