"""Utterance sharding across the GPUs of one node (new functionality; the reference has no
multi-device path).  Utterances are independent, so the data path needs no
collective: each rank (one process per GPU) synthesises a contiguous slice of the batch.  The only
exchange is the OPTIONAL gather of the finished waveforms (RCCL over xGMI when the process group
is ``nccl``; ``gloo`` in the CPU tests)."""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced partition: the first ``n_items % world`` ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(n_items: int, world: int) -> List[int]:
    return [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]


def take_shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def gather_utterances(local: torch.Tensor, n_total: int, dst: Optional[int] = 0, group=None) -> Optional[torch.Tensor]:
    """Collect the per-rank ``[b_local, T]`` waveforms into ``[n_total, T]``.

    ``dst=None`` -> all-gather (every rank gets the batch); otherwise only ``dst`` receives it and the
    others return ``None``.  Equal shards go through one ``all_gather_into_tensor`` / ``gather``;
    ragged shards are padded to the largest shard (at most one utterance of padding per rank).
    On an 8-GPU MI355X node each peer's shard travels over its own xGMI link, so the gather is
    link-parallel; no ring is forced."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = shard_counts(n_total, world)
    if local.shape[0] != counts[rank]:
        raise ValueError("local shard has %d utterances, expected %d" % (local.shape[0], counts[rank]))
    T = local.shape[1]
    mx = max(counts)
    send = local.contiguous()
    if send.shape[0] < mx:
        pad = torch.zeros(mx - send.shape[0], T, dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    if dst is None:
        out = torch.empty(world * mx, T, dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(out, send, group=group)
        parts = out.view(world, mx, T)
    else:
        bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, bufs, dst=dst, group=group)
        if rank != dst:
            return None
        parts = torch.stack(bufs, 0)
    if all(c == mx for c in counts):
        return parts.reshape(world * mx, T)
    return torch.cat([parts[r, :counts[r]] for r in range(world)], 0)
