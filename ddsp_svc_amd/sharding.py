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


def gather_utterances(local: torch.Tensor, n_total: int, dst: Optional[int] = 0, group=None, async_op: bool = False,
                      out: Optional[torch.Tensor] = None):
    """Collect the per-rank ``[b_local, T]`` waveforms into ``[n_total, T]``.

    ``dst=None`` -> all-gather (every rank gets the batch); otherwise only the rank whose rank INSIDE ``group`` is
    ``dst`` receives it and the others return ``None``.  Equal shards go through one ``all_gather_into_tensor`` /
    ``gather`` whose receive buffers are slices of the result tensor (no second copy on the root); ragged shards are
    padded to the largest shard (at most one utterance of padding per rank).  On an 8-GPU MI355X node each peer's shard
    travels over its own xGMI link, so the gather is link-parallel; no ring is forced.

    ``out`` (equal shards only): the result tensor ``[n_total, T]`` of a receiving rank, kept by the caller from step to step.  A
    rank whose ``local`` IS its slice of it (``out[rank * b : (rank + 1) * b]`` -- e.g. the synthesis was handed that slice as
    ``signal_out``) sends from there: its own shard is never copied (RCCL's in-place all-gather; a root's own slot of a gather),
    and with one rank nothing moves at all.

    ``async_op=True`` returns ``(finish, work)`` instead: the collective is enqueued and ``finish()`` waits for it and
    returns the tensor (or None) -- a caller that synthesises in chunks can enqueue the gather of chunk i and start
    chunk i+1 before waiting."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)                                  # rank inside ``group``: shards are numbered in group order
    counts = shard_counts(n_total, world)
    if local.dim() != 2:
        raise ValueError("local must be [b_local, T]")
    if local.shape[0] != counts[rank]:
        raise ValueError("local shard has %d utterances, expected %d" % (local.shape[0], counts[rank]))
    if dst is not None and not (0 <= dst < world):
        raise ValueError("dst must be a rank inside the group")
    T = local.shape[1]
    mx = max(counts)
    send = local.contiguous()
    if send.shape[0] < mx:
        pad = torch.zeros(mx - send.shape[0], T, dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    receiver = dst is None or rank == dst
    in_place = False
    if out is not None and receiver:
        if any(c != mx for c in counts):
            raise ValueError("out= needs equal shards")
        if tuple(out.shape) != (world * mx, T) or out.dtype != send.dtype or out.device != send.device or not out.is_contiguous():
            raise ValueError("out must be a contiguous [%d, %d] tensor of the shards' dtype on their device" % (world * mx, T))
        in_place = send.data_ptr() == out[rank * mx:(rank + 1) * mx].data_ptr()
    else:
        out = torch.empty(world * mx, T, dtype=send.dtype, device=send.device) if receiver else None
    if world == 1 and in_place:
        work = None                                              # the one shard is already where the result lives
    elif dst is None:
        work = dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)
    else:
        # (in place: the root's own slot is the very tensor it sends -- the backend's copy of a root's own shard is then a copy of a
        # tensor onto itself, which torch skips, not a device copy between two views of the same memory)
        bufs = [send if (in_place and r == rank) else out[r * mx:(r + 1) * mx] for r in range(world)] if receiver else None
        # torch.distributed addresses the destination by GLOBAL rank; translate the group-local one
        gdst = dist.get_global_rank(group, dst) if group is not None else dst
        work = dist.gather(send, bufs, dst=gdst, group=group, async_op=async_op)

    def finish():
        if async_op and work is not None:
            work.wait()
        if not receiver:
            return None
        if all(c == mx for c in counts):
            return out
        parts = out.view(world, mx, T)
        return torch.cat([parts[r, :counts[r]] for r in range(world)], 0)

    if async_op:
        return finish, work
    return finish()


def synth_sharded(fn, n_total: int, rank: int, world: int, *batched, logical_shards: int = 1):
    """Run ``fn(*slices) -> [b, T]`` on this rank's contiguous slice of every ``[n_total, ...]`` tensor in ``batched``
    (utterances are independent: no collective on the data path).  ``logical_shards > 1`` further splits the rank's
    slice into that many contiguous pieces run back to back -- what a 1-GPU box uses to exercise the partition
    arithmetic of a G-GPU job -- and concatenates the results."""
    lo, hi = shard_bounds(n_total, rank, world)
    outs = []
    for g in range(logical_shards):
        a, b = shard_bounds(hi - lo, g, logical_shards)
        if b > a:
            outs.append(fn(*(t[lo + a:lo + b] for t in batched)))
    if not outs:
        return None
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)
