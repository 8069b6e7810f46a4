"""Collectives of the frame-sharded path (one process per GPU, RCCL over xGMI) — the only data-path exchanges there are:

* `all_gather_rows`: the north-star's "all-gather of visual tokens" — every rank contributes the rows it encoded (its contiguous frame /
  30-s-window range of ONE video) and ends up with all rows in reference order.  What the reference's sequence-parallel path does with
  `Gather.forward` (`Vidi1.5_9B/vidi/model/lmm/dattn/sequence_parallel/all_to_all.py:361`: all_gather along the token axis, then the ranks'
  pieces concatenated in rank order) and `merge_data` (`.../dattn/split.py:73-93`: concat + narrow to the true length).  Ragged shards (the
  frame count does not divide by the world size; the last audio window is clipped by the global floors) are padded to the longest shard
  for the collective and narrowed afterwards — data movement only, bit-exact.
* `all_gather_packed`: the per-layer exchange of the key-sharded cross-attention — same primitive, fixed sizes; with
  `async_op` the collective runs on the backend's own stream beside whatever the caller launches next (the T2T attention), and `wait()`
  orders the consumer behind it.

Backends: `nccl` (= RCCL on ROCm) moves device buffers directly.  `gloo` is the CPU transport of the tests (several ranks sharing the one
GPU of a test box, or the CPU-only host-logic tests): device tensors take a round trip through host memory there."""
from __future__ import annotations

from typing import Optional, Sequence

import torch


class _Done:
    """handle of a collective that already completed (gloo test transport)"""

    def wait(self):
        return True


def backend_of(group) -> str:
    import torch.distributed as dist
    return dist.get_backend(group)


def all_gather_packed(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """`out` [world, *inp.shape] <- every rank's `inp`.  Returns a handle with `.wait()` (which makes the CURRENT stream wait; no host
    block under RCCL)."""
    import torch.distributed as dist
    cat_shape = (out.shape[0] * inp.shape[0],) + tuple(inp.shape[1:])          # the concatenated-along-dim-0 form both backends accept
    if backend_of(group) == "gloo" and inp.is_cuda:
        o, i = out.cpu().view(cat_shape), inp.cpu().contiguous()
        dist.all_gather_into_tensor(o, i, group=group)
        out.copy_(o.view(out.shape))
        return _Done()
    work = dist.all_gather_into_tensor(out.view(cat_shape), inp, group=group, async_op=async_op)
    return work if async_op and work is not None else _Done()


def all_gather_rows(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """rows of all ranks in rank order: `local` is this rank's [counts[rank], ...] piece; -> [sum(counts), ...] on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [int(c) for c in counts]
    if len(counts) != world or int(local.shape[0]) != counts[rank]:
        raise ValueError(f"all_gather_rows: rank {rank} holds {int(local.shape[0])} rows, the partition says {counts}")
    tail = tuple(local.shape[1:])
    total, longest = sum(counts), max(counts)
    if total == 0:
        return local.new_empty((0,) + tail)
    local = local.contiguous()
    if min(counts) == longest:                                   # even shards: the collective's output IS the result
        out = local.new_empty((world,) + (longest,) + tail)
        all_gather_packed(out, local, group)
        return out.view((total,) + tail)
    send = local
    if counts[rank] < longest:                                   # ragged: pad to the longest shard (the pad rows are never read back)
        send = local.new_zeros((longest,) + tail)
        send[: counts[rank]].copy_(local)
    recv = local.new_empty((world, longest) + tail)
    all_gather_packed(recv, send, group)
    return torch.cat([recv[r, : counts[r]] for r in range(world) if counts[r] > 0], dim=0)


def broadcast0(t: torch.Tensor, group=None) -> torch.Tensor:
    """rank 0's value of a small tensor on every rank of the group"""
    import torch.distributed as dist
    src = dist.get_global_rank(group, 0) if group is not None else 0
    if backend_of(group) == "gloo" and t.is_cuda:
        c = t.cpu()
        dist.broadcast(c, src=src, group=group)
        return c.to(t.device)
    dist.broadcast(t, src=src, group=group)
    return t


def all_reduce_host_ints(values: Sequence[int], group=None, device: Optional[torch.device] = None) -> list:
    """SUM over the ranks of a few host integers (modality presence, valid-key counts): one tiny collective, result back on the host"""
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.int64, device="cpu" if backend_of(group) == "gloo" or device is None else device)
    dist.all_reduce(t, group=group)
    return [int(x) for x in t.tolist()]
