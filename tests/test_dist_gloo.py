"""world_size-2 (gloo, CPU) test of the multi-GPU decomposition: keys sharded by frame/time across
ranks, per-rank partial softmax (numerator, running max, denominator), all-gather, exact LSE merge —
the same identity `VidiEngine._cross` applies to the kernels' partials on RCCL.  The per-rank compute
here is the CPU oracle (tests may use it); what is under test is the sharding + merge logic."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vidi_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(n, world, rank):
    base, extra = divmod(n, world)
    s = rank * base + min(rank, extra)
    return s, s + base + (1 if rank < extra else 0)


def _partial(q, k, v, scale, softcap):
    """un-normalised partial attention over a key shard: (num[H,Lq,D], m[H,Lq], l[H,Lq]) in base-e units"""
    if k.shape[1] == 0:
        H, Lq, D = q.shape
        return torch.zeros(H, Lq, D), torch.full((H, Lq), float("-inf")), torch.zeros(H, Lq)
    s = torch.matmul(q, k.transpose(1, 2)) * scale
    s = torch.tanh(s / softcap) * softcap
    m = s.max(-1).values
    p = torch.exp(s - m[..., None])
    return torch.matmul(p, v), m, p.sum(-1)


def _worker(rank, world, port, frames, tok_per_frame, q, k, v, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f0, f1 = _shard(frames, world, rank)                       # contiguous frame range -> contiguous token range
    ks, ke = f0 * tok_per_frame, f1 * tok_per_frame
    num, m, l = _partial(q, k[:, ks:ke], v[:, ks:ke], 0.25, 50.0)
    gnum = torch.zeros((world,) + num.shape); gm = torch.zeros((world,) + m.shape); gl = torch.zeros((world,) + l.shape)
    # concatenated-along-dim-0 form (accepted by both gloo and nccl/RCCL)
    dist.all_gather_into_tensor(gnum.view(-1, *num.shape[1:]), num.contiguous())
    dist.all_gather_into_tensor(gm.view(-1, *m.shape[1:]), m.contiguous())
    dist.all_gather_into_tensor(gl.view(-1, *l.shape[1:]), l.contiguous())
    out = O.merge_partials(gnum, gm, gl)
    # every rank must hold the identical merged result
    chk = [torch.zeros_like(out) for _ in range(world)]
    dist.all_gather(chk, out)
    assert all(torch.equal(c, chk[0]) for c in chk)
    if rank == 0:
        ret.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("frames,world", [(7, 2), (1, 2)])
def test_sharded_cross_attention_merge(frames, world):
    torch.manual_seed(0)
    H, Lq, D, tpf = 4, 5, 16, 9
    N = frames * tpf
    q, k, v = torch.randn(H, Lq, D), torch.randn(H, N, D), torch.randn(H, N, D)
    ref = O.sdpa_reference(q[None], k[None], v[None], 0.25, 50.0)[0]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, frames, tpf, q, k, v, ret)) for r in range(world)]
    for p in procs:
        p.start()
    out = ret.get(timeout=60)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)


def test_frame_and_window_shards_tile_the_video():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for n, world in [(3600, 8), (3600, 1), (120, 8), (7, 8), (300, 4)]:
        cuts = [bench.shard(n, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        assert max(e - s for s, e in cuts) - min(e - s for s, e in cuts) <= 1
