"""Frame/window sharding of one video over ranks — the host bookkeeping of the PRODUCT code on CPU:

* vidi_amd/shard.py: frame / window ranges, the audio-token range a window shard owns, the packed all-gather layout;
* vidi_amd/model.py `VidiForCausalLM.encode_mm_state / generate` under `engine.set_dist()` with world 2 and 3 over gloo: every rank
  is handed the same video, keeps its share, and must produce the single-rank answer.  The numerics engine of this CPU test is the
  oracle behind VidiEngine's interface (tests/oracle_engine.py), which also asserts the offsets / totals / whole-sample flags the
  product passes; the HIP engine runs the same product path in tests/test_gpu_dist.py."""
import os
import socket

import numpy as np
import pytest
import torch

from vidi_amd import shard as S


def test_shards_tile_the_range():
    for n, world in [(3600, 8), (3600, 1), (120, 8), (7, 8), (300, 4), (0, 2), (1, 2)]:
        cuts = [S.shard(n, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        sizes = [e - s for s, e in cuts]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        S.shard(4, 2, 2)


@pytest.mark.parametrize("audio_size,world", [(360000, 8), (173, 2), (173, 3), (2999, 2), (3001, 2), (90000, 7), (100, 4)])
def test_audio_window_shards_tile_the_global_token_range(audio_size, world):
    """rank shards of 30-s windows own disjoint, ordered, complete ranges of the GLOBAL pooled audio tokens
    (floor(floor(size * 1500 / 3000) / 5), multimodal.py:226-235), including the clipped last window and empty shards."""
    rows, pool = 1500, 5
    C = -(-audio_size // 3000)
    s1 = int(np.floor(np.array([audio_size]) * (1500 / 3000)).astype(int)[0])
    total = int(np.floor(np.array([s1]) / pool).astype(int)[0])
    nxt = 0
    for r in range(world):
        sh = S.video_shard(0, C, world, r)
        tok0, n = S.audio_shard_tokens(sh.c0, sh.windows, rows, pool, total)
        assert n >= 0
        if n:
            assert tok0 == nxt
            nxt = tok0 + n
    assert nxt == total
    with pytest.raises(ValueError):
        S.audio_shard_tokens(0, 1, 1501, 5, 10)


def test_packed_partial_layout():
    nkv, R, hd = 8, 78, 256
    tot = S.packed_partial_floats(2, nkv, R, hd)
    o0, m0 = S.packed_offsets(0, nkv, R, hd)
    o1, m1 = S.packed_offsets(1, nkv, R, hd)
    assert (o0, m0, o1, m1) == (0, nkv * R * hd, nkv * R * (hd + 2), nkv * R * (hd + 2) + nkv * R * hd)
    assert m1 + nkv * R * 2 == tot
    assert tot * 4 == 2 * 8 * 78 * 258 * 4                      # 1.29 MB per rank per layer at the 39-token prefill; 33 KB at decode


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, frames, windows, audio_size, ret, over=None):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from oracle_engine import OracleEngine
    from util import seeded
    from vidi_amd.config import tiny
    from vidi_amd.model import VidiForCausalLM
    from vidi_amd.weights import init_random_weights
    torch.set_num_threads(2)
    cfg = tiny(**(over or {}))
    w = init_random_weights(cfg, seed=3, dtype=torch.float32, device="cpu")
    eng = OracleEngine(cfg, w)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        eng.set_dist(None)
    model = VidiForCausalLM(cfg, w, dtype=torch.float32, device="cpu", engine=eng)
    px = seeded((abs(frames), 3, cfg.vis_image_size, cfg.vis_image_size), 200, 0.5).clamp(-1, 1)
    mel = seeded((windows, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 201, 0.3)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24, 25]], dtype=torch.int64)
    if frames < 0:
        # BASELINE configs[4] shape on the host logic: 8 ragged prompts (right-padded + attention mask) answered together against ONE
        # sharded video, greedy and sampled (every rank seeds its own RNG differently: rank 0's draw must reach all of them)
        g = torch.Generator().manual_seed(5)
        lens = [5, 6, 7, 8, 9, 10, 11, 12]
        bids = torch.randint(20, cfg.vocab_size - 1, (8, max(lens)), generator=g)
        bids[:, 0], bids[:, 3] = 2, -200
        bmask = torch.arange(max(lens))[None, :] < torch.tensor(lens)[:, None]
        mm = model.encode_mm_state([px], [mel], [audio_size])
        out = model.generate(bids, mm_state=mm, attention_mask=bmask, max_new_tokens=4, do_sample=False)
        smp = model.generate(bids[:2], mm_state=mm, attention_mask=bmask[:2], max_new_tokens=4, do_sample=True, top_k=5,
                             generator=torch.Generator().manual_seed(1000 + rank))
        # beam search over the sharded video: the beams' text caches are re-gathered on every rank alike (replicated text stream)
        bm = model.generate(bids[:2], mm_state=mm, attention_mask=bmask[:2], max_new_tokens=4, num_beams=3, num_return_sequences=2)
        ret.put((rank, out.tolist(), smp.tolist(), bm.tolist()))
    else:
        out = model.generate(ids, images=[px], audios=[mel], audio_sizes=[audio_size], max_new_tokens=5, do_sample=False)
        ret.put((rank, out.tolist(), getattr(eng, "last_shard", None)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, frames, windows, audio_size, over=None, timeout=240):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, frames, windows, audio_size, ret, over)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(ret.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,frames", [(2, 5), (3, 2)])
def test_generate_under_set_dist_shards_the_video_and_reproduces_single_rank(world, frames):
    """world 3 with 2 frames leaves rank 2 without a frame (empty shard); 2 windows over 3 ranks likewise"""
    windows, audio_size = 2, 173
    ref = _run(1, frames, windows, audio_size)[0][1]
    got = _run(world, frames, windows, audio_size)
    for rank, toks, sh in got:
        assert toks == ref, (rank, toks, ref)
        f0, f1 = S.shard(frames, world, rank)
        assert sh == dict(kind="img", local=f1 - f0, off=f0, total=frames)          # what the product handed this rank's engine


@pytest.mark.parametrize("windows,clip", [(120, 37), (60, 0)], ids=["configs3_60min_1fps", "configs4_30min_2fps"])
def test_world8_with_the_baseline_partition_shapes(windows, clip):
    """BASELINE configs[3] / [4] on the host logic, WORLD 8: 3 600 frames -> 450 per rank; 120 windows -> 15 per rank with the last
    window clipped by the global floors (multimodal.py:226-235: rank 7 owns fewer audio tokens than its windows hold), 60 windows ->
    8, 8, 8, 8, 7, 7, 7, 7.  Small towers (42-px frames, 4 tokens each) keep it a host-logic test: the token-budget rule sees the GLOBAL
    3 600 frames (base 1 000 -> the resize branch, as the 60-min config takes it), pos_t the global indices, and `generate()` on every one
    of the 8 ranks must return the single-rank tokens."""
    over = dict(vis_image_size=42, mm_max_tokens_base=1000)
    frames, audio_size = 3600, windows * 100 - clip
    ref = _run(1, frames, windows, audio_size, over, timeout=600)[0][1]
    got = _run(8, frames, windows, audio_size, over, timeout=600)
    assert len(got) == 8
    for rank, toks, sh in got:
        assert toks == ref, (rank, toks, ref)
        assert sh == dict(kind="img", local=450, off=450 * rank, total=3600)
    # the audio-token ranges of the 8 window shards, as the product computes them (50 encoder rows per window, pool 5)
    s2_total = (audio_size * 50 // 100) // 5
    owned = [S.audio_shard_tokens(S.shard(windows, 8, r)[0], S.shard(windows, 8, r)[1] - S.shard(windows, 8, r)[0], 50, 5, s2_total) for r in range(8)]
    assert sum(n for _, n in owned) == s2_total and all(owned[i][0] + owned[i][1] == owned[i + 1][0] for i in range(7))
    if clip:
        assert owned[7][1] < 15 * 10                                 # the last rank's windows are clipped


def test_batch_of_eight_ragged_queries_under_set_dist():
    """8 prompts of different lengths share one sharded video (world 2): greedy tokens equal the single-rank run on every rank, and a
    SAMPLED generation ends with identical tokens on both ranks although their RNG states differ (rank 0's draw is broadcast); beam
    search (3 beams, 2 returned per prompt) returns the single-rank sequences on every rank."""
    ref = _run(1, -5, 2, 173)[0]
    got = _run(2, -5, 2, 173)
    for rank, toks, smp, bm in got:
        assert toks == ref[1], (rank, toks, ref[1])
        assert bm == ref[3] and len(bm) == 4, (rank, bm, ref[3])        # beam search: the single-rank sequences on every rank
    assert got[0][2] == got[1][2], (got[0][2], got[1][2])


# ---------------------------------------------------------------------------------------------------------------------------------
# dist mode "gather_tokens": the north-star's literal collective (BASELINE configs[3]) — frame-sharded encode + all-gather of the tokens
# ---------------------------------------------------------------------------------------------------------------------------------
def _gather_worker(rank, world, port, frames, windows, audio_size, ret, two_videos):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from oracle_engine import OracleEngine
    from util import seeded
    from vidi_amd.config import tiny
    from vidi_amd.model import VidiForCausalLM
    from vidi_amd.weights import init_random_weights
    torch.set_num_threads(1)
    cfg = tiny()
    w = init_random_weights(cfg, seed=3, dtype=torch.float32, device="cpu")
    eng = OracleEngine(cfg, w)
    eng.per_unit = True                      # frames / windows one at a time on every world size: bit-comparable (see OracleEngine)
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        eng.set_dist(None, mode="gather_tokens")
    model = VidiForCausalLM(cfg, w, dtype=torch.float32, device="cpu", engine=eng)
    px = seeded((frames, 3, cfg.vis_image_size, cfg.vis_image_size), 200, 0.5).clamp(-1, 1)
    mel = seeded((windows, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 201, 0.3)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24, 25]], dtype=torch.int64)
    fi, mi, fa, ma = model.encode_videos([px], [mel], [audio_size])
    rec = dict(rank=rank, fi=fi.numpy(), mi=mi.numpy(), fa=fa.numpy(), ma=ma.numpy(), gathers=getattr(eng, "n_token_gathers", 0))
    rec["tokens"] = model.generate(ids, images=[px], audios=[mel], audio_sizes=[audio_size], max_new_tokens=4, do_sample=False).tolist()
    if two_videos:
        # a batch of two videos of different lengths (one prompt each): every video is sharded on its own, the token budget is the batch's
        px2 = seeded((3, 3, cfg.vis_image_size, cfg.vis_image_size), 202, 0.5).clamp(-1, 1)
        mel2 = seeded((1, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 203, 0.3)
        b = model.encode_videos([px, px2], [mel, mel2], [audio_size, 90])
        rec["batch"] = [t.numpy() for t in b]
    ret.put(rec)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_gather(world, frames, windows, audio_size, two_videos=False, timeout=300):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, frames, windows, audio_size, ret, two_videos)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((ret.get(timeout=timeout) for _ in range(world)), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("world,frames,windows,audio_size,two", [(2, 5, 2, 173, True), (3, 2, 2, 173, False), (8, 11, 3, 273, False)],
                         ids=["world2_ragged+batch", "world3_empty_shards", "world8_ragged_clipped"])
def test_gather_tokens_mode_reproduces_single_rank_encode_videos_bit_for_bit(world, frames, windows, audio_size, two):
    """`encode_videos` under `set_dist(mode="gather_tokens")`: every rank encodes its frame / window range with the global positions, the
    tokens + masks are all-gathered (ragged shards padded for the collective, narrowed after) and EVERY rank must hold the single-rank
    tensors bit for bit — features and masks of both modalities, reference order (multimodal.py:254-265).  world 3 with 2 frames / 2
    windows leaves rank 2 with empty shards; world 8 with 11 frames is ragged (2,2,2,1,...) and 3 windows leave five ranks without audio,
    the last window clipped by the global floors.  generate() in that mode returns the single-rank tokens (decoder replicated)."""
    ref = _run_gather(1, frames, windows, audio_size, two)[0]
    got = _run_gather(world, frames, windows, audio_size, two)
    assert len(got) == world
    for r in got:
        for k in ("fi", "mi", "fa", "ma"):
            assert r[k].shape == ref[k].shape and r[k].dtype == ref[k].dtype, (k, r[k].shape, ref[k].shape)
            assert np.array_equal(r[k], ref[k]), f"rank {r['rank']}: {k} differs from the single-rank encode"
        assert r["tokens"] == ref["tokens"]
        assert r["gathers"] >= 4                                      # features + mask per modality, per encode
        if two:
            for a, b in zip(r["batch"], ref["batch"]):
                assert np.array_equal(a, b)
    assert ref["gathers"] == 0
    # ... and the per-unit evaluation those bits come from IS the oracle's encode (the batched restatement of multimodal.py:156-252)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import vidi_oracle as O
    from oracle_engine import oracle_config
    from util import seeded
    from vidi_amd.config import tiny
    from vidi_amd.weights import init_random_weights
    cfg = tiny()
    w = {k: v.float() for k, v in init_random_weights(cfg, seed=3, dtype=torch.float32, device="cpu").items()}
    px = seeded((frames, 3, cfg.vis_image_size, cfg.vis_image_size), 200, 0.5).clamp(-1, 1)
    mel = seeded((windows, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 201, 0.3)
    f, m = O.encode_video_images([px], w, oracle_config(cfg))
    fa, ma = O.encode_video_audios([mel], [audio_size], w, oracle_config(cfg))
    assert np.array_equal(m.numpy(), ref["mi"]) and np.array_equal(ma.numpy(), ref["ma"])
    np.testing.assert_allclose(ref["fi"], f.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ref["fa"], fa.numpy(), rtol=1e-4, atol=1e-5)


def test_all_gather_rows_ragged_and_even():
    """vidi_amd/dist.py on gloo, world 3: ragged row counts incl. an empty shard, 1-D masks, and the even fast path"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, 3, port, ret)) for r in range(3)]
    for p in procs:
        p.start()
    got = [ret.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got)


def _rows_worker(rank, world, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import torch.distributed as dist
    from vidi_amd.dist import all_gather_rows
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for counts in ([3, 0, 2], [2, 2, 2], [0, 0, 0], [1, 4, 4]):
        full = torch.arange(sum(counts) * 5, dtype=torch.float32).view(sum(counts), 5)
        offs = [sum(counts[:r]) for r in range(world)]
        out = all_gather_rows(full[offs[rank]: offs[rank] + counts[rank]].clone(), counts, None)
        ok &= torch.equal(out, full)
        m = (torch.arange(sum(counts)) % 3).to(torch.uint8)
        ok &= torch.equal(all_gather_rows(m[offs[rank]: offs[rank] + counts[rank]].clone(), counts, None), m)
    try:
        all_gather_rows(torch.zeros(2, 5), [1, 1, 1], None)                 # a rank whose piece disagrees with the partition
        ok = False
    except ValueError:
        pass
    ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_split_key_slices_of_the_dual_cross_attention_launch():
    """host logic of engine._cross_dual: slices in proportion to the keys, never empty, never more than the launch has, short modalities
    capped at one slice per 8 sub-tiles"""
    from vidi_amd.shard import split_key_slices
    assert split_key_slices(32, 2813, 1125) == (23, 9)                 # the 60-min config at decode: 90 000 / 36 000 keys
    assert split_key_slices(10, 2813, 1125) == (7, 3)                  # ... at the 39-token prompt (3 row tiles)
    for slices in (2, 3, 10, 32, 64):
        for a in (1, 7, 8, 9, 100, 2813, 30000):
            for b in (1, 5, 64, 1125, 12000):
                za, zb = split_key_slices(slices, a, b)
                assert za >= 1 and zb >= 1 and za + zb <= slices
                assert za <= max(1, (a + 7) // 8) and zb <= max(1, (b + 7) // 8)
                if a >= 8 * slices and b >= 8 * slices:                # both long: every slice is used, and the longer slice of
                    assert za + zb == slices                           # the two modalities is within one slice of a perfect split
                    per = lambda n, z: -(-n // (4 * z))
                    ideal = -(-(a + b) // (4 * slices))
                    assert max(per(a, za), per(b, zb)) <= ideal * (1 + 1.0 / min(za, zb)) + 1
    import pytest
    with pytest.raises(ValueError):
        split_key_slices(1, 10, 10)
    with pytest.raises(ValueError):
        split_key_slices(8, 0, 10)



def test_dist_plan_reproduces_the_design_figures():
    """tools/dist_plan.py (the expectation table for the first real multi-GPU run) computes the 8-rank partition of the 60-min configuration
    with the product's own host logic and must land on the figures DESIGN.md section 6 quotes; other configurations tile without gaps."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dist_plan
    dist_plan.check_design_figures()
    for world, frames, fps, q in ((8, 3600, 2.0, 8), (4, 3600, 1.0, 1), (2, 300, 1.0, 1), (8, 301, 1.0, 1), (3, 25, 1.0, 1)):
        p = dist_plan.plan(world, frames, fps, 39, q)
        assert p["ranks"][-1]["video_tokens"][1] == p["video_tokens"] and p["ranks"][-1]["audio_tokens"][1] == p["audio_tokens"]
        assert sum(r["frames"][1] - r["frames"][0] for r in p["ranks"]) == frames
