"""Run-to-run BITWISE reproducibility of the hot entry points under concurrency ("race detection").

The kernels order their LDS rings, DMA queues and accumulator hand-overs by hand-counted `s_waitcnt vmcnt(n)` / barriers instead of
compiler-inserted full drains (gemm_w4.h, attn_self_rm.hip, attn_cross_rows.hip, gemv_mfma.hip).  A wait that is one too loose does not crash:
it reads a tile a few cycles early — sometimes — and perturbs a few outputs without moving any tolerance-level check.  What it cannot do
is reproduce the same bits 50 times while a second stream competes for the same CUs, LDS and memory pipes.  So: two engines (same weights),
one per HIP stream, run the whole hot path AT THE REAL DIMENSIONS 25 times each, concurrently — towers (SigLIP d = 72 / N = 729, Whisper),
the diagonal stream + K/V cache fill over 20 000 keys, the text prefill of one prompt and of 8 ragged prompts (many-row cross-attention),
single-row and 8-row decode steps (split-KV cross-attention, weight-streaming projections, lm_head) — and every one of the 50 results must
equal the first bit for bit (exact integer checksums of the raw bits, position-weighted).  bench.py applies the same rule to the
first-token logits of every timed step at BASELINE size."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def bits(t: torch.Tensor) -> torch.Tensor:
    """two exact integer checksums of a tensor's raw bits (order-independent sums: deterministic themselves); int64[2] on the device"""
    raw = t.contiguous().view(-1)
    if raw.element_size() == 2:
        v = raw.view(torch.int16).to(torch.int64)
    elif raw.element_size() == 4:
        v = raw.view(torch.int32).to(torch.int64)
    elif raw.element_size() == 1:
        v = raw.view(torch.uint8).to(torch.int64)
    else:
        v = raw.view(torch.int64)
    w = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521 + 1
    return torch.stack([v.sum(), (v * w).sum()])


def hot_path(model, px, mel, feats, ids1, ids8, am8):
    """one pass over the hot entry points; -> {name: checksum}"""
    from vidi_amd.model import strip_image_token
    eng = model.engine
    out = {}
    out["siglip_forward"] = bits(eng.siglip_forward(px))
    out["whisper_forward"] = bits(eng.whisper_forward(mel))
    fi, mi = eng.encode_video_images(px, normalizer=eng.normalizer)
    fa, ma = eng.encode_video_audios(mel, 1000, normalizer=eng.normalizer)
    out["video_tokens"], out["audio_tokens"] = bits(fi), bits(fa)
    # the decoder over MANY keys (the towers above give 392 + 100): 20 000 + 5 000 rows of token embeddings
    mm = eng.mm_stream_prefill(feats[:20000], None, feats[20000:], None, pre_normalized=True, check_masks=False)
    out["k_cache"], out["v_cache"] = bits(mm.kc), bits(mm.vtc)
    for name, ids, am, steps in (("1 prompt", ids1, None, 3), ("8 ragged prompts", ids8, am8, 3)):
        idt, mask, pos = strip_image_token(ids, am)
        ts, last = model._prefill(idt, mask, pos, mm, steps + 1)
        out[f"prefill hidden ({name})"] = bits(last)
        logits, nxt = eng.logits_argmax(last)
        out[f"prefill logits ({name})"] = bits(logits)
        for s in range(steps):
            emb = eng.embed_tokens(nxt)
            posn = ts.n_valid.clone(); ts.n_valid += 1
            hn = eng.text_forward(emb, posn, ts, mm, Lq=1)
            logits, nxt = eng.logits_argmax(hn)
            out[f"decode step {s} logits ({name})"] = bits(logits)
    return out


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_fifty_concurrent_runs_on_two_streams_are_bit_identical(dt):
    import make_golden_realdims as MR
    from vidi_amd.model import VidiForCausalLM
    from vidi_amd.weights import init_random_weights
    cfg = MR.realdims_config()
    w = init_random_weights(cfg, seed=MR.WEIGHT_SEED, dtype=torch.float32, device="cpu")
    wt = {k: (v if ".mm_rand_pos_" in k else v.to(dt)) for k, v in w.items()}
    del w
    models = [VidiForCausalLM(cfg, dict(wt), dtype=dt, device="cuda") for _ in range(2)]        # one engine (workspaces!) per stream
    del wt
    px, mel, ids1 = MR.make_inputs(cfg)
    px, mel = px[0].to(dt).cuda(), mel[0].to(dt).cuda()
    g = torch.Generator().manual_seed(5)
    feats = (torch.randn((25000, cfg.hidden_size), generator=g) * 1.7).to(dt).cuda()             # ~ normalizer x mm_std-scaled rows
    lens = [24, 28, 32, 36, 40, 44, 48, 52]                                                       # BASELINE configs[4]'s ragged batch
    ids8 = torch.randint(10, cfg.vocab_size, (8, max(lens) + 1), generator=g)
    ids8[:, 0], ids8[:, 4] = cfg.bos_token_id, -200
    am8 = torch.arange(max(lens) + 1)[None, :] < (torch.tensor(lens)[:, None] + 1)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    results = []
    torch.cuda.synchronize()
    with torch.no_grad():
        for it in range(25):
            for m, s in zip(models, streams):                      # both streams hold queued work at the same time: the launches interleave on the GPU
                with torch.cuda.stream(s):
                    results.append(hot_path(m, px, mel, feats, ids1, ids8, am8))
    torch.cuda.synchronize()
    assert len(results) == 50
    first = {k: v.cpu() for k, v in results[0].items()}
    assert len(first) >= 16, sorted(first)
    bad = {}
    for i, r in enumerate(results[1:], 1):
        for k, v in r.items():
            if not torch.equal(v.cpu(), first[k]):
                bad.setdefault(k, []).append(i)
    assert not bad, f"results that differ from run 0 (entry point -> runs): {bad}"
