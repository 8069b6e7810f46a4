"""Parity at FULL DEPTH and REAL DIMS — the configuration the bench runs, where bf16 error compounds over the layers:

* SigLIP-so400m tower as Vidi1.5 uses it (H 1152, I 4304, 16 heads x 72, 729 tokens, 26 of 27 layers -> hidden_states[-2],
  mm_vision/siglip.py:29-34; HF modeling_siglip.py:250-357), both arms of the LayerNorm fold;
* Whisper-large-v3 encoder (d 1280, 20 heads x 64, ffn 5120, 1 500 positions, 32 layers; mm_audio/whisper.py:26-27);
* the D-Attn decoder at Gemma2-9B dims, ALL 42 layers (gemma.py:125-244, 267-424), at BASELINE configs[2]'s sizes (90 000 image +
  36 000 audio keys): K/V caches of layers 0, 1, 20 and 41, the 39-token text prefill and four teacher-forced decode steps.

Frames / windows / stream rows never interact inside their towers / the diagonal stream, so the GPU runs enough of them to reach the
production kernels (the persistent GEMM serves >= 192 tiles) and the fp32 CPU oracle is evaluated on a SAMPLE of them; for the decoder
the key masks make the restriction exact: every key outside the sample is masked on the GPU, so the text stream attends to exactly the
keys the oracle holds.  bf16 (the bench dtype); the bound of every check is written where it is made.

TWO oracle arms per check.  (1) fp32 oracle: the distance to exact arithmetic — dominated by bf16 rounding NOISE that the reference
itself has (drift report; 5-sigma bounds from the measured noise growth).  (2) SAME-ROUNDING oracle: `oracle/vidi_oracle.py` is
dtype-generic, so fed bf16 weights and inputs it rounds where the reference's eager modules round (every nn.Module output to bf16,
norms in fp32 inside).  Against that arm the noise cancels and what is left is the kernels' own deviation (different summation
order -> occasional one-ulp flips of an output, the documented folds: LayerNorm / softmax scale folded into weights, o_proj over
the summed repeat_kv column blocks), so its bounds are several times tighter and a real kernel error of a few % at depth fails."""
import os

import numpy as np
import pytest
import torch

import vidi_oracle as O
from util import perm_positions, report

pytestmark = pytest.mark.gpu


class LazyF32:
    """state dict view whose tensors are converted to fp32 on the host when the oracle asks for them (a 9B-parameter fp32 copy would
    take 33 GB of host memory; one layer's projections are 0.8 GB)"""

    def __init__(self, w):
        self.w = w

    def __getitem__(self, k):
        return self.w[k].float()

    def __contains__(self, k):
        return k in self.w

    def get(self, k, default=None):
        return self.w[k].float() if k in self.w else default


# Bounds of the same-rounding arm (fraction of the reference's spread, relative part); set from the measured use of each
# (profiles/r4_tolerance_audit.jsonl), about 2x the observed maximum.
SAME_ROUNDING = {
    "siglip": (4e-2, 2e-2), "siglip_rms": 1.5e-2,
    "whisper": (4e-2, 2e-2), "whisper_rms": 1.5e-2,
    "kv": (2e-2, 1e-2), "kv_rms": 5e-3,
    "hidden": (6e-2, 2e-2), "hidden_rms": 1.5e-2,
}


def _tower_cfg(**over):
    from vidi_amd.config import tiny
    return tiny(**over)


@pytest.mark.parametrize("fold", ["1", "1u", "0"], ids=["ln_fold", "ln_fold_unscaled_q", "ln_plain"])
def test_siglip_tower_real_dims_full_depth(fold, monkeypatch):
    """(ln_fold: the default arm — LayerNorms folded, softmax scale folded into q, maximum inside the contraction; ln_fold_unscaled_q:
    VIDI_ATTN_PRESCALE=0; ln_plain: no folds)"""
    from test_gpu_model import make, oracle_cfg
    dt = torch.bfloat16
    if fold == "1u":
        monkeypatch.setenv("VIDI_ATTN_PRESCALE", "0"); fold = "1"
    monkeypatch.setenv("VIDI_LN_FOLD", fold)
    cfg = _tower_cfg(vis_image_size=384, vis_patch_size=14, vis_hidden_size=1152, vis_intermediate_size=4304, vis_num_layers=27,
                     vis_num_heads=16, vis_frames_per_chunk=16)
    eng, w32 = make(cfg, dt, seed=11)
    assert eng.ln_fold == (fold == "1") and cfg.vis_select_layers == 26 and cfg.vis_side ** 2 == 729
    T = 16                                                  # 11 664 rows: every projection of the tower takes the persistent GEMM
    g = torch.Generator().manual_seed(300)
    px = (torch.randn((T, 3, 384, 384), generator=g) * 0.5).clamp(-1, 1).to(dt)
    got = eng.siglip_forward(px.cuda())
    assert got.shape == (T, 729, 1152)
    sample = [0, 15]                                        # first / last frame of the chunk (first and last row tiles)
    ref = O.siglip_forward(px[sample].float(), w32, oracle_cfg(cfg))
    # 26 layers of bf16 residual-stream roundings against the fp32 oracle: 7 % of the spread + 4 % relative (1.06 M values; 0.7 used)
    report(f"siglip real dims x26 layers (fold={fold})", got[sample], ref, 7e-2 * ref.std().item(), 4e-2)
    # same-rounding arm: the oracle in bf16 with the eager rounding points (HF modeling_siglip.py:310-357 executed in bf16)
    w16 = {k: v.to(dt) for k, v in w32.items()}
    ref16 = O.siglip_forward(px[sample], w16, oracle_cfg(cfg)).float()
    report(f"siglip real dims x26 layers (fold={fold}) vs the bf16-rounding oracle", got[sample], ref16, SAME_ROUNDING["siglip"][0] * ref16.std().item(),
           SAME_ROUNDING["siglip"][1])
    rms = float((got[sample].float().cpu() - ref16).pow(2).mean().sqrt() / ref16.pow(2).mean().sqrt())
    rms32 = float((got[sample].float().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    rms_ref = float((ref16 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    report(f"siglip x26 (fold={fold}) rms error / rms vs bf16 oracle [vs fp32: {rms32:.4f}; bf16 oracle vs fp32 oracle: {rms_ref:.4f}]",
           torch.tensor([rms]), torch.tensor([0.0]), SAME_ROUNDING["siglip_rms"], 0.0)


def test_whisper_encoder_real_dims_full_depth():
    from test_gpu_model import make, oracle_cfg
    dt = torch.bfloat16
    cfg = _tower_cfg(aud_num_mel_bins=128, aud_d_model=1280, aud_num_layers=32, aud_num_heads=20, aud_ffn_dim=5120,
                     aud_max_source_positions=1500, aud_nb_max_frames=3000, aud_chunks_per_batch=8)
    eng, w32 = make(cfg, dt, seed=12)
    C = 8                                                   # 12 000 rows: the persistent GEMM on every projection
    g = torch.Generator().manual_seed(301)
    mel = (torch.randn((C, 128, 3000), generator=g) * 0.3).to(dt)
    got = eng.whisper_forward(mel.cuda())
    assert got.shape == (C, 1500, 1280)
    sample = [7]
    ref = O.whisper_encoder_forward(mel[sample].float(), w32, oracle_cfg(cfg))
    # 32 layers: 8 % of the spread + 4 % relative (1.9 M values; 0.7 used)
    report("whisper real dims x32 layers", got[sample], ref, 8e-2 * ref.std().item(), 4e-2)
    w16 = {k: v.to(dt) for k, v in w32.items()}
    ref16 = O.whisper_encoder_forward(mel[sample], w16, oracle_cfg(cfg)).float()
    report("whisper real dims x32 layers vs the bf16-rounding oracle", got[sample], ref16, SAME_ROUNDING["whisper"][0] * ref16.std().item(),
           SAME_ROUNDING["whisper"][1])
    rms = float((got[sample].float().cpu() - ref16).pow(2).mean().sqrt() / ref16.pow(2).mean().sqrt())
    rms32 = float((got[sample].float().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    rms_ref = float((ref16 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    report(f"whisper x32 rms error / rms vs bf16 oracle [vs fp32: {rms32:.4f}; bf16 oracle vs fp32 oracle: {rms_ref:.4f}]",
           torch.tensor([rms]), torch.tensor([0.0]), SAME_ROUNDING["whisper_rms"], 0.0)


def _unpack_rows(mm, li, rows, nkv, hd):
    rows_t = torch.as_tensor(rows, dtype=torch.int64, device=mm.kc.device)
    k = mm.kc[li].reshape(nkv, -1, hd)[:, rows_t].permute(1, 0, 2).reshape(len(rows), nkv * hd)
    pos = torch.as_tensor(perm_positions(32)[np.asarray(rows) & 31], dtype=torch.int64, device=mm.kc.device)
    v = mm.vtc[li][:, rows_t >> 5, :, pos]
    return k.float().cpu(), v.reshape(len(rows), nkv * hd).float().cpu()


def test_decoder_42_layers_real_dims_at_the_60_min_sizes():
    from test_gpu_model import oracle_cfg
    from vidi_amd.config import tiny
    from vidi_amd.engine import VidiEngine, _round_up
    from vidi_amd.model import strip_image_token
    from vidi_amd.weights import init_random_weights
    dt = torch.bfloat16
    cfg = tiny(hidden_size=3584, intermediate_size=14336, num_attention_heads=16, num_key_value_heads=8, head_dim=256,
               query_pre_attn_scalar=256.0, sliding_window=4096, num_hidden_layers=42, vocab_size=1024)
    w = init_random_weights(cfg, seed=5, dtype=dt, device="cuda")              # 8.3 G parameters: drawn on the GPU
    w_host = {k: v.cpu() for k, v in w.items()}                                # bf16 host copy for the oracle (16.6 GB)
    eng = VidiEngine(cfg, w, dtype=dt, device="cuda", free_source=True)
    del w
    torch.cuda.empty_cache()
    wl = LazyF32(w_host)
    ocfg = oracle_cfg(cfg)
    H, nkv, hd = cfg.hidden_size, cfg.num_key_value_heads, cfg.head_dim
    Nv, Na = 90000, 36000
    g = torch.Generator(device="cuda").manual_seed(123)
    img = (torch.randn((Nv, H), generator=g, device="cuda") * cfg.mm_std).to(dt)      # un-normalised features (gemma.py:353-356 scales them)
    aud = (torch.randn((Na, H), generator=g, device="cuda") * cfg.mm_std).to(dt)
    # the SAMPLE: first / last rows, 64-row tile edges, the modality boundary, and random rows; everything else is masked
    rs = np.random.RandomState(7)
    img_rows = sorted(set([0, 1, 31, 32, 63, 64, 65, 4095, 4096, 44999, 89983, 89984, 89998, 89999] + rs.randint(0, Nv, 1010).tolist()))
    aud_rows = sorted(set([0, 1, 63, 64, 35967, 35968, 35998, 35999] + rs.randint(0, Na, 248).tolist()))
    imask = torch.zeros(Nv, dtype=torch.uint8, device="cuda"); imask[torch.as_tensor(img_rows, device="cuda")] = 1
    amask = torch.zeros(Na, dtype=torch.uint8, device="cuda"); amask[torch.as_tensor(aud_rows, device="cuda")] = 1
    mm = eng.mm_stream_prefill(img, imask, aud, amask, pre_normalized=False)
    aud_start = _round_up(Nv, 64)
    assert mm.ntile64 * 64 == 126080 and mm.img_mask is not None and mm.aud_mask is not None

    # ---- GPU: 39-token prompt + 4 teacher-forced decode steps over the 42-layer caches ----
    gi = torch.Generator().manual_seed(2)
    ids = torch.randint(10, cfg.vocab_size, (1, 40), generator=gi)
    ids[0, 0], ids[0, 4] = cfg.bos_token_id, -200
    forced = [11, 12, 13, 14]
    idt, mask, pos = strip_image_token(ids)
    L = idt.shape[1]
    ts = eng.new_text_state(1, L + len(forced) + 1)
    hn = eng.text_forward(eng.embed_tokens(idt.cuda()), pos.reshape(-1).cuda(), ts, mm, Lq=L, new_mask=mask.cuda()).float().cpu()
    dec = []
    for i, t in enumerate(forced):
        nxt = torch.tensor([t], dtype=torch.int64, device="cuda")
        dec.append(eng.text_forward(eng.embed_tokens(nxt), torch.tensor([L + i], device="cuda"), ts, mm, Lq=1).float().cpu())

    # ---- oracle on the sampled keys only (all valid there) ----
    xi = img[torch.as_tensor(img_rows, device="cuda")].float().cpu()[None]
    xa = aud[torch.as_tensor(aud_rows, device="cuda")].float().cpu()[None]
    mi = torch.ones((1, len(img_rows)), dtype=torch.bool); ma = torch.ones((1, len(aud_rows)), dtype=torch.bool)
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    idl, am, opos = O.strip_image_token(ids)
    emb = O.embed_text(idl, am, wl)
    caches = O.OracleCaches()
    href = O.model_forward(emb, opos, am, xi, mi, xa, ma, wl, ocfg, caches, 0)

    failures = []

    def check(*a):
        try:                                                    # every comparison is evaluated (and audited) before the test fails
            report(*a)
        except AssertionError as e:
            failures.append(str(e))

    def check_rms(name, got, ref, frac):
        """root-mean-square error as a fraction of the reference's rms: the stable statistic behind the 5-sigma element bounds"""
        g, r = got.float().cpu(), ref.float().cpu()
        e = float((g - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
        check(name + " [rms error / rms]", torch.tensor([e]), torch.tensor([0.0]), frac, 0.0)

    # K/V caches of the sampled rows: one bf16 GEMM over K = 3 584 of the (li times updated, bf16-rounded) stream rows against the fp32
    # oracle.  Every element bound below is a 5-sigma bound over 2.1 M (image) / 0.5 M (audio) values; the error is bf16 rounding noise
    # that accumulates in the residual stream, in quadrature over the layers: 2.4 % of the spread at layer 0, +2.6 % per stream update
    # (o_proj, two norm pairs, GeGLU, down_proj, each rounded to bf16) -> 3.5 % at layer 1, 11.9 % at layer 20, 16.8 % at layer 41
    # (measured maxima: 1.9 / 2.7 / 9.3 / 12.7 %), + 1.5 % relative.  The rms error stays 5x lower (checked against 1/4 of the bound).
    for li in (0, 1, 20, 41):
        for name, rows, start, cache in (("image", img_rows, 0, caches.image), ("audio", aud_rows, aud_start, caches.audio)):
            kg, vg = _unpack_rows(mm, li, [start + r for r in rows], nkv, hd)
            kref, vref = cache[li]
            a = (2.4e-2 ** 2 + li * 2.6e-2 ** 2) ** 0.5
            check(f"42-layer stream: layer {li} {name} K rows", kg, kref[0], a * kref.std().item(), 1.5e-2)
            check(f"42-layer stream: layer {li} {name} V rows", vg, vref[0], a * vref.std().item(), 1.5e-2)
            check_rms(f"42-layer stream: layer {li} {name} K rows", kg, kref[0], a / 4)
    # text hidden states after 42 layers of T2T + T2V + T2A (final norm applied), 140 k values: the same noise, through the cross-attention
    # over 42 layers of caches: rms error 3.2 % of the rms (bound 5 %), 5-sigma element bound 21 % of the spread + 5 % relative
    check("42-layer text prefill hidden (39 tokens)", hn, href[0], 21e-2 * href.std().item(), 5e-2)
    check_rms("42-layer text prefill hidden (39 tokens)", hn, href[0], 5e-2)
    tm = am
    for i, t in enumerate(forced):
        e = torch.nn.functional.embedding(torch.tensor([[t]]), wl["model.embed_tokens.weight"])
        tm = torch.cat([tm, torch.ones(1, 1, dtype=torch.bool)], dim=1)
        r = O.model_forward(e, torch.tensor([[L + i]]), tm, xi, mi, xa, ma, wl, ocfg, caches, L + i)
        check(f"42-layer teacher-forced decode step {i}", dec[i], r[0], 21e-2 * r.std().item(), 5e-2)
        check_rms(f"42-layer teacher-forced decode step {i}", dec[i], r[0], 5e-2)
    # ---- same-rounding arm: the oracle again, in bf16 with the eager rounding points, on the same sampled keys ----
    class Lazy16:
        def __init__(self, w): self.w = w
        def __getitem__(self, k): return self.w[k]
        def __contains__(self, k): return k in self.w
        def get(self, k, default=None): return self.w.get(k, default)

    w16 = Lazy16(w_host)
    caches16 = O.OracleCaches()
    emb16 = O.embed_text(idl, am, w16)
    href16 = O.model_forward(emb16, opos, am, xi.to(dt), mi, xa.to(dt), ma, w16, ocfg, caches16, 0).float()
    kvb, kvr = SAME_ROUNDING["kv"]
    for li in (0, 1, 20, 41):
        for name, rows, start, cache, c32 in (("image", img_rows, 0, caches16.image, caches.image), ("audio", aud_rows, aud_start, caches16.audio, caches.audio)):
            kg, vg = _unpack_rows(mm, li, [start + r for r in rows], nkv, hd)
            kref, vref = cache[li][0].float(), cache[li][1].float()
            check(f"42-layer stream vs bf16-rounding oracle: layer {li} {name} K rows", kg, kref[0], kvb * kref.std().item(), kvr)
            check(f"42-layer stream vs bf16-rounding oracle: layer {li} {name} V rows", vg, vref[0], kvb * vref.std().item(), kvr)
            check_rms(f"42-layer stream vs bf16-rounding oracle: layer {li} {name} K rows", kg, kref[0], SAME_ROUNDING["kv_rms"])
            # how far the two oracle arms are from each other (the noise the fp32 comparison above has to allow for)
            d = float((kref[0] - c32[li][0][0].float()).abs().max() / c32[li][0].float().std())
            print(f"[drift] layer {li} {name}: bf16-rounding oracle vs fp32 oracle, worst K element = {100 * d:.2f} % of the spread")
    hb, hr = SAME_ROUNDING["hidden"]
    check("42-layer text prefill hidden (39 tokens) vs bf16-rounding oracle", hn, href16[0], hb * href16.std().item(), hr)
    check_rms("42-layer text prefill hidden (39 tokens) vs bf16-rounding oracle", hn, href16[0], SAME_ROUNDING["hidden_rms"])
    tm = am
    for i, t in enumerate(forced):
        e = torch.nn.functional.embedding(torch.tensor([[t]]), w16["model.embed_tokens.weight"])
        tm = torch.cat([tm, torch.ones(1, 1, dtype=torch.bool)], dim=1)
        r = O.model_forward(e, torch.tensor([[L + i]]), tm, xi.to(dt), mi, xa.to(dt), ma, w16, ocfg, caches16, L + i).float()
        check(f"42-layer teacher-forced decode step {i} vs bf16-rounding oracle", dec[i], r[0], hb * r.std().item(), hr)
        check_rms(f"42-layer teacher-forced decode step {i} vs bf16-rounding oracle", dec[i], r[0], SAME_ROUNDING["hidden_rms"])
    assert not failures, "\n".join(failures)
