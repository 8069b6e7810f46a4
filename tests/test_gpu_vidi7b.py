"""Vidi-7B (Mistral D-Attn, Vidi_7B/model/lmm/dattn/mistral.py) on the GPU against the fp32 CPU oracle: the
kernels that only this model uses (SiLU-GLU epilogue, learned Conv2DPool as im2col + GEMM + align_corners
resize, plain RMSNorm) and the engine wiring (no post-norms / softcaps / normalizer, untied head)."""
import pytest
import torch
import torch.nn.functional as F

import vidi_oracle as O
from util import logit_tol, report, seeded
from test_gpu_model import make, oracle_cfg, tol as mtol

pytestmark = pytest.mark.gpu
DTYPES = [torch.bfloat16, torch.float16]


def ktol(dt, scale):
    """single kernels: 1 % of the spread + 1 % relative (bf16), 0.2 % / 0.2 % (fp16) — see test_gpu_kernels.tol"""
    return ((1e-2 if dt == torch.bfloat16 else 2e-3) * scale, 1e-2 if dt == torch.bfloat16 else 2e-3)


@pytest.fixture(scope="module")
def hip():
    from vidi_amd import hip as H
    H.load_library()
    return H


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M", [3, 200])
def test_glu_silu(hip, dt, M):
    I, K = 256, 128
    x = seeded((M, K), 31, dtype=dt); g = seeded((I, K), 32, 0.1, dtype=dt); u = seeded((I, K), 33, 0.1, dtype=dt)
    wgu = torch.stack([g.view(I // 32, 32, K), u.view(I // 32, 32, K)], dim=1).reshape(2 * I, K).contiguous()
    ref = F.silu(x.float() @ g.float().T) * (x.float() @ u.float().T)
    if M > 8:
        y = hip.gemm_glu(x.cuda(), wgu.cuda(), act=hip.ACT_SILU)
    else:                                                        # decode path: GEMV + unpack
        yp = hip.gemv(x.cuda(), wgu.cuda())
        y = hip.glu_unpack(yp, torch.empty((M, I), dtype=dt, device="cuda"), hip.ACT_SILU)
    report("glu silu", y, ref, *[2 * t for t in ktol(dt, ref.std().item())])    # fused activation: a second rounding inside


@pytest.mark.parametrize("dt", DTYPES)
def test_mistral_rmsnorm_is_norm_mm(hip, dt):
    x = seeded((37, 256), 34, 3.0, dtype=dt); w = (1.0 + seeded((256,), 35, 0.3)).to(dt)
    ref = O.mistral_rmsnorm(x, w, 1e-5)                           # same dtype => same rounding points
    y = hip.norm(hip.NORM_MM, x.cuda(), w.cuda(), eps=1e-5)
    report("mistral rmsnorm", y, ref.float(), *ktol(dt, 0.05))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("side,s_out,C", [(7, 2, 64), (27, 2, 64), (27, 3, 64)])
def test_learned_conv_pool(hip, dt, side, s_out, C):
    T = 3
    k = -(-side // s_out)
    oc = side - k + 1
    x = seeded((T, side * side, C), 36, dtype=dt)                 # NHWC tower features
    w = seeded((C, C, k, k), 37, (C * k * k) ** -0.5, dtype=dt)
    ref = O.learned_conv2d_pool(x.float().view(T, side, side, C).permute(0, 3, 1, 2), w.float(), s_out).permute(0, 2, 3, 1)
    col = torch.empty((T * oc * oc, k * k * C), dtype=dt, device="cuda")
    hip.im2col_nhwc(x.cuda(), col, T=T, side=side, C=C, k=k)
    w2 = w.permute(0, 2, 3, 1).reshape(C, -1).contiguous().cuda()
    conv = hip.gemm(col, w2, None)
    out = torch.empty((T * s_out * s_out, C), dtype=dt, device="cuda")
    hip.resize_bilinear_ac(conv, out, T=T, s_in=oc, s_out=s_out, C=C)
    report("learned conv pool", out.view(T, s_out, s_out, C), ref, *ktol(dt, ref.std().item()))


@pytest.fixture(scope="module", params=DTYPES, ids=["bf16", "fp16"])
def setup7b(request):
    from vidi_amd.config import tiny_7b
    cfg = tiny_7b()
    eng, w32 = make(cfg, request.param)
    return cfg, eng, w32, request.param


def test_7b_encode_video_images(setup7b):
    cfg, eng, w32, dt = setup7b
    assert eng.mistral and cfg.arch == "mistral"
    px = seeded((3, 3, cfg.vis_image_size, cfg.vis_image_size), 140, 0.5).clamp(-1, 1).to(dt)
    ref, rmask = O.encode_video_images([px.float()], w32, oracle_cfg(cfg))
    got, mask = eng.encode_video_images(px.cuda())
    assert got.shape[0] == 3 * cfg.mm_image_pool_size ** 2
    assert torch.equal(mask.bool().cpu(), rmask[0])
    atol, rtol = mtol(dt, ref.std().item())
    report("7b encode images", got, ref[0], atol, rtol)


def test_7b_prefill_logits_and_generate(setup7b):
    cfg, eng, w32, dt = setup7b
    from vidi_amd.model import VidiForCausalLM
    from types import SimpleNamespace
    ocfg = oracle_cfg(cfg)
    px = seeded((3, 3, cfg.vis_image_size, cfg.vis_image_size), 141, 0.5).clamp(-1, 1).to(dt)
    mel = seeded((1, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 142, 0.3).to(dt)
    ids = torch.tensor([[1, 21, 22, 23, -200, 24, 25, 26]], dtype=torch.int64)
    n_new = 6
    ref_ids, dbg = O.generate_greedy(ids, [px.float()], [mel.float()], [100], w32, ocfg, n_new, return_debug=True)
    model = VidiForCausalLM.__new__(VidiForCausalLM)
    model.config, model.dtype, model.device, model.engine = cfg, dt, torch.device("cuda"), eng
    model.generation_config = SimpleNamespace(eos_token_id=cfg.eos_token_id, pad_token_id=0)
    model.model = None
    out = model.forward(ids, images=px[None].cuda(), audios=mel[None].cuda(), audio_sizes=[100], logits_to_keep=1)
    ref_logits = dbg["prefill_logits"]
    ltol = logit_tol(dt, ref_logits)
    report("7b prefill logits", out.logits[:, -1], ref_logits, ltol, 0.0)
    got = model.generate(ids, images=px[None].cuda(), audios=mel[None].cuda(), audio_sizes=[100], max_new_tokens=n_new,
                         do_sample=False, use_cache=True).cpu()
    top2 = torch.topk(ref_logits[0].float(), 2).values
    if float(top2[0] - top2[1]) > 2 * ltol:
        assert int(got[0, 0]) == int(ref_ids[0, 0])
    assert got.shape[1] <= n_new and got.dtype == torch.int64
    # text-only query (no video): plain Mistral path, mistral.py:169-174
    ref_t = O.generate_greedy(torch.tensor([[1, 21, 22, 23]]), None, None, None, w32, ocfg, 1, return_debug=True)[1]["prefill_logits"]
    out_t = model.forward(torch.tensor([[1, 21, 22, 23]]), logits_to_keep=1)
    report("7b text-only logits", out_t.logits[:, -1], ref_t, logit_tol(dt, ref_t), 0.0)
    # batched right padding is rejected like the reference does (mistral.py:366-373)
    with pytest.raises(ValueError):
        model.generate(torch.tensor([[1, 5, -200, 6], [1, -200, 7, 0]]), attention_mask=torch.tensor([[1, 1, 1, 1], [1, 1, 1, 0]]),
                       images=px[None].cuda(), max_new_tokens=2)


def test_7b_real_dims_two_layers():
    """Mistral-7B layer dims (H=4096, 32/8 heads x 128, I=14336), 2 layers, tiny towers: mm stream + text + one decode step"""
    from vidi_amd.config import tiny_7b
    from vidi_amd.model import strip_image_token
    from test_gpu_model import _run_oracle_prefill
    dt = torch.bfloat16
    cfg = tiny_7b(hidden_size=4096, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
                  query_pre_attn_scalar=128.0, sliding_window=4096, num_hidden_layers=2, vocab_size=1024)
    eng, w32 = make(cfg, dt, seed=6)
    ocfg = oracle_cfg(cfg)
    H = cfg.hidden_size
    Nv, Na = 300, 40
    img = seeded((1, Nv, H), 118, cfg.mm_std).to(dt); aud = seeded((1, Na, H), 119, cfg.mm_std).to(dt)
    imask = torch.ones((1, Nv), dtype=torch.bool); amask = torch.ones((1, Na), dtype=torch.bool)
    ids = torch.tensor([[1, 31, -200, 32, 33, 34, 35, 36, 37, 38, 39]], dtype=torch.int64)
    href, caches, am = _run_oracle_prefill(w32, ocfg, ids, img.float(), imask, aud.float(), amask)
    mm = eng.mm_stream_prefill(img[0].cuda(), imask[0].to(torch.uint8).cuda(), aud[0].cuda(), amask[0].to(torch.uint8).cuda(),
                               pre_normalized=False)
    nkv, hd = cfg.num_key_value_heads, cfg.head_dim
    kref, _ = caches.image[1]
    kc = mm.kc[1].reshape(nkv, -1, hd)[:, :Nv].permute(1, 0, 2).reshape(Nv, -1)
    report("7b real-dims image K cache L1", kc, kref[0], 3.6e-2 * kref.std().item(), 3e-2)   # 0.73 used (K = 4 096, one un-normalised stream update)
    idt, mask, pos_ids = strip_image_token(ids)
    ts = eng.new_text_state(1, 16)
    hn = eng.text_forward(eng.embed_tokens(idt.cuda()), pos_ids.reshape(-1).cuda(), ts, mm, Lq=idt.shape[1], new_mask=mask.cuda())
    # bf16 vs the fp32 oracle after 2 layers of un-normalised residual adds (no post-norms in Mistral): 7 % of rms
    report("7b real-dims text hidden", hn, href[0], 7e-2 * href.std().item(), 4e-2)
    nxt = torch.tensor([41], dtype=torch.int64)
    e = torch.nn.functional.embedding(nxt[:, None], w32["model.embed_tokens.weight"])
    tm = torch.cat([am, torch.ones(1, 1, dtype=torch.bool)], dim=1)
    p = torch.tensor([[idt.shape[1]]])
    href2 = O.model_forward(e, p, tm, img.float(), imask, aud.float(), amask, w32, ocfg, caches, idt.shape[1])
    hn2 = eng.text_forward(eng.embed_tokens(nxt.cuda()), p.reshape(-1).cuda(), ts, mm, Lq=1)
    report("7b real-dims decode hidden", hn2, href2[0], 7e-2 * href2.std().item(), 4e-2)


def test_7b_left_padded_batch_equals_single_rows(setup7b):
    """batched generation the way Vidi-7B accepts it (left padding, mistral.py:366-373): each row == that prompt alone"""
    cfg, eng, w32, dt = setup7b
    from vidi_amd.model import VidiForCausalLM
    from types import SimpleNamespace
    model = VidiForCausalLM.__new__(VidiForCausalLM)
    model.config, model.dtype, model.device, model.engine = cfg, dt, torch.device("cuda"), eng
    model.generation_config = SimpleNamespace(eos_token_id=cfg.eos_token_id, pad_token_id=0)
    model.model = None
    px = seeded((3, 3, cfg.vis_image_size, cfg.vis_image_size), 150, 0.5).clamp(-1, 1).to(dt)
    mel = seeded((1, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), 151, 0.3).to(dt)
    st = model.encode_mm_state(px[None].cuda(), mel[None].cuda(), [100])
    ids = torch.tensor([[0, 0, 1, 21, -200, 22], [1, 31, 32, -200, 33, 34]])
    am = torch.tensor([[0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1]])
    both = model.generate(ids, attention_mask=am, mm_state=st, max_new_tokens=4, eos_token_id=-1).cpu()
    for i in range(2):
        one = model.generate(ids[i][am[i].bool()][None], mm_state=st, max_new_tokens=4, eos_token_id=-1).cpu()
        assert torch.equal(both[i], one[0])
