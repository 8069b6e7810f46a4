"""HIP path vs golden vectors produced by EXECUTING the reference's own model code on CPU/fp32
(tests/golden/make_golden_dattn.py, make_golden_dattn_7b.py — third-party stand-ins only).  No oracle in between:
`VidiForCausalLM.forward/generate` on the GPU against what `DattnGemma2ForCausalLM` / `DattnMistralForCausalLM` returned.
Tolerances: the GPU model computes in bf16/fp16 with the reference's rounding points, the goldens are fp32 — 5 % of the
tensor's rms + 3 % relative for bf16 (1 % / 0.6 % fp16) on activations, 3x that on logits; masks bit-exact; greedy tokens
equal wherever the golden top-2 margin exceeds the tolerance."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from util import report

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def tol(dt, k=1.0):
    return (5e-2 * k, 3e-2) if dt == torch.bfloat16 else (1e-2 * k, 6e-3)


def build(cfg, dt, seed=3):
    from vidi_amd.engine import VidiEngine
    from vidi_amd.model import VidiForCausalLM
    from vidi_amd.weights import init_random_weights
    w = init_random_weights(cfg, seed=seed, dtype=torch.float32, device="cpu")     # the weights the goldens were made with
    eng = VidiEngine(cfg, {k: v.to(dt) if not k.count(".mm_rand_pos_") else v for k, v in w.items()}, dtype=dt, device="cuda")
    model = VidiForCausalLM.__new__(VidiForCausalLM)
    model.config, model.dtype, model.device, model.engine = cfg, dt, torch.device("cuda"), eng
    model.generation_config = SimpleNamespace(eos_token_id=cfg.eos_token_id, pad_token_id=0)
    model.model = None
    return model


def check_case(model, D, case, dt, n_new, with_mask=False):
    cfg = model.config
    t = lambda n: torch.from_numpy(D[f"{case}_{n}"])                                 # noqa: E731
    ids = t("input_ids")
    am = t("attention_mask") if with_mask else None
    px, mel = t("images").to(dt).cuda(), t("audios").to(dt).cuda()
    sizes = D[f"{case}_audio_sizes"].tolist()
    kw = {} if am is None else {"attention_mask": am}
    # the golden inputs are fp32; the GPU model sees them rounded to its dtype (as inference.py does with `.to(dtype)`)
    ref = t("prefill_logits")
    atol, rtol = tol(dt, ref.std().item())
    if with_mask:                                    # padded batch: each row's last VALID position, all positions checked too
        out = model.forward(ids, images=px, audios=mel, audio_sizes=sizes, logits_to_keep=0, **kw)
        tm = t("text_mask").bool()
        lens = tm.sum(-1)
        got = out.logits[torch.arange(ids.shape[0]), (lens - 1).cuda()]
        report(f"{case} all valid positions' logits", out.logits.cpu()[tm], t("prefill_logits_all")[tm], 3 * atol, rtol)
    else:
        got = model.forward(ids, images=px, audios=mel, audio_sizes=sizes, logits_to_keep=1, **kw).logits[:, -1]
    report(f"{case} prefill logits vs reference execution", got, ref, 3 * atol, rtol)
    if n_new:
        got = model.generate(ids, images=px, audios=mel, audio_sizes=sizes, max_new_tokens=n_new, do_sample=False).cpu()
        ref_tok = D[f"{case}_tokens"]
        logits = [ref[0]] + [x for x in torch.from_numpy(D[f"{case}_step_logits"])[0]]
        for i in range(min(got.shape[1], ref_tok.shape[1])):
            top2 = torch.topk(logits[i].float(), 2).values
            if float(top2[0] - top2[1]) <= 6 * atol:
                break                                                                # low-margin step: later tokens may diverge
            assert int(got[0, i]) == int(ref_tok[0, i]), f"{case}: token {i}: {int(got[0, i])} != reference {int(ref_tok[0, i])}"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_vidi15_against_reference_execution(dt):
    from vidi_amd.config import tiny
    D = np.load(os.path.join(GOLD, "reference_dattn.npz"))
    model = build(tiny(sliding_window=64), dt)
    # encode pipeline: masks bit-exact, features within tolerance
    eng = model.engine
    px = torch.from_numpy(D["A_images"])[0].to(dt).cuda()
    feats, mask = eng.encode_video_images(px)
    assert torch.equal(mask.bool().cpu(), torch.from_numpy(D["A_image_mask"])[0])
    ref = torch.from_numpy(D["A_image_embeds"])[0]
    report("A image_embeds vs reference execution", feats, ref, *tol(dt, ref.std().item()))
    check_case(model, D, "A", dt, n_new=6)
    check_case(model, D, "B", dt, n_new=0, with_mask=True)
    check_case(model, D, "C", dt, n_new=2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_vidi7b_against_reference_execution(dt):
    from vidi_amd.config import tiny_7b
    D = np.load(os.path.join(GOLD, "reference_dattn_7b.npz"))
    model = build(tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64), dt)
    eng = model.engine
    mel = torch.from_numpy(D["A_audios"])[0].to(dt).cuda()
    feats, mask = eng.encode_video_audios(mel, int(D["A_audio_sizes"][0]))
    assert torch.equal(mask.bool().cpu(), torch.from_numpy(D["A_audio_mask"])[0])
    ref = torch.from_numpy(D["A_audio_embeds"])[0]
    report("7B audio_embeds vs reference execution", feats, ref, *tol(dt, ref.std().item()))
    check_case(model, D, "A", dt, n_new=5)


def test_vidi15_token_budget_branch_against_reference_execution():
    """case D of the goldens: 3 760 frames cross the token budget inside the reference's own encode_video_images ((10, 10) via
    resize_by_tokens, bilinear up-sampling in Conv2DPool, 25 tokens/frame).  HIP path: same token count (bit-exact integer rule),
    sampled embeddings and prefill logits within the bf16 tolerances."""
    from vidi_amd.config import tiny
    dt = torch.bfloat16
    D = np.load(os.path.join(GOLD, "reference_dattn.npz"))
    cfg = tiny(sliding_window=64)
    model = build(cfg, dt)
    T = int(D["D_n_frames"][0])
    S, M, Fr = cfg.vis_image_size, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames
    gd = torch.Generator().manual_seed(777)
    px = (torch.randn((1, T, 3, S, S), generator=gd) * 0.5).clamp(-1, 1)
    mel = torch.randn((1, 1, M, Fr), generator=gd) * 0.3
    feats, mask = model.engine.encode_video_images(px[0].to(dt).cuda())
    assert feats.shape[0] == int(D["D_n_tokens"][0]) == 25 * T and int(mask.sum()) == int(D["D_mask_sum"][0])
    for name, sl in (("head", slice(0, 50)), ("mid", slice(47000, 47050)), ("tail", slice(-50, None))):
        ref = torch.from_numpy(D[f"D_embeds_{name}"])
        report(f"D embeds {name} vs reference execution", feats[sl], ref, *tol(dt, ref.std().item()))
    out = model.forward(torch.from_numpy(D["D_input_ids"]), images=px.to(dt).cuda(), audios=mel.to(dt).cuda(),
                        audio_sizes=D["D_audio_sizes"].tolist(), logits_to_keep=1)
    ref = torch.from_numpy(D["D_prefill_logits"])
    atol, rtol = tol(dt, ref.std().item())
    report("D prefill logits vs reference execution", out.logits[:, -1], ref, 3 * atol, rtol)


def test_vidi15_generate_against_reference_generate():
    """cases E/F: the reference's own generate() (HF greedy loop threaded by gemma.py:657-687).  Ours must return the same NEW tokens
    wherever the reference's top-2 score margin exceeds the bf16 tolerance, and stop at EOS like it does."""
    from vidi_amd.config import tiny
    dt = torch.bfloat16
    D = np.load(os.path.join(GOLD, "reference_dattn.npz"))
    cfg = tiny(sliding_window=64)
    px = torch.from_numpy(D["A_images"]).to(dt).cuda(); mel = torch.from_numpy(D["A_audios"]).to(dt).cuda()
    m6 = build(cfg, dt, seed=6)
    got = m6.generate(torch.from_numpy(D["E_input_ids"]), images=px, audios=mel, audio_sizes=[100], max_new_tokens=8, do_sample=False,
                      use_cache=True, pad_token_id=0).cpu()
    scores = torch.from_numpy(D["E_scores"])[0]
    atol, _ = tol(dt, scores.std().item())
    for i in range(got.shape[1]):
        top2 = torch.topk(scores[i], 2).values
        if float(top2[0] - top2[1]) <= 6 * atol:
            break
        assert int(got[0, i]) == int(D["E_tokens"][0, i]), f"step {i}: {int(got[0, i])} != reference generate() {int(D['E_tokens'][0, i])}"
    m3 = build(cfg, dt, seed=3)
    got = m3.generate(torch.from_numpy(D["F_input_ids"]), images=px, audios=mel, audio_sizes=[100], max_new_tokens=8, do_sample=False).cpu()
    assert got.tolist() == D["F_tokens"].tolist()                       # [[eos]]: one new token, then stop


def test_vidi7b_generate_against_reference_generate():
    """Vidi-7B case E: six different greedy tokens from the reference's own generate(); ours must agree wherever the margin allows"""
    from vidi_amd.config import tiny_7b
    dt = torch.bfloat16
    D = np.load(os.path.join(GOLD, "reference_dattn_7b.npz"))
    cfg = tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64)
    m6 = build(cfg, dt, seed=6)
    px = torch.from_numpy(D["A_images"]).to(dt).cuda(); mel = torch.from_numpy(D["A_audios"]).to(dt).cuda()
    got = m6.generate(torch.from_numpy(D["A_input_ids"]), images=px, audios=mel, audio_sizes=[100], max_new_tokens=6, do_sample=False).cpu()
    scores = torch.from_numpy(D["E_scores"])[0]
    atol, _ = tol(dt, scores.std().item())
    agreed = 0
    for i in range(got.shape[1]):
        top2 = torch.topk(scores[i], 2).values
        if float(top2[0] - top2[1]) <= 6 * atol:
            break
        assert int(got[0, i]) == int(D["E_tokens"][0, i]), f"step {i}: {int(got[0, i])} != reference generate() {int(D['E_tokens'][0, i])}"
        agreed += 1
    assert agreed >= 1
