"""The oracle vs golden vectors produced by EXECUTING the reference's importable modules
(tests/golden/make_golden.py -> reference_modules.npz).  CPU only; runs everywhere."""
import os

import numpy as np
import torch

import vidi_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_modules.npz"))


def t(name):
    return torch.from_numpy(G[name])


def test_rms_norm_and_RMSNorm():
    x, w = t("norm_x"), t("norm_w")
    assert torch.equal(O.mm_rms_norm(x), t("rms_norm_f32"))
    assert torch.equal(O.mm_RMSNorm(x, torch.ones(64) * 0.02898), t("RMSNorm_std_f32"))
    assert torch.equal(O.mm_RMSNorm(x, w), t("RMSNorm_w_f32"))
    got = O.mm_RMSNorm(x.to(torch.bfloat16), w.to(torch.bfloat16)).float()
    assert torch.equal(got, t("RMSNorm_w_bf16")), "weight multiply must happen after the cast back (norm.py:25)"


def test_space_to_depth_and_pool():
    x = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 6)
    assert torch.equal(O.space_to_depth(x, 2), t("s2d"))
    f = t("pool_x")
    assert torch.equal(O.conv2d_pool(f, (28, 28), 2), t("pool_28"))
    torch.testing.assert_close(O.conv2d_pool(f, (10, 10), 2), t("pool_10"), rtol=0, atol=0)
    torch.testing.assert_close(O.conv2d_pool(f, (26, 26), 2), t("pool_26"), rtol=0, atol=0)
    f7 = t("pool7_x")
    assert torch.equal(O.conv2d_pool(f7, (28, 28), 2), t("pool7_28"))
    torch.testing.assert_close(O.conv2d_pool(f7, (10, 10), 2), t("pool7_10"), rtol=0, atol=0)


def test_token_budget_table_bit_exact():
    """T -> (h,w): 25,300,306 stay 28; 307 -> 26 (the cliff); 3600 -> 10 (floor) — SURVEY.md §8c"""
    tab = G["budget_table"]
    for T, h, w in tab.tolist():
        assert O.token_budget_hw(T, 27, 2, 60000) == (h, w), T
    expect = {25: 28, 300: 28, 306: 28, 307: 26, 400: 24, 600: 20, 1200: 14, 3600: 10, 7200: 10}
    for T, h in expect.items():
        assert O.token_budget_hw(T, 27, 2, 60000)[0] == h


def test_sinusoid_and_learnable_pos():
    assert torch.equal(O.fractional_sinusoid(t("sin_p"), 32), t("sin_pe"))
    w = {"p.mlp.0.weight": t("pos_mlp.0.weight"), "p.mlp.0.bias": t("pos_mlp.0.bias"),
         "p.mlp.2.weight": t("pos_mlp.2.weight"), "p.mlp.2.bias": t("pos_mlp.2.bias")}
    pe0 = O.learnable_pos_embd(7, 100, 32, w, "p.", torch.bfloat16).float()
    assert torch.equal(pe0.reshape(7, 1, 1, 32), t("pos_dim0_bf16"))
    pe2 = O.learnable_pos_embd(5, 100, 32, w, "p.", torch.bfloat16).float()
    assert torch.equal(pe2.reshape(1, 1, 5, 32), t("pos_dim2_bf16"))


def test_projector_mlp():
    w = {"m.model.0.weight": t("mlp_model.0.weight"), "m.model.0.bias": t("mlp_model.0.bias"),
         "m.model.2.weight": t("mlp_model.2.weight"), "m.model.2.bias": t("mlp_model.2.bias")}
    torch.testing.assert_close(O.projector_mlp(t("mlp_x"), w, "m."), t("mlp_y"), rtol=0, atol=0)


def test_vidi7b_learned_conv2d_pool():
    """Vidi-7B Conv2DPool (learned conv + align_corners=True resize) against the reference module's own output."""
    g7 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_modules_7b.npz"))
    for tag in "abcd":
        d_in, d_out, s_in, s_out = [int(v) for v in g7[f"{tag}_cfg"]]
        y = O.learned_conv2d_pool(torch.from_numpy(g7[f"{tag}_x"]), torch.from_numpy(g7[f"{tag}_w"]), s_out)
        assert y.shape == (2, d_out, s_out, s_out)
        np.testing.assert_allclose(y.numpy(), g7[f"{tag}_y"], rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# The D-Attn path itself: golden vectors from EXECUTING the reference's own model code (gemma.py / multimodal.py / xattn.py /
# split.py with third-party stand-ins only — tests/golden/ref_harness.py, make_golden_dattn.py -> reference_dattn.npz)
# ---------------------------------------------------------------------------------------------------------------------
import dataclasses  # noqa: E402

import pytest  # noqa: E402

D = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_dattn.npz"))
# fp32 on both sides; the residual differences are summation order (fused vs separate matmuls, softmax formulation)
DATTN_ATOL, DATTN_RTOL = 2e-5, 2e-5


def d(name):
    return torch.from_numpy(D[name])


@pytest.fixture(scope="module")
def dattn_setup():
    from vidi_amd.config import tiny
    from vidi_amd.weights import init_random_weights
    cfg = tiny(sliding_window=64)                                          # = make_golden_dattn.golden_config()
    w = init_random_weights(cfg, seed=3, dtype=torch.float32, device="cpu")
    names = {f.name for f in dataclasses.fields(O.OracleConfig)}
    ocfg = O.OracleConfig(**{k: v for k, v in cfg.to_dict().items() if k in names}, vis_select_layer=cfg.mm_vision_select_layer)
    return cfg, ocfg, w


def _run_oracle(case, ocfg, w, n_new, with_mask=False):
    ids = d(case + "_input_ids")
    images = list(d(case + "_images"))
    audios = list(d(case + "_audios"))
    sizes = D[case + "_audio_sizes"].tolist()
    am = d(case + "_attention_mask") if with_mask else None
    return O.generate_greedy(ids, images, audios, sizes, w, ocfg, max(n_new, 1), attention_mask=am, return_debug=True)


def _close(name, got, ref, atol=DATTN_ATOL, rtol=DATTN_RTOL):
    from util import report
    return report(name, got, ref, atol, rtol)


def test_dattn_encode_matches_reference_execution(dattn_setup):
    """encode_video_images / encode_video_audios (multimodal.py:156-252) incl. masks (bit-exact) and the audio floors"""
    cfg, ocfg, w = dattn_setup
    for case, with_mask in (("A", False), ("B", True)):
        _, dbg = _run_oracle(case, ocfg, w, 1, with_mask)
        assert torch.equal(dbg["image_mask"], d(case + "_image_mask"))
        assert torch.equal(dbg["audio_mask"], d(case + "_audio_mask"))
        _close(case + " image_embeds", dbg["image_embeds"], d(case + "_image_embeds"))
        _close(case + " audio_embeds", dbg["audio_embeds"], d(case + "_audio_embeds"))
    assert d("B_audio_mask").sum(-1).tolist() == [20, 13]                   # floor(floor(200*.5)/5), floor(floor(130*.5)/5)


def test_dattn_text_prep_matches_reference_execution(dattn_setup):
    """prepare_inputs_labels_for_multimodal: mask / positions after deleting the -200 placeholder (bit-exact on valid slots)"""
    cfg, ocfg, w = dattn_setup
    _, dbg = _run_oracle("B", ocfg, w, 1, True)
    ref_mask = d("B_text_mask").bool()
    assert torch.equal(dbg["text_mask"], ref_mask)
    assert torch.equal(dbg["position_ids"][ref_mask], d("B_position_ids")[ref_mask])


def test_dattn_prefill_caches_hidden_logits(dattn_setup):
    """DattnGemma2Model.forward / DecoderLayer.forward / forward_xattn (gemma.py:50-424): every layer's image/audio K/V
    (i.e. the diagonal stream feeding them), the final hidden state on valid rows, and the logits"""
    cfg, ocfg, w = dattn_setup
    for case, with_mask in (("A", False), ("B", True)):
        _, dbg = _run_oracle(case, ocfg, w, 1, with_mask)
        for li in range(cfg.num_hidden_layers):
            for mod, cache in (("img", dbg["caches"].image), ("aud", dbg["caches"].audio)):
                mask = d(f"{case}_{'image' if mod == 'img' else 'audio'}_mask")[..., None]
                _close(f"{case} {mod} K layer {li}", cache[li][0] * mask, d(f"{case}_{mod}_k_{li}") * mask)
                _close(f"{case} {mod} V layer {li}", cache[li][1] * mask, d(f"{case}_{mod}_v_{li}") * mask)
        tm = d(case + "_text_mask").bool()
        _close(case + " final hidden (valid rows)", dbg["prefill_hidden"][tm], d(case + "_prefill_hidden_last")[tm])
        _close(case + " prefill logits", dbg["prefill_logits"], d(case + "_prefill_logits"))


def test_dattn_greedy_decode_matches_reference_execution(dattn_setup):
    """decode steps served from the three caches (use_image_cache branch gemma.py:179-195, T2T cache) and greedy tokens"""
    cfg, ocfg, w = dattn_setup
    toks, dbg = _run_oracle("A", ocfg, w, 6)
    assert toks.tolist() == D["A_tokens"].tolist()
    got = torch.stack(dbg["step_logits"], dim=1)
    _close("A step logits", got, d("A_step_logits"))


def test_dattn_all_zero_video_branch(dattn_setup):
    """a sample whose frames are all zero: mask row forced valid for the kernel and the branch output zeroed (gemma.py:180-192)"""
    cfg, ocfg, w = dattn_setup
    toks, dbg = _run_oracle("C", ocfg, w, 2)
    assert not d("C_image_mask").any() and not dbg["image_mask"].any()
    _close("C prefill logits", dbg["prefill_logits"], d("C_prefill_logits"))
    assert toks.tolist() == D["C_tokens"].tolist()


# ---- Vidi-7B: the reference's mistral.py / multimodal.py (learned Conv2DPool, Da->Da audio pool) executed the same way
D7 = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_dattn_7b.npz"))


def test_dattn_7b_matches_reference_execution():
    from vidi_amd.config import tiny_7b
    from vidi_amd.weights import init_random_weights
    cfg = tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64)      # = make_golden_dattn_7b.golden_config()
    w = init_random_weights(cfg, seed=3, dtype=torch.float32, device="cpu")
    names = {f.name for f in dataclasses.fields(O.OracleConfig)}
    ocfg = O.OracleConfig(**{k: v for k, v in cfg.to_dict().items() if k in names}, vis_select_layer=cfg.mm_vision_select_layer, arch="mistral")
    t7 = lambda n: torch.from_numpy(D7[n])                                          # noqa: E731
    toks, dbg = O.generate_greedy(t7("A_input_ids"), list(t7("A_images")), list(t7("A_audios")), D7["A_audio_sizes"].tolist(),
                                  w, ocfg, 5, return_debug=True)
    assert torch.equal(dbg["image_mask"], t7("A_image_mask")) and torch.equal(dbg["audio_mask"], t7("A_audio_mask"))
    _close("7B image_embeds", dbg["image_embeds"], t7("A_image_embeds"))
    _close("7B audio_embeds", dbg["audio_embeds"], t7("A_audio_embeds"))
    for li in range(cfg.num_hidden_layers):
        _close(f"7B img K layer {li}", dbg["caches"].image[li][0], t7(f"A_img_k_{li}"))
        _close(f"7B img V layer {li}", dbg["caches"].image[li][1], t7(f"A_img_v_{li}"))
    _close("7B final hidden", dbg["prefill_hidden"], t7("A_prefill_hidden_last"))
    _close("7B prefill logits", dbg["prefill_logits"], t7("A_prefill_logits"))
    assert toks.tolist() == D7["A_tokens"].tolist()
    _close("7B step logits", torch.stack(dbg["step_logits"], dim=1), t7("A_step_logits"))


def test_dattn_token_budget_branch_end_to_end(dattn_setup):
    """3 760 tiny frames cross `max_tokens = 60000 * pool^2` (multimodal.py:175-180): the reference's own encode_video_images chose
    (10, 10) via resize_by_tokens and up-sampled inside Conv2DPool -> 25 tokens per frame; the oracle must land on the same tokens,
    the same embeddings (sampled rows) and the same prefill logits.  Frames are regenerated from the generator's seed."""
    cfg, ocfg, w = dattn_setup
    T = int(D["D_n_frames"][0])
    S, M, Fr = cfg.vis_image_size, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames
    gd = torch.Generator().manual_seed(777)
    px = (torch.randn((1, T, 3, S, S), generator=gd) * 0.5).clamp(-1, 1)
    mel = torch.randn((1, 1, M, Fr), generator=gd) * 0.3
    toks, dbg = O.generate_greedy(d("D_input_ids"), list(px), list(mel), D["D_audio_sizes"].tolist(), w, ocfg, 1, return_debug=True)
    emb = dbg["image_embeds"]
    assert emb.shape[1] == int(D["D_n_tokens"][0]) == 25 * T
    assert int(dbg["image_mask"].sum()) == int(D["D_mask_sum"][0])
    _close("D embeds head", emb[0, :50], d("D_embeds_head"))
    _close("D embeds mid", emb[0, 47000:47050], d("D_embeds_mid"))
    _close("D embeds tail", emb[0, -50:], d("D_embeds_tail"))
    assert abs(float(emb.abs().mean()) - float(D["D_embeds_abs_mean"][0])) < 1e-6
    _close("D prefill logits", dbg["prefill_logits"], d("D_prefill_logits"), 1e-4, 1e-4)


def test_dattn_reference_generate_loop(dattn_setup):
    """the reference's own generate() driving HF's greedy loop (gemma.py:603-687): new tokens only, per-step scores, EOS stop"""
    cfg, ocfg, w = dattn_setup
    from vidi_amd.weights import init_random_weights
    w6 = init_random_weights(cfg, seed=6, dtype=torch.float32, device="cpu")
    images, audios = list(d("A_images")), list(d("A_audios"))
    toks, dbg = O.generate_greedy(d("E_input_ids"), images, audios, [100], w6, ocfg, 8, return_debug=True)
    assert toks.tolist() == D["E_tokens"].tolist() and len(set(toks[0].tolist())) > 1
    got = torch.stack([dbg["prefill_logits"]] + dbg["step_logits"], dim=1)
    _close("E per-step scores", got, d("E_scores"), 1e-4, 1e-4)
    toks = O.generate_greedy(d("F_input_ids"), images, audios, [100], w, ocfg, 8)
    assert toks.tolist() == D["F_tokens"].tolist() == [[cfg.eos_token_id]]


def test_dattn_7b_reference_generate_loop():
    """Vidi-7B: the reference's own generate() under HF's loop (mistral.py:629-716), weights seed 6 — six different greedy tokens"""
    from vidi_amd.config import tiny_7b
    from vidi_amd.weights import init_random_weights
    cfg = tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64)
    w6 = init_random_weights(cfg, seed=6, dtype=torch.float32, device="cpu")
    names = {f.name for f in dataclasses.fields(O.OracleConfig)}
    ocfg = O.OracleConfig(**{k: v for k, v in cfg.to_dict().items() if k in names}, vis_select_layer=cfg.mm_vision_select_layer, arch="mistral")
    t7 = lambda n: torch.from_numpy(D7[n])                                          # noqa: E731
    toks, dbg = O.generate_greedy(t7("A_input_ids"), list(t7("A_images")), list(t7("A_audios")), D7["A_audio_sizes"].tolist(),
                                  w6, ocfg, 6, return_debug=True)
    assert toks.tolist() == D7["E_tokens"].tolist() and len(set(toks[0].tolist())) >= 4
    _close("7B E per-step scores", torch.stack([dbg["prefill_logits"]] + dbg["step_logits"], dim=1), t7("E_scores"), 1e-4, 1e-4)
