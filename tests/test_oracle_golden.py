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
