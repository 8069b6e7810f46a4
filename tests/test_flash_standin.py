"""The flash-attn stand-in of the golden harness (tests/golden/ref_harness.py:_fa_core / flash_attn_func / flash_attn_varlen_func) is the one link of
the reference-executed goldens that is RESTATED rather than executed: flash-attn is CUDA-only.  This file pins it on third-party code that IS
installed and executable: transformers' own eager attention of Gemma2 (`modeling_gemma2.eager_attention_forward`: scale -> softcap·tanh(s / softcap) ->
additive mask -> fp32 softmax -> PV, GQA by `repeat_kv` — the path HF itself holds equal to its flash-attention path) and PyTorch's
`scaled_dot_product_attention`.  Covered: the non-causal cross-attention form with and without a softcap and with a key-padding mask (the T2V / T2A
call, xattn.py:253), the causal form with flash-attn's bottom-right alignment and a sliding window (the T2T call), GQA, and the varlen packing
(`flash_attn_varlen_func` over `cu_seqlens` == the padded call row by row).  CPU only, no reference checkout needed."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_harness as RH  # noqa: E402

mg = pytest.importorskip("transformers.models.gemma2.modeling_gemma2")


def _hf(q, k, v, scale, softcap, add_mask):
    """transformers' eager Gemma2 attention: q [B,Lq,H,D], k / v [B,Lk,Hk,D] -> [B,Lq,H,D]"""
    mod = SimpleNamespace(head_dim=q.shape[-1], num_key_value_groups=q.shape[2] // k.shape[2], training=False)
    out, _ = mg.eager_attention_forward(mod, q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), add_mask, dropout=0.0,
                                        scaling=scale, softcap=softcap)
    return out


@pytest.mark.parametrize("softcap", [None, 50.0, 5.0])
@pytest.mark.parametrize("H,Hk", [(4, 4), (4, 2), (8, 1)])
def test_cross_attention_form_equals_hf_eager(softcap, H, Hk):
    torch.manual_seed(1)
    B, Lq, Lk, D = 2, 5, 37, 16
    q, k, v = torch.randn(B, Lq, H, D) * 3, torch.randn(B, Lk, Hk, D) * 3, torch.randn(B, Lk, Hk, D)
    got = RH.flash_attn_func(q, k, v, softmax_scale=0.3, causal=False, softcap=softcap or 0.0)
    torch.testing.assert_close(got, _hf(q, k, v, 0.3, softcap, None), rtol=1e-5, atol=1e-6)
    # with key padding (what pad_input / unpad_input + the varlen call amount to): keys beyond each row's length are invisible
    lens = [37, 20]
    key_ok = torch.arange(Lk)[None, :] < torch.tensor(lens)[:, None]
    add = torch.zeros(B, 1, 1, Lk).masked_fill(~key_ok[:, None, None, :], float("-inf"))
    ref = _hf(q, k, v, 0.3, softcap, add)
    got = RH._fa_core(q, k, v, 0.3, False, (-1, -1), softcap or 0.0, key_ok)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)
    # ... and the varlen entry point over the packed rows gives the padded call's rows
    qp = torch.cat([q[0], q[1]])
    kp, vp = torch.cat([k[0, : lens[0]], k[1, : lens[1]]]), torch.cat([v[0, : lens[0]], v[1, : lens[1]]])
    cu_q = torch.tensor([0, Lq, 2 * Lq], dtype=torch.int32)
    cu_k = torch.tensor([0, lens[0], lens[0] + lens[1]], dtype=torch.int32)
    var = RH.flash_attn_varlen_func(qp, kp, vp, cu_q, cu_k, Lq, max(lens), softmax_scale=0.3, causal=False, softcap=softcap or 0.0)
    torch.testing.assert_close(var.view(B, Lq, H, D), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("window", [None, 4])
@pytest.mark.parametrize("Lq,Lk", [(6, 6), (1, 9), (3, 9)])
def test_causal_form_is_bottom_right_aligned_like_hf_with_a_cache(window, Lq, Lk):
    """flash-attn aligns the causal mask to the bottom-right corner (query i of Lq sees keys j <= i + Lk - Lq): exactly HF's mask for the last Lq
    positions of a cache of Lk keys; a sliding window W keeps keys with i - j <= W (window_size = (W, W) under causal)"""
    torch.manual_seed(2)
    B, H, Hk, D = 1, 4, 2, 8
    q, k, v = torch.randn(B, Lq, H, D), torch.randn(B, Lk, Hk, D), torch.randn(B, Lk, Hk, D)
    i = torch.arange(Lq)[:, None] + (Lk - Lq)
    j = torch.arange(Lk)[None, :]
    ok = j <= i
    if window is not None:
        ok &= j >= i - window
    add = torch.zeros(1, 1, Lq, Lk).masked_fill(~ok[None, None], float("-inf"))
    ref = _hf(q, k, v, D ** -0.5, 30.0, add)
    got = RH.flash_attn_func(q, k, v, causal=True, window_size=(-1, -1) if window is None else (window, window), softcap=30.0)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)
    # without a softcap: PyTorch's own kernel as a second witness
    sd = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2).repeat_interleave(H // Hk, 1),
                                                          v.transpose(1, 2).repeat_interleave(H // Hk, 1), attn_mask=ok[None, None]).transpose(1, 2)
    got = RH.flash_attn_func(q, k, v, causal=True, window_size=(-1, -1) if window is None else (window, window))
    torch.testing.assert_close(got, sd, rtol=1e-5, atol=1e-6)


def test_bert_padding_helpers_round_trip():
    torch.manual_seed(3)
    x = torch.randn(3, 7, 5)
    m = torch.tensor([[1, 1, 1, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0, 0]], dtype=torch.bool)
    flat, idx, cu, mx = RH.unpad_input(x, m)
    assert flat.shape[0] == 10 and cu.tolist() == [0, 3, 10, 10] and mx == 7
    assert torch.equal(RH.index_first_axis(x.flatten(0, 1), idx), flat)
    back = RH.pad_input(flat, idx, 3, 7)
    assert torch.equal(back[m], x[m]) and float(back[~m].abs().sum()) == 0.0
