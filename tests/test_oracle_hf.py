"""The oracle's third-party blocks vs the installed `transformers` eager implementations (Gemma2 text
decoder, SigLIP vision tower, Whisper encoder) on tiny random models.  The reference pins
transformers 4.50.0; the installed 5.x implements the same arithmetic for these blocks.
CPU only."""
import pytest
import torch

import vidi_oracle as O

transformers = pytest.importorskip("transformers")


def _ocfg(**kw):
    return O.OracleConfig(**kw)


def test_gemma2_text_decoder_matches_hf():
    from transformers import Gemma2Config, Gemma2Model
    torch.manual_seed(0)
    hc = Gemma2Config(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=32, query_pre_attn_scalar=32, sliding_window=64, rms_norm_eps=1e-6,
                      attn_logit_softcapping=50.0, final_logit_softcapping=30.0, max_position_embeddings=128, pad_token_id=0,
                      attn_implementation="eager")
    m = Gemma2Model(hc).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.05)
    w = {"model." + k: v.detach() for k, v in m.state_dict().items()}
    cfg = _ocfg(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                head_dim=32, query_pre_attn_scalar=32.0, sliding_window=64, vocab_size=128)
    ids = torch.tensor([[5, 9, 33, 7, 100, 42, 17]])
    with torch.no_grad():
        ref = m(input_ids=ids, use_cache=False).last_hidden_state
    emb = torch.nn.functional.embedding(ids, w["model.embed_tokens.weight"])
    pos = torch.arange(ids.shape[1])[None]
    am = torch.ones_like(ids, dtype=torch.bool)
    got = O.model_forward(emb, pos, am, None, None, None, None, w, cfg, O.OracleCaches(), 0)
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)


def test_gemma2_blocks_match_hf():
    from transformers.models.gemma2 import modeling_gemma2 as G
    torch.manual_seed(1)
    x = torch.randn(3, 5, 64)
    n = G.Gemma2RMSNorm(64, eps=1e-6)
    with torch.no_grad():
        n.weight.normal_(0, 0.3)
    torch.testing.assert_close(O.gemma_rmsnorm(x, n.weight.detach(), 1e-6), n(x), rtol=0, atol=0)
    xb = x.to(torch.bfloat16)
    assert torch.equal(O.gemma_rmsnorm(xb, n.weight.detach().to(torch.bfloat16), 1e-6), n.to(torch.bfloat16)(xb))
    q, k = torch.randn(2, 4, 5, 32), torch.randn(2, 2, 5, 32)
    pos = torch.arange(5)[None].repeat(2, 1)
    cos, sin = O.rope_cos_sin(pos, 32, 10000.0, torch.float32)
    qh, kh = G.apply_rotary_pos_emb(q, k, cos, sin)
    qo, ko = O.apply_rope(q, k, cos, sin)
    assert torch.equal(qo, qh) and torch.equal(ko, kh)


def test_mistral_text_decoder_matches_hf():
    """Vidi-7B's text-only path is plain MistralModel (mistral.py:169-174 falls through to super().forward)."""
    from transformers import MistralConfig, MistralModel
    torch.manual_seed(4)
    hc = MistralConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2, head_dim=16, sliding_window=64, rms_norm_eps=1e-5, rope_theta=10000.0,
                       max_position_embeddings=128, pad_token_id=0, attn_implementation="eager")
    m = MistralModel(hc).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.05)
    w = {"model." + k: v.detach() for k, v in m.state_dict().items()}
    cfg = _ocfg(arch="mistral", hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, head_dim=16, query_pre_attn_scalar=16.0, sliding_window=64, vocab_size=128,
                rms_norm_eps=1e-5, attn_logit_softcapping=None, final_logit_softcapping=None)
    ids = torch.tensor([[5, 9, 33, 7, 100, 42, 17]])
    with torch.no_grad():
        ref = m(input_ids=ids, use_cache=False).last_hidden_state
    emb = torch.nn.functional.embedding(ids, w["model.embed_tokens.weight"])
    pos = torch.arange(ids.shape[1])[None]
    am = torch.ones_like(ids, dtype=torch.bool)
    got = O.model_forward(emb, pos, am, None, None, None, None, w, cfg, O.OracleCaches(), 0)
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)


def test_mistral_blocks_match_hf():
    from transformers.models.mistral import modeling_mistral as M
    torch.manual_seed(5)
    x = torch.randn(3, 5, 64)
    n = M.MistralRMSNorm(64, eps=1e-5)
    with torch.no_grad():
        n.weight.normal_(1.0, 0.3)
    torch.testing.assert_close(O.mistral_rmsnorm(x, n.weight.detach(), 1e-5), n(x), rtol=0, atol=0)
    xb = x.to(torch.bfloat16)
    assert torch.equal(O.mistral_rmsnorm(xb, n.weight.detach().to(torch.bfloat16), 1e-5), n.to(torch.bfloat16)(xb))


def test_siglip_tower_matches_hf():
    from transformers import SiglipVisionConfig, SiglipVisionModel
    torch.manual_seed(2)
    hc = SiglipVisionConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=3, num_attention_heads=4, image_size=98,
                            patch_size=14, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh", attn_implementation="eager")
    m = SiglipVisionModel(hc).eval()
    # transformers 4.50 (the reference's pin) nests the tower under `vision_model.`; 5.x flattened it
    w = {"model.mm_vis." + (k if k.startswith("vision_model.") else "vision_model." + k): v.detach() for k, v in m.state_dict().items()}
    cfg = _ocfg(vis_image_size=98, vis_patch_size=14, vis_hidden_size=64, vis_intermediate_size=176, vis_num_layers=3, vis_num_heads=4)
    px = torch.randn(2, 3, 98, 98) * 0.5
    with torch.no_grad():
        hs = m(px, output_hidden_states=True).hidden_states
    got = O.siglip_forward(px, w, cfg)
    torch.testing.assert_close(got, hs[-2], rtol=2e-5, atol=2e-5)          # mm_vision/siglip.py:32 select_layer=-2


def test_whisper_encoder_matches_hf():
    from transformers import WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    torch.manual_seed(3)
    hc = WhisperConfig(num_mel_bins=64, d_model=64, encoder_layers=2, encoder_attention_heads=4, encoder_ffn_dim=128,
                       max_source_positions=50, attn_implementation="eager")
    m = WhisperEncoder(hc).eval()
    w = {"model.mm_aud.encoder." + k: v.detach() for k, v in m.state_dict().items()}
    cfg = _ocfg(aud_num_mel_bins=64, aud_d_model=64, aud_num_layers=2, aud_num_heads=4, aud_ffn_dim=128,
                aud_max_source_positions=50, aud_nb_max_frames=100)
    mel = torch.randn(2, 64, 100) * 0.3
    with torch.no_grad():
        ref = m(mel)[0]
    got = O.whisper_encoder_forward(mel, w, cfg)
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)
