"""Golden vectors for the Vidi-7B (Mistral D-Attn) path produced by EXECUTING THE REFERENCE'S OWN MODEL CODE
(`DattnMistralForCausalLM.forward` -> Vidi_7B/model/lmm/dattn/{mistral,multimodal,xattn,split}.py, learned Conv2DPool) on CPU
in fp32, third-party dependencies replaced only (tests/golden/ref_harness.py:install_7b).  Run in the build container:

    python tests/golden/make_golden_dattn_7b.py        ->  tests/golden/reference_dattn_7b.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
OUT = os.path.join(HERE, "reference_dattn_7b.npz")

import ref_harness as RH  # noqa: E402


class _Proc:
    def __init__(self, **k):
        self.__dict__.update(k)


def golden_config():
    """tiny Mistral-arch config with head_dim = hidden/heads = 128 (transformers 4.44 Mistral has no separate head_dim; 128 is the real Mistral-7B head size the kernels are built for) and the
    sliding window opened (see make_golden_dattn.golden_config)"""
    from vidi_amd.config import tiny_7b
    return tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64)


def build_reference_model(cfg):
    M7, mm = RH.install_7b()
    import model.mm_vision.siglip as RS
    import model.mm_audio.whisper as RW

    def vis_from_pretrained(name, select_layer=-2, attn_implementation=None, **k):
        c = RS.SiglipVisionTowerConfig(
            hidden_size=cfg.vis_hidden_size, intermediate_size=cfg.vis_intermediate_size, num_hidden_layers=cfg.vis_num_layers,
            num_attention_heads=cfg.vis_num_heads, image_size=cfg.vis_image_size, patch_size=cfg.vis_patch_size,
            layer_norm_eps=cfg.vis_ln_eps, hidden_act="gelu_pytorch_tanh")
        c.select_layer = select_layer
        c._attn_implementation = "eager"
        return RS.SiglipVisionTower(c)

    def aud_from_pretrained(name, attn_implementation=None, **k):
        c = RW.WhisperAudioTowerConfig(
            num_mel_bins=cfg.aud_num_mel_bins, d_model=cfg.aud_d_model, encoder_layers=cfg.aud_num_layers,
            encoder_attention_heads=cfg.aud_num_heads, encoder_ffn_dim=cfg.aud_ffn_dim,
            max_source_positions=cfg.aud_max_source_positions, decoder_layers=1, decoder_attention_heads=cfg.aud_num_heads,
            decoder_ffn_dim=cfg.aud_ffn_dim, vocab_size=64)
        c._attn_implementation = "eager"
        return RW.WhisperAudioTower(c)

    RS.SiglipImageProcessor = type("P", (), {"from_pretrained": staticmethod(lambda *a, **k: _Proc(size={"height": cfg.vis_image_size}))})
    RS.SiglipVisionTower.from_pretrained = staticmethod(vis_from_pretrained)
    RW.WhisperFeatureExtractor = type("P", (), {"from_pretrained": staticmethod(lambda *a, **k: _Proc(nb_max_frames=cfg.aud_nb_max_frames))})
    RW.WhisperAudioTower.from_pretrained = staticmethod(aud_from_pretrained)
    M7.AutoTokenizer = type("T", (), {"from_pretrained": staticmethod(
        lambda *a, **k: _Proc(padding_side=k.get("padding_side", "right"), model_max_length=4096, unk_token="<unk>",
                              pad_token=None, pad_token_id=cfg.pad_token_id))})
    c = M7.DattnMistralConfig(
        hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
        num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
        rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
        vocab_size=cfg.vocab_size, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id,
        bos_token_id=cfg.bos_token_id, tie_word_embeddings=False)
    for k in ("mm_input_type", "mm_projector_type", "mm_image_aspect_ratio", "mm_image_pool_size", "mm_audio_pool_size",
              "mm_std", "mm_time_interval", "mm_vision_tower", "mm_audio_tower", "mm_vision_select_layer"):
        setattr(c, k, getattr(cfg, k))
    c.mm_splits = 32          # what eval/inference.py:87 sets before asking (inputs smaller than 32 go through split_data's repeat path)
    c.train_vis = False
    c.train_aud = False
    c._attn_implementation = "flash_attention_2"
    model = M7.DattnMistralForCausalLM(c).float().eval()
    return model, M7


def load_weights(model, w):
    sd = model.state_dict()
    if not any(".mm_vis.vision_model." in k for k in sd):               # transformers 5.x flattened SiglipVisionModel
        w = {k.replace(".mm_vis.vision_model.", ".mm_vis."): v for k, v in w.items()}
        type(model.model.mm_vis).vision_model = property(lambda self: self)
    missing_in_ref = [k for k in w if k not in sd]
    assert not missing_in_ref, f"names the reference model does not have: {missing_in_ref[:5]}"
    for k, v in w.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    res = model.load_state_dict({k: v.float() for k, v in w.items()}, strict=False)
    assert not res.unexpected_keys
    last = model.model.mm_vis.config.num_hidden_layers - 1
    for k in res.missing_keys:
        assert (".head." in k or ".post_layernorm." in k or f".encoder.layers.{last}." in k), k


def run_case(model, cfg, input_ids, images, audios, audio_sizes, n_new):
    out = {}
    attention_mask = torch.ones_like(input_ids, dtype=torch.bool)       # what HF generate() supplies
    with torch.no_grad():
        r = model(input_ids=input_ids, attention_mask=attention_mask, images=images, audios=audios,
                  audio_sizes=audio_sizes, use_cache=True, output_hidden_states=True, return_dict=True)
        (_, _, am, _, inputs_embeds, _, img, imask, aud, amask) = model.prepare_inputs_labels_for_multimodal(
            input_ids, None, attention_mask, None, None, images, None, audios, audio_sizes)
    out["image_embeds"] = img.numpy(); out["image_mask"] = imask.numpy()
    out["audio_embeds"] = aud.numpy(); out["audio_mask"] = amask.numpy()
    out["prefill_hidden_last"] = r.hidden_states[-1].numpy()
    for li in range(cfg.num_hidden_layers):
        k, v = r.past_image_key_values[li]
        out[f"img_k_{li}"] = k.numpy(); out[f"img_v_{li}"] = v.numpy()
    logits = r.logits[:, -1]
    out["prefill_logits"] = logits.numpy()
    toks, step_logits = [], []
    text_mask = am.clone()
    pkv, pik, pak = r.past_key_values, r.past_image_key_values, r.past_audio_key_values
    for step in range(n_new):
        nxt = torch.argmax(logits, dim=-1)
        toks.append(nxt)
        if step == n_new - 1:
            break
        text_mask = torch.cat([text_mask, torch.ones(1, 1, dtype=text_mask.dtype)], dim=1)
        pos = (text_mask.long().sum(-1) - 1)[:, None]
        with torch.no_grad():
            r = model(input_ids=nxt[:, None], attention_mask=text_mask, position_ids=pos, past_key_values=pkv,
                      past_image_key_values=pik, past_audio_key_values=pak, image_embeds=img, image_attention_mask=imask,
                      audio_embeds=aud, audio_attention_mask=amask, use_cache=True,
                      cache_position=torch.tensor([text_mask.shape[1] - 1]), return_dict=True)
        # HF `_update_model_kwargs_for_generation` (mistral.py:181-195): thread the returned caches into the next step.
        # (past_audio_key_values comes back None: the layer call passes it under a misspelt keyword, mistral.py:412, so the
        # audio K/V are recomputed from audio_embeds every step — same values)
        pkv, pik, pak = r.past_key_values, r.past_image_key_values, r.past_audio_key_values
        logits = r.logits[:, -1]
        step_logits.append(logits.numpy())
    out["tokens"] = torch.stack(toks, dim=1).numpy()
    out["step_logits"] = np.stack(step_logits, axis=1)
    return out


def main():
    from vidi_amd.weights import init_random_weights
    cfg = golden_config()
    model, M7 = build_reference_model(cfg)
    w = init_random_weights(cfg, seed=3, dtype=torch.float32, device="cpu")
    load_weights(model, w)
    S, M, Fr = cfg.vis_image_size, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames
    g = torch.Generator().manual_seed(20260925)
    px = (torch.randn((1, 3, 3, S, S), generator=g) * 0.5).clamp(-1, 1)
    mel = torch.randn((1, 1, M, Fr), generator=g) * 0.3
    ids = torch.tensor([[1, 21, 22, 23, -200, 24, 25, 26, 300, 301]], dtype=torch.int64)
    a = run_case(model, cfg, ids, px, mel, [100], n_new=5)
    res = {"A_" + k: v for k, v in a.items()}
    res["A_input_ids"] = ids.numpy(); res["A_images"] = px.numpy(); res["A_audios"] = mel.numpy()
    res["A_audio_sizes"] = np.array([100])
    # ---- case E: the reference's OWN generate() (mistral.py:629-681 -> HF greedy loop threaded by :683-716), weights seed 6:
    # six different greedy tokens, per-step scores
    model.generation_config.eos_token_id = cfg.eos_token_id
    load_weights(model, init_random_weights(cfg, seed=6, dtype=torch.float32, device="cpu"))
    with torch.no_grad():
        g = model.generate(ids, images=px, audios=mel, audio_sizes=[100], do_sample=False, max_new_tokens=6, use_cache=True,
                           pad_token_id=0, output_scores=True, return_dict_in_generate=True)
    res["E_tokens"] = g.sequences.numpy(); res["E_scores"] = torch.stack(g.scores, dim=1).numpy()
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, f"{os.path.getsize(OUT) / 1e6:.2f} MB;", len(res), "arrays")


if __name__ == "__main__":
    main()
