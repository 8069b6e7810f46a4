"""Model configuration for the MI355X Vidi path.

Field names follow the reference's `DattnGemma2Config` (Vidi1.5_9B/vidi/model/lmm/dattn/gemma.py:427-448)
plus the HF tower configs it pulls in (SigLIP / Whisper).  Real checkpoints provide these through
`config.json`; nothing in the kernels hard-codes them."""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass, fields
from typing import Optional


@dataclass
class VidiConfig:
    model_type: str = "dattn_gemma2"
    # ---- LLM (Gemma2) ----
    hidden_size: int = 3584
    intermediate_size: int = 14336
    num_hidden_layers: int = 42
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 256
    query_pre_attn_scalar: float = 256.0
    attn_logit_softcapping: Optional[float] = 50.0
    final_logit_softcapping: Optional[float] = 30.0
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    sliding_window: int = 4096
    vocab_size: int = 256000
    eos_token_id: int = 107
    pad_token_id: int = 0
    bos_token_id: int = 2
    tie_word_embeddings: bool = True
    # ---- multimodal glue (mm_* keys of the reference config) ----
    mm_input_type: str = "video"
    mm_projector_type: str = "mlp2x_gelu"
    mm_image_aspect_ratio: str = "resize"
    mm_image_pool_size: int = 2
    mm_audio_pool_size: int = 5
    mm_std: float = 0.02898
    mm_time_interval: int = 10000
    mm_splits: int = 1                      # activation-chunking knob of the reference; results are chunk-invariant
    mm_max_tokens_base: int = 60000         # literal at multimodal.py:176
    mm_vision_tower: str = "google/siglip2-so400m-patch14-384"
    mm_audio_tower: str = "openai/whisper-large-v3"
    mm_vision_select_layer: int = -2
    # ---- vision tower ----
    vis_image_size: int = 384
    vis_patch_size: int = 14
    vis_hidden_size: int = 1152
    vis_intermediate_size: int = 4304
    vis_num_layers: int = 27
    vis_num_heads: int = 16
    vis_ln_eps: float = 1e-6
    # ---- audio tower ----
    aud_num_mel_bins: int = 128
    aud_d_model: int = 1280
    aud_num_layers: int = 32
    aud_num_heads: int = 20
    aud_ffn_dim: int = 5120
    aud_max_source_positions: int = 1500
    aud_nb_max_frames: int = 3000
    aud_ln_eps: float = 1e-5
    aud_sampling_rate: int = 16000
    aud_hop_length: int = 160
    # ---- engine knobs (ours) ----
    vis_frames_per_chunk: int = 720         # SigLIP activation chunk (frames): ~2050 M-tiles per GEMM => <1% tail waves (720 vs 360: +1% measured)
    aud_chunks_per_batch: int = 60          # Whisper activation chunk (30-s windows)

    @property
    def arch(self) -> str:
        """'gemma2' (Vidi1.5-9B, DattnGemma2*) or 'mistral' (Vidi-7B, DattnMistral*: Vidi_7B/model/lmm/dattn/mistral.py)."""
        return "mistral" if "mistral" in self.model_type else "gemma2"

    @property
    def vis_side(self) -> int:
        return self.vis_image_size // self.vis_patch_size

    @property
    def img_pool_kernel(self) -> int:
        """Vidi-7B learned Conv2DPool kernel size: ceil(s_in / s_out) (Vidi_7B/model/mm_vision/pool.py:15-17)."""
        return -(-self.vis_side // self.mm_image_pool_size)

    @property
    def vis_select_layers(self) -> int:
        """number of encoder layers actually needed for hidden_states[select_layer]"""
        return self.mm_vision_select_layer % (self.vis_num_layers + 1)

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_dict(cls, d: dict) -> "VidiConfig":
        known = {f.name for f in fields(cls)}
        d = dict(d)
        if "mistral" in str(d.get("model_type", "")):
            # a Vidi-7B config.json is a MistralConfig + mm_* keys: no head_dim / query_pre_attn_scalar / softcaps
            hd = d.get("head_dim") or d["hidden_size"] // d["num_attention_heads"]
            d.setdefault("head_dim", hd)
            d["head_dim"] = hd
            d.setdefault("query_pre_attn_scalar", float(hd))
            d.setdefault("attn_logit_softcapping", None)
            d.setdefault("final_logit_softcapping", None)
            d.setdefault("tie_word_embeddings", False)
            d.setdefault("rms_norm_eps", 1e-5)
            if d.get("sliding_window") is None:
                d["sliding_window"] = 1 << 30
        return cls(**{k: v for k, v in d.items() if k in known})

    @classmethod
    def from_pretrained(cls, path: str) -> "VidiConfig":
        """config.json of a Vidi checkpoint (a Gemma2 / Mistral config + mm_* keys).  The tower dimensions are not in it: the
        reference builds the towers from `mm_vision_tower` / `mm_audio_tower` (multimodal.py:44-57); when those resolve to local
        directories their config.json provides the dimensions, otherwise the published dims of the default towers apply."""
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        cfg = cls.from_dict(raw)
        from .weights import resolve_tower_dir
        vdir = resolve_tower_dir(cfg.mm_vision_tower, path)
        if vdir and os.path.exists(os.path.join(vdir, "config.json")):
            t = json.load(open(os.path.join(vdir, "config.json")))
            t = t.get("vision_config", t)
            for ours, theirs in (("vis_image_size", "image_size"), ("vis_patch_size", "patch_size"), ("vis_hidden_size", "hidden_size"),
                                 ("vis_intermediate_size", "intermediate_size"), ("vis_num_layers", "num_hidden_layers"),
                                 ("vis_num_heads", "num_attention_heads"), ("vis_ln_eps", "layer_norm_eps")):
                if theirs in t and ours not in raw:
                    setattr(cfg, ours, t[theirs])
        adir = resolve_tower_dir(cfg.mm_audio_tower, path)
        if adir and os.path.exists(os.path.join(adir, "config.json")):
            t = json.load(open(os.path.join(adir, "config.json")))
            for ours, theirs in (("aud_num_mel_bins", "num_mel_bins"), ("aud_d_model", "d_model"), ("aud_num_layers", "encoder_layers"),
                                 ("aud_num_heads", "encoder_attention_heads"), ("aud_ffn_dim", "encoder_ffn_dim"),
                                 ("aud_max_source_positions", "max_source_positions")):
                if theirs in t and ours not in raw:
                    setattr(cfg, ours, t[theirs])
            if "max_source_positions" in t and "aud_nb_max_frames" not in raw:
                cfg.aud_nb_max_frames = 2 * t["max_source_positions"]
        return cfg

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=1)


def vidi15_9b() -> VidiConfig:
    """External dims of Vidi1.5-9B: google/gemma-2-9b + siglip2-so400m-patch14-384 + whisper-large-v3
    (Vidi1.5_9B/scripts/finetune.sh:18,22,25)."""
    return VidiConfig()


def vidi_7b() -> VidiConfig:
    """External dims of Vidi-7B: Mistral-7B (32 layers, 32/8 heads x 128, SiLU-GLU 14336, no softcaps, untied
    lm_head) + siglip-so400m-patch14-384 + whisper-large-v3 with the learned Conv2DPool
    (Vidi_7B/model/lmm/dattn/mistral.py:456-470, multimodal.py:63-94).  mm_image_pool_size / vocab come from the
    checkpoint's config.json; the values here are the ones used for synthetic runs."""
    return VidiConfig(
        model_type="dattn_mistral", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
        num_key_value_heads=8, head_dim=128, query_pre_attn_scalar=128.0, attn_logit_softcapping=None,
        final_logit_softcapping=None, rms_norm_eps=1e-5, rope_theta=10000.0, sliding_window=4096, vocab_size=32000,
        eos_token_id=2, pad_token_id=0, bos_token_id=1, tie_word_embeddings=False,
        mm_image_pool_size=2, mm_vision_tower="google/siglip-so400m-patch14-384")


def tiny(**over) -> VidiConfig:
    """Small config exercising every padding path (odd patch grid, tower head_dim 16, fc1 pad 176->192, GQA 4/2)."""
    cfg = VidiConfig(
        hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
        head_dim=128, query_pre_attn_scalar=128.0, sliding_window=16, vocab_size=512, eos_token_id=7,
        mm_time_interval=100, mm_max_tokens_base=60000,
        vis_image_size=98, vis_patch_size=14, vis_hidden_size=64, vis_intermediate_size=176, vis_num_layers=3, vis_num_heads=4,
        aud_num_mel_bins=64, aud_d_model=64, aud_num_layers=2, aud_num_heads=4, aud_ffn_dim=128,
        aud_max_source_positions=50, aud_nb_max_frames=100,
        vis_frames_per_chunk=4, aud_chunks_per_batch=2,
    )
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def tiny_7b(**over) -> VidiConfig:
    """Small Mistral-arch (Vidi-7B) config: plain RMSNorm, SiLU-GLU, untied head, learned conv pool (7 -> 2, k = 4)."""
    cfg = tiny(model_type="dattn_mistral", attn_logit_softcapping=None, final_logit_softcapping=None, rms_norm_eps=1e-5,
               tie_word_embeddings=False, mm_image_pool_size=2, head_dim=128, query_pre_attn_scalar=128.0,
               num_attention_heads=4, num_key_value_heads=2)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg
