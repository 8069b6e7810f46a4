"""Weight containers for the Vidi path: HF-style state-dict names (what a real checkpoint's
safetensors hold), random initialisation for synthetic runs, and safetensors loading.

Names follow the reference modules: `model.layers.N.*` (Gemma2), `model.mm_vis.vision_model.*`
(SiglipVisionModel), `model.mm_aud.encoder.*` (WhisperEncoder) and the `model.mm_rand_*` glue of
Vidi1.5_9B/vidi/model/lmm/dattn/multimodal.py:63-94."""
from __future__ import annotations

import glob
import os
from typing import Dict

import torch

from .config import VidiConfig


def weight_shapes(cfg: VidiConfig) -> Dict[str, tuple]:
    H, I, nq, nkv, hd = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    s: Dict[str, tuple] = {"model.embed_tokens.weight": (cfg.vocab_size, H), "model.norm.weight": (H,)}
    if not cfg.tie_word_embeddings:
        s["lm_head.weight"] = (cfg.vocab_size, H)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        s[p + "self_attn.q_proj.weight"] = (nq * hd, H)
        s[p + "self_attn.k_proj.weight"] = (nkv * hd, H)
        s[p + "self_attn.v_proj.weight"] = (nkv * hd, H)
        s[p + "self_attn.o_proj.weight"] = (H, nq * hd)
        s[p + "mlp.gate_proj.weight"] = (I, H)
        s[p + "mlp.up_proj.weight"] = (I, H)
        s[p + "mlp.down_proj.weight"] = (H, I)
        norms = ("input_layernorm", "post_attention_layernorm") if cfg.arch == "mistral" else \
            ("input_layernorm", "post_attention_layernorm", "pre_feedforward_layernorm", "post_feedforward_layernorm")
        for n in norms:
            s[p + n + ".weight"] = (H,)
    # SigLIP
    Hv, Iv, P = cfg.vis_hidden_size, cfg.vis_intermediate_size, cfg.vis_patch_size
    v = "model.mm_vis.vision_model."
    s[v + "embeddings.patch_embedding.weight"] = (Hv, 3, P, P)
    s[v + "embeddings.patch_embedding.bias"] = (Hv,)
    s[v + "embeddings.position_embedding.weight"] = (cfg.vis_side ** 2, Hv)
    for i in range(cfg.vis_select_layers):
        p = f"{v}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (Hv, Hv)
            s[p + f"self_attn.{n}.bias"] = (Hv,)
        s[p + "mlp.fc1.weight"] = (Iv, Hv); s[p + "mlp.fc1.bias"] = (Iv,)
        s[p + "mlp.fc2.weight"] = (Hv, Iv); s[p + "mlp.fc2.bias"] = (Hv,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (Hv,); s[p + n + ".bias"] = (Hv,)
    # Whisper encoder
    Da, Fa = cfg.aud_d_model, cfg.aud_ffn_dim
    a = "model.mm_aud.encoder."
    s[a + "conv1.weight"] = (Da, cfg.aud_num_mel_bins, 3); s[a + "conv1.bias"] = (Da,)
    s[a + "conv2.weight"] = (Da, Da, 3); s[a + "conv2.bias"] = (Da,)
    s[a + "embed_positions.weight"] = (cfg.aud_max_source_positions, Da)
    s[a + "layer_norm.weight"] = (Da,); s[a + "layer_norm.bias"] = (Da,)
    for i in range(cfg.aud_num_layers):
        p = f"{a}layers.{i}."
        for n in ("q_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (Da, Da); s[p + f"self_attn.{n}.bias"] = (Da,)
        s[p + "self_attn.k_proj.weight"] = (Da, Da)
        s[p + "fc1.weight"] = (Fa, Da); s[p + "fc1.bias"] = (Fa,)
        s[p + "fc2.weight"] = (Da, Fa); s[p + "fc2.bias"] = (Da,)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            s[p + n + ".weight"] = (Da,); s[p + n + ".bias"] = (Da,)
    # mm glue
    m = "model."
    pool = cfg.mm_image_pool_size
    if cfg.arch == "mistral":                                   # learned Conv2DPool feeds the projector directly
        k = cfg.img_pool_kernel
        s[m + "mm_rand_img_pool.conv.weight"] = (Hv, Hv, k, k)
        s[m + "mm_rand_img_projector.model.0.weight"] = (H, Hv)
    else:
        s[m + "mm_rand_img_projector.model.0.weight"] = (H, Hv * pool * pool)
    s[m + "mm_rand_img_projector.model.0.bias"] = (H,)
    s[m + "mm_rand_img_projector.model.2.weight"] = (H, H); s[m + "mm_rand_img_projector.model.2.bias"] = (H,)
    # Vidi1.5: Conv1d(Da -> H) then MLP(H -> H) (Vidi1.5_9B/.../multimodal.py:85-92); Vidi-7B: Conv1d(Da -> Da) then
    # MLP(Da -> H) (Vidi_7B/.../multimodal.py:80-87)
    Dp = Da if cfg.arch == "mistral" else H
    s[m + "mm_rand_aud_pool.weight"] = (Dp, Da, cfg.mm_audio_pool_size)
    s[m + "mm_rand_aud_projector.model.0.weight"] = (H, Dp); s[m + "mm_rand_aud_projector.model.0.bias"] = (H,)
    s[m + "mm_rand_aud_projector.model.2.weight"] = (H, H); s[m + "mm_rand_aud_projector.model.2.bias"] = (H,)
    for n in ("mm_rand_img_norm", "mm_rand_aud_norm", "mm_rand_llm_norm"):
        s[m + n + ".weight"] = (H,)
    for n in ("mm_rand_pos_h", "mm_rand_pos_w", "mm_rand_pos_t"):
        s[m + n + ".mlp.0.weight"] = (H, H); s[m + n + ".mlp.0.bias"] = (H,)
        s[m + n + ".mlp.2.weight"] = (H, H); s[m + n + ".mlp.2.bias"] = (H,)
    return s


def is_fp32_param(name: str) -> bool:
    """LearnablePosEmbd's MLP is built with dtype=float32 and computes in fp32 (mm_vision/pos.py:36-39)."""
    return ".mm_rand_pos_" in name


def init_random_weights(cfg: VidiConfig, seed: int = 3, dtype: torch.dtype = torch.bfloat16, device="cpu") -> Dict[str, torch.Tensor]:
    """SURVEY.md §8(d) synthetic weights: Linear ~ N(0,0.02), biases small, norm weights near their
    identity value with a perturbation so the weight paths are exercised, llm_norm = mm_std."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in weight_shapes(cfg).items():
        dt = torch.float32 if is_fp32_param(name) else dtype
        if name.endswith("mm_rand_llm_norm.weight"):
            t = torch.full(shape, cfg.mm_std, dtype=torch.float32, device=dev)
        elif ("layernorm.weight" in name or name == "model.norm.weight") and cfg.arch != "mistral":   # Gemma (1+w) form
            t = torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * 0.1
        elif "layernorm.weight" in name or name == "model.norm.weight":        # Mistral plain-weight form
            t = 1.0 + torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * 0.1
        elif name.endswith("norm.weight") or "layer_norm" in name and name.endswith(".weight"):
            t = 1.0 + torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * 0.1
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * 0.02
        elif name.endswith("position_embedding.weight") or name.endswith("embed_positions.weight"):
            t = torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * 0.02
        elif name == "model.embed_tokens.weight":
            t = torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * 0.02
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * min(0.02 * (3584 / max(fan_in, 1)) ** 0.5, 0.08)
        out[name] = t.to(dt)
    return out


def load_safetensors_dir(path: str, device="cpu") -> Dict[str, torch.Tensor]:
    """Load every *.safetensors shard under `path` into one state dict (real checkpoints)."""
    from safetensors.torch import load_file
    out: Dict[str, torch.Tensor] = {}
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        # older exports: pytorch_model.bin / pytorch_model-0000x-of-0000y.bin (what HF's from_pretrained falls back to, builder.py:41-43)
        bins = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
        if not bins:
            raise FileNotFoundError(f"no *.safetensors (or pytorch_model*.bin) under {path}")
        for f in bins:
            out.update(torch.load(f, map_location=str(device), weights_only=True))
        return out
    for f in files:
        out.update(load_file(f, device=str(device)))
    return out


# ------------------------------------------------------------------------------------------------
# real checkpoints: the reference builds both towers from their own repositories (`mm_vision_tower` / `mm_audio_tower`,
# multimodal.py:44-57) and tolerates their absence from the Vidi checkpoint (gemma.py:469 `_keys_to_ignore_on_load_missing`)
# ------------------------------------------------------------------------------------------------
def resolve_tower_dir(name: str, model_path: str):
    """A tower reference that is reachable offline: the name itself if it is a directory, else <model_path>/<name> or
    <model_path>/<basename(name)>; None otherwise (hub names cannot be fetched here)."""
    if not name:
        return None
    for cand in (name, os.path.join(model_path, name), os.path.join(model_path, os.path.basename(name.rstrip("/")))):
        if os.path.isdir(cand):
            return cand
    return None


def _tower_state(tower_dir: str, kind: str) -> Dict[str, torch.Tensor]:
    """HF tower checkpoint -> the names the Vidi state dict uses.  SigLIP: `vision_model.*` -> `model.mm_vis.vision_model.*`;
    Whisper: `model.encoder.*` or `encoder.*` -> `model.mm_aud.encoder.*`."""
    sd = load_safetensors_dir(tower_dir)
    out = {}
    for k, v in sd.items():
        if kind == "vis":
            if k.startswith("vision_model."):
                out["model.mm_vis." + k] = v
        else:
            kk = k[len("model."):] if k.startswith("model.") else k
            if kk.startswith("encoder."):
                out["model.mm_aud." + kk] = v
    return out


def load_checkpoint(path: str, cfg: VidiConfig) -> Dict[str, torch.Tensor]:
    """State dict of a Vidi checkpoint directory with the reference's parameter names; tower weights missing from it are taken
    from the tower directories named by the config.  Raises with the list of absent parameters otherwise (shapes are checked)."""
    sd = load_safetensors_dir(path)
    shapes = weight_shapes(cfg)
    for kind, prefix, name in (("vis", "model.mm_vis.", cfg.mm_vision_tower), ("aud", "model.mm_aud.", cfg.mm_audio_tower)):
        if any(k.startswith(prefix) and k not in sd for k in shapes):
            tdir = resolve_tower_dir(name, path)
            if tdir is not None:
                for k, v in _tower_state(tdir, kind).items():
                    sd.setdefault(k, v)
    missing = [k for k in shapes if k not in sd]
    if missing:
        raise KeyError(f"{len(missing)} parameters absent from {path} (first: {missing[:6]}).  Tower weights may live in a local copy of "
                       f"config.mm_vision_tower={cfg.mm_vision_tower!r} / mm_audio_tower={cfg.mm_audio_tower!r}: a directory of that name "
                       f"(absolute, or under the checkpoint directory) with HF-format *.safetensors")
    bad = [(k, tuple(sd[k].shape), shapes[k]) for k in shapes if tuple(sd[k].shape) != tuple(shapes[k])]
    if bad:
        raise ValueError(f"{len(bad)} parameters with unexpected shapes, e.g. {bad[:4]} (checkpoint vs config)")
    return sd
