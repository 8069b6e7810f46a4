"""CPU oracle for the Vidi inference hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain eager PyTorch on the CPU, the arithmetic of the reference's
multimodal inference forward (bytedance/vidi, `Vidi1.5_9B/vidi/...`).  It exists so that the HIP
kernels in `vidi_amd/csrc` can be checked against something that follows the reference line by
line.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
nothing under `vidi_amd/` does.  It is never the product path.

Parity status: the reference ships no tests, golden vectors or known-answer values and cannot
run as shipped in this environment (flash-attn is CUDA-only; liger / deepspeed / the transformers-4.50 pin are absent), so
this oracle is pinned on OUTPUTS OF THE REFERENCE'S OWN CODE EXECUTED IN THE BUILD CONTAINER:
  * `tests/golden/ref_harness.py` replaces only the absent third-party packages and then imports and runs the reference's
    model code unmodified on CPU/fp32; `make_golden_dattn.py` / `make_golden_dattn_7b.py` drive
    `DattnGemma2ForCausalLM.forward` / `DattnMistralForCausalLM.forward` end to end (encode pipelines, every decoder layer,
    all three caches, greedy decode steps, padded batches) and commit the results; `tests/test_oracle_golden.py` holds this
    file to them — bit-exact for masks/indices/tokens, 2e-5 for floats (observed 5e-7);
  * the small reference modules importable by file path (`mm_layer/norm.py`, `mm_layer/mlp.py`, `mm_vision/pool.py`,
    `mm_vision/pos.py`, `vidi/utils.py`) are pinned bit-exactly the same way (`make_golden.py`, `make_golden_7b.py`);
  * the third-party blocks (Gemma2/Mistral RMSNorm/MLP/RoPE/attention, SigLIP and Whisper encoders) are additionally checked
    against the installed `transformers` eager implementations in `tests/test_oracle_hf.py`;
  * restated, not executed: flash-attn's kernels (CUDA-only) — `flash_attn_func(softcap=..)` follows its published definition
    `softmax(softcap * tanh(q k^T * scale / softcap)) v` with fp32 softmax, both here and in the harness stand-in.

All functions are dtype-generic: fed fp32 tensors they are the fp32 oracle; fed bf16/fp16 tensors
they reproduce the reference's eager rounding points (every nn.Module output rounds to the model
dtype, norms compute in fp32 internally) which is what the GPU path imitates.

Citations are relative to /root/reference/Vidi1.5_9B/vidi/ unless noted (TP/ = transformers).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
W = Dict[str, Tensor]

IGNORE_INDEX = -100          # constants.py:10
IMAGE_TOKEN_INDEX = -200     # constants.py:11


# --------------------------------------------------------------------------------------------
# configuration (mirrors DattnGemma2Config + tower configs; values come from config.json in real
# checkpoints — gemma.py:427-448; nothing here is hard-coded into the math)
# --------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    # "gemma2" = Vidi1.5-9B (Vidi1.5_9B/vidi/model/lmm/dattn/gemma.py); "mistral" = Vidi-7B
    # (Vidi_7B/model/lmm/dattn/mistral.py): plain RMSNorm, no post-norms, SiLU-GLU, no softcaps, no embedding
    # normalizer, learned Conv2DPool (Vidi_7B/model/mm_vision/pool.py) instead of the token-budget resize
    arch: str = "gemma2"
    # LLM
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
    eos_token_id: int = 107                      # gemma.py:461
    # multimodal glue
    mm_image_pool_size: int = 2
    mm_audio_pool_size: int = 5
    mm_std: float = 0.02898
    mm_time_interval: int = 10000
    mm_max_tokens_base: int = 60000              # literal at multimodal.py:176
    # vision tower (SigLIP)
    vis_image_size: int = 384
    vis_patch_size: int = 14
    vis_hidden_size: int = 1152
    vis_intermediate_size: int = 4304
    vis_num_layers: int = 27
    vis_num_heads: int = 16
    vis_ln_eps: float = 1e-6
    vis_select_layer: int = -2                   # siglip.py:17
    # audio tower (Whisper encoder)
    aud_num_mel_bins: int = 128
    aud_d_model: int = 1280
    aud_num_layers: int = 32
    aud_num_heads: int = 20
    aud_ffn_dim: int = 5120
    aud_max_source_positions: int = 1500
    aud_nb_max_frames: int = 3000
    aud_ln_eps: float = 1e-5

    @property
    def vis_side(self) -> int:
        return self.vis_image_size // self.vis_patch_size


# --------------------------------------------------------------------------------------------
# norms / activations
# --------------------------------------------------------------------------------------------
def gemma_rmsnorm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """Gemma2RMSNorm.forward — TP/models/gemma2/modeling_gemma2.py:55-63.
    x*rsqrt(mean(x^2)+eps)*(1+w), all in fp32, then cast back."""
    xf = x.float()
    out = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    out = out * (1.0 + weight.float())
    return out.type_as(x)


def mistral_rmsnorm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """MistralRMSNorm (transformers modeling_mistral, used by Vidi_7B/.../mistral.py:131-134): variance in
    fp32, normalised tensor cast back to the input dtype, THEN multiplied by the weight."""
    xf = x.float()
    h = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return weight * h.to(x.dtype)


def llm_rmsnorm(x: Tensor, weight: Tensor, cfg: "OracleConfig") -> Tensor:
    if cfg.arch == "mistral":
        return mistral_rmsnorm(x, weight, cfg.rms_norm_eps)
    return gemma_rmsnorm(x, weight, cfg.rms_norm_eps)


def mm_rms_norm(x: Tensor, eps: float = 1e-5) -> Tensor:
    """rms_norm — model/mm_layer/norm.py:9-16 (weight-free, fp32 inside, cast back)."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    return (xf * torch.rsqrt(var + eps)).to(dt)


def mm_RMSNorm(x: Tensor, weight: Tensor, eps: float = 1e-5) -> Tensor:
    """RMSNorm.forward — model/mm_layer/norm.py:19-25: weight * rms_norm(x); the weight multiply
    happens AFTER the cast back to the input dtype (in the parameter/input promoted dtype)."""
    return weight * mm_rms_norm(x, eps)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu_tanh(x: Tensor) -> Tensor:
    return F.gelu(x, approximate="tanh")


def gelu_erf(x: Tensor) -> Tensor:
    return F.gelu(x)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    return F.linear(x, w, b)


# --------------------------------------------------------------------------------------------
# attention primitives
# --------------------------------------------------------------------------------------------
def sdpa_reference(q: Tensor, k: Tensor, v: Tensor, scale: float, softcap: Optional[float] = None,
                   add_mask: Optional[Tensor] = None) -> Tensor:
    """softmax(softcap*tanh(q k^T*scale/softcap) + mask) v with fp32 softmax.
    q:[B,H,Lq,D] k,v:[B,H,Lk,D].  This is flash_attn_func/flash_attn_varlen_func's math
    (xattn.py:123,253; flash-attn 2.8.3 `softcap` argument) and HF eager attention
    (TP/models/gemma2/modeling_gemma2.py:184-215)."""
    s = torch.matmul(q, k.transpose(2, 3)) * scale
    if softcap is not None:
        s = torch.tanh(s / softcap) * softcap
    if add_mask is not None:
        s = s + add_mask
    p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(p, v)


def repeat_kv(x: Tensor, n_rep: int) -> Tensor:
    """TP/models/gemma2/modeling_gemma2.py:171-181.  x:[B,Hkv,L,D] -> [B,Hkv*n_rep,L,D]."""
    b, h, l, d = x.shape
    if n_rep == 1:
        return x
    return x[:, :, None].expand(b, h, n_rep, l, d).reshape(b, h * n_rep, l, d)


def rope_cos_sin(position_ids: Tensor, dim: int, theta: float, dtype: torch.dtype) -> Tuple[Tensor, Tensor]:
    """Gemma2RotaryEmbedding.forward — TP/models/gemma2/modeling_gemma2.py:118-136 (fp32, then cast).
    position_ids:[B,L] -> cos,sin:[B,L,dim]."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: Tensor) -> Tensor:
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> Tuple[Tensor, Tensor]:
    """apply_rotary_pos_emb — TP/models/gemma2/modeling_gemma2.py:146-168 (heads at dim 1)."""
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


# --------------------------------------------------------------------------------------------
# SigLIP vision tower (HF SiglipVisionModel; features = hidden_states[-2])  — siglip.py:29-34
# --------------------------------------------------------------------------------------------
def siglip_forward(pixel: Tensor, w: W, cfg: OracleConfig, prefix: str = "model.mm_vis.vision_model.") -> Tensor:
    """pixel:[T,3,S,S] -> hidden_states[select_layer]:[T,side^2,Hv].
    hidden_states = (embeddings, layer1_out, ..., layerL_out); index -2 = output of layer L-1
    (no post-LN) — TP/models/siglip/modeling_siglip.py:116-187 (embed), 250-357 (layer)."""
    p = prefix
    x = F.conv2d(pixel, w[p + "embeddings.patch_embedding.weight"], w[p + "embeddings.patch_embedding.bias"],
                 stride=cfg.vis_patch_size)
    x = x.flatten(2).transpose(1, 2)                                  # [T, N, Hv]
    x = x + w[p + "embeddings.position_embedding.weight"][None]
    n_states = cfg.vis_num_layers + 1
    target = cfg.vis_select_layer % n_states                          # -2 -> L-1 layers applied
    for i in range(target):
        x = siglip_layer(x, w, cfg, f"{p}encoder.layers.{i}.")
    return x


def siglip_layer(x: Tensor, w: W, cfg: OracleConfig, lp: str) -> Tensor:
    """one SiglipEncoderLayer on x:[T,N,Hv] — TP/models/siglip/modeling_siglip.py:250-357 (pre-LN attention + pre-LN MLP)."""
    nh = cfg.vis_num_heads
    hd = cfg.vis_hidden_size // nh
    r = x
    h = layer_norm(x, w[lp + "layer_norm1.weight"], w[lp + "layer_norm1.bias"], cfg.vis_ln_eps)
    T, N, _ = h.shape
    q = linear(h, w[lp + "self_attn.q_proj.weight"], w[lp + "self_attn.q_proj.bias"]).view(T, N, nh, hd).transpose(1, 2)
    k = linear(h, w[lp + "self_attn.k_proj.weight"], w[lp + "self_attn.k_proj.bias"]).view(T, N, nh, hd).transpose(1, 2)
    v = linear(h, w[lp + "self_attn.v_proj.weight"], w[lp + "self_attn.v_proj.bias"]).view(T, N, nh, hd).transpose(1, 2)
    a = sdpa_reference(q, k, v, hd ** -0.5).transpose(1, 2).reshape(T, N, nh * hd)
    a = linear(a, w[lp + "self_attn.out_proj.weight"], w[lp + "self_attn.out_proj.bias"])
    x = r + a
    r = x
    h = layer_norm(x, w[lp + "layer_norm2.weight"], w[lp + "layer_norm2.bias"], cfg.vis_ln_eps)
    h = linear(h, w[lp + "mlp.fc1.weight"], w[lp + "mlp.fc1.bias"])
    h = gelu_tanh(h)                                                  # hidden_act = gelu_pytorch_tanh
    h = linear(h, w[lp + "mlp.fc2.weight"], w[lp + "mlp.fc2.bias"])
    return r + h


# --------------------------------------------------------------------------------------------
# Whisper encoder (HF WhisperEncoder) — mm_audio/whisper.py:26-27, TP/models/whisper/modeling_whisper.py
# --------------------------------------------------------------------------------------------
def whisper_encoder_forward(mel: Tensor, w: W, cfg: OracleConfig, prefix: str = "model.mm_aud.encoder.") -> Tensor:
    """mel:[C,n_mels,3000] -> [C,1500,d_model]."""
    p = prefix
    x = gelu_erf(F.conv1d(mel, w[p + "conv1.weight"], w[p + "conv1.bias"], padding=1))
    x = gelu_erf(F.conv1d(x, w[p + "conv2.weight"], w[p + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    x = x + w[p + "embed_positions.weight"][None]
    for i in range(cfg.aud_num_layers):
        x = whisper_layer(x, w, cfg, f"{p}layers.{i}.")
    return layer_norm(x, w[p + "layer_norm.weight"], w[p + "layer_norm.bias"], cfg.aud_ln_eps)


def whisper_layer(x: Tensor, w: W, cfg: OracleConfig, lp: str) -> Tensor:
    """one WhisperEncoderLayer on x:[C,N,d] — TP/models/whisper/modeling_whisper.py:279-333, 380-411."""
    nh = cfg.aud_num_heads
    hd = cfg.aud_d_model // nh
    r = x
    h = layer_norm(x, w[lp + "self_attn_layer_norm.weight"], w[lp + "self_attn_layer_norm.bias"], cfg.aud_ln_eps)
    C, N, _ = h.shape
    # q is scaled BEFORE the attention call (modeling_whisper.py:309), scaling=1.0 inside
    q = (linear(h, w[lp + "self_attn.q_proj.weight"], w[lp + "self_attn.q_proj.bias"]) * (hd ** -0.5))
    q = q.view(C, N, nh, hd).transpose(1, 2)
    k = linear(h, w[lp + "self_attn.k_proj.weight"], None).view(C, N, nh, hd).transpose(1, 2)
    v = linear(h, w[lp + "self_attn.v_proj.weight"], w[lp + "self_attn.v_proj.bias"]).view(C, N, nh, hd).transpose(1, 2)
    a = sdpa_reference(q, k, v, 1.0).transpose(1, 2).reshape(C, N, nh * hd)
    a = linear(a, w[lp + "self_attn.out_proj.weight"], w[lp + "self_attn.out_proj.bias"])
    x = r + a
    r = x
    h = layer_norm(x, w[lp + "final_layer_norm.weight"], w[lp + "final_layer_norm.bias"], cfg.aud_ln_eps)
    h = gelu_erf(linear(h, w[lp + "fc1.weight"], w[lp + "fc1.bias"]))
    h = linear(h, w[lp + "fc2.weight"], w[lp + "fc2.bias"])
    x = r + h
    if x.dtype == torch.float16:                                      # modeling_whisper.py:409-411
        cv = torch.finfo(x.dtype).max - 1000
        x = torch.clamp(x, min=-cv, max=cv)
    return x


# --------------------------------------------------------------------------------------------
# frame pooling / token budget  — mm_vision/pool.py, vidi/utils.py
# --------------------------------------------------------------------------------------------
def space_to_depth(x: Tensor, m: int = 2) -> Tensor:
    """utils.py:134-150.  [B,C,H,W] -> [B,C*m*m,H/m,W/m], out channel = c*m*m + dy*m + dx."""
    B, C, H, Wd = x.shape
    assert H % m == 0 and Wd % m == 0
    x = x.reshape(B, C, H // m, m, Wd // m, m).permute(0, 1, 3, 5, 2, 4)
    return x.reshape(B, C * m * m, H // m, Wd // m)


def resize_by_tokens_hw(B: int, H: int, Wd: int, max_tokens: int) -> Tuple[int, int]:
    """utils.py:152-171 on the padded (H,W)=(side+1,side+1) grid.  Pure integer/float rule."""
    ratio = math.sqrt(max_tokens / (B * H * Wd))
    th, tw = int(H * ratio), int(Wd * ratio)
    return max(10, th - th % 2), max(10, tw - tw % 2)


def token_budget_hw(T: int, side: int, pool: int, base: int = 60000) -> Tuple[int, int]:
    """multimodal.py:175-180.  Returns the `hw` handed to Conv2DPool; the literal (28,28) means
    'no resize' for the reference's 27-patch tower and is kept as the sentinel."""
    n_tokens = T * (side + 1) * (side + 1)
    max_tokens = base * pool * pool
    if n_tokens > max_tokens:
        return resize_by_tokens_hw(T, side + 1, side + 1, max_tokens)
    return 28, 28


def learned_conv2d_pool(x: Tensor, weight: Tensor, s_out: int) -> Tensor:
    """Vidi-7B Conv2DPool.forward — Vidi_7B/model/mm_vision/pool.py:19-26: bias-free conv with kernel
    ceil(s_in/s_out) (stride 1), then bilinear resize to s_out x s_out with align_corners=True."""
    x = F.conv2d(x, weight)
    assert x.shape[-1] >= s_out
    return F.interpolate(x, size=s_out, mode="bilinear", align_corners=True)


def conv2d_pool(x: Tensor, hw: Tuple[int, int], merge: int = 2) -> Tensor:
    """Conv2DPool.forward — mm_vision/pool.py:23-32.  x:[B,C,side,side]."""
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    if hw[0] != 28:
        x = F.interpolate(x, size=hw, mode="bilinear", align_corners=False)
    return space_to_depth(x, merge)


# --------------------------------------------------------------------------------------------
# learnable positional embedding — mm_vision/pos.py
# --------------------------------------------------------------------------------------------
def fractional_sinusoid(position: Tensor, d: int) -> Tensor:
    """FractionalSinusoidalEmbedding — pos.py:11-26: pe[:,0::2]=sin, pe[:,1::2]=cos (fp32)."""
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float) * -(math.log(10000.0) / d))
    pe = torch.zeros(len(position), d, dtype=torch.float)
    pos = position.float().unsqueeze(1)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def learnable_pos_embd(l: int, N: int, d: int, w: W, prefix: str, out_dtype: torch.dtype) -> Tensor:
    """LearnablePosEmbd.forward (eval) — pos.py:41-65.  Returns [l,d] in out_dtype; the MLP runs
    in fp32 on fp32-cast weights (mlp.py:31-40) with exact-erf GELU."""
    assert l > 1
    p = torch.arange(l, dtype=torch.float)
    p = p / (l - 1) * (N - 1)
    pe = fractional_sinusoid(p, d)
    h = F.linear(pe, w[prefix + "mlp.0.weight"].float(), w[prefix + "mlp.0.bias"].float())
    h = F.gelu(h)
    h = F.linear(h, w[prefix + "mlp.2.weight"].float(), w[prefix + "mlp.2.bias"].float())
    return h.to(out_dtype)


def projector_mlp(x: Tensor, w: W, prefix: str) -> Tensor:
    """MLP('mlp2x_gelu') — mm_layer/mlp.py:9-28: Linear, GELU(erf), Linear."""
    h = linear(x, w[prefix + "model.0.weight"], w[prefix + "model.0.bias"])
    h = gelu_erf(h)
    return linear(h, w[prefix + "model.2.weight"], w[prefix + "model.2.bias"])


# --------------------------------------------------------------------------------------------
# encode_video_images / encode_video_audios — lmm/dattn/multimodal.py:156-252
# --------------------------------------------------------------------------------------------
def encode_video_images(images: Sequence[Tensor], w: W, cfg: OracleConfig,
                        vis_features: Optional[Tensor] = None, budget_frames: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """images: list (batch) of [T_i,3,S,S].  Returns (features[B,Nv,H], mask[B,Nv] bool).
    `vis_features` lets a caller inject precomputed tower outputs (for slice tests); `budget_frames` overrides the frame count the
    token-budget rule sees (:175-180 counts the concatenated frames of the WHOLE batch) so that one video of a batch can be evaluated
    on its own with the batch's budget."""
    m = "model."
    split_sizes = [im.shape[0] for im in images]
    concat = torch.cat(list(images), dim=0)
    feats = siglip_forward(concat, w, cfg) if vis_features is None else vis_features   # :163-169
    side = cfg.vis_side
    feats = feats.reshape(len(feats), side, side, -1).permute(0, 3, 1, 2)              # :171-173
    if cfg.arch == "mistral":                                                           # Vidi_7B/.../multimodal.py:165-170
        feats = learned_conv2d_pool(feats, w[m + "mm_rand_img_pool.conv.weight"], cfg.mm_image_pool_size)
    else:
        hw = token_budget_hw(feats.size(0) if budget_frames is None else budget_frames, side, cfg.mm_image_pool_size,
                             cfg.mm_max_tokens_base)                                   # :175-180
        feats = conv2d_pool(feats, hw, cfg.mm_image_pool_size)                          # :182-189
    feats = feats.permute(0, 2, 3, 1)                                                   # :190
    feats = projector_mlp(feats, w, m + "mm_rand_img_projector.")                       # :192
    feats = mm_RMSNorm(feats, w[m + "mm_rand_img_norm.weight"])                         # :193
    d = cfg.hidden_size
    ph = learnable_pos_embd(feats.shape[1], cfg.mm_image_pool_size, d, w, m + "mm_rand_pos_h.", feats.dtype)
    feats = feats + mm_rms_norm(ph.reshape(1, -1, 1, d))                                # :194
    pw = learnable_pos_embd(feats.shape[2], cfg.mm_image_pool_size, d, w, m + "mm_rand_pos_w.", feats.dtype)
    feats = feats + mm_rms_norm(pw.reshape(1, 1, -1, d))                                # :195
    per = torch.split(feats, split_sizes, dim=0)                                        # :196
    outs = []
    for f in per:                                                                       # :197-198
        pt = learnable_pos_embd(f.shape[0], cfg.mm_time_interval, d, w, m + "mm_rand_pos_t.", f.dtype)
        f = f + mm_rms_norm(pt.reshape(-1, 1, 1, d))
        outs.append(f.flatten(0, 2))
    feats = torch.nn.utils.rnn.pad_sequence(outs, batch_first=True)                    # :199
    mask = torch.sum(torch.abs(feats), dim=-1) != 0                                     # :201
    sample_mask = torch.stack([torch.sum(torch.abs(x)) for x in images]) != 0           # :202
    mask = mask * sample_mask.unsqueeze(-1)                                             # :204
    feats = mm_RMSNorm(feats, w[m + "mm_rand_llm_norm.weight"])                         # :205
    feats = feats * mask.unsqueeze(-1)                                                  # :206
    return feats, mask


def audio_token_counts(audio_sizes: Sequence[int], cfg: OracleConfig) -> Tuple[np.ndarray, List[int]]:
    """The integer floors of multimodal.py:226-227, 234-235 (bit-exact index math)."""
    pool_ratio = cfg.aud_max_source_positions / cfg.aud_nb_max_frames
    s1 = np.floor(np.array(audio_sizes) * pool_ratio).astype(int)
    s2 = np.floor(s1 / cfg.mm_audio_pool_size).astype(int).tolist()
    return s1, s2


def encode_video_audios(audios: Sequence[Tensor], audio_sizes: Sequence[int], w: W, cfg: OracleConfig,
                        aud_features: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """audios: list (batch) of [C_i,n_mels,3000]; audio_sizes: mel frames per sample."""
    m = "model."
    concat = torch.cat(list(audios), dim=0)
    feats = whisper_encoder_forward(concat, w, cfg) if aud_features is None else aud_features   # :216-222
    split_sizes = [len(a) for a in audios]
    per = torch.split(feats, split_sizes, dim=0)                                        # :224-225
    s1, s2 = audio_token_counts(audio_sizes, cfg)                                       # :226-227
    per = [f.flatten(0, 1)[:s] for f, s in zip(per, s1)]                                # :228
    feats = torch.nn.utils.rnn.pad_sequence(per, batch_first=True)                     # :229
    feats = feats.permute(0, 2, 1)
    feats = F.conv1d(feats, w[m + "mm_rand_aud_pool.weight"], None,
                     stride=cfg.mm_audio_pool_size)                                     # :231-233
    feats = feats.permute(0, 2, 1)
    per = [f[:s] for f, s in zip(feats, s2)]                                            # :234-236
    feats = torch.cat(per, dim=0)                                                       # :238
    feats = projector_mlp(feats, w, m + "mm_rand_aud_projector.")                       # :239
    feats = mm_RMSNorm(feats, w[m + "mm_rand_aud_norm.weight"])                         # :240
    per = torch.split(feats, s2, dim=0)                                                 # :241
    d = cfg.hidden_size
    outs = []
    for f in per:                                                                       # :242
        pt = learnable_pos_embd(f.shape[0], cfg.mm_time_interval, d, w, m + "mm_rand_pos_t.", f.dtype)
        outs.append(f + mm_rms_norm(pt))
    feats = torch.nn.utils.rnn.pad_sequence(outs, batch_first=True)                    # :243
    mask = torch.sum(torch.abs(feats), dim=-1) != 0                                     # :245
    sample_mask = torch.stack([torch.sum(torch.abs(x)) for x in audios]) != 0           # :246
    mask = mask * sample_mask.unsqueeze(-1)                                             # :248
    feats = mm_RMSNorm(feats, w[m + "mm_rand_llm_norm.weight"])                         # :249
    feats = feats * mask.unsqueeze(-1)                                                  # :250
    return feats, mask


# --------------------------------------------------------------------------------------------
# text input preparation — multimodal.py:339-451 (integer/index path, bit-exact)
# --------------------------------------------------------------------------------------------
def strip_image_token(input_ids: Tensor, attention_mask: Optional[Tensor] = None,
                      padding_side: str = "right") -> Tuple[List[Tensor], Tensor, Tensor]:
    """Returns (per-sample id lists with -200 removed, attention_mask[B,Lmax] bool,
    position_ids[B,Lmax] long).  multimodal.py:352-430."""
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    else:
        attention_mask = attention_mask.bool()
    ids = [row[m] for row, m in zip(input_ids, attention_mask)]                         # :362
    out = []
    for cur in ids:
        n_img = int((cur == IMAGE_TOKEN_INDEX).sum())
        assert n_img <= 1, "only support at most one image for now."                   # :369
        out.append(cur[cur != IMAGE_TOKEN_INDEX])                                       # :377-397 (delete, not expand)
    max_len = max(x.shape[0] for x in out)
    B = len(out)
    am = torch.zeros((B, max_len), dtype=torch.bool)
    pos = torch.zeros((B, max_len), dtype=torch.long)
    for i, cur in enumerate(out):
        n = cur.shape[0]
        if n > 0:
            if padding_side == "left":
                am[i, -n:] = True
                pos[i, -n:] = torch.arange(n)
            else:
                am[i, :n] = True
                pos[i, :n] = torch.arange(n)
    return out, am, pos


def embed_text(ids: List[Tensor], am: Tensor, w: W, padding_side: str = "right") -> Tensor:
    """embed_tokens + zero padding — multimodal.py:385, 411-432."""
    E = w["model.embed_tokens.weight"]
    B, L = am.shape
    out = torch.zeros((B, L, E.shape[1]), dtype=E.dtype)
    for i, cur in enumerate(ids):
        n = cur.shape[0]
        if n == 0:
            continue
        e = F.embedding(cur, E)
        if padding_side == "left":
            out[i, -n:] = e
        else:
            out[i, :n] = e
    return out


# --------------------------------------------------------------------------------------------
# D-Attn decoder — lmm/dattn/gemma.py
# --------------------------------------------------------------------------------------------
@dataclass
class OracleCaches:
    """text: per-layer (k,v) [B,Hkv,L,D] post-RoPE; image/audio: per-layer (k,v) [B,N,Hkv*D]
    stored pre-repeat_kv as the reference's DynamicCache does (gemma.py:59-65)."""
    text: List[Tuple[Tensor, Tensor]] = field(default_factory=list)
    image: List[Tuple[Tensor, Tensor]] = field(default_factory=list)
    audio: List[Tuple[Tensor, Tensor]] = field(default_factory=list)


def _key_padding_add_mask(mask: Tensor, dtype: torch.dtype) -> Tensor:
    """bool [B,N] -> additive [B,1,1,N] (0 / -inf): what varlen unpadding achieves (xattn.py:36-138)."""
    add = torch.zeros(mask.shape, dtype=torch.float32)
    add[~mask] = float("-inf")
    return add[:, None, None, :]


def forward_xattn(h: Tensor, kv_in: Optional[Tensor], kv_mask: Tensor, w: W, lp: str, cfg: OracleConfig,
                  cache: List[Tuple[Tensor, Tensor]], layer_idx: int) -> Tuple[Tensor, Optional[Tensor]]:
    """DattnGemma2Attention.forward_xattn — gemma.py:50-96.
    Returns (o_proj(attn)[B,Lq,H], V_repeated[B,N,Hq,D] or None when served from the cache)."""
    nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    q = linear(h, w[lp + "self_attn.q_proj.weight"])                                    # :58 (no RoPE)
    fresh = len(cache) <= layer_idx
    if fresh:                                                                           # :60-63
        k = linear(kv_in, w[lp + "self_attn.k_proj.weight"])
        v = linear(kv_in, w[lp + "self_attn.v_proj.weight"])
        cache.append((k, v))
    else:
        k, v = cache[layer_idx]                                                         # :65
    B, Lq, _ = q.shape
    N = k.shape[1]
    qh = q.view(B, Lq, nq, hd).transpose(1, 2)
    kh = repeat_kv(k.view(B, N, nkv, hd).transpose(1, 2), nq // nkv)                    # :73-78
    vh = repeat_kv(v.view(B, N, nkv, hd).transpose(1, 2), nq // nkv)
    scale = cfg.query_pre_attn_scalar ** -0.5
    add = _key_padding_add_mask(kv_mask, torch.float32)
    a = sdpa_reference(qh, kh, vh, scale, cfg.attn_logit_softcapping, add)              # :81-91
    a = a.transpose(1, 2).reshape(B, Lq, nq * hd)
    out = linear(a, w[lp + "self_attn.o_proj.weight"])                                  # :94
    return out, (vh.transpose(1, 2) if fresh else None)                                 # :96


def text_self_attn(h: Tensor, cos: Tensor, sin: Tensor, text_mask: Tensor, w: W, lp: str, cfg: OracleConfig,
                   cache: List[Tuple[Tensor, Tensor]], layer_idx: int, past_len: int, sliding: bool) -> Tensor:
    """Gemma2Attention.forward under FA2 — gemma.py:165-175 -> TP/.../modeling_gemma2.py:248-288.
    Causal over text, RoPE, softcap, right-padding key mask, sliding window on even layers
    (FA2 window_size=(W,W): key j visible to query i iff i-W <= j <= i)."""
    nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    B, Lq, _ = h.shape
    q = linear(h, w[lp + "self_attn.q_proj.weight"]).view(B, Lq, nq, hd).transpose(1, 2)
    k = linear(h, w[lp + "self_attn.k_proj.weight"]).view(B, Lq, nkv, hd).transpose(1, 2)
    v = linear(h, w[lp + "self_attn.v_proj.weight"]).view(B, Lq, nkv, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if len(cache) <= layer_idx:
        cache.append((k, v))
    else:
        pk, pv = cache[layer_idx]
        k, v = torch.cat([pk, k], dim=2), torch.cat([pv, v], dim=2)
        cache[layer_idx] = (k, v)
    Lk = k.shape[2]
    kh, vh = repeat_kv(k, nq // nkv), repeat_kv(v, nq // nkv)
    qi = torch.arange(past_len, past_len + Lq)[:, None]
    kj = torch.arange(Lk)[None, :]
    allowed = kj <= qi
    if sliding:
        allowed = allowed & (kj >= qi - cfg.sliding_window)
    allowed = allowed[None, None] & text_mask[:, None, None, :Lk]
    add = torch.zeros(allowed.shape, dtype=torch.float32)
    add[~allowed] = float("-inf")
    a = sdpa_reference(q, kh, vh, cfg.query_pre_attn_scalar ** -0.5, cfg.attn_logit_softcapping, add)
    a = torch.nan_to_num(a, nan=0.0)          # fully padded query rows (FA2 never computes them)
    a = a.transpose(1, 2).reshape(B, Lq, nq * hd)
    return linear(a, w[lp + "self_attn.o_proj.weight"])


def gemma_mlp(x: Tensor, w: W, lp: str) -> Tensor:
    """Gemma2MLP — TP/.../modeling_gemma2.py:79-82: down(gelu_tanh(gate(x)) * up(x))."""
    g = gelu_tanh(linear(x, w[lp + "mlp.gate_proj.weight"]))
    u = linear(x, w[lp + "mlp.up_proj.weight"])
    return linear(g * u, w[lp + "mlp.down_proj.weight"])


def mistral_mlp(x: Tensor, w: W, lp: str) -> Tensor:
    """MistralMLP: down(silu(gate(x)) * up(x))."""
    g = F.silu(linear(x, w[lp + "mlp.gate_proj.weight"]))
    u = linear(x, w[lp + "mlp.up_proj.weight"])
    return linear(g * u, w[lp + "mlp.down_proj.weight"])


def feed_forward(x: Tensor, w: W, lp: str, cfg: OracleConfig) -> Tensor:
    """DattnGemma2DecoderLayer.feed_foward — gemma.py:116-123; Vidi-7B: mistral.py:131-137."""
    if cfg.arch == "mistral":
        h = mistral_rmsnorm(x, w[lp + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        return x + mistral_mlp(h, w, lp)
    h = gemma_rmsnorm(x, w[lp + "pre_feedforward_layernorm.weight"], cfg.rms_norm_eps)
    h = gemma_mlp(h, w, lp)
    h = gemma_rmsnorm(h, w[lp + "post_feedforward_layernorm.weight"], cfg.rms_norm_eps)
    return x + h


def mm_stream_layer(x: Tensor, w: W, lp: str, cfg: OracleConfig) -> Tuple[Tensor, Tensor, Tensor]:
    """The per-token 'diagonal' stream of one layer on mm tokens x:[B,N,H] — gemma.py:183-184,
    61-62, 196-202.  Returns (x_next, k[B,N,Hkv*D], v[B,N,Hkv*D])."""
    nq, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    h = llm_rmsnorm(x, w[lp + "input_layernorm.weight"], cfg)
    k = linear(h, w[lp + "self_attn.k_proj.weight"])
    v = linear(h, w[lp + "self_attn.v_proj.weight"])
    B, N, _ = v.shape
    vrep = repeat_kv(v.view(B, N, nkv, hd).transpose(1, 2), nq // nkv).transpose(1, 2).flatten(2, 3)
    o = linear(vrep, w[lp + "self_attn.o_proj.weight"])
    if cfg.arch != "mistral":                                                           # 7B: mistral.py:224-227 (no post norm)
        o = gemma_rmsnorm(o, w[lp + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
    x = x + o
    return feed_forward(x, w, lp, cfg), k, v


def decoder_layer(hidden: Tensor, cos: Tensor, sin: Tensor, text_mask: Tensor,
                  image_embeds: Optional[Tensor], image_mask: Optional[Tensor],
                  audio_embeds: Optional[Tensor], audio_mask: Optional[Tensor],
                  w: W, cfg: OracleConfig, caches: OracleCaches, layer_idx: int, past_len: int
                  ) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    """DattnGemma2DecoderLayer.forward, multimodal branch — gemma.py:153-244."""
    lp = f"model.layers.{layer_idx}."
    mistral = cfg.arch == "mistral"
    sliding = True if mistral else (not bool(layer_idx % 2))                            # :104; Mistral: every layer
    residual = hidden
    h = llm_rmsnorm(hidden, w[lp + "input_layernorm.weight"], cfg)                      # :162 / mistral.py:187
    t = text_self_attn(h, cos, sin, text_mask, w, lp, cfg, caches.text, layer_idx, past_len, sliding)   # :165-175

    def branch(embeds, mask, cache):
        use_cache = len(cache) > layer_idx                                              # :179
        n_valid = torch.sum(mask, dim=-1)                                               # :180
        m2 = mask.clone()
        m2[n_valid == 0] = True                                                         # :181-182
        kv_in = embeds if use_cache else llm_rmsnorm(embeds, w[lp + "input_layernorm.weight"], cfg)  # :183-184
        o, vrep = forward_xattn(h, kv_in, m2, w, lp, cfg, cache, layer_idx)             # :185-191
        o = o * (n_valid != 0)[:, None, None]                                           # :192
        if not use_cache:                                                               # :195-202
            vflat = vrep.flatten(2, 3)
            u = linear(vflat, w[lp + "self_attn.o_proj.weight"])
            if not mistral:
                u = gemma_rmsnorm(u, w[lp + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
            embeds = embeds + u
            embeds = feed_forward(embeds, w, lp, cfg)
        return o, embeds

    if image_embeds is not None:
        i_out, image_embeds = branch(image_embeds, image_mask, caches.image)
    else:
        i_out = 0.0                                                                     # :204
    if audio_embeds is not None:
        a_out, audio_embeds = branch(audio_embeds, audio_mask, caches.audio)
    else:
        a_out = 0.0                                                                     # :233
    if mistral:
        hidden = residual + t + i_out + a_out                                           # mistral.py:261 (left to right)
    else:
        s = t + i_out + a_out                                                           # :236
        hidden = residual + gemma_rmsnorm(s, w[lp + "post_attention_layernorm.weight"], cfg.rms_norm_eps)  # :237
    hidden = feed_forward(hidden, w, lp, cfg)                                           # :238
    return hidden, image_embeds, audio_embeds


def model_forward(inputs_embeds: Tensor, position_ids: Tensor, text_mask: Tensor,
                  image_embeds: Optional[Tensor], image_mask: Optional[Tensor],
                  audio_embeds: Optional[Tensor], audio_mask: Optional[Tensor],
                  w: W, cfg: OracleConfig, caches: OracleCaches, past_len: int) -> Tensor:
    """DattnGemma2Model.forward — gemma.py:267-424.  Returns last_hidden_state [B,Lq,H]."""
    dt = inputs_embeds.dtype
    cos, sin = rope_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta, dt)             # :348
    hidden = inputs_embeds
    if cfg.arch != "mistral":                                                           # Mistral has no embedding normalizer (mistral.py:369)
        normalizer = torch.tensor(cfg.hidden_size ** 0.5, dtype=dt)                      # :353 (rounds in fp16/bf16)
        hidden = inputs_embeds * normalizer                                             # :354
        if image_embeds is not None:
            image_embeds = image_embeds * normalizer                                    # :355
        if audio_embeds is not None:
            audio_embeds = audio_embeds * normalizer                                    # :356
    for li in range(cfg.num_hidden_layers):                                             # :362-406
        hidden, image_embeds, audio_embeds = decoder_layer(
            hidden, cos, sin, text_mask, image_embeds, image_mask, audio_embeds, audio_mask,
            w, cfg, caches, li, past_len)
    return llm_rmsnorm(hidden, w["model.norm.weight"], cfg)                             # :411 / mistral.py:423


def lm_logits(hidden: Tensor, w: W, cfg: OracleConfig) -> Tensor:
    """lm_head + final-logit softcap — gemma.py:565-569."""
    # gemma-2 ties lm_head to embed_tokens (tie_word_embeddings=True): checkpoints carry no lm_head.weight
    logits = linear(hidden, w.get("lm_head.weight", w["model.embed_tokens.weight"]))
    if cfg.final_logit_softcapping is not None:
        logits = torch.tanh(logits / cfg.final_logit_softcapping) * cfg.final_logit_softcapping
    return logits


def generate_greedy(input_ids: Tensor, images: Optional[Sequence[Tensor]], audios: Optional[Sequence[Tensor]],
                    audio_sizes: Optional[Sequence[int]], w: W, cfg: OracleConfig, max_new_tokens: int,
                    attention_mask: Optional[Tensor] = None, return_debug: bool = False):
    """DattnGemma2ForCausalLM.generate(do_sample=False) — gemma.py:603-655 + HF greedy loop.
    Returns new token ids [B,n_new] (HF semantics when driven by inputs_embeds); finished rows are
    padded with eos... (HF pads with pad_token_id; callers compare up to the first eos)."""
    ids, am, pos = strip_image_token(input_ids, attention_mask)
    emb = embed_text(ids, am, w)
    img = imask = aud = amask = None
    if images is not None:
        img, imask = encode_video_images(images, w, cfg)
    if audios is not None:
        aud, amask = encode_video_audios(audios, audio_sizes, w, cfg)
    caches = OracleCaches()
    B, L = am.shape
    lens = am.sum(-1)
    hidden = model_forward(emb, pos, am, img, imask, aud, amask, w, cfg, caches, 0)
    last = hidden[torch.arange(B), lens - 1]                                            # last valid token per row
    logits = lm_logits(last[:, None, :], w, cfg)[:, 0]
    debug = {"prefill_logits": logits.clone(), "image_embeds": img, "image_mask": imask,
             "audio_embeds": aud, "audio_mask": amask, "prefill_hidden": hidden, "text_mask": am, "position_ids": pos,
             "caches": caches, "step_logits": []}
    out = []
    finished = torch.zeros(B, dtype=torch.bool)
    text_mask = am.clone()
    cur_len = L
    for step in range(max_new_tokens):
        nxt = torch.argmax(logits.float(), dim=-1)
        nxt = torch.where(finished, torch.full_like(nxt, cfg.eos_token_id), nxt)
        out.append(nxt)
        finished = finished | (nxt == cfg.eos_token_id)
        if bool(finished.all()) or step == max_new_tokens - 1:
            break
        e = F.embedding(nxt[:, None], w["model.embed_tokens.weight"])
        text_mask = torch.cat([text_mask, torch.ones(B, 1, dtype=torch.bool)], dim=1)
        # HF: position of the new token = number of valid tokens so far (cumsum of the mask - 1)
        p = (text_mask.long().sum(-1) - 1)[:, None]
        hidden = model_forward(e, p, text_mask, img, imask, aud, amask, w, cfg, caches, cur_len)
        cur_len += 1
        logits = lm_logits(hidden, w, cfg)[:, 0]
        debug["step_logits"].append(logits.clone())
    ids_out = torch.stack(out, dim=1)
    return (ids_out, debug) if return_debug else ids_out


# --------------------------------------------------------------------------------------------
# ask() post-processing — eval/inference.py:52-66 (integer/string path, bit-exact)
# --------------------------------------------------------------------------------------------
def format_time_ranges(text: str, length: float) -> str:
    import re
    pattern = re.compile(r"(\d\.\d+)-(\d\.\d+)")
    outs = []
    for a, b in pattern.findall(text.strip()):
        t0, t1 = float(a) * length, float(b) * length
        outs.append("{:02d}:{:02d}:{:02d}-{:02d}:{:02d}:{:02d}".format(
            int(t0 / 3600), (int(t0) % 3600) // 60, int(t0) % 60,
            int(t1 / 3600), (int(t1) % 3600) // 60, int(t1) % 60))
    return ", ".join(outs)


def tensor_split_bounds(n: int, parts: int) -> List[Tuple[int, int]]:
    """torch.tensor_split chunk bounds used by splitted_call (split.py:12-22)."""
    base, extra = divmod(n, parts)
    out, s = [], 0
    for i in range(parts):
        e = s + base + (1 if i < extra else 0)
        out.append((s, e))
        s = e
    return out


# --------------------------------------------------------------------------------------------
# partial-softmax merge identity used by the split-KV / multi-GPU cross-attention (ours, not the
# reference's): exact because the tanh softcap is applied per logit before the softmax.
# --------------------------------------------------------------------------------------------
def merge_partials(o: Tensor, m: Tensor, l: Tensor) -> Tensor:
    """o:[S,...,D] un-normalised partial outputs (sum p*v with p=exp(s-m_s)), m,l:[S,...]."""
    mg = m.max(dim=0).values
    sc = torch.exp(m - mg[None])
    sc = torch.where(torch.isinf(m) & (m < 0), torch.zeros_like(sc), sc)
    num = (o * sc[..., None]).sum(0)
    den = (l * sc).sum(0)
    return num / den[..., None]
