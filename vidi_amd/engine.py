"""VidiEngine — host orchestration of the HIP kernels for the Vidi1.5 (Gemma2 D-Attn) inference path.

Everything numeric goes through `vidi_amd.hip` (the C ABI); torch is used for device buffers,
views, copies and index tensors.  The structure follows the reference call stack:

  encode_video_images / encode_video_audios   <- lmm/dattn/multimodal.py:156-252
  mm_stream_prefill (diagonal V2V/A2A stream)  <- lmm/dattn/gemma.py:183-202 (x42), 59-65 (caches)
  text_forward (T2T + T2V + T2A)               <- lmm/dattn/gemma.py:160-238, 267-424, 562-569

Design choices that differ from the reference on purpose (the first three give identical results; the weight folds re-round
weights once at load time and are tolerance-level deviations — each has a switch whose off arm keeps the reference's rounding
points, and both arms are held to the goldens):
  * the multimodal stream is query-independent, so it is run ONCE per video for all layers
    (`MMState`) and shared by every query/decoding step; the reference interleaves it with the
    text prefill layer by layer (gemma.py:362-406) and re-multiplies the embeds every step.
  * K/V of the multimodal tokens are written by the projection GEMM's epilogue directly in the
    cross-attention kernel's tile layout; `repeat_kv` is never materialised.
  * layer L-1's stream update (gemma.py:196-202 on the last layer) is dead in the reference and
    skipped here.
  * VIDI_LN_FOLD (default on): the towers' LayerNorm weights are multiplied into the consuming projection, Wf = T(W * gamma) —
    one extra rounding per weight; the reference multiplies the T-rounded LayerNorm output by the unfused weights.
  * VIDI_ATTN_PRESCALE (default on, SigLIP's d = 72 heads): the softmax scale (and log2 e) is multiplied into the q projection before that
    fold's single rounding, so q is rounded at a different value than the reference's (same relative precision); the attention kernel then
    keeps its running maximum rounded to T.  The reference scales the T-rounded scores.
  * VIDI_FOLD_REPKV (default on): the multimodal stream's o_proj(repeat_kv(V)) uses wo_kv = T(sum_g Wo block) — the G column blocks
    summed in fp32 and rounded once; the error (about one ulp per weight) feeds every later layer's K/V cache.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import hip
from .config import VidiConfig


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


from .shard import audio_token_counts, token_budget_hw  # noqa: E402,F401  (host integer rules; re-exported: bench.py and the tests import them from here)


@dataclass
class MMState:
    """Per-video, query-independent state: cross-attention caches for all layers."""
    n_img: int = 0
    n_aud: int = 0
    img_start: int = 0
    aud_start: int = 0
    ntile64: int = 0
    kc: Optional[torch.Tensor] = None        # [L, nkv, ntile64, 64, hd]
    vtc: Optional[torch.Tensor] = None       # [L, nkv, 2*ntile64, hd, 32]  (32-key sub-tiles of 64-byte rows)
    img_mask: Optional[torch.Tensor] = None  # uint8 [>= n_img] or None when every key is valid
    aud_mask: Optional[torch.Tensor] = None
    img_any_valid: bool = True
    aud_any_valid: bool = True
    g_img: int = 0                           # global (all ranks) key counts; == n_img/n_aud on one GPU
    g_aud: int = 0
    # reference-facing views (encode_videos outputs)
    image_features: Optional[torch.Tensor] = None
    image_attention_mask: Optional[torch.Tensor] = None
    audio_features: Optional[torch.Tensor] = None
    audio_attention_mask: Optional[torch.Tensor] = None


@dataclass
class TextState:
    """Text KV cache (the reference's HybridCache, gemma.py:308-315) for B rows."""
    B: int
    Lmax: int
    kc: torch.Tensor                         # [L, B, Lmax, nkv*hd]
    vc: torch.Tensor
    kmask: torch.Tensor                      # uint8 [B, Lmax]
    past_len: int = 0
    n_valid: Optional[torch.Tensor] = None   # int64 [B] number of valid tokens per row
    pos_dev: Optional[torch.Tensor] = None   # int32 [1] device mirror of past_len (graph-captured decode)
    pos_idx: Optional[torch.Tensor] = None   # int64 [1] same value, as an index tensor


class VidiEngine:
    def __init__(self, cfg: VidiConfig, weights: Dict[str, torch.Tensor], dtype: torch.dtype = torch.bfloat16,
                 device: str = "cuda", free_source: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("VidiEngine needs a HIP device: the kernels have no CPU path")
        hip.load_library()
        self.cfg = cfg
        self.dtype = dtype
        self.dev = torch.device(device)
        self.mistral = cfg.arch == "mistral"                                                  # Vidi-7B wiring (mistral.py)
        # gemma.py:353 multiplies every embedding stream by sqrt(H) rounded to the model dtype; Mistral has no normalizer
        self.normalizer = 1.0 if self.mistral else float(torch.tensor(cfg.hidden_size ** 0.5, dtype=dtype).float())
        self.glu_act = hip.ACT_SILU if self.mistral else hip.ACT_GELU_TANH
        # towers: LayerNorm folded into the q/k/v and fc1 projections (default) or run as its own row pass (VIDI_LN_FOLD=0: the A/B arm)
        # Diagnostic probe (None in production: one attribute test per layer).  A dict asks the layer loops to keep copies of their
        # per-layer INPUT rows, so a checker can evaluate every layer on exactly the input the kernels saw ("teacher-forced" parity:
        # one layer's rounding per comparison instead of the drift of all the layers before it):
        #   {"vis_frames": [frame idx], "vis_x": []}   siglip_forward: rows of those frames before each layer + after the last
        #   {"aud_windows": [window idx], "aud_x": []} whisper_forward: likewise (before the final LayerNorm)
        #   {"stream_rows": LongTensor, "stream_x": []}  mm_stream_prefill: residual-stream rows at every layer's input
        #   {"text_h": []}                              text_forward: the text residual stream at every layer's input + after the last
        self.probe = None
        self.ln_fold = os.environ.get("VIDI_LN_FOLD", "1") != "0"
        # towers: q | k | v as one row-major buffer + the transpose-read attention kernel (VIDI_ATTN_RM=1) or Q|K row-major + V transposed by
        # the projection's epilogue + the Vt attention kernel (0)
        self.attn_rm = os.environ.get("VIDI_ATTN_RM", "1") != "0"
        # SigLIP tower (d = 72): softmax scale * log2(e) folded into the q projection (weights and bias, before the LayerNorm fold: still ONE
        # rounding per weight) and vidi_attn_self_rm called with scale = 0 — its d = 72 body then carries the running maximum inside the
        # QK^T contraction and the exponent needs no FMA (VIDI_ATTN_PRESCALE=0: unscaled q, scale applied to the fp32 scores)
        self.attn_prescale = os.environ.get("VIDI_ATTN_PRESCALE", "1") != "0" and self.ln_fold and self.attn_rm
        # multimodal stream: each (post-norm + residual, next pre-norm) pair as one launch (VIDI_STREAM_NORM2=0: two launches)
        self.stream_norm2 = os.environ.get("VIDI_STREAM_NORM2", "1") != "0"
        # decode step: rope + cache append + T2T as one launch (VIDI_DECODE_ATTN=0: rope_cache + attn_text), T2V + T2A partial passes as
        # one launch (VIDI_CROSS_DUAL=0: one launch per modality) — the A/B arms of tools/ab_decode.py
        # ragged batch of prompts (right- / left-padded + attention mask): the text stream runs on the VALID positions only — a row map as the
        # reference's `_unpad_xattn_input` builds (xattn.py:36-103) carried through every projection, norm and the cross-attention; only the
        # T2T attention sees the padded [B, Lq] frame (VIDI_TEXT_VARLEN=0: every kernel computes the pad positions too, the A/B arm)
        self.text_varlen = os.environ.get("VIDI_TEXT_VARLEN", "1") != "0"
        self.skinny_gemm = os.environ.get("VIDI_SKINNY_GEMM", "1") != "0"      # prompts of 9..128 rows: csrc/gemm_skinny.h instead of the tile GEMM
        self.decode_attn = os.environ.get("VIDI_DECODE_ATTN", "1") != "0"
        self.cross_dual = os.environ.get("VIDI_CROSS_DUAL", "1") != "0"
        # decode step: the Gemma2 norm pairs folded into the gate/up and the next layer's q/k/v projections (VIDI_DECODE_NORM_GEMV=0:
        # vidi_resid_norm2 + vidi_gemv[_glu] as separate launches)
        self.decode_norm_gemv = os.environ.get("VIDI_DECODE_NORM_GEMV", "1") != "0"
        # a BATCH of decode rows (several queries on one video): from this many rows on, the weight-streaming projections run on the matrix
        # pipe (csrc/gemv_mfma.hip) instead of the VALU GEMV, whose FMAs saturate near 8 rows; 0 = never (the A/B arm)
        self.gemv_mfma_rows = int(os.environ.get("VIDI_GEMV_MFMA_MIN_ROWS", "5"))
        # decode step: the T2T launch and the merge of the cross-attention partials as one launch (VIDI_DECODE_TAIL=0: two launches)
        self.decode_tail = os.environ.get("VIDI_DECODE_TAIL", "1") != "0"
        # multimodal stream: o_proj over repeat_kv(V) as one GEMM over V with the G column blocks of o_proj summed at load time
        # (VIDI_FOLD_REPKV=0: the repeat done by the GEMM's operand read, K twice as long)
        self.fold_repkv = os.environ.get("VIDI_FOLD_REPKV", "1") != "0"
        # Vidi-7B's learned Conv2DPool: window gather in the GEMM loader (VIDI_POOL_LOADER=0: im2col + GEMM)
        self.pool_loader = os.environ.get("VIDI_POOL_LOADER", "1") != "0"
        self.norm_mode = hip.NORM_MM if self.mistral else hip.NORM_GEMMA            # MistralRMSNorm == w * T(x_hat)
        self._pack(weights, free_source)
        self._rope_cache: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._ws: Dict[str, torch.Tensor] = {}
        self.pg, self.world, self.rank = None, 1, 0
        self.sharded = False                            # set_dist(): the key-sharded cross-attention path (world > 1, or forced for a one-rank RCCL test)
        self.shard_encode = False                       # set_dist(): every rank encodes its own frame / window range (both dist modes)
        self.dist_mode = "sharded_stream"
        # sharded stream: the per-layer all-gather is issued asynchronously (it runs on the backend's stream) BEFORE the T2T launch and
        # waited for after it, so the text self-attention hides it (VIDI_DIST_OVERLAP=0: T2T, then the exchange, strictly serial)
        self.dist_overlap = os.environ.get("VIDI_DIST_OVERLAP", "1") != "0"
        self.n_collectives = 0                          # data-path all-gathers issued (one per decoder layer per forward when sharded)

    # -----------------------------------------------------------------------------------------
    # weight packing (one-time repack into kernel-preferred layouts; owned by this module)
    # -----------------------------------------------------------------------------------------
    def _pack(self, w: Dict[str, torch.Tensor], free_source: bool):
        cfg, dt, dev = self.cfg, self.dtype, self.dev

        def g(name, fp32=False):
            t = w[name]
            t = t.to(device=dev, dtype=torch.float32 if fp32 else dt)
            return t.contiguous()

        def pop(name):
            if free_source:
                w.pop(name, None)

        def fold_ln(W, b, gamma, beta):
            """LayerNorm folded into the projection that consumes it (include/vidi_hip.h: vidi_gemm_ln): returns
            (Wf = T(W * gamma), colsum[n] = sum_k Wf[n, k] of the ROUNDED Wf (so the mean term cancels exactly against the MFMA
            products), shift[n] = sum_k W[n, k] * beta[k] + b[n]) — fp32 vectors, computed once at load time."""
            Wf = (W.float() * gamma.float()[None, :]).to(dt).contiguous()
            colsum = Wf.float().sum(dim=1).contiguous()
            shift = (W.float() @ beta.float() + (b.float() if b is not None else 0.0)).contiguous()
            return Wf, colsum, shift

        H, I = cfg.hidden_size, cfg.intermediate_size
        self.embed = g("model.embed_tokens.weight")
        self.lm_head = self.embed if cfg.tie_word_embeddings or "lm_head.weight" not in w else g("lm_head.weight")
        self.final_norm = g("model.norm.weight")
        self.layers: List[Dict[str, torch.Tensor]] = []
        for i in range(cfg.num_hidden_layers):
            p = f"model.layers.{i}."
            L: Dict[str, torch.Tensor] = {}
            L["wqkv"] = torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                                   g(p + "self_attn.v_proj.weight")], dim=0).contiguous()
            nqd = cfg.num_attention_heads * cfg.head_dim
            L["wkv"] = L["wqkv"][nqd:]                                        # [Wk; Wv] view
            L["wo"] = g(p + "self_attn.o_proj.weight")
            if self.fold_repkv and i + 1 < cfg.num_hidden_layers:
                # multimodal stream: o_proj(repeat_kv(V)) (gemma.py:196-197) == V (sum_g Wo[:, head kvh*G+g])^T — the G column blocks of
                # o_proj that repeat_kv feeds with the same values, summed in fp32 and rounded once: half the K of that GEMM
                Gq = cfg.num_attention_heads // cfg.num_key_value_heads
                L["wo_kv"] = L["wo"].view(H, cfg.num_key_value_heads, Gq, cfg.head_dim).float().sum(2).reshape(H, -1).to(dt).contiguous()
            gate, up = g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")
            # 32-row interleave [gate 0..31 | up 0..31 | gate 32..63 | ...] for the fused GeGLU epilogue
            L["wgu"] = torch.stack([gate.view(I // 32, 32, H), up.view(I // 32, 32, H)], dim=1).reshape(2 * I, H).contiguous()
            del gate, up
            L["wdown"] = g(p + "mlp.down_proj.weight")
            norm_names = (("input_layernorm", "ln_in"), ("post_attention_layernorm", "ln_post_attn")) if cfg.arch == "mistral" else \
                (("input_layernorm", "ln_in"), ("post_attention_layernorm", "ln_post_attn"),
                 ("pre_feedforward_layernorm", "ln_pre_ffn"), ("post_feedforward_layernorm", "ln_post_ffn"))
            for n, k in norm_names:
                L[k] = g(p + n + ".weight")
            for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj",
                      "mlp.up_proj", "mlp.down_proj"):
                pop(p + n + ".weight")
            self.layers.append(L)

        # ---- SigLIP ----
        v = "model.mm_vis.vision_model."
        Hv, Iv, P = cfg.vis_hidden_size, cfg.vis_intermediate_size, cfg.vis_patch_size
        Ivp = _round_up(Iv, 64)
        kp = _round_up(3 * P * P, 64)
        conv_w = g(v + "embeddings.patch_embedding.weight")
        # patch embedding straight from the NCHW pixels (vidi_patch_embed: the persistent GEMM's loader gathers the patches; default) or
        # through an im2col buffer + the generic GEMM (VIDI_PATCH_LOADER=0, the A/B arm; also patches wider than the loader's 16-pixel run)
        self.patch_loader = os.environ.get("VIDI_PATCH_LOADER", "1") != "0" and P <= 16 and 3 * P * 16 >= 192 and (P | cfg.vis_image_size) % 2 == 0
        if self.patch_loader:
            pw = hip.patch_embed_weight(conv_w, P)
        else:
            pw = torch.zeros((Hv, kp), dtype=dt, device=dev)
            pw[:, : 3 * P * P] = conv_w.reshape(Hv, -1)
        del conv_w
        self.vis = {"patch_w": pw, "patch_b": g(v + "embeddings.patch_embedding.bias"),
                    "pos": g(v + "embeddings.position_embedding.weight"), "kpad": kp, "ipad": Ivp, "layers": []}
        for i in range(cfg.vis_select_layers):
            p = f"{v}encoder.layers.{i}."
            L = {}
            L["wqkv"] = torch.cat([g(p + f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")], dim=0).contiguous()
            L["bqkv"] = torch.cat([g(p + f"self_attn.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")], dim=0).contiguous()
            L["wo"], L["bo"] = g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias")
            fc1 = torch.zeros((Ivp, Hv), dtype=dt, device=dev); fc1[:Iv] = g(p + "mlp.fc1.weight")
            b1 = torch.zeros((Ivp,), dtype=dt, device=dev); b1[:Iv] = g(p + "mlp.fc1.bias")
            fc2 = torch.zeros((Hv, Ivp), dtype=dt, device=dev); fc2[:, :Iv] = g(p + "mlp.fc2.weight")
            L["fc1"], L["b1"], L["fc2"], L["b2"] = fc1, b1, fc2, g(p + "mlp.fc2.bias")
            for n, k in (("layer_norm1", "ln1"), ("layer_norm2", "ln2")):
                L[k + "w"], L[k + "b"] = g(p + n + ".weight"), g(p + n + ".bias")
            if self.ln_fold:    # layer_norm1 -> q/k/v_proj and layer_norm2 -> fc1 with the LayerNorm folded in (the plain weights are dropped)
                Wqkv, bqkv = L["wqkv"], L.pop("bqkv")
                self.vis_prescaled = self.attn_prescale and cfg.vis_hidden_size // cfg.vis_num_heads == 72
                if self.vis_prescaled:
                    qs = torch.ones((3 * Hv, 1), dtype=torch.float32, device=dev)
                    qs[:Hv] = (cfg.vis_hidden_size // cfg.vis_num_heads) ** -0.5 * math.log2(math.e)
                    Wqkv, bqkv = Wqkv.float() * qs, bqkv.float() * qs[:, 0]             # fp32: fold_ln rounds the product once
                L["wqkv"], L["sqkv"], L["cqkv"] = fold_ln(Wqkv, bqkv, L["ln1w"], L["ln1b"])
                L["fc1"], L["s1"], L["c1"] = fold_ln(L["fc1"], L.pop("b1"), L["ln2w"], L["ln2b"])
            self.vis["layers"].append(L)

        # ---- Whisper encoder ----
        a = "model.mm_aud.encoder."
        Da, nm = cfg.aud_d_model, cfg.aud_num_mel_bins
        k1 = _round_up(3 * nm, 64)
        c1 = torch.zeros((Da, k1), dtype=dt, device=dev)
        c1[:, : 3 * nm] = g(a + "conv1.weight").permute(0, 2, 1).reshape(Da, 3 * nm)      # k = tap*nmel + c
        c2 = g(a + "conv2.weight").permute(0, 2, 1).reshape(Da, 3 * Da).contiguous()       # k = tap*Da + c
        self.aud = {"conv1_w": c1, "conv1_b": g(a + "conv1.bias"), "conv2_w": c2, "conv2_b": g(a + "conv2.bias"),
                    "pos": g(a + "embed_positions.weight"), "lnw": g(a + "layer_norm.weight"), "lnb": g(a + "layer_norm.bias"),
                    "k1": k1, "layers": []}
        for i in range(cfg.aud_num_layers):
            p = f"{a}layers.{i}."
            L = {}
            L["wqkv"] = torch.cat([g(p + f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")], dim=0).contiguous()
            L["bqkv"] = torch.cat([g(p + "self_attn.q_proj.bias"), torch.zeros((Da,), dtype=dt, device=dev),
                                   g(p + "self_attn.v_proj.bias")], dim=0).contiguous()
            L["wo"], L["bo"] = g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias")
            L["fc1"], L["b1"], L["fc2"], L["b2"] = g(p + "fc1.weight"), g(p + "fc1.bias"), g(p + "fc2.weight"), g(p + "fc2.bias")
            for n, k in (("self_attn_layer_norm", "ln1"), ("final_layer_norm", "ln2")):
                L[k + "w"], L[k + "b"] = g(p + n + ".weight"), g(p + n + ".bias")
            if self.ln_fold:
                L["wqkv"], L["sqkv"], L["cqkv"] = fold_ln(L["wqkv"], L.pop("bqkv"), L["ln1w"], L["ln1b"])
                L["fc1"], L["s1"], L["c1"] = fold_ln(L["fc1"], L.pop("b1"), L["ln2w"], L["ln2b"])
            self.aud["layers"].append(L)

        # ---- multimodal glue ----
        m = "model."
        self.mm = {
            "img_w0": g(m + "mm_rand_img_projector.model.0.weight"), "img_b0": g(m + "mm_rand_img_projector.model.0.bias"),
            "img_w2": g(m + "mm_rand_img_projector.model.2.weight"), "img_b2": g(m + "mm_rand_img_projector.model.2.bias"),
            # Conv1d weight [d_out, Da, k] -> GEMM weight [d_out, tap*Da + c]; d_out = H (Vidi1.5) or Da (Vidi-7B)
            "aud_pool": (lambda t: t.permute(0, 2, 1).reshape(t.shape[0], -1).contiguous())(g(m + "mm_rand_aud_pool.weight")),
            "aud_w0": g(m + "mm_rand_aud_projector.model.0.weight"), "aud_b0": g(m + "mm_rand_aud_projector.model.0.bias"),
            "aud_w2": g(m + "mm_rand_aud_projector.model.2.weight"), "aud_b2": g(m + "mm_rand_aud_projector.model.2.bias"),
            "img_norm": g(m + "mm_rand_img_norm.weight"), "aud_norm": g(m + "mm_rand_aud_norm.weight"),
            "llm_norm": g(m + "mm_rand_llm_norm.weight"),
        }
        if cfg.arch == "mistral":
            # learned Conv2DPool kernel [d_out, d_in, k, k] -> GEMM weight [d_out, (dy*k+dx)*d_in + c] (im2col_nhwc order)
            cw = g(m + "mm_rand_img_pool.conv.weight")
            self.mm["img_pool_w"] = cw.permute(0, 2, 3, 1).reshape(cw.shape[0], -1).contiguous()
            pop(m + "mm_rand_img_pool.conv.weight")
        for n in ("h", "w", "t"):
            for k in ("mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias"):
                self.mm[f"pos_{n}.{k}"] = g(f"{m}mm_rand_pos_{n}.{k}", fp32=True)
        d = H
        self.pos_div = torch.exp(torch.arange(0, d, 2, dtype=torch.float) * -(math.log(10000.0) / d)).to(dev)   # pos.py:15

    # -----------------------------------------------------------------------------------------
    # small helpers
    # -----------------------------------------------------------------------------------------
    def _buf(self, name: str, shape, dtype=None, zero: bool = False) -> torch.Tensor:
        """Workspace cache keyed by name; reallocated when the shape grows/changes."""
        dtype = dtype or self.dtype
        t = self._ws.get(name)
        n = 1
        for s in shape:
            n *= s
        if t is None or t.numel() < n or t.dtype != dtype:
            # workspaces outlive the call: allocate them as normal tensors even when the caller runs under torch.inference_mode()
            # (the reference CLI does, inference.py:40) — an inference tensor cannot be updated in place by a later call outside it
            with torch.inference_mode(False):
                t = (torch.zeros if zero else torch.empty)(n, dtype=dtype, device=self.dev)
            self._ws[name] = t
        return t[:n].view(*shape)

    def proj(self, x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """bias-free projection: weight-streaming GEMV for M <= 8 (decode), the split-K weight-streaming MFMA kernel for a prompt's
        9..128 rows (VIDI_SKINNY_GEMM=0: the tile GEMM, the A/B arm), the tile / persistent MFMA GEMM otherwise."""
        M = x.shape[0]
        if self.gemv_mfma_rows and self.gemv_mfma_rows <= M <= 32 and hip.gemv_mfma_fits(M, w.shape[0], x.shape[1], False, x, w):
            return hip.gemv_mfma(x, w, out)
        if M <= 8:
            return hip.gemv(x, w, out)
        if M <= 128 and self.skinny_gemm and w.shape[0] <= 65536:         # (lm_head's 256 000 rows run unsplit: an fp32 round trip of M x N for nothing)
            need = hip.gemm_skinny_workspace_bytes(M, w.shape[0], x.shape[1])
            have = self._ws.get("skinny_ws")
            # (a workspace that would have to be allocated in the middle of a graph capture — a decode step of 9+ rows whose prefill ran on
            # the tile kernel: stay on the tile kernel for that step)
            if need and ((have is not None and have.numel() * 4 >= need) or not torch.cuda.is_current_stream_capturing()):
                return hip.gemm_skinny(x, w, self._buf("skinny_ws", (need // 4,), torch.float32), out)
        return hip.gemm(x, w, None, out)

    def proj_glu(self, x: torch.Tensor, wgu: torch.Tensor, out: torch.Tensor, act: int) -> torch.Tensor:
        """gated-MLP front half of a few decode rows (M <= 8): the matrix-pipe kernel from `gemv_mfma_rows` rows on, else the VALU GEMV"""
        M = x.shape[0]
        if self.gemv_mfma_rows and M >= self.gemv_mfma_rows and hip.gemv_mfma_fits(M, wgu.shape[0] // 2, x.shape[1], True, x, wgu):
            return hip.gemv_mfma(x, wgu, out, glu_act=act)
        return hip.gemv_glu(x, wgu, out, act)

    def sample_flag(self, x: torch.Tensor) -> torch.Tensor:
        """int32[1] device flag "the sample holds any non-zero value" (`torch.sum(torch.abs(x)) != 0`, multimodal.py:202, 246) of a
        WHOLE sample; the sharded encode passes it in because a rank only sees its own frames / windows."""
        flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        if x.numel() == 0:                          # an empty shard (more ranks than frames / windows) holds no non-zero value
            return flag
        if x.is_cuda:
            hip.any_nonzero(x.to(self.dtype).contiguous().view(-1), flag)
        elif bool((x != 0).any()):                  # host tensor (the CLI hands over CPU frames): a host reduction, no full-video upload
            flag.fill_(1)
        return flag

    # ---- encoder-layer pieces with the LayerNorms folded away (SigLIP and Whisper share the layer structure) ----
    #   x -> LN1 -> q/k/v -> attention -> out_proj + x -> LN2 -> fc1 + GELU -> fc2 + x
    # Folded (default): LayerNorm(x) is never written and x is never re-read for statistics.  The GEMM that PRODUCES x (out_proj / fc2
    # with the residual add) leaves per-row partial sums of what it stored (vidi_gemm_res_stats), a tiny launch turns them into
    # (mean, rstd) per row (vidi_ln_finalize) and the GEMM that CONSUMES LayerNorm(x) applies them in its epilogue (vidi_gemm_ln:
    # Linear(LayerNorm(x)) == rstd * (x Wf^T - mean * colsum) + shift).  Only the tower's first LayerNorm needs a pass (row_stats).
    def _tower_layer(self, x, L, ws, eps, act, attn_kw, qkv_kw):
        M = x.shape[0]
        st, part, h, yqk, vt, ao, f1 = ws["st"], ws["part"], ws["h"], ws["yqk"], ws["vt"], ws["ao"], ws["f1"]
        Hd = x.shape[1]
        if self.ln_fold and self.attn_rm:
            # q | k | v from ONE plain projection, written head-major ([3][frame][head][token][d]: a head's key rows contiguous -> whole
            # 128-byte lines for the attention's K / V tiles); the attention kernel transposes V on the fly (LDS transpose read), so the
            # GEMM has no scattered V^T stores
            qkv = ws["qkv"]
            hip.gemm_ln_heads(x, L["wqkv"], st, L["sqkv"], L["cqkv"], qkv, seq=attn_kw["N"], hd=attn_kw["D"])
            hip.attn_self_rm(qkv, ao[:M], B=attn_kw["B"], N=attn_kw["N"], H=attn_kw["H"], D=attn_kw["D"], scale=attn_kw["scale"], head_major=True)
            hip.gemm_res_stats(ao[:M], L["wo"], L["bo"], x, x, part)
            hip.ln_finalize(part, st, M, Hd, eps)
            hip.gemm_ln(x, L["fc1"], st, L["s1"], L["c1"], f1[:M], act=act)
            hip.gemm_res_stats(f1[:M], L["fc2"], L["b2"], x, x, part)
            hip.ln_finalize(part, st, M, Hd, eps)
        elif self.ln_fold:
            hip.gemm_qkv_vt_ln(x, L["wqkv"], st, L["sqkv"], L["cqkv"], yqk[:M], vt, **qkv_kw)
            hip.attn_self(yqk[:M], vt, ao[:M], **attn_kw)
            hip.gemm_res_stats(ao[:M], L["wo"], L["bo"], x, x, part)
            hip.ln_finalize(part, st, M, Hd, eps)
            hip.gemm_ln(x, L["fc1"], st, L["s1"], L["c1"], f1[:M], act=act)
            hip.gemm_res_stats(f1[:M], L["fc2"], L["b2"], x, x, part)
            hip.ln_finalize(part, st, M, Hd, eps)                     # statistics of the NEXT layer's first LayerNorm
        else:
            hip.norm(hip.NORM_LAYER, x, L["ln1w"], eps=eps, bias=L["ln1b"], out=h[:M])
            hip.gemm_qkv_vt(h[:M], L["wqkv"], L["bqkv"], yqk[:M], vt, **qkv_kw)
            hip.attn_self(yqk[:M], vt, ao[:M], **attn_kw)
            hip.gemm(ao[:M], L["wo"], L["bo"], x, residual=x)
            hip.norm(hip.NORM_LAYER, x, L["ln2w"], eps=eps, bias=L["ln2b"], out=h[:M])
            hip.gemm(h[:M], L["fc1"], L["b1"], f1[:M], act=act)
            hip.gemm(f1[:M], L["fc2"], L["b2"], x, residual=x)

    def pos_table(self, which: str, l: int, N: int, i0: int = 0, rows: Optional[int] = None) -> torch.Tensor:
        """rms_norm(LearnablePosEmbd(...)) rows [i0, i0+rows) of l — pos.py:41-65, multimodal.py:194-197."""
        rows = l if rows is None else rows
        d = self.cfg.hidden_size
        pe = torch.empty((rows, d), dtype=torch.float32, device=self.dev)
        hip.sinusoid(pe, self.pos_div, rows=rows, i0=i0, l=l, N=N, d=d)
        h = hip.gemm_f32(pe, self.mm[f"pos_{which}.mlp.0.weight"], self.mm[f"pos_{which}.mlp.0.bias"], hip.ACT_GELU_ERF)
        h = hip.gemm_f32(h, self.mm[f"pos_{which}.mlp.2.weight"], self.mm[f"pos_{which}.mlp.2.bias"], hip.ACT_NONE)
        return hip.norm(hip.NORM_MM_NOW, None, None, eps=1e-5, x_f32=h, dtype=self.dtype)

    # -----------------------------------------------------------------------------------------
    # SigLIP tower: features = hidden_states[-2]  (mm_vision/siglip.py:29-34)
    # -----------------------------------------------------------------------------------------
    def siglip_forward(self, pixel: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        T = pixel.shape[0]
        S, P, side = cfg.vis_image_size, cfg.vis_patch_size, cfg.vis_side
        N, Hv, nh = side * side, cfg.vis_hidden_size, cfg.vis_num_heads
        hd = Hv // nh
        Npad = _round_up(N, 64)
        V = self.vis
        out = torch.empty((T * N, Hv), dtype=self.dtype, device=self.dev)
        fc = max(1, cfg.vis_frames_per_chunk)
        Mmax = min(T, fc) * N
        A = None if self.patch_loader else self._buf("vis_A", (Mmax, V["kpad"]))
        fold = self.ln_fold
        ws = {"st": self._buf("vis_stats", (2 * Mmax,), dtype=torch.float32) if fold else None,
              "part": self._buf("vis_part", (2 * Mmax * hip.stat_strips(Hv),), dtype=torch.float32) if fold else None,
              "h": None if fold else self._buf("vis_h", (Mmax, Hv)),
              # Q|K rows + V^T planes of the Vt attention arm; the default arm (head-major q|k|v + transpose-read attention) needs neither
              "yqk": None if (fold and self.attn_rm) else self._buf("vis_qk", (Mmax, 2 * Hv)),
              "vt": None if (fold and self.attn_rm) else self._buf("vis_vt", (min(T, fc), nh, hd, Npad), zero=True),
              "qkv": self._buf("vis_qkv", (Mmax, 3 * Hv)) if (fold and self.attn_rm) else None,
              "ao": self._buf("vis_ao", (Mmax, Hv)), "f1": self._buf("vis_f1", (Mmax, V["ipad"]))}
        pixel = pixel.to(self.dtype).contiguous()
        for c0 in range(0, T, fc):
            c1 = min(T, c0 + fc)
            Tc, M = c1 - c0, (c1 - c0) * N
            x = out[c0 * N: c1 * N]
            if self.patch_loader:
                hip.patch_embed(pixel[c0:c1], V["patch_w"], V["patch_b"], V["pos"], x, T=Tc, S=S, P=P)       # TP siglip:124-130, 178
            else:
                hip.im2col_patch(pixel[c0:c1], A[:M], T=Tc, S=S, P=P, Kpad=V["kpad"])
                hip.gemm(A[:M], V["patch_w"], V["patch_b"], x, residual=V["pos"], rmod=N)
            if fold:
                hip.row_stats(x, ws["st"], cfg.vis_ln_eps)
            pr = self.probe if (self.probe is not None and "vis_frames" in self.probe) else None
            grab = (lambda: pr["vis_x"].append(torch.stack([x[(f - c0) * N: (f - c0 + 1) * N].clone() for f in pr["vis_frames"] if c0 <= f < c1]))) \
                if pr is not None and any(c0 <= f < c1 for f in pr["vis_frames"]) else None
            for L in V["layers"]:
                if grab is not None:
                    grab()
                self._tower_layer(x, L, ws, cfg.vis_ln_eps, hip.ACT_GELU_TANH,
                                  dict(B=Tc, N=N, Npad=Npad, H=nh, D=hd, koff=Hv, scale=0.0 if getattr(self, "vis_prescaled", False) else hd ** -0.5),
                                  dict(vstart=2 * Hv, hd=hd, seq=N, seqpad=Npad, nheads=nh))
            if grab is not None:
                grab()
        return out.view(T, N, Hv)

    # -----------------------------------------------------------------------------------------
    # encode_video_images — multimodal.py:156-208 (one video; frame shard [f0, f0+T) of T_total)
    # -----------------------------------------------------------------------------------------
    def encode_video_images(self, pixel: torch.Tensor, frame_offset: int = 0, total_frames: Optional[int] = None,
                            normalizer: Optional[float] = None, sample_flag: Optional[torch.Tensor] = None,
                            vis_features: Optional[torch.Tensor] = None, budget_frames: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """`pixel`: this call's frames [T,3,S,S] of ONE video — all of it, or (sharded) frames frame_offset .. of its total_frames.
        `budget_frames`: the frame count the token-budget rule sees.  The reference applies the rule to the CONCATENATED frames of
        the whole batch (multimodal.py:157-158, 175-180: `n_tokens = image_features.size(0) * ...`), so two 200-frame videos answered
        in one batch are pooled like a 400-frame video; default = this video's total frames (batch of one)."""
        cfg = self.cfg
        T = pixel.shape[0]
        Ttot = total_frames if total_frames is not None else T
        side, pool, Hv, H = cfg.vis_side, cfg.mm_image_pool_size, cfg.vis_hidden_size, cfg.hidden_size
        if T == 0:                                  # a rank whose frame shard is empty (more ranks than frames)
            return (torch.empty((0, H), dtype=self.dtype, device=self.dev), torch.empty((0,), dtype=torch.uint8, device=self.dev))
        f = self.siglip_forward(pixel) if vis_features is None else vis_features
        if self.mistral:
            # Vidi-7B: learned conv (k = ceil(side/pool), stride 1) -> bilinear(align_corners=True) to pool x pool
            # (Vidi_7B/.../multimodal.py:165-170, mm_vision/pool.py:19-26); no token-budget rule
            k = cfg.img_pool_kernel
            oc = side - k + 1
            oh = ow = pool
            pooled = torch.empty((T * oh * ow, Hv), dtype=self.dtype, device=self.dev)
            f3 = f.reshape(T, side * side, Hv)
            # the k x k window gathered by the GEMM's loader (vidi_conv_window; default) or an im2col buffer + the generic GEMM
            # (VIDI_POOL_LOADER=0, the A/B arm: 88 MB of im2col rows per frame at k = 14, C = 1152)
            loader = self.pool_loader and Hv % 64 == 0 and k * k * Hv >= 192
            fc = max(1, min(T, 1024)) if loader else max(1, min(T, (1 << 30) // (oc * oc * k * k * Hv * 2)))      # one descriptor (< 4 GB) / im2col chunk <= 1 GiB
            for t0 in range(0, T, fc):
                t1 = min(T, t0 + fc)
                if loader:
                    conv = self._buf("pool_conv", ((t1 - t0) * oc * oc, Hv))
                    hip.conv_window(f3[t0:t1], self.mm["img_pool_w"], conv, T=t1 - t0, side=side, C=Hv, k=k)
                else:
                    col = self._buf("pool_col", ((t1 - t0) * oc * oc, k * k * Hv))
                    hip.im2col_nhwc(f3[t0:t1], col, T=t1 - t0, side=side, C=Hv, k=k)
                    conv = hip.gemm(col, self.mm["img_pool_w"], None)
                hip.resize_bilinear_ac(conv, pooled[t0 * oh * ow: t1 * oh * ow], T=t1 - t0, s_in=oc, s_out=pool, C=Hv)
        else:
            hw = token_budget_hw(Ttot if budget_frames is None else budget_frames, side, pool, cfg.mm_max_tokens_base)   # global T (of the batch) decides
            resize = hw[0] != 28
            h, w = hw if resize else (side + 1, side + 1)
            oh, ow = h // pool, w // pool
            C4 = Hv * pool * pool
            pooled = torch.empty((T * oh * ow, C4), dtype=self.dtype, device=self.dev)
            hip.pool_s2d(f, pooled, T=T, side=side, C=Hv, h=h, w=w, m=pool, resize=resize)
        p1 = hip.gemm(pooled, self.mm["img_w0"], self.mm["img_b0"], act=hip.ACT_GELU_ERF)
        p2 = hip.gemm(p1, self.mm["img_w2"], self.mm["img_b2"])
        x = hip.norm(hip.NORM_MM, p2, self.mm["img_norm"], eps=1e-5)
        ph = self.pos_table("h", oh, pool)
        pw_ = self.pos_table("w", ow, pool)
        pt = self.pos_table("t", Ttot, cfg.mm_time_interval, i0=frame_offset, rows=T)
        hip.add_pos(x, ph, pw_, pt, T=T, oh=oh, ow=ow, H=H)
        if sample_flag is None:
            sample_flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
            hip.any_nonzero(pixel.to(self.dtype).contiguous().view(-1), sample_flag)
        mask = torch.empty((T * oh * ow,), dtype=torch.uint8, device=self.dev)
        feats = hip.norm(hip.NORM_LLM, x, self.mm["llm_norm"], eps=1e-5, mask_out=mask, sample_flag=sample_flag,
                         normalizer=1.0 if normalizer is None else normalizer)
        return feats, mask

    # -----------------------------------------------------------------------------------------
    # Whisper encoder (mm_audio/whisper.py:26-27; TP whisper/modeling_whisper.py:540-648)
    # -----------------------------------------------------------------------------------------
    def whisper_forward(self, mel: torch.Tensor) -> torch.Tensor:
        cfg, A = self.cfg, self.aud
        C, nm, Lm = mel.shape
        Da, nh = cfg.aud_d_model, cfg.aud_num_heads
        hd = Da // nh
        N = Lm // 2
        Npad = _round_up(N, 64)
        out = torch.empty((C * N, Da), dtype=self.dtype, device=self.dev)
        cb = max(1, cfg.aud_chunks_per_batch)
        nb = min(C, cb)
        # rows 0 and L+1 of every chunk are the conv padding; 4 spare zero rows at the very end absorb the
        # (zero-weighted) K padding overrun of the last row view
        melT = self._buf("aud_melT", (nb * (Lm + 2) + 4, nm), zero=True)[: nb * (Lm + 2)].view(nb, Lm + 2, nm)
        y1 = self._buf("aud_y1", (nb, Lm + 1, Da), zero=True)              # row 0 = left zero pad of conv2
        y1[:, 0].zero_()
        fold = self.ln_fold
        ws = {"st": self._buf("aud_stats", (2 * nb * N,), dtype=torch.float32) if fold else None,
              "part": self._buf("aud_part", (2 * nb * N * hip.stat_strips(Da),), dtype=torch.float32) if fold else None,
              "h": None if fold else self._buf("aud_h", (nb * N, Da)),
              "yqk": None if (fold and self.attn_rm) else self._buf("aud_qk", (nb * N, 2 * Da)),
              "vt": None if (fold and self.attn_rm) else self._buf("aud_vt", (nb, nh, hd, Npad), zero=True),
              "qkv": self._buf("aud_qkv", (nb * N, 3 * Da)) if (fold and self.attn_rm) else None,
              "ao": self._buf("aud_ao", (nb * N, Da)), "f1": self._buf("aud_f1", (nb * N, cfg.aud_ffn_dim))}
        mel = mel.to(self.dtype).contiguous()
        for c0 in range(0, C, cb):
            c1 = min(C, c0 + cb)
            Cc, M = c1 - c0, (c1 - c0) * N
            x = out[c0 * N: c1 * N]
            hip.mel_transpose_pad(mel[c0:c1], melT[:Cc])
            # conv1 (k3,p1) as a GEMM over overlapping rows of melT; GELU(erf)
            hip.gemm(melT[0], A["conv1_w"], A["conv1_b"], y1[:Cc, 1:], act=hip.ACT_GELU_ERF, M=Lm, K=A["k1"], ldx=nm,
                     batch=Cc, bsX=(Lm + 2) * nm, bsY=(Lm + 1) * Da)
            # conv2 (k3,s2,p1): row t reads y1 rows 2t..2t+2; GELU(erf); + embed_positions
            hip.gemm(y1[0], A["conv2_w"], A["conv2_b"], x.view(Cc, N, Da), act=hip.ACT_GELU_ERF, residual=A["pos"], rmod=N,
                     M=N, K=3 * Da, ldx=2 * Da, batch=Cc, bsX=(Lm + 1) * Da, bsY=N * Da, bsR=0)
            if fold:
                hip.row_stats(x, ws["st"], cfg.aud_ln_eps)
            pr = self.probe if (self.probe is not None and "aud_windows" in self.probe) else None
            grab = (lambda: pr["aud_x"].append(torch.stack([x[(c - c0) * N: (c - c0 + 1) * N].clone() for c in pr["aud_windows"] if c0 <= c < c1]))) \
                if pr is not None and any(c0 <= c < c1 for c in pr["aud_windows"]) else None
            for L in A["layers"]:
                if grab is not None:
                    grab()
                self._tower_layer(x, L, ws, cfg.aud_ln_eps, hip.ACT_GELU_ERF,
                                  dict(B=Cc, N=N, Npad=Npad, H=nh, D=hd, koff=Da, scale=hd ** -0.5),
                                  dict(vstart=2 * Da, hd=hd, seq=N, seqpad=Npad, nheads=nh))
                if self.dtype == torch.float16:                     # TP whisper:409-411 overflow guard
                    cv = torch.finfo(torch.float16).max - 1000
                    x.clamp_(min=-cv, max=cv)
                    if fold:                                        # the clamp may have changed rows: their statistics again
                        hip.row_stats(x, ws["st"], cfg.aud_ln_eps)
            if grab is not None:
                grab()
            hip.norm(hip.NORM_LAYER, x, A["lnw"], eps=cfg.aud_ln_eps, bias=A["lnb"], out=x)
        return out.view(C, N, Da)

    # -----------------------------------------------------------------------------------------
    # encode_video_audios — multimodal.py:210-252 (one sample)
    # -----------------------------------------------------------------------------------------
    def encode_video_audios(self, mel: torch.Tensor, audio_size: int, normalizer: Optional[float] = None,
                            aud_features: Optional[torch.Tensor] = None, chunk_offset: int = 0,
                            sample_flag: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """`mel` may be a shard of the sample's 30-s windows starting at window `chunk_offset`;
        `audio_size` is always the GLOBAL mel-frame count (the floors of multimodal.py:226-235 are global)."""
        cfg = self.cfg
        H, Da, pool = cfg.hidden_size, cfg.aud_d_model, cfg.mm_audio_pool_size
        if mel.shape[0] == 0 and aud_features is None:          # a rank whose window shard is empty
            if audio_token_counts(audio_size, cfg)[1] < 2:
                raise ValueError("LearnablePosEmbd requires more than one audio token (pos.py:42)")
            return (torch.empty((0, H), dtype=self.dtype, device=self.dev), torch.empty((0,), dtype=torch.uint8, device=self.dev))
        f = self.whisper_forward(mel) if aud_features is None else aud_features              # [C, 1500, Da]
        s1, s2_total = audio_token_counts(audio_size, cfg)
        flat = f.reshape(-1, Da)
        if s2_total < 2:
            raise ValueError("LearnablePosEmbd requires more than one audio token (pos.py:42)")
        from .shard import audio_shard_tokens
        tok0, s2 = audio_shard_tokens(chunk_offset, f.shape[0], f.shape[1], pool, s2_total)   # first global token / count of this shard
        if s2 == 0:
            return (torch.empty((0, H), dtype=self.dtype, device=self.dev), torch.empty((0,), dtype=torch.uint8, device=self.dev))
        # Conv1d(k=pool, s=pool, no bias) over the first s1 rows == GEMM on a [s2, pool*Da] view
        pooled = hip.gemm(flat, self.mm["aud_pool"], None, M=s2, K=pool * Da, ldx=pool * Da)
        p1 = hip.gemm(pooled, self.mm["aud_w0"], self.mm["aud_b0"], act=hip.ACT_GELU_ERF)
        p2 = hip.gemm(p1, self.mm["aud_w2"], self.mm["aud_b2"])
        x = hip.norm(hip.NORM_MM, p2, self.mm["aud_norm"], eps=1e-5)
        pt = self.pos_table("t", s2_total, cfg.mm_time_interval, i0=tok0, rows=s2)
        hip.add_pos(x, None, None, pt, T=s2, oh=1, ow=1, H=H)
        flag = sample_flag
        if flag is None:
            flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
            hip.any_nonzero(mel.to(self.dtype).contiguous().view(-1), flag)
        mask = torch.empty((s2,), dtype=torch.uint8, device=self.dev)
        feats = hip.norm(hip.NORM_LLM, x, self.mm["llm_norm"], eps=1e-5, mask_out=mask, sample_flag=flag,
                         normalizer=1.0 if normalizer is None else normalizer)
        return feats, mask

    # -----------------------------------------------------------------------------------------
    # multimodal stream through all layers (diagonal V2V/A2A + cache fill)
    # -----------------------------------------------------------------------------------------
    def mm_stream_prefill(self, img: Optional[torch.Tensor], img_mask: Optional[torch.Tensor],
                          aud: Optional[torch.Tensor], aud_mask: Optional[torch.Tensor],
                          pre_normalized: bool = True, check_masks: bool = True) -> MMState:
        """img/aud: [N, H] features (already multiplied by the normalizer when pre_normalized)."""
        cfg = self.cfg
        H, nkv, hd, G = cfg.hidden_size, cfg.num_key_value_heads, cfg.head_dim, cfg.num_attention_heads // cfg.num_key_value_heads
        kvd = nkv * hd
        n_img = 0 if img is None else img.shape[0]
        n_aud = 0 if aud is None else aud.shape[0]
        aud_start = _round_up(n_img, 64)
        ntot = aud_start + _round_up(n_aud, 64)
        ntile = ntot // 64
        st = MMState(n_img=n_img, n_aud=n_aud, img_start=0, aud_start=aud_start, ntile64=ntile)
        st.g_img, st.g_aud = n_img, n_aud
        if self.sharded:                                # modality presence is a global property
            import torch.distributed as dist
            t = torch.tensor([n_img, n_aud], dtype=torch.int64, device=self.dev if dist.get_backend(self.pg) != "gloo" else "cpu")
            dist.all_reduce(t, group=self.pg)
            st.g_img, st.g_aud = int(t[0]), int(t[1])
        if ntot == 0 and not self.sharded:
            return st
        Lr = cfg.num_hidden_layers
        X = torch.zeros((ntot, H), dtype=self.dtype, device=self.dev)
        for src, off, n in ((img, 0, n_img), (aud, aud_start, n_aud)):
            if n:
                if pre_normalized:
                    X[off: off + n].copy_(src)
                else:
                    hip.scale(src.contiguous(), X[off: off + n], self.normalizer)
        st.kc = torch.empty((Lr, nkv, ntile, 64, hd), dtype=self.dtype, device=self.dev)
        st.vtc = torch.empty((Lr, nkv, 2 * ntile, hd, 32), dtype=self.dtype, device=self.dev)
        hbuf = self._buf("mm_h", (ntot, H))
        vrow = self._buf("mm_vrow", (ntot, kvd))
        u = self._buf("mm_u", (ntot, H))
        gt = self._buf("mm_g", (ntot, cfg.intermediate_size))
        eps = cfg.rms_norm_eps
        fused = self.stream_norm2 and not self.mistral
        pr = self.probe if (self.probe is not None and "stream_rows" in self.probe) else None
        for li, L in enumerate(self.layers if ntot > 0 else []):
            if pr is not None:
                pr["stream_x"].append(X.index_select(0, pr["stream_rows"]))
            if not fused or li == 0:                                                                 # fused: produced by the previous layer's second pass
                hip.norm(self.norm_mode, X, L["ln_in"], eps=eps, out=hbuf)                           # gemma.py:183-184 / mistral.py:204-205
            hip.gemm_kv_cache(hbuf, L["wkv"], st.kc[li], st.vtc[li], vrow, kvd=kvd, hd=hd, ntile64=ntile, tok0=0)   # :61-63
            if li == Lr - 1:
                break                                                                                # dead update on the last layer
            if self.mistral:
                if self.fold_repkv:
                    hip.gemm(vrow, L["wo_kv"], None, X, residual=X)                                  # mistral.py:219-221: x += o_proj(repeat_kv(V))
                else:
                    hip.gemm(vrow, L["wo"], None, X, repkv=(hd, G), K=G * kvd, residual=X)
                hip.norm(hip.NORM_MM, X, L["ln_post_attn"], eps=eps, out=hbuf)                       # :131-134 feed_foward
                hip.gemm_glu(hbuf, L["wgu"], gt, act=hip.ACT_SILU)
                hip.gemm(gt, L["wdown"], None, X, residual=X)                                        # :135
                continue
            if self.fold_repkv:
                hip.gemm(vrow, L["wo_kv"], None, u)                                                  # :196-197 o_proj(repeat_kv(V)), K folded
            else:
                hip.gemm(vrow, L["wo"], None, u, repkv=(hd, G), K=G * kvd)
            if fused:
                # residual + post-norm and the following pre-norm in ONE pass over the rows (vidi_resid_norm2, the phase-by-phase
                # form for many rows: the arithmetic and rounding points of NORM_GEMMA_ADD followed by NORM_GEMMA, bit-identical; 4
                # instead of 5 row transfers per pair)
                hip.resid_norm2(u, None, None, X, L["ln_post_attn"], L["ln_pre_ffn"], X, hbuf, eps=eps)   # :198-201 + :118
            else:
                hip.norm(hip.NORM_GEMMA_ADD, u, L["ln_post_attn"], eps=eps, residual=X, out=X)       # :198-201
                hip.norm(hip.NORM_GEMMA, X, L["ln_pre_ffn"], eps=eps, out=hbuf)                      # :118
            hip.gemm_geglu(hbuf, L["wgu"], gt)                                                       # :119 gate/up + GeGLU
            hip.gemm(gt, L["wdown"], None, u)                                                        # :119 down_proj
            if fused:
                hip.resid_norm2(u, None, None, X, L["ln_post_ffn"], self.layers[li + 1]["ln_in"], X, hbuf, eps=eps)   # :120-121 + next :183
            else:
                hip.norm(hip.NORM_GEMMA_ADD, u, L["ln_post_ffn"], eps=eps, residual=X, out=X)        # :120-121
        if check_masks:
            # one host sync per VIDEO (the reference syncs per layer per step: xattn.py:214-215)
            for name, m in (("img", img_mask), ("aud", aud_mask)):
                if m is None and not self.sharded:
                    continue
                nv = int(m.sum().item()) if m is not None and m.numel() else 0
                nv_global = nv
                if self.sharded:                       # "sample has any valid key" is a global property
                    import torch.distributed as dist
                    t = torch.tensor([nv], dtype=torch.int64, device=self.dev if dist.get_backend(self.pg) != "gloo" else "cpu")
                    dist.all_reduce(t, group=self.pg)
                    nv_global = int(t.item())
                setattr(st, f"{name}_any_valid", nv_global > 0)
                if m is not None and 0 < nv < m.numel() or (m is not None and nv == 0 and m.numel() > 0 and nv_global > 0):
                    pad = torch.zeros(_round_up(m.numel(), 64), dtype=torch.uint8, device=self.dev)
                    pad[: m.numel()] = m
                    setattr(st, f"{name}_mask", pad)
        return st

    # -----------------------------------------------------------------------------------------
    # text stream
    # -----------------------------------------------------------------------------------------
    def new_text_state(self, B: int, Lmax: int) -> TextState:
        cfg = self.cfg
        kvd = cfg.num_key_value_heads * cfg.head_dim
        Lr = cfg.num_hidden_layers
        return TextState(B=B, Lmax=Lmax,
                         kc=torch.zeros((Lr, B, Lmax, kvd), dtype=self.dtype, device=self.dev),
                         vc=torch.zeros((Lr, B, Lmax, kvd), dtype=self.dtype, device=self.dev),
                         kmask=torch.zeros((B, Lmax), dtype=torch.uint8, device=self.dev))

    def reorder_text_state(self, ts: TextState, parents: torch.Tensor) -> None:
        """Beam search: row i of the text K/V cache continues from row parents[i] (HF `Cache.reorder_cache(beam_idx)`, which the reference
        gets through gemma.py:646-655).  The rows of a prompt's beams share their mask and length, the video / audio K/V belong to the
        prompt, so only the filled part of the text K/V moves (a gather in place: data movement, a few MB per step)."""
        n = ts.past_len
        idx = parents.to(self.dev, torch.int64)
        ts.kc[:, :, :n] = ts.kc[:, :, :n].index_select(1, idx)
        ts.vc[:, :, :n] = ts.vc[:, :, :n].index_select(1, idx)
        ts.kmask.copy_(ts.kmask.index_select(0, idx))
        if ts.n_valid is not None:
            ts.n_valid = ts.n_valid.index_select(0, idx)

    def _rope_tables(self, max_pos: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos/sin rows per position, fp32 math then cast (TP gemma2:118-136); host-precomputed table."""
        if self._rope_cache is None or self._rope_cache[0].shape[0] < max_pos:
            with torch.inference_mode(False):                        # cached across calls: a normal tensor (see _buf)
                hd = self.cfg.head_dim
                n = max(max_pos, 2048)
                inv = 1.0 / (self.cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
                fr = torch.arange(n, dtype=torch.float)[:, None] * inv[None, :]
                emb = torch.cat((fr, fr), dim=-1)
                self._rope_cache = (emb.cos().to(self.dtype).to(self.dev), emb.sin().to(self.dtype).to(self.dev))
        return self._rope_cache

    def _cross_local(self, q: torch.Tensor, li: int, mm: MMState, which: str, R: int):
        """Split-KV cross-attention of R query rows over this rank's keys of one modality -> (Opart, ML, zsplit, n_local_keys)."""
        cfg = self.cfg
        nkv, hd = cfg.num_key_value_heads, cfg.head_dim
        G = cfg.num_attention_heads // nkv
        n = mm.n_img if which == "img" else mm.n_aud
        start = mm.img_start if which == "img" else mm.aud_start
        mask = mm.img_mask if which == "img" else mm.aud_mask
        Rpad = _round_up(R, 32)
        nsub = (n + 31) // 32
        row_blocks = -(-(Rpad // 32) // hip.attn_cross_row_tiles_per_block(Rpad, cfg.attn_logit_softcapping, self.dtype))      # blocks along the rows (a block covers 1 or 4 row tiles)
        zsplit = max(1, min(256 // max(1, nkv * row_blocks), (nsub + 7) // 8))
        key = f"xattn_ws_{which}_{zsplit}_{Rpad}"        # one workspace per modality: both partial sets live until the merge
        if key not in self._ws:
            with torch.inference_mode(False):
                self._ws[key] = hip.attn_cross_workspace(zsplit, nkv, Rpad, hd, self.dev)
        opart, ml = self._ws[key]
        if n > 0:
            hip.attn_cross(q, mm.kc[li], mm.vtc[li], mask, opart, ml, R=R, Rpad=Rpad, G=G, nkv=nkv, HD=hd, ntile64=mm.ntile64,
                           key_start=start, n_keys=n, scale=cfg.query_pre_attn_scalar ** -0.5,
                           softcap=cfg.attn_logit_softcapping, zsplit=zsplit)
        return opart, ml, zsplit, n

    def _cross(self, q: torch.Tensor, li: int, mm: MMState, which: str, out: torch.Tensor, R: int, defer_merge: bool = False):
        """single-GPU T2V / T2A: all keys are local"""
        cfg = self.cfg
        nkv, hd = cfg.num_key_value_heads, cfg.head_dim
        G = cfg.num_attention_heads // nkv
        any_valid = mm.img_any_valid if which == "img" else mm.aud_any_valid
        Rpad = _round_up(R, 32)
        opart, ml, zsplit, n = self._cross_local(q, li, mm, which, R)
        if defer_merge and n > 0:
            return (opart, ml, out, zsplit, not any_valid)      # merged together with the other modality (attn_merge2)
        hip.attn_merge(opart, ml, out, W=zsplit, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=hd, zero_out=not any_valid)
        return None

    def _cross_dual(self, q: torch.Tensor, li: int, mm: MMState, out_img: torch.Tensor, out_aud: torch.Tensor, R: int,
                    defer_merge: bool = False):
        """single-GPU T2V + T2A of one layer: ONE split-KV launch over both key regions (vidi_attn_cross2, the key slices shared out
        in proportion to the modalities' keys) and ONE merge launch.  False: shape not eligible (the caller runs them separately)."""
        cfg = self.cfg
        nkv, hd = cfg.num_key_value_heads, cfg.head_dim
        G = cfg.num_attention_heads // nkv
        Rpad = _round_up(R, 32)
        Z = 256 // max(1, nkv * -(-(Rpad // 32) // hip.attn_cross_row_tiles_per_block(Rpad, cfg.attn_logit_softcapping, self.dtype)))
        if mm.n_img <= 0 or mm.n_aud <= 0 or Z < 2:
            return False
        from .shard import split_key_slices
        zs = dict(zip(("img", "aud"), split_key_slices(Z, (mm.n_img + 31) // 32, (mm.n_aud + 31) // 32)))
        sets = {}
        for which in ("img", "aud"):
            key = f"xattn_ws_{which}_{zs[which]}_{Rpad}"
            if key not in self._ws:
                with torch.inference_mode(False):
                    self._ws[key] = hip.attn_cross_workspace(zs[which], nkv, Rpad, hd, self.dev)
            opart, ml = self._ws[key]
            sets[which] = dict(mask=mm.img_mask if which == "img" else mm.aud_mask, opart=opart, ml=ml,
                               key_start=mm.img_start if which == "img" else mm.aud_start,
                               n_keys=mm.n_img if which == "img" else mm.n_aud, zsplit=zs[which])
        hip.attn_cross2(q, mm.kc[li], mm.vtc[li], sets["img"], sets["aud"], R=R, Rpad=Rpad, G=G, nkv=nkv, HD=hd, ntile64=mm.ntile64,
                        scale=cfg.query_pre_attn_scalar ** -0.5, softcap=cfg.attn_logit_softcapping)
        merge = ((sets["img"]["opart"], sets["img"]["ml"], out_img, zs["img"], not mm.img_any_valid),
                 (sets["aud"]["opart"], sets["aud"]["ml"], out_aud, zs["aud"], not mm.aud_any_valid))
        if defer_merge:                                  # the caller merges (together with the decode step's T2T launch)
            return merge
        hip.attn_merge2(*merge[0], *merge[1], nkv=nkv, R=R, Rpad=Rpad, G=G, HD=hd)
        return True

    def _cross_sharded_begin(self, q: torch.Tensor, li: int, mm: MMState, outs: Dict[str, torch.Tensor], R: int, async_op: bool = False):
        """T2V + T2A of one layer with the keys sharded over ranks, first half.  Per rank: split-KV partials over the local
        keys -> ONE launch folds them into the partial form (numerator, m, l) of both modalities, written straight into a packed send
        buffer -> ONE all-gather (RCCL) per layer, issued here.  `async_op`: the collective runs on the backend's stream and the caller
        goes on launching (the T2T attention) before `_cross_sharded_finish` waits for it.
        `outs`: {"img": [M, nq*hd] slice, "aud": ...} for the modalities the SAMPLE has (a global property)."""
        from .shard import packed_offsets, packed_partial_floats
        cfg = self.cfg
        nkv, hd = cfg.num_key_value_heads, cfg.head_dim
        G = cfg.num_attention_heads // nkv
        Rpad = _round_up(R, 32)
        mods = [w for w in ("img", "aud") if w in outs]
        total = packed_partial_floats(len(mods), nkv, R, hd)
        pk = f"xattn_pack_{len(mods)}_{R}"
        if pk not in self._ws:
            with torch.inference_mode(False):
                self._ws[pk] = (torch.zeros((total,), dtype=torch.float32, device=self.dev),
                                torch.zeros((self.world, total), dtype=torch.float32, device=self.dev))
        send, recv = self._ws[pk]
        flat = recv.view(-1)
        local_sets, final_sets = [None, None], [None, None]
        ldo = None
        for slot, which in enumerate(mods):
            opart, ml, zsplit, n = self._cross_local(q, li, mm, which, R)
            o_off, ml_off = packed_offsets(slot, nkv, R, hd)
            any_valid = mm.img_any_valid if which == "img" else mm.aud_any_valid
            # a rank that holds no key of the modality contributes the neutral partial (W = 0 -> m = -inf, l = 0)
            local_sets[slot] = dict(opart=opart if n > 0 else None, ml=ml if n > 0 else None, ws_o=nkv * Rpad * hd, ws_ml=nkv * Rpad * 2,
                                    W=zsplit if n > 0 else 0, out_f32=send[o_off:], out_ml=send[ml_off:])
            final_sets[slot] = dict(opart=flat[o_off:], ml=flat[ml_off:], ws_o=total, ws_ml=total, W=self.world, out=outs[which],
                                    zero=not any_valid)
            ldo = outs[which].stride(0)
        dt = hip._dt(outs[mods[0]])
        hip.attn_merge2_sharded(local_sets, nkv=nkv, R=R, Rpad=Rpad, rpo=R, G=G, HD=hd, ldo=ldo, dtype=dt)
        work = self._all_gather(recv, send, async_op=async_op)
        self.n_collectives += 1
        return work, final_sets, dict(nkv=nkv, R=R, Rpad=R, rpo=R, G=G, HD=hd, ldo=ldo, dtype=dt)

    def _cross_sharded_finish(self, pending) -> None:
        """second half: wait for the layer's all-gather (a stream dependency, no host block under RCCL), then ONE launch merges the
        world's partials of both modalities (exact: the tanh softcap is per logit, the merge is the LSE identity flash-attn itself uses
        between key blocks)."""
        work, final_sets, kw = pending
        work.wait()
        hip.attn_merge2_sharded(final_sets, **kw)

    def _cross_sharded(self, q: torch.Tensor, li: int, mm: MMState, outs: Dict[str, torch.Tensor], R: int):
        self._cross_sharded_finish(self._cross_sharded_begin(q, li, mm, outs, R))

    # -----------------------------------------------------------------------------------------
    # multi-GPU (one process per GPU, RCCL): frame/chunk-sharded keys, replicated text stream
    # -----------------------------------------------------------------------------------------
    def set_dist(self, group=None, mode: Optional[str] = None):
        """One process per GPU: every rank is handed the SAME video and encodes its contiguous range of frames / 30-s windows with the
        global positions (vidi_amd/shard.py).  What happens next is `mode` (default: $VIDI_DIST_MODE or "sharded_stream"):

        "gather_tokens"   the north-star's literal collective (BASELINE configs[3]): RCCL all-gather of the visual and audio tokens
                          (645 MB + 258 MB at 60 min) -> reference-order `[Nv, H]` on every rank, then the decoder replicated: the
                          diagonal stream, the K/V caches and the cross-attention are those of a single GPU.  Scales the towers only
                          (Amdahl: the 1.9 PFLOP stream stays on every rank).
        "sharded_stream"  the shard stays resident through the decoder: each rank streams ITS tokens through the 42 layers and keeps its
                          K/V shard; per layer one all-gather of packed cross-attention partials + exact LSE merge."""
        import torch.distributed as dist
        mode = mode or os.environ.get("VIDI_DIST_MODE", "sharded_stream")
        if mode not in ("gather_tokens", "sharded_stream"):
            raise ValueError(f"dist mode {mode!r}: expected 'gather_tokens' or 'sharded_stream'")
        self.pg = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dist_mode = mode
        # VIDI_FORCE_SHARDED=1: a one-rank group still goes through shard -> (token all-gather | partial form -> all-gather -> merge), so
        # the RCCL branch of either exchange can be exercised on a one-GPU box (tests/test_gpu_dist.py)
        self.shard_encode = self.world > 1 or os.environ.get("VIDI_FORCE_SHARDED", "0") == "1"
        self.sharded = self.shard_encode and mode == "sharded_stream"

    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        from .dist import all_gather_packed
        return all_gather_packed(out, inp, self.pg, async_op=async_op)

    def text_forward(self, hidden: torch.Tensor, positions: torch.Tensor, ts: TextState, mm: Optional[MMState],
                     Lq: int, new_mask: Optional[torch.Tensor] = None, dyn: bool = False) -> torch.Tensor:
        """hidden: [B*Lq, H] embeds already multiplied by the normalizer; positions: int64 [B*Lq].
        Appends Lq positions to the text cache.  Returns the final-norm hidden states [B*Lq, H].
        dyn=True (Lq == 1): the cache slot comes from the device scalars ts.pos_dev / ts.pos_idx, so nothing in the
        launch sequence depends on the decode position (hipGraph capture); the caller advances them."""
        cfg = self.cfg
        B = ts.B
        H, nq, nkv, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        nqd, kvd = nq * hd, nkv * hd
        M = B * Lq
        eps = cfg.rms_norm_eps
        p0 = ts.past_len
        if p0 + Lq > ts.Lmax:
            raise RuntimeError("text KV cache exhausted")
        cos_t, sin_t = self._rope_tables(int(ts.Lmax) + 1)
        cos = cos_t.index_select(0, positions).contiguous()
        sin = sin_t.index_select(0, positions).contiguous()
        if dyn:
            if Lq != 1 or new_mask is not None or ts.pos_dev is None:
                raise RuntimeError("dyn text_forward is the single-token decode step")
            ts.kmask.index_fill_(1, ts.pos_idx, 1)
        elif new_mask is None:
            ts.kmask[:, p0: p0 + Lq] = 1
        else:
            ts.kmask[:, p0: p0 + Lq] = new_mask.to(torch.uint8)
        has_img = mm is not None and mm.g_img > 0
        has_aud = mm is not None and mm.g_aud > 0
        nstream = 1 + int(has_img) + int(has_aud)
        # ---- varlen: drop the pad positions of a ragged batch (flash_attn_varlen_func's view of the batch, xattn.py:36-103, 123) ----
        # Every text-side kernel but the T2T attention is row-wise (or, the cross-attention, row-wise on the query side): they run on the
        # packed valid rows.  The T2T launch keeps the [B, Lq] frame: q | k | v rows are scattered into a zeroed padded buffer in front of it
        # and its output rows gathered back (two small copies per layer).  8 prompts of 24..52 tokens: 304 rows instead of 416.
        Mp, rows_map = M, None
        prt = self.probe if (self.probe is not None and "text_h" in self.probe) else None
        if self.text_varlen and new_mask is not None and Lq > 1 and B > 1 and not dyn and prt is None:
            valid = new_mask.reshape(-1).to(torch.bool)
            rows_map = torch.nonzero(valid).squeeze(1)               # one host sync per ragged PREFILL (the reference syncs per layer: xattn.py:214-215)
            if 0 < rows_map.numel() < M:
                M = int(rows_map.numel())
                hidden = hidden.index_select(0, rows_map)
                cos_pad, sin_pad = cos, sin
                cos, sin = cos.index_select(0, rows_map), sin.index_select(0, rows_map)
                qkv_pad = self._buf("t_qkv_pad", (Mp, nqd + 2 * kvd), zero=True)          # pad rows: zeros for good (only valid rows are ever copied in)
                att_pad = self._buf("t_att_pad", (Mp, nqd))
            else:
                rows_map = None
        hn = self._buf("t_h", (M, H))
        qkv = self._buf("t_qkv", (M, nqd + 2 * kvd))
        qr = self._buf("t_qr", (M, nqd))
        att = self._buf("t_att", (3 * M, nqd))
        oall = self._buf("t_o", (3 * M, H))
        gt = self._buf("t_g", (M, cfg.intermediate_size))
        dn = self._buf("t_d", (M, H))
        sc = cfg.query_pre_attn_scalar ** -0.5
        tmp = self._buf("t_tmp", (M, H)) if self.mistral else None
        fused = not self.mistral       # Gemma2 wiring: add3 + post-norm/residual + next pre-norm run as one launch (vidi_resid_norm2)
        # decode (few rows): that launch is folded into the skinny projection that consumes it (vidi_gemv[_glu]_norm2); the residual
        # stream then ping-pongs between two buffers (the fused launch reads the old one in every block while block 0 writes the new one)
        fuse_proj = fused and self.decode_norm_gemv and hip.gemv_norm2_fits(M, H)
        other = self._buf("t_hid2", (M, H)) if fuse_proj else None
        qkv_ready = False
        nL = len(self.layers)
        final_out = None
        for li, L in enumerate(self.layers):
            if prt is not None:
                prt["text_h"].append(hidden.clone())
            if not qkv_ready:
                if not (fused and li > 0):                                                           # fused: produced by the previous layer's FFN side
                    hip.norm(self.norm_mode, hidden, L["ln_in"], eps=eps, out=hn)                   # gemma.py:162 / mistral.py:187
                self.proj(hn, L["wqkv"], qkv)
            qkv_ready = False
            # RoPE'd q for T2T (raw q stays in qkv for the cross-attention, gemma.py:58), RoPE'd k and v appended to the cache
            window = cfg.sliding_window if (self.mistral or li % 2 == 0) else 0                       # gemma.py:104; Mistral: every layer
            t2t_fused = Lq == 1 and self.decode_attn and hip.attn_text_decode_fits(nq=nq, nkv=nkv, HD=hd, Lmax=ts.Lmax, window=window,
                                                                                   pos0=None if dyn else p0)
            t2t_kw = dict(B=B, Lmax=ts.Lmax, nq=nq, nkv=nkv, HD=hd, window=window, scale=sc, softcap=cfg.attn_logit_softcapping, pos0=p0,
                          pos_dev=ts.pos_dev if dyn else None)
            # decode with both modalities on one GPU: the T2T launch is held back and issued together with the merge of the
            # cross-attention partials (vidi_attn_text_decode_merge2) after the T2V + T2A partial pass
            t2t_with_merge = (t2t_fused and self.decode_tail and self.cross_dual and not self.sharded and hd in (128, 256)
                              and mm is not None and mm.g_img > 0 and mm.g_aud > 0)
            k = 1
            pending = outs = None
            if self.sharded:
                outs = {}
                if has_img:
                    outs["img"] = att[k * M: (k + 1) * M]; k += 1
                if has_aud:
                    outs["aud"] = att[k * M: (k + 1) * M]; k += 1
                if outs and self.dist_overlap:
                    # local partials + pack + the layer's all-gather go out FIRST (they need only the raw q of the projection); the
                    # collective then runs on RCCL's stream while the T2T launches below execute — same launches, same operands,
                    # same bits as the serial order
                    pending = self._cross_sharded_begin(qkv[:, :nqd], li, mm, outs, R=M * (nq // nkv), async_op=True)
            if t2t_with_merge:
                pass
            elif t2t_fused:
                # decode: rope + cache append + T2T in one launch (vidi_attn_text_decode)
                hip.attn_text_decode(qkv, ts.kc[li], ts.vc[li], ts.kmask, cos, sin, att[:M], **t2t_kw)
            elif dyn:
                hip.rope_cache(qkv, qr, ts.kc[li], ts.vc[li], cos, sin, B=B, Lq=1, Lmax=ts.Lmax, nq=nq, nkv=nkv, HD=hd,
                               pos_dev=ts.pos_dev)
                hip.attn_text_dyn(qr, ts.kc[li], ts.vc[li], ts.kmask, att[:M], B=B, Lq=1, Lmax=ts.Lmax, nq=nq, nkv=nkv, HD=hd,
                                  past_len_dev=ts.pos_dev, window=window, scale=sc, softcap=cfg.attn_logit_softcapping)
            elif rows_map is not None:
                qkv_pad.index_copy_(0, rows_map, qkv)                                               # packed rows -> their [B, Lq] slots
                qr_pad = self._buf("t_qr_pad", (Mp, nqd))
                hip.rope_cache(qkv_pad, qr_pad, ts.kc[li], ts.vc[li], cos_pad, sin_pad, B=B, Lq=Lq, Lmax=ts.Lmax, nq=nq, nkv=nkv, HD=hd, pos0=p0)
                hip.attn_text(qr_pad, ts.kc[li], ts.vc[li], ts.kmask, att_pad, B=B, Lq=Lq, Lmax=ts.Lmax, nq=nq, nkv=nkv, HD=hd,
                              past_len=p0, window=window, scale=sc, softcap=cfg.attn_logit_softcapping)
                torch.index_select(att_pad, 0, rows_map, out=att[:M])
            else:
                hip.rope_cache(qkv, qr, ts.kc[li], ts.vc[li], cos, sin, B=B, Lq=Lq, Lmax=ts.Lmax, nq=nq, nkv=nkv, HD=hd, pos0=p0)
                hip.attn_text(qr, ts.kc[li], ts.vc[li], ts.kmask, att[:M], B=B, Lq=Lq, Lmax=ts.Lmax, nq=nq, nkv=nkv, HD=hd,
                              past_len=p0, window=window, scale=sc, softcap=cfg.attn_logit_softcapping)
            qraw = qkv[:, :nqd]
            G = nq // nkv
            if self.sharded:
                if pending is not None:
                    self._cross_sharded_finish(pending)
                elif outs:
                    self._cross_sharded(qraw, li, mm, outs, R=M * G)
            else:
                both = has_img and has_aud
                pend = []
                dual = both and self.cross_dual and self._cross_dual(qraw, li, mm, att[M: 2 * M], att[2 * M: 3 * M], R=M * G,
                                                                     defer_merge=t2t_with_merge)
                if t2t_with_merge and dual:
                    hip.attn_text_decode_merge2(qkv, ts.kc[li], ts.vc[li], ts.kmask, cos, sin, att[:M], dual[0], dual[1], R=M * G,
                                                Rpad=_round_up(M * G, 32), **t2t_kw)
                elif t2t_with_merge:                                    # the dual launch was not eligible after all: T2T on its own
                    hip.attn_text_decode(qkv, ts.kc[li], ts.vc[li], ts.kmask, cos, sin, att[:M], **t2t_kw)
                if dual:
                    k = 3                                               # T2V + T2A partials in one launch, merged by one launch
                    both = False
                    has_img_l = has_aud_l = False
                else:
                    has_img_l, has_aud_l = has_img, has_aud
                if has_img_l:
                    pend.append(self._cross(qraw, li, mm, "img", att[k * M: (k + 1) * M], R=M * G, defer_merge=both)); k += 1
                if has_aud_l:
                    pend.append(self._cross(qraw, li, mm, "aud", att[k * M: (k + 1) * M], R=M * G, defer_merge=both)); k += 1
                if both:
                    if pend[0] is not None and pend[1] is not None:     # T2V and T2A partials merged by one launch
                        (oa, mla, outa, wa, za), (ob, mlb, outb, wb, zb) = pend
                        hip.attn_merge2(oa, mla, outa, wa, za, ob, mlb, outb, wb, zb, nkv=nkv, R=M * G, Rpad=_round_up(M * G, 32), G=G, HD=hd)
                    else:
                        for pd in pend:
                            if pd is not None:
                                hip.attn_merge(pd[0], pd[1], pd[2], W=pd[3], nkv=nkv, R=M * G, Rpad=_round_up(M * G, 32), G=G, HD=hd, zero_out=pd[4])
            # one o_proj pass over the stacked [text; image; audio] attention outputs (gemma.py:94 x3)
            self.proj(att[: nstream * M], L["wo"], oall[: nstream * M])
            if self.mistral:
                # mistral.py:261: residual + text + image + audio, evaluated left to right (each sum rounds), then
                # feed_foward = x + mlp(post_attention_layernorm(x)) (:131-137)
                if nstream <= 2:
                    hip.add3(hidden, oall[:M], oall[M: 2 * M] if nstream == 2 else None, hidden)
                else:
                    hip.add3(hidden, oall[:M], oall[M: 2 * M], tmp)
                    hip.add3(tmp, oall[2 * M: 3 * M], None, hidden)
                hip.norm(hip.NORM_MM, hidden, L["ln_post_attn"], eps=eps, out=hn)
                if M <= 8:
                    self.proj_glu(hn, L["wgu"], gt, hip.ACT_SILU)
                    self.proj(gt, L["wdown"], dn)
                    hip.add3(hidden, dn, None, hidden)
                else:
                    hip.gemm_glu(hn, L["wgu"], gt, act=hip.ACT_SILU)
                    hip.gemm(gt, L["wdown"], None, hidden, residual=hidden)
                continue
            # text + image + audio (:236), residual + post_attention_layernorm (:237), pre_feedforward_layernorm (:118)
            o_b = oall[M: 2 * M] if nstream >= 2 else None
            o_c = oall[2 * M: 3 * M] if nstream == 3 else None
            if fuse_proj:
                hip.gemv_glu_norm2(oall[:M], o_b, o_c, hidden, L["ln_post_attn"], L["ln_pre_ffn"], other, L["wgu"], gt, eps=eps,
                                   act=hip.ACT_GELU_TANH)
                hidden, other = other, hidden
            else:
                hip.resid_norm2(oall[:M], o_b, o_c, hidden, L["ln_post_attn"], L["ln_pre_ffn"], hidden, hn, eps=eps)
                if M <= 8:
                    self.proj_glu(hn, L["wgu"], gt, hip.ACT_GELU_TANH)
                else:
                    hip.gemm_geglu(hn, L["wgu"], gt)
            self.proj(gt, L["wdown"], dn)
            # residual + post_feedforward_layernorm (:120-121), then the NEXT layer's input_layernorm (:162) or the final norm (:411)
            if li + 1 < nL and fuse_proj:
                nxt = self.layers[li + 1]
                hip.gemv_norm2(dn, None, None, hidden, L["ln_post_ffn"], nxt["ln_in"], other, nxt["wqkv"], qkv, eps=eps)   # + next :57-60
                hidden, other = other, hidden
                qkv_ready = True
            elif li + 1 < nL:
                hip.resid_norm2(dn, None, None, hidden, L["ln_post_ffn"], self.layers[li + 1]["ln_in"], hidden, hn, eps=eps)
            else:
                final_out = torch.empty_like(hidden)
                hip.resid_norm2(dn, None, None, hidden, L["ln_post_ffn"], self.final_norm, hidden, final_out, eps=eps)
        if not dyn:
            ts.past_len = p0 + Lq
        if prt is not None:
            prt["text_h"].append(hidden.clone())                    # the residual stream after the last layer (before the final norm)
        if final_out is None:
            final_out = hip.norm(self.norm_mode, hidden, self.final_norm, eps=eps)                      # gemma.py:411 / mistral.py:423
        if rows_map is not None:                                    # back into the caller's [B * Lq, H] frame (pad rows: zeros)
            full = torch.zeros((Mp, H), dtype=final_out.dtype, device=self.dev)
            full.index_copy_(0, rows_map, final_out)
            return full
        return final_out

    # ---- graph-captured greedy decode -------------------------------------------------
    def decode_step_dyn(self, ids: torch.Tensor, ts: TextState, mm: Optional[MMState]) -> torch.Tensor:
        """One greedy decode step whose launches do not depend on the position: embed(ids) -> 42 layers ->
        lm_head/softcap/argmax; advances the device-side position scalars.  Returns next ids [B]."""
        emb = self.embed_tokens(ids)
        posn = ts.n_valid.clone()                                  # HF: position = cumsum(mask) - 1 of the new token
        ts.n_valid += 1
        hn = self.text_forward(emb, posn, ts, mm, Lq=1, dyn=True)
        _, nxt = self.logits_argmax(hn)
        ts.pos_dev += 1
        ts.pos_idx += 1
        return nxt

    def make_decode_graph(self, ts: TextState, mm: Optional[MMState], first_ids: torch.Tensor):
        """Runs ONE decode step eagerly on a side stream (warm-up, consumes `first_ids`), then captures the step in
        a hipGraph.  Returns (next_ids_after_warmup, replay) where replay(ids) -> next ids advances the caches by
        one token per call.  Sharded (one process per GPU): the per-layer all-gathers of the partials are RCCL launches on the
        capture stream, so the whole step — 42 exchanges included — is one graph launch per token; the gloo test transport moves
        the partials through the host and cannot be captured."""
        if self.sharded:
            import torch.distributed as dist
            if dist.get_backend(self.pg) != "nccl":
                raise RuntimeError("graph-captured decode over shards needs the RCCL backend (the gloo test transport is host-driven)")
        if ts.past_len + 2 > ts.Lmax:
            raise RuntimeError("text KV cache exhausted")
        ts.pos_dev = torch.tensor([ts.past_len], dtype=torch.int32, device=self.dev)
        ts.pos_idx = torch.tensor([ts.past_len], dtype=torch.int64, device=self.dev)
        g_in = first_ids.clone()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            warm = self.decode_step_dyn(g_in, ts, mm)
        torch.cuda.current_stream().wait_stream(side)
        ts.past_len += 1
        graph = torch.cuda.CUDAGraph()
        # sharded: other threads of the process (the process group's watchdog) may touch the runtime while this thread captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if self.sharded else "global"):
            g_out = self.decode_step_dyn(g_in, ts, mm)

        def replay(ids: torch.Tensor) -> torch.Tensor:
            if ts.past_len + 1 > ts.Lmax:
                raise RuntimeError("text KV cache exhausted")
            g_in.copy_(ids)
            graph.replay()
            ts.past_len += 1
            return g_out
        replay.graph = graph
        return warm, replay

    def logits_argmax(self, hn_last: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """lm_head + final softcap (gemma.py:565-569) + greedy argmax.  hn_last: [B, H]."""
        B = hn_last.shape[0]
        logits = self.proj(hn_last.contiguous(), self.lm_head)
        idx = torch.empty((B,), dtype=torch.int64, device=self.dev)
        hip.softcap_argmax(logits, idx, self.cfg.final_logit_softcapping, self.argmax_workspace(B))
        return logits, idx

    def argmax_workspace(self, B: int) -> torch.Tensor:
        """the engine's scratch for vidi_softcap_argmax (the C ABI keeps no state; the engine issues its work on one stream at a time)"""
        ws = getattr(self, "_am_ws", None)
        if ws is None or ws.numel() * 8 < 16 * B:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("argmax workspace must exist before graph capture (run one eager step first)")
            ws = self._am_ws = hip.softcap_argmax_workspace(max(256, B), self.dev)
        return ws

    def embed_tokens(self, ids: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """embed_tokens(ids) * normalizer (normalize=False: the raw table rows); ids < 0 give zero rows (padding)."""
        ids = ids.reshape(-1).to(torch.int64).contiguous()
        out = torch.empty((ids.numel(), self.cfg.hidden_size), dtype=self.dtype, device=self.dev)
        return hip.embed(ids, self.embed, out, normalizer=self.normalizer if normalize else 1.0)
