"""Harness that makes the reference's OWN model code (Vidi1.5_9B/vidi/model/lmm/dattn/{gemma,multimodal,xattn,split}.py,
mm_vision/*, mm_audio/*, mm_layer/*) executable on CPU in the build container, so golden vectors can be produced by running
it instead of by restating it.  TEST INFRASTRUCTURE ONLY; needs /root/reference, never runs on the GPU box.

Nothing of the reference is modified or copied.  Only its absent THIRD-PARTY dependencies are replaced:

  * flash-attn 2.8.3 (CUDA-only)      -> `flash_attn_func` / `flash_attn_varlen_func` / `bert_padding.*` restated in eager
                                          fp32 PyTorch from the published definition
                                          softmax(softcap*tanh(q k^T*scale/softcap) + causal/window mask) v
  * liger-kernel (in-place swaps of Gemma2RMSNorm/GeGLU by fused equivalents) -> no-op (HF modules stay)
  * deepspeed.comm                     -> torch.distributed (never initialised: single process)
  * transformers 4.50 -> installed 5.x: `HybridCache`/`DynamicCache` with the 4.50 list-of-(k,v) protocol the reference
    uses (`len(c) <= layer`, `c.update(k, v, layer)`, `c[layer]`), an attention function registered under the name
    "flash_attention_2" that does what 4.50's FA2 path did for a 2-D padding mask (cache update, causal, sliding window,
    softcap), and shims for keyword/attribute renames
  * tokenizer/processor/tower `from_pretrained` (network) -> random-init towers of a given tiny config
  * langid, decord, orjson, ...        -> empty modules (dataset code is imported by package __init__ but not used)
"""
from __future__ import annotations

import importlib.machinery
import sys
import types

import torch

REF = "/root/reference/Vidi1.5_9B"


class _Any(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {})


def _stub(name, **attrs):
    m = _Any(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ------------------------------------------------------------------------------------------------ flash-attn restated
def _fa_core(q, k, v, softmax_scale, causal, window_size, softcap, key_ok=None):
    """q [B,Lq,H,D], k/v [B,Lk,Hk,D] -> [B,Lq,H,D]; flash-attn semantics: bottom-right aligned causal mask,
    window (left,right) with -1 = unbounded, GQA by head repetition, fp32 softmax."""
    B, Lq, H, D = q.shape
    Lk, Hk = k.shape[1], k.shape[2]
    if softmax_scale is None:
        softmax_scale = D ** -0.5
    qf, kf, vf = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    if Hk != H:
        kf = kf.repeat_interleave(H // Hk, dim=1)
        vf = vf.repeat_interleave(H // Hk, dim=1)
    s = qf @ kf.transpose(-1, -2) * softmax_scale
    if softcap and softcap > 0:
        s = softcap * torch.tanh(s / softcap)
    i = torch.arange(Lq)[:, None] + (Lk - Lq)
    j = torch.arange(Lk)[None, :]
    ok = torch.ones((Lq, Lk), dtype=torch.bool)
    left, right = window_size
    if causal:
        right = 0 if right < 0 else min(right, 0)
    if right >= 0:
        ok &= j <= i + right
    if left >= 0:
        ok &= j >= i - left
    ok = ok[None, None]
    if key_ok is not None:
        ok = ok & key_ok[:, None, None, :]
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)
    return (p @ vf).transpose(1, 2).to(q.dtype)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    assert dropout_p == 0.0 and alibi_slopes is None
    return _fa_core(q, k, v, softmax_scale, causal, window_size, softcap)


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                           deterministic=False, return_attn_probs=False, block_table=None):
    assert dropout_p == 0.0
    out = torch.empty_like(q)
    for b in range(len(cu_seqlens_q) - 1):
        q0, q1 = int(cu_seqlens_q[b]), int(cu_seqlens_q[b + 1])
        k0, k1 = int(cu_seqlens_k[b]), int(cu_seqlens_k[b + 1])
        if q1 > q0:
            out[q0:q1] = _fa_core(q[None, q0:q1], k[None, k0:k1], v[None, k0:k1], softmax_scale, causal, window_size, softcap)[0]
    return out


def index_first_axis(x, indices):
    return x[indices]


def unpad_input(hidden_states, attention_mask, unused_mask=None):
    seqlens = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    cu = torch.nn.functional.pad(torch.cumsum(seqlens, dim=0, dtype=torch.int32), (1, 0))
    # 4 values, as xattn.py:94 unpacks them (flash-attn < 2.6.2; 2.8.3 returns a 5th `seqused` value, with which the
    # reference's padded-batch prefill would raise — the arithmetic intended is the same)
    return hidden_states.flatten(0, 1)[indices], indices, cu, int(seqlens.max())


def pad_input(hidden_states, indices, batch, seqlen):
    out = torch.zeros((batch * seqlen, *hidden_states.shape[1:]), dtype=hidden_states.dtype, device=hidden_states.device)
    out[indices] = hidden_states
    return out.view(batch, seqlen, *hidden_states.shape[1:])


# ------------------------------------------------------------------------------ transformers-4.50 cache protocol restated
class ListCache:
    """DynamicCache of transformers 4.50 as the reference uses it: per-layer (k, v), concatenated along the sequence."""

    def __init__(self, *a, **k):
        self.key_cache, self.value_cache = [], []

    def __len__(self):
        return len(self.key_cache)

    def __getitem__(self, i):
        return self.key_cache[i], self.value_cache[i]

    def update(self, k, v, layer_idx, cache_kwargs=None):
        if len(self.key_cache) <= layer_idx:
            self.key_cache.append(k)
            self.value_cache.append(v)
        else:
            self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], k], dim=-2)
            self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], v], dim=-2)
        return self.key_cache[layer_idx], self.value_cache[layer_idx]

    def get_seq_length(self, layer_idx=0):
        return 0 if len(self.key_cache) <= layer_idx else self.key_cache[layer_idx].shape[-2]

    @classmethod
    def from_legacy_cache(cls, past_key_values=None):
        c = cls()
        for i, (k, v) in enumerate(past_key_values or ()):
            c.update(k, v, i)
        return c

    def to_legacy_cache(self):
        return tuple(zip(self.key_cache, self.value_cache))


class TextCache(ListCache):
    """Stands in for HybridCache(config, max_batch_size, max_cache_len, dtype): the reference only constructs it and hands
    it to Gemma2Attention; a growing list cache holds the same keys (the static/sliding layout is an HF storage detail —
    the sliding window itself is applied by the attention function below)."""


def _t2t_attention(module, query, key, value, attention_mask, dropout=0.0, scaling=None, sliding_window=None,
                   softcap=None, **kwargs):
    """What transformers 4.50 `Gemma2Attention.forward` + its flash_attention_2 interface do for the text stream:
    update the cache with the rotated K/V, causal attention over all cached keys, sliding window on the layers that have
    one (keys j with i - j < window), logit softcap, 2-D padding mask = key validity.  query [B,H,L,D], key/value [B,Hk,L,D]."""
    cache = kwargs.get("past_key_value", None)
    if cache is not None:
        key, value = cache.update(key, value, module.layer_idx)
    q, k, v = query.transpose(1, 2), key.transpose(1, 2), value.transpose(1, 2)
    Lk = k.shape[1]
    key_ok = None
    if attention_mask is not None and attention_mask.dim() == 2:
        key_ok = attention_mask[:, -Lk:].bool()
        if key_ok.shape[1] < Lk:                                       # mask shorter than the cache: left part is valid
            key_ok = torch.cat([torch.ones((key_ok.shape[0], Lk - key_ok.shape[1]), dtype=torch.bool), key_ok], dim=1)
    window = (-1, -1)
    if sliding_window is not None:
        window = (sliding_window, sliding_window)                      # transformers 4.50 `_flash_attention_forward`: window_size=(W, W)
    out = _fa_core(q, k, v, scaling, True, window, softcap or 0.0, key_ok)
    return out, None


def install():
    """Install the third-party stand-ins and import the reference package.  Returns the reference modules."""
    import transformers  # noqa: F401
    import transformers.generation.utils  # noqa: F401  (must import before the deepspeed stub exists)
    import transformers.models.gemma2.modeling_gemma2 as mg
    import transformers.models.siglip.modeling_siglip  # noqa: F401
    import transformers.models.whisper.modeling_whisper  # noqa: F401
    import transformers.models.clip.modeling_clip  # noqa: F401
    from transformers import AutoConfig, AutoModelForCausalLM, AutoTokenizer  # noqa: F401
    import transformers.utils as tu
    import transformers.cache_utils as cu
    import torch.distributed as tdist

    lk = _stub("liger_kernel"); lk.__path__ = []
    lt = _stub("liger_kernel.transformers"); lt.__path__ = []
    _stub("liger_kernel.transformers.monkey_patch", apply_liger_kernel_to_gemma2=lambda *a, **k: None,
          LigerRMSNorm=type("LigerRMSNorm", (torch.nn.Module,), {}))
    ds = _stub("deepspeed"); ds.__path__ = []
    ds.comm = _stub("deepspeed.comm", **{k: getattr(tdist, k) for k in dir(tdist) if not k.startswith("_")})
    for n in ("langid", "decord", "cv2", "orjson", "ffmpeg", "av", "moviepy", "soundfile", "librosa"):
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                _stub(n)
    fa = _stub("flash_attn", flash_attn_func=flash_attn_func, flash_attn_varlen_func=flash_attn_varlen_func)
    fa.__path__ = []
    _stub("flash_attn.bert_padding", index_first_axis=index_first_axis, pad_input=pad_input, unpad_input=unpad_input)
    tu.is_flash_attn_2_available = lambda: True
    tu.is_flash_attn_greater_or_equal = lambda v: True
    cu.HybridCache = TextCache
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
    ALL_ATTENTION_FUNCTIONS["flash_attention_2"] = _t2t_attention
    from transformers.modeling_utils import PreTrainedModel
    # 5.x probes for the real flash-attn package / hub kernels when a model is built with "flash_attention_2": keep the
    # requested name (it resolves to `_t2t_attention` above; the towers are built with "eager")
    PreTrainedModel._check_and_adjust_attn_implementation = lambda self, attn_implementation, *a, **k: attn_implementation or "eager"
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import vidi.model.lmm.dattn.gemma as G
    import vidi.model.lmm.dattn.multimodal as MM
    import vidi.model.lmm.dattn.xattn as X
    G.DynamicCache = ListCache                                         # the 4.50 protocol (`cache[layer]`, `len(cache)`)
    G.HybridCache = TextCache
    return G, MM, X, mg


REF_7B = "/root/reference/Vidi_7B"


def install_7b():
    """Same for Vidi_7B/model (package name `model`; pinned transformers 4.44.2, flash-attn 2.6.3).  Extra stand-in:
    `modeling_mistral.MistralFlashAttention2` (removed in 5.x) = the installed `MistralAttention` (same projections, RoPE,
    GQA arithmetic) behind the 4.44 call signature `(hidden_states, attention_mask, position_ids, past_key_value, ...)
    -> (out, None, past_key_value)`, attending through `_t2t_attention` above."""
    import transformers  # noqa: F401
    import transformers.generation.utils  # noqa: F401
    import transformers.models.mistral.modeling_mistral as mm
    import transformers.models.siglip.modeling_siglip  # noqa: F401
    import transformers.models.whisper.modeling_whisper  # noqa: F401
    import transformers.models.clip.modeling_clip  # noqa: F401
    import transformers.utils as tu
    import transformers.cache_utils as cu
    _RealCache = cu.Cache

    fa = _stub("flash_attn", flash_attn_func=flash_attn_func, flash_attn_varlen_func=flash_attn_varlen_func)
    fa.__path__ = []
    _stub("flash_attn.bert_padding", index_first_axis=index_first_axis, pad_input=pad_input, unpad_input=unpad_input)
    for n in ("decord", "cv2", "ffmpeg"):
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                _stub(n)
    tu.is_flash_attn_2_available = lambda: True
    tu.is_flash_attn_greater_or_equal = lambda v: True
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS, PreTrainedModel
    ALL_ATTENTION_FUNCTIONS["flash_attention_2"] = _t2t_attention
    PreTrainedModel._check_and_adjust_attn_implementation = lambda self, attn_implementation, *a, **k: attn_implementation or "eager"

    class MistralFlashAttention2(mm.MistralAttention):
        def __init__(self, config, layer_idx=None):
            super().__init__(config, layer_idx)
            self.hidden_size = config.hidden_size
            self.num_heads = config.num_attention_heads
            self.num_key_value_heads = config.num_key_value_heads
            self.rotary_emb = mm.MistralRotaryEmbedding(config=config)

        def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                    output_attentions=False, use_cache=False, cache_position=None, **kwargs):
            pe = self.rotary_emb(hidden_states, position_ids)
            out, _ = super().forward(hidden_states, pe, attention_mask, past_key_value=past_key_value,
                                     cache_position=cache_position)
            return out, None, past_key_value

    mm.MistralFlashAttention2 = MistralFlashAttention2
    cu.DynamicCache = ListCache
    if REF_7B not in sys.path:
        sys.path.insert(0, REF_7B)
    import model.lmm.dattn.mistral as M7
    M7.DynamicCache = ListCache
    # `isinstance(past_key_values, Cache)` (mistral.py:338): our list cache, or the cache object HF's generate() loop creates
    M7.Cache = (ListCache, _RealCache)
    return M7, mm
