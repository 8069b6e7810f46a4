"""`do_sample=True` token selection for `VidiForCausalLM.generate` — the logits warpers HF's `GenerationMixin` applies for
the keyword arguments the reference forwards to it (gemma.py:646-655 `**kwargs`: temperature, top_k, top_p), then one
multinomial draw per row.  Host-side glue on the [B, vocab] logits the kernels produce (the reference CLI is greedy; this
path is off the hot loop's critical kernels and uses torch tensor ops only for sort/cumsum/multinomial)."""
from __future__ import annotations

from typing import Optional

import torch


def warp_logits(logits: torch.Tensor, temperature: Optional[float] = None, top_k: Optional[int] = None,
                top_p: Optional[float] = None, min_tokens_to_keep: int = 1) -> torch.Tensor:
    """TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper, in HF's order and with its tie/threshold rules.
    logits: [B, V] float32 (already soft-capped).  Filtered entries become -inf."""
    x = logits.float()
    if temperature is not None and temperature != 1.0:
        if temperature <= 0:
            raise ValueError("`temperature` has to be a strictly positive float")
        x = x / temperature
    if top_k is not None and top_k > 0:
        k = min(max(int(top_k), min_tokens_to_keep), x.shape[-1])
        kth = torch.topk(x, k, dim=-1).values[..., -1, None]
        x = x.masked_fill(x < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        if not 0.0 <= top_p <= 1.0:
            raise ValueError("`top_p` has to be a float in [0, 1]")
        sorted_x, sorted_idx = torch.sort(x, descending=False, dim=-1)
        cum = sorted_x.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1.0 - top_p)                                   # the low-probability tail whose mass is <= 1 - top_p
        remove[..., -min_tokens_to_keep:] = False
        x = x.masked_fill(remove.scatter(-1, sorted_idx, remove), float("-inf"))
    return x


def sample(logits: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """one draw per row from softmax(logits) -> [B] int64"""
    probs = torch.softmax(logits.float(), dim=-1)
    return torch.multinomial(probs, num_samples=1, generator=generator).squeeze(-1)
