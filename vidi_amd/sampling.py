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


def generation_kwargs_processors(kwargs: dict, eos_token_ids, device) -> tuple:
    """The logits processors / stopping criteria HF's `GenerationMixin` builds from plain `generate()` keyword arguments — the reference
    forwards `**kwargs` to `super().generate()` (gemma.py:646-655), so `repetition_penalty`, `no_repeat_ngram_size`, `bad_words_ids`,
    `min_length`, `min_new_tokens`, `suppress_tokens` and `max_time` work there without the caller building any object.  Returned in HF's
    order (`_get_logits_processor`); they are transformers' own classes (imported only when one of the arguments is given), applied to the
    NEW tokens only, as under `inputs_embeds` in the reference (prompt length 0).  -> (processors, criteria)"""
    procs, crits = [], []
    rp = kwargs.get("repetition_penalty")
    ng = kwargs.get("no_repeat_ngram_size")
    bad = kwargs.get("bad_words_ids")
    min_len = kwargs.get("min_length")
    min_new = kwargs.get("min_new_tokens")
    sup = kwargs.get("suppress_tokens")
    max_time = kwargs.get("max_time")
    want = (rp is not None and rp != 1.0) or (ng or 0) > 0 or bad or (min_len or 0) > 0 or (min_new or 0) > 0 or sup or max_time is not None
    if not want:
        return procs, crits
    from transformers.generation import logits_process as lp
    eos = list(eos_token_ids) if eos_token_ids else None
    dev = str(device)
    if rp is not None and rp != 1.0:
        procs.append(lp.RepetitionPenaltyLogitsProcessor(penalty=float(rp)))
    if (ng or 0) > 0:
        procs.append(lp.NoRepeatNGramLogitsProcessor(int(ng)))
    if bad:
        procs.append(lp.NoBadWordsLogitsProcessor(bad, eos_token_id=eos))
    if (min_len or 0) > 0 and eos:
        procs.append(lp.MinLengthLogitsProcessor(int(min_len), eos, device=dev))
    if (min_new or 0) > 0 and eos:
        procs.append(lp.MinNewTokensLengthLogitsProcessor(0, int(min_new), eos, device=dev))
    if sup:
        procs.append(lp.SuppressTokensLogitsProcessor(list(sup), device=dev))
    if max_time is not None:
        from transformers import MaxTimeCriteria
        crits.append(MaxTimeCriteria(max_time=float(max_time)))
    return procs, crits
