"""`num_beams > 1` for `VidiForCausalLM.generate`.  The reference hands `**kwargs` to HF's `GenerationMixin.generate`
(Vidi1.5_9B/vidi/model/lmm/dattn/gemma.py:646-655), so `num_beams`, `length_penalty`, `early_stopping` and `num_return_sequences` reach
HF's beam search there; this module restates that search (transformers' `GenerationMixin._beam_search`, the vectorised form of 4.50+)
over this package's engine interface:

  * every batch row is expanded to `num_beams` rows BEFORE the text prefill (HF's `_expand_inputs_for_generation`); the beams of a row
    share the resident video / audio K/V (`mm_state`), each has its own text K/V rows, re-gathered by `engine.reorder_text_state`
    when a step's survivors descend from other rows;
  * the reference drives HF with `inputs_embeds`, so the sequences the search (and every logits processor / stopping criterion) sees
    are the NEW tokens only: prompt length 0, `max_length == max_new_tokens`, and the length penalty divides by the number of new tokens;
  * per step: log-softmax of the soft-capped logits -> processors -> + the beam's running score -> top `K = max(2, 1 + n_eos) *
    num_beams` of a row's `num_beams * vocab` continuations; a continuation whose token is an EOS (or that reaches `max_new_tokens`, or
    that a stopping criterion flags) cannot run on; among the first `num_beams` of the K those are candidates for the row's finished
    list (score = accumulated log-prob / length ** length_penalty; best `num_beams` kept); the best `num_beams` of the others run on;
  * the loop ends when no row can improve (HF's heuristic: the best running score, length-normalised at the current length — at
    `max_new_tokens` for `early_stopping="never"` with a positive penalty — against the worst finished one), or, with
    `early_stopping=True`, when every row holds `num_beams` finished sequences, or when no continuation can run on.

Host-side selection logic on the [B * num_beams, vocab] logits the kernels produce; torch tensor ops (log_softmax / topk / gather) only,
like `sampling.py`.  The reference CLI decodes greedily; this path is off the benchmarked loop."""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple, Union

import torch

NEG = -1.0e9            # HF's "cannot be chosen" offset (added, not assigned: scores stay finite)


def _take(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """t[b, idx[b, j], ...] for every row b"""
    ix = idx
    while ix.dim() < t.dim():
        ix = ix.unsqueeze(-1)
    return torch.take_along_dim(t, ix, dim=1)


def beam_search(step_logits: Callable[[torch.Tensor, Union[torch.Tensor, None]], torch.Tensor], first_logits: torch.Tensor, batch: int,
                num_beams: int, vocab: int, max_new: int, eos_ids: Sequence[int], fill: int, processors: List, criteria: List,
                length_penalty: float = 1.0, early_stopping: Union[bool, str] = False,
                num_return_sequences: int = 1, do_sample: bool = False, generator=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`first_logits` [batch * num_beams, vocab]: the prefill's next-token logits of the expanded rows.  `step_logits(tokens, parents)`
    feeds one token per expanded row ([batch * num_beams] int64) after re-gathering the rows' text caches from `parents` (global row
    indices, or None when every row continues itself) and returns the next logits.  `do_sample`: beam-search multinomial sampling — the K
    continuations of a row are DRAWN without replacement from softmax(accumulated log-probabilities) instead of taken from its top (HF's
    `_get_top_k_continuations`; the sampling warpers belong to `processors`, applied to the log-probabilities like every other processor).
    -> (sequences [batch * num_return_sequences, n], their scores [batch * num_return_sequences])."""
    dev = first_logits.device
    nb, B, V = num_beams, batch, vocab
    K = max(2, 1 + len(eos_ids)) * nb
    head = torch.arange(K, device=dev) < nb                                       # only the first num_beams of the K may finish
    eos_t = torch.tensor(list(eos_ids), dtype=torch.int64, device=dev)
    run_seq = torch.full((B, nb, max_new), int(fill), dtype=torch.int64, device=dev)
    fin_seq = run_seq.clone()
    run_score = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    run_score[:, 1:] = NEG                                                        # the expanded rows are identical: one live beam at step 0
    fin_score = torch.full((B, nb), NEG, dtype=torch.float32, device=dev)
    fin_done = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    fin_len = torch.zeros((B, nb), dtype=torch.int64, device=dev)
    can_improve = torch.ones((B, 1), dtype=torch.bool, device=dev)
    row0 = (torch.arange(B, device=dev) * nb)[:, None]
    logits = first_logits
    for cur in range(max_new):
        logp = torch.log_softmax(logits.float(), dim=-1)
        flat = run_seq.view(B * nb, max_new)[:, :cur]
        for proc in processors:
            logp = proc(flat, logp)
        acc = (logp.view(B, nb, V) + run_score[:, :, None]).reshape(B, nb * V)
        if do_sample:
            top_i = torch.multinomial(torch.softmax(acc, dim=-1), num_samples=K, generator=generator)
            top_lp = torch.gather(acc, 1, top_i)
        else:
            top_lp, top_i = torch.topk(acc, K, dim=1)
        parent = torch.div(top_i, V, rounding_mode="floor")
        tok = top_i - parent * V
        cand = _take(run_seq, parent)
        cand[:, :, cur] = tok
        stops = torch.isin(tok, eos_t) | torch.full_like(tok, cur + 1 >= max_new, dtype=torch.bool)
        for crit in criteria:
            s = crit(cand.view(B * K, max_new)[:, : cur + 1], None)
            s = s.to(dev).bool() if torch.is_tensor(s) else torch.full((B * K,), bool(s), device=dev)
            stops = stops | s.view(B, K)
        # ---- the beams that run on: best num_beams of the continuations that did not stop
        live_lp = top_lp + stops.float() * NEG
        keep = torch.topk(live_lp, nb, dim=1).indices
        run_seq, run_score, run_parent = _take(cand, keep), _take(live_lp, keep), _take(parent, keep)
        # ---- the finished list: previous entries + this step's stopping continuations among the first num_beams
        just = stops & head[None, :]
        sc = top_lp / (float(cur + 1) ** length_penalty)
        sc = sc + (fin_done.all(dim=-1, keepdim=True) & (early_stopping is True)).float() * NEG
        sc = sc + (~can_improve).float() * NEG
        sc = sc + (~just).float() * NEG
        m_score = torch.cat((fin_score, sc), dim=1)
        best = torch.topk(m_score, nb, dim=1).indices
        fin_seq = _take(torch.cat((fin_seq, cand), dim=1), best)
        fin_score = _take(m_score, best)
        fin_done = _take(torch.cat((fin_done, just), dim=1), best)
        fin_len = _take(torch.cat((fin_len, torch.full((B, K), cur + 1, dtype=torch.int64, device=dev)), dim=1), best)
        # ---- can a running beam still beat the worst finished one?
        n = cur + 1
        ref_len = max_new if (early_stopping == "never" and length_penalty > 0.0) else n
        best_running = run_score[:, :1] / (float(ref_len) ** length_penalty)
        worst = torch.where(fin_done, fin_score.min(dim=1, keepdim=True).values, torch.full_like(fin_score, NEG))
        can_improve = can_improve & (best_running > worst).any(dim=-1, keepdim=True)
        go_on = bool(can_improve.any()) and not (bool(fin_done.all()) and early_stopping is True) and not bool(stops.all())
        if not go_on:
            break
        parents = (run_parent + row0).reshape(-1)
        same = bool((parents == torch.arange(B * nb, device=dev)).all())
        logits = step_logits(run_seq[:, :, cur].reshape(-1), None if same else parents)
    nrs = int(num_return_sequences)
    n_out = int(fin_len[:, :nrs].max())
    return fin_seq[:, :nrs, :n_out].reshape(B * nrs, n_out), fin_score[:, :nrs].reshape(B * nrs)
