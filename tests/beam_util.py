"""Test helper: the log-probabilities a model assigns to a GIVEN new-token sequence, step by step, as beam search scores them
(log-softmax of the soft-capped logits, then the logits processors built from the generate kwargs — HF's order under `num_beams > 1`)."""
import torch


def forced_log_probs(model, ids, mm, seq, kwargs, eos):
    from vidi_amd.sampling import generation_kwargs_processors
    procs, _ = generation_kwargs_processors(kwargs, eos, model.engine.dev)
    seen = []

    def force(input_ids, scores):
        step = input_ids.shape[1]
        lp = torch.log_softmax(scores.float(), -1)
        for p in procs:
            lp = p(input_ids, lp)
        seen.append(lp[0].clone())
        out = torch.full_like(scores, float("-inf"))
        out[0, int(seq[step])] = 0.0
        return out

    model.generate(ids, mm_state=mm, do_sample=False, max_new_tokens=len(seq), eos_token_id=None, pad_token_id=0, logits_processor=[force])
    return seen


def hypothesis_score(lps, seq, length_penalty=1.0):
    n = len(seq)
    return sum(float(lps[i][seq[i]]) for i in range(n)) / (n ** length_penalty)


def trim_at_eos(seq, eos):
    n = len(seq)
    for e in eos:
        if e in seq:
            n = min(n, seq.index(e) + 1)
    return seq[:n]
