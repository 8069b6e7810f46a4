"""vidi_amd/beam.py against transformers' own `GenerationMixin._beam_search`, differentially: a tiny randomly initialised GPT-2 on CPU is
decoded by HF's `generate(num_beams=...)` and by `beam_search` driven with the same model's logits (a callback that re-gathers the running
sequences from their parents), over seeds, beam counts, EOS sets, length penalties, `early_stopping` modes and `num_return_sequences`.
(The reference reaches this HF code through gemma.py:646-655; tests/test_generate_api.py pins the end-to-end cases on the reference's
own generate().)"""
import itertools

import pytest
import torch

transformers = pytest.importorskip("transformers")

from vidi_amd.beam import beam_search  # noqa: E402

V = 37


def tiny_lm(seed):
    torch.manual_seed(seed)
    cfg = transformers.GPT2Config(n_layer=1, n_embd=32, n_head=2, vocab_size=V, n_positions=64, bos_token_id=0, eos_token_id=1)
    m = transformers.GPT2LMHeadModel(cfg).eval()
    with torch.no_grad():                                               # spread the logits: a near-uniform LM makes every beam a tie
        m.lm_head.weight.mul_(12.0)
    return m


CASES = [dict(num_beams=nb, eos=eos, length_penalty=lp, early_stopping=es, nrs=nrs)
         for nb, eos, lp, es, nrs in itertools.product((2, 4), ([1], [1, 5, 9], []), (1.0, 0.0, 2.0, -0.5), (False, True, "never"), (1, 2))]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_beam_search_equals_hf(seed):
    lm = tiny_lm(seed)
    g = torch.Generator().manual_seed(100 + seed)
    prompt = torch.randint(2, V, (2, 5), generator=g)                  # two prompts of equal length
    B, L = prompt.shape
    max_new = 9
    bad = []
    for c in CASES:
        nb, eos = c["num_beams"], c["eos"]
        with torch.no_grad():
            ref = lm.generate(prompt, attention_mask=torch.ones_like(prompt), do_sample=False, num_beams=nb, max_new_tokens=max_new,
                              eos_token_id=eos if eos else None, pad_token_id=0, length_penalty=c["length_penalty"], early_stopping=c["early_stopping"],
                              num_return_sequences=c["nrs"], output_scores=True, return_dict_in_generate=True)
        # the running sequences as beam_search keeps them are private to it: mirror them here from (tokens, parents)
        state = {"seq": prompt.repeat_interleave(nb, dim=0)}

        def logits_of(seq):
            with torch.no_grad():
                return lm(seq).logits[:, -1].float()

        def step(tokens, parents):
            if parents is not None:
                state["seq"] = state["seq"][parents]
            state["seq"] = torch.cat((state["seq"], tokens[:, None]), dim=1)
            return logits_of(state["seq"])

        fill = 0 or (eos[0] if eos else -1)                             # HF: `pad_token_id or eos_token_id[0]`, pad 0 -> the first EOS
        seqs, scores = beam_search(step, logits_of(state["seq"]), B, nb, V, max_new, eos, fill, [], [], c["length_penalty"], c["early_stopping"],
                                   c["nrs"])
        want = ref.sequences[:, L:]
        ok = seqs.shape == want.shape and torch.equal(seqs, want) and torch.allclose(scores, ref.sequences_scores, atol=1e-5, rtol=0)
        if not ok:
            bad.append((c, seqs.tolist(), want.tolist(), scores.tolist(), ref.sequences_scores.tolist()))
    assert not bad, f"{len(bad)} of {len(CASES)} configurations differ; first: {bad[0]}"


@pytest.mark.parametrize("seed", [0, 1])
def test_beam_sampling_equals_hf_draw_for_draw(seed):
    """`num_beams > 1` with `do_sample=True`: HF draws a row's continuations from softmax(accumulated log-probabilities) after its warpers;
    the same seed, the same draws (the warpers — temperature, top-k, top-p with HF's `min_tokens_to_keep = n_eos + 1` under beams — go
    through vidi_amd.sampling.warp_logits, as in the product)."""
    from vidi_amd.sampling import warp_logits
    lm = tiny_lm(seed)
    g = torch.Generator().manual_seed(200 + seed)
    prompt = torch.randint(2, V, (2, 4), generator=g)
    B, L = prompt.shape
    max_new = 7
    bad = []
    for nb, eos, kw in itertools.product((2, 3), ([1], [1, 5]), (dict(top_k=8), dict(top_k=0, top_p=0.8, temperature=1.5), dict(top_k=12, temperature=0.7))):
        torch.manual_seed(500 + seed)
        with torch.no_grad():
            ref = lm.generate(prompt, attention_mask=torch.ones_like(prompt), do_sample=True, num_beams=nb, max_new_tokens=max_new, eos_token_id=eos, pad_token_id=0,
                              output_scores=True, return_dict_in_generate=True, **kw)
        state = {"seq": prompt.repeat_interleave(nb, dim=0)}

        def logits_of(seq):
            with torch.no_grad():
                return lm(seq).logits[:, -1].float()

        def step(tokens, parents):
            if parents is not None:
                state["seq"] = state["seq"][parents]
            state["seq"] = torch.cat((state["seq"], tokens[:, None]), dim=1)
            return logits_of(state["seq"])

        keep = max(2, len(eos) + 1)
        warper = lambda ids, sc: warp_logits(sc, kw.get("temperature"), kw.get("top_k", 50), kw.get("top_p"), keep)        # noqa: E731
        torch.manual_seed(500 + seed)
        seqs, scores = beam_search(step, logits_of(state["seq"]), B, nb, V, max_new, eos, eos[0], [warper], [], 1.0, False, 1, do_sample=True)
        want = ref.sequences[:, L:]
        if not (seqs.shape == want.shape and torch.equal(seqs, want) and torch.allclose(scores, ref.sequences_scores, atol=1e-5, rtol=0)):
            bad.append((nb, eos, kw, seqs.tolist(), want.tolist()))
    assert not bad, f"{len(bad)} configurations differ; first: {bad[0]}"
