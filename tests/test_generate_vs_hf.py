"""`VidiForCausalLM.generate()`'s host logic against transformers' own `GenerationMixin.generate`, differentially and broadly: the product
class is driven over an adapter "engine" whose numerics are a tiny randomly initialised GPT-2 (CPU), HF generates from `inputs_embeds` of the
same prompt — how the reference drives it (gemma.py:646-655): the sequences, every logits processor and every stopping criterion see the
NEW tokens only.  Swept: greedy with EOS lists / pad fill, the plain generation kwargs HF turns into processors (repetition penalty,
n-gram ban, bad words, suppressed tokens, min_new_tokens / min_length), `max_length`, sampling (temperature / top-k / top-p /
num_return_sequences, HF's defaults for absent knobs), beam search and beam-search sampling, `return_dict_in_generate` scores.
(tests/test_generate_api.py pins the end-to-end cases on the REFERENCE's own generate(); this file covers the combinations.)"""
from types import SimpleNamespace

import pytest
import torch

transformers = pytest.importorskip("transformers")

from vidi_amd.config import tiny  # noqa: E402
from vidi_amd.model import VidiForCausalLM  # noqa: E402

V = 41


class LMEngine:
    """VidiEngine's text interface over a causal LM: "embeddings" are the token ids themselves, "hidden states" the logits."""

    def __init__(self, lm):
        self.lm, self.dev, self.dtype = lm, torch.device("cpu"), torch.float32
        self.mistral, self.world, self.rank, self.pg, self.normalizer = False, 1, 0, None, 1.0

    def new_text_state(self, B, Lmax):
        return SimpleNamespace(B=B, Lmax=Lmax, ids=torch.zeros((B, 0), dtype=torch.int64), mask=torch.zeros((B, 0), dtype=torch.bool), past_len=0, n_valid=None)

    def embed_tokens(self, ids, normalize=True):
        return ids.reshape(-1, 1).float()

    def text_forward(self, hidden, positions, ts, mm, Lq, new_mask=None, dyn=False):
        ids = hidden.view(ts.B, Lq).long()
        nm = torch.ones((ts.B, Lq), dtype=torch.bool) if new_mask is None else new_mask.bool()
        ts.ids = torch.cat((ts.ids, ids.clamp(min=0)), dim=1)
        ts.mask = torch.cat((ts.mask, nm), dim=1)
        ts.past_len += Lq
        with torch.no_grad():
            pos = (ts.mask.long().cumsum(-1) - 1).clamp(min=0)
            logits = self.lm(input_ids=ts.ids, attention_mask=ts.mask.long(), position_ids=pos).logits[:, -Lq:].float()
        return logits.reshape(ts.B * Lq, -1)

    def logits_argmax(self, h):
        return h, torch.argmax(h, dim=-1)

    def reorder_text_state(self, ts, parents):
        ts.ids, ts.mask = ts.ids[parents], ts.mask[parents]
        if ts.n_valid is not None:
            ts.n_valid = ts.n_valid[parents]


def build(seed):
    torch.manual_seed(seed)
    cfg_lm = transformers.GPT2Config(n_layer=1, n_embd=32, n_head=2, vocab_size=V, n_positions=96, bos_token_id=0, eos_token_id=1)
    lm = transformers.GPT2LMHeadModel(cfg_lm).eval()
    with torch.no_grad():
        lm.lm_head.weight.mul_(10.0)                                   # a peaked LM: EOS tokens and repeats do occur, ties do not
    cfg = tiny(vocab_size=V, eos_token_id=1, pad_token_id=0)
    model = VidiForCausalLM(cfg, {}, dtype=torch.float32, device="cpu", engine=LMEngine(lm))
    return lm, model


def hf(lm, prompt, seed=None, **kw):
    if seed is not None:
        torch.manual_seed(seed)
    with torch.no_grad():
        return lm.generate(inputs_embeds=lm.transformer.wte(prompt), attention_mask=torch.ones_like(prompt), pad_token_id=0, **kw)


def ours(model, prompt, seed=None, **kw):
    if seed is not None:
        torch.manual_seed(seed)
    return model.generate(prompt, pad_token_id=0, **kw)


GREEDY = [dict(max_new_tokens=12), dict(max_new_tokens=12, eos_token_id=[1, 7, 9]), dict(max_new_tokens=10, eos_token_id=5, min_new_tokens=4),
          dict(max_new_tokens=10, eos_token_id=[5, 6], min_length=6), dict(max_new_tokens=12, repetition_penalty=1.4), dict(max_new_tokens=12, no_repeat_ngram_size=2),
          dict(max_new_tokens=10, bad_words_ids=[[3], [4, 8], [11]]), dict(max_new_tokens=10, suppress_tokens=[2, 3, 4, 5, 6, 7]),
          dict(max_length=15), dict(max_length=15, max_new_tokens=4), dict(max_new_tokens=9, repetition_penalty=1.2, no_repeat_ngram_size=3, eos_token_id=[1, 12]),
          dict()]
SAMPLED = [dict(max_new_tokens=10), dict(max_new_tokens=10, temperature=0.7), dict(max_new_tokens=10, top_k=5), dict(max_new_tokens=10, top_k=0, top_p=0.7),
           dict(max_new_tokens=8, temperature=1.6, top_k=12, top_p=0.9), dict(max_new_tokens=8, num_return_sequences=3, top_k=9),
           dict(max_new_tokens=10, top_k=None, temperature=1.3), dict(max_new_tokens=10, repetition_penalty=1.3, top_k=10, eos_token_id=[1, 4])]
BEAMS = [dict(num_beams=2, max_new_tokens=8), dict(num_beams=4, max_new_tokens=8, eos_token_id=[1, 6]), dict(num_beams=3, max_new_tokens=9, length_penalty=2.0, early_stopping=True),
         dict(num_beams=3, max_new_tokens=9, length_penalty=0.0, early_stopping="never", num_return_sequences=2), dict(num_beams=3, max_new_tokens=8, repetition_penalty=1.3),
         dict(num_beams=2, max_new_tokens=8, no_repeat_ngram_size=2, min_new_tokens=3, eos_token_id=5), dict(num_beams=4, max_length=14, num_return_sequences=4),
         dict(num_beams=3, max_new_tokens=8, do_sample=True, top_k=8), dict(num_beams=2, max_new_tokens=8, do_sample=True, temperature=1.4, top_p=0.9, num_return_sequences=2)]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_generate_equals_hf_over_the_argument_space(seed):
    lm, model = build(seed)
    g = torch.Generator().manual_seed(300 + seed)
    bad = []
    for B in (1, 2):
        prompt = torch.randint(2, V, (B, 6), generator=g)
        for kw in GREEDY:
            a, b = hf(lm, prompt, do_sample=False, **kw), ours(model, prompt, do_sample=False, **kw)
            if a.tolist() != b.tolist():
                bad.append(("greedy", B, kw, a.tolist(), b.tolist()))
        for i, kw in enumerate(SAMPLED):
            a, b = hf(lm, prompt, seed=700 + i, do_sample=True, **kw), ours(model, prompt, seed=700 + i, do_sample=True, **kw)
            if a.tolist() != b.tolist():
                bad.append(("sampled", B, kw, a.tolist(), b.tolist()))
        for i, kw in enumerate(BEAMS):
            a = hf(lm, prompt, seed=900 + i, output_scores=True, return_dict_in_generate=True, **kw)
            b = ours(model, prompt, seed=900 + i, output_scores=True, return_dict_in_generate=True, **kw)
            if a.sequences.tolist() != b.sequences.tolist() or not torch.allclose(a.sequences_scores, b.sequences_scores, atol=1e-5, rtol=0):
                bad.append(("beams", B, kw, a.sequences.tolist(), b.sequences.tolist()))
    assert not bad, f"{len(bad)} configurations differ from HF; first: {bad[0]}"


def test_return_dict_scores_equal_hf():
    lm, model = build(5)
    prompt = torch.tensor([[3, 9, 14, 2, 30]])
    for kw in (dict(do_sample=False, max_new_tokens=6, repetition_penalty=1.3), dict(do_sample=True, max_new_tokens=6, top_k=7, temperature=0.8)):
        a = hf(lm, prompt, seed=11, output_scores=True, output_logits=True, return_dict_in_generate=True, **kw)
        b = ours(model, prompt, seed=11, output_scores=True, output_logits=True, return_dict_in_generate=True, **kw)
        assert a.sequences.tolist() == b.sequences.tolist()
        assert len(a.scores) == len(b.scores) and len(a.logits) == len(b.logits)
        for x, y in zip(a.scores, b.scores):
            assert torch.equal(torch.isinf(x), torch.isinf(y))
            assert torch.allclose(torch.nan_to_num(x, neginf=0.0), torch.nan_to_num(y, neginf=0.0), atol=1e-5)
        for x, y in zip(a.logits, b.logits):
            assert torch.allclose(x, y, atol=1e-5)


class _StopOn(transformers.StoppingCriteria):
    """stop a row once it has emitted `tok` twice"""

    def __init__(self, tok):
        self.tok = tok

    def __call__(self, input_ids, scores, **kw):
        return (input_ids == self.tok).sum(-1) >= 2


class _Collect:
    def __init__(self):
        self.calls = []

    def put(self, x):
        self.calls.append(x.tolist())

    def end(self):
        self.calls.append("end")


@pytest.mark.parametrize("seed", [0, 1])
def test_hooks_padding_and_streams_equal_hf(seed):
    """caller-supplied logits processors / stopping criteria (they see the new tokens only), a streamer's call sequence, pad / EOS
    conventions (`pad_token_id=None` -> the first EOS), and a LEFT-padded ragged batch (HF positions from the mask; here every row is decoded
    unpadded against the same prompt) — all against HF"""
    lm, model = build(10 + seed)
    g = torch.Generator().manual_seed(40 + seed)
    prompt = torch.randint(2, V, (1, 7), generator=g)
    LP = transformers.LogitsProcessorList
    bad = []
    cases = [dict(max_new_tokens=10, logits_processor=LP([transformers.SuppressTokensLogitsProcessor([2, 3, 4, 5], device="cpu")])),
             dict(max_new_tokens=12, stopping_criteria=transformers.StoppingCriteriaList([_StopOn(int(prompt[0, 0]))])),
             dict(max_new_tokens=12, logits_processor=LP([transformers.NoRepeatNGramLogitsProcessor(2)]), repetition_penalty=1.2, eos_token_id=[1, 8]),
             dict(max_new_tokens=8, do_sample=True, top_k=6, logits_processor=LP([transformers.TemperatureLogitsWarper(0.5)]))]
    for i, kw in enumerate(cases):
        kw = dict(kw); kw.setdefault("do_sample", False)
        a, b = hf(lm, prompt, seed=60 + i, **kw), ours(model, prompt, seed=60 + i, **kw)
        if a.tolist() != b.tolist():
            bad.append((kw, a.tolist(), b.tolist()))
    # the stopping-criterion case on a batch of two different prompts: rows stop independently, finished rows are padded
    p2 = torch.randint(2, V, (2, 7), generator=g)
    crit = lambda: transformers.StoppingCriteriaList([_StopOn(int(p2[0, 0])), _StopOn(int(p2[1, 1]))])          # noqa: E731
    a, b = hf(lm, p2, do_sample=False, max_new_tokens=14, stopping_criteria=crit()), ours(model, p2, do_sample=False, max_new_tokens=14, stopping_criteria=crit())
    if a.tolist() != b.tolist():
        bad.append(("criteria, batch 2", a.tolist(), b.tolist()))
    # pad conventions: no pad_token_id -> HF pads finished rows with the first EOS
    with torch.no_grad():
        a = lm.generate(inputs_embeds=lm.transformer.wte(p2), attention_mask=torch.ones_like(p2), do_sample=False, max_new_tokens=12, eos_token_id=[int(p2[0, 0]), 1])
    b = model.generate(p2, do_sample=False, max_new_tokens=12, eos_token_id=[int(p2[0, 0]), 1])
    if a.tolist() != b.tolist():
        bad.append(("pad = first EOS", a.tolist(), b.tolist()))
    # streamer: the same sequence of put() / end() calls (B = 1: HF's streamers take one row)
    sa, sb = _Collect(), _Collect()
    hf(lm, prompt, do_sample=False, max_new_tokens=6, streamer=sa)
    ours(model, prompt, do_sample=False, max_new_tokens=6, streamer=sb)
    norm = lambda calls: [c if c == "end" else [t for row in (c if isinstance(c[0], list) else [c]) for t in row] if c else [] for c in calls]      # noqa: E731
    if norm(sa.calls) != norm(sb.calls):
        bad.append(("streamer", sa.calls, sb.calls))
    # left-padded ragged batch
    lens = [7, 4, 6]
    rag = torch.randint(2, V, (3, 7), generator=g)
    am = torch.zeros((3, 7), dtype=torch.long)
    for r, n in enumerate(lens):
        am[r, 7 - n:] = 1
        rag[r, : 7 - n] = 0
    with torch.no_grad():
        pos = (am.cumsum(-1) - 1).clamp(min=0)
        a = lm.generate(inputs_embeds=lm.transformer.wte(rag), attention_mask=am, position_ids=pos, do_sample=False, max_new_tokens=8, pad_token_id=0, eos_token_id=[1, 5])
    b = model.generate(rag, attention_mask=am, do_sample=False, max_new_tokens=8, pad_token_id=0, eos_token_id=[1, 5])
    if a.tolist() != b.tolist():
        bad.append(("left-padded ragged batch", a.tolist(), b.tolist()))
    # ... with `min_length` / `max_length` ABOVE the prompt length: HF resolves both once against the batch's padded prompt (7 positions),
    # not per row (the row-by-row path once subtracted each row's own length a second time)
    with torch.no_grad():
        free = lm.generate(inputs_embeds=lm.transformer.wte(rag), attention_mask=am, position_ids=pos, do_sample=False, max_new_tokens=4, pad_token_id=0)
    first = sorted({int(t) for t in free[:, 0]})                # every row's favourite first token as EOS: only `min_length` keeps a row going
    for kw in (dict(min_length=9, max_new_tokens=8, eos_token_id=first), dict(min_length=12, max_length=18, eos_token_id=first),
               dict(min_length=14, max_new_tokens=12, eos_token_id=first + [int(free[0, 1])])):
        with torch.no_grad():
            a = lm.generate(inputs_embeds=lm.transformer.wte(rag), attention_mask=am, position_ids=pos, do_sample=False, pad_token_id=0, **kw)
        b = model.generate(rag, attention_mask=am, do_sample=False, pad_token_id=0, **kw)
        if a.tolist() != b.tolist():
            bad.append(("left-padded ragged batch, lengths counted on the padded prompt", kw, a.tolist(), b.tolist()))
    assert not bad, f"{len(bad)} differ from HF; first: {bad[0]}"
