"""`generate(num_beams > 1)` on the HIP engine: (1) `engine.reorder_text_state` — a decode step after re-gathering the text K/V rows equals
the same step after a fresh prefill of the re-gathered histories; (2) the search over the kernels' logits against the reference's own
`generate()` outputs (tests/golden/reference_beams.json): the hypothesis found is re-scored by the fp32 CPU oracle (forced decoding) and
must carry the score the HIP path reported, and must be as good as the reference's best within the logit tolerance; where every
decision of the reference's search is separated by more than that tolerance the tokens must be the reference's."""
import json
import os

import numpy as np
import pytest
import torch

from oracle_engine import OracleEngine
from beam_util import forced_log_probs, hypothesis_score, trim_at_eos
from util import logit_tol

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_beams.json")))


def build(seed, dt, arch="vidi15"):
    from types import SimpleNamespace
    from vidi_amd.config import tiny, tiny_7b
    from vidi_amd.engine import VidiEngine
    from vidi_amd.model import VidiForCausalLM
    from vidi_amd.weights import init_random_weights
    cfg = (tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64) if arch == "vidi7b"
           else tiny(sliding_window=64))                                # the golden configs of tests/golden/make_golden_dattn*.py
    w = init_random_weights(cfg, seed=seed, dtype=dt, device="cpu")
    eng = VidiEngine(cfg, dict(w), dtype=dt, device="cuda", free_source=False)
    model = VidiForCausalLM.__new__(VidiForCausalLM)
    model.config, model.dtype, model.device, model.engine = cfg, dt, torch.device("cuda"), eng
    model.generation_config = SimpleNamespace(eos_token_id=cfg.eos_token_id, pad_token_id=0)
    model.model = None
    w32 = {k: v.float() for k, v in w.items()}
    ref = VidiForCausalLM(cfg, w32, dtype=torch.float32, device="cpu", engine=OracleEngine(cfg, w32))
    return cfg, model, ref


def video(nrow, dt, dev, out=None, arch="vidi15"):
    """case A's frames / mel rounded to the model dtype (the oracle gets the same rounded values, as fp32)"""
    d = np.load(os.path.join(HERE, "golden", "reference_dattn_7b.npz" if arch == "vidi7b" else "reference_dattn.npz"))
    px, mel = torch.from_numpy(d["A_images"]).to(dt).to(out or dt), torch.from_numpy(d["A_audios"]).to(dt).to(out or dt)
    return dict(images=px.repeat(nrow, 1, 1, 1, 1).to(dev), audios=mel.repeat(nrow, 1, 1, 1).to(dev), audio_sizes=[100] * nrow)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_reorder_text_state_equals_a_fresh_prefill(dt):
    cfg, model, _ = build(6, dt)
    eng = model.engine
    mm = model.encode_mm_state(**video(1, dt, "cuda"))
    from vidi_amd.model import strip_image_token
    prompts = torch.tensor([[2, 21, -200, 22, 23], [2, 30, -200, 31, 32], [2, 40, -200, 41, 42]], dtype=torch.int64)
    nxt = torch.tensor([50, 60, 70], dtype=torch.int64, device="cuda")
    parents = torch.tensor([2, 0, 0], dtype=torch.int64, device="cuda")
    nxt2 = torch.tensor([51, 61, 61], dtype=torch.int64, device="cuda")

    def run(p, first, reorder):
        ids, mask, pos = strip_image_token(p, None)
        ts, _ = model._prefill(ids, mask, pos, mm, 4)
        outs = []
        for step, tok in enumerate((first, nxt2)):
            if step == 1 and reorder is not None:
                eng.reorder_text_state(ts, reorder)
            emb = eng.embed_tokens(tok)
            posn = ts.n_valid.clone()
            ts.n_valid += 1
            outs.append(eng.logits_argmax(eng.text_forward(emb, posn, ts, mm, Lq=1))[0].float().clone())
        return outs

    got = run(prompts, nxt, parents)[1]
    want = run(prompts[parents.cpu()], nxt[parents], None)[1]
    assert torch.equal(got, want), float((got - want).abs().max())       # the same kernels on the same rows: bit-equal
    assert not torch.equal(got[0], got[1])                                # row 0 continues prompt 2; rows 1 and 2 both continue prompt 0 with the same token
    assert torch.equal(got[1], got[2])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", [c for c in GOLD["cases"] if len(c["input_ids"]) == 1], ids=lambda c: c["name"])
def test_beam_search_on_the_hip_engine(case, dt):
    cfg, model, ref = build(case["seed"], dt, case["arch"])
    ids = torch.tensor(case["input_ids"], dtype=torch.int64)
    eos = case["eos_token_id"]
    calls = []
    inner = model.engine.reorder_text_state
    model.engine.reorder_text_state = lambda ts, parents: (calls.append(1), inner(ts, parents))[1]
    g = model.generate(ids, do_sample=False, pad_token_id=0, eos_token_id=eos if len(eos) > 1 else eos[0], output_scores=True,
                       return_dict_in_generate=True, **video(1, dt, "cuda", arch=case["arch"]), **case["kwargs"])
    seqs, scores = g.sequences.cpu(), g.sequences_scores.cpu()
    assert seqs.shape[0] == len(case["sequences"]) and seqs.shape[1] <= case["kwargs"]["max_new_tokens"]
    mm32 = ref.encode_mm_state(**video(1, dt, "cpu", out=torch.float32, arch=case["arch"]))
    lpen = float(case["kwargs"].get("length_penalty", 1.0))
    for r in range(seqs.shape[0]):
        seq = trim_at_eos(seqs[r].tolist(), eos)                        # an EOS ends the hypothesis; what follows is padding
        n = len(seq)
        lps = forced_log_probs(ref, ids, mm32, seq, case["kwargs"], eos)
        tol = logit_tol(dt, lps[0][torch.isfinite(lps[0])])            # per-step error bound of a log-probability (their spread = the logits')
        want = hypothesis_score(lps, seq, lpen)
        slack = 2 * tol * n / (n ** lpen)
        # the reported score is the hypothesis's own: an error in the cache gather would score another history
        assert abs(float(scores[r]) - want) <= slack, (case["name"], r, float(scores[r]), want, tol)
        # and it is as good as the reference's r-th best
        assert float(scores[r]) >= case["sequences_scores"][r] - slack, (float(scores[r]), case["sequences_scores"][r])
    same = seqs.tolist() == case["sequences"]
    log = os.environ.get("VIDI_TEST_REPORT")
    if log:
        with open(log, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "name": "beam tokens equal the reference's", "same": same,
                                "score": [float(x) for x in scores], "reference_score": case["sequences_scores"], "cache_gathers": len(calls)}) + "\n")
