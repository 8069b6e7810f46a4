"""`generate()`'s HF surface against what the REFERENCE's own `generate()` returned for the same weights, video and arguments (the product
class driven over the CPU oracle engine; goldens written by tests/golden/make_golden_beams.py / make_golden_sampling.py / make_golden_dattn.py
from gemma.py:603-655 / mistral.py:629-681 -> HF `GenerationMixin`): beam search token for token with the sequence scores; `do_sample=True`
draw for draw under the same `torch.manual_seed`; `return_dict_in_generate` / `output_scores`; `max_length`."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

from beam_util import forced_log_probs, hypothesis_score, trim_at_eos  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402

GOLD = json.load(open(os.path.join(HERE, "golden", "reference_beams.json")))
_models = {}


def golden_cfg(arch):
    from vidi_amd.config import tiny, tiny_7b
    if arch == "vidi7b":                                                # make_golden_dattn_7b.golden_config()
        return tiny_7b(num_attention_heads=2, num_key_value_heads=1, head_dim=128, query_pre_attn_scalar=128.0, sliding_window=64)
    return tiny(sliding_window=64)                                      # make_golden_dattn.golden_config()


def golden_model(seed, arch="vidi15"):
    if (arch, seed) not in _models:
        from vidi_amd.model import VidiForCausalLM
        from vidi_amd.weights import init_random_weights
        cfg = golden_cfg(arch)
        w = init_random_weights(cfg, seed=seed, dtype=torch.float32, device="cpu")
        _models[(arch, seed)] = VidiForCausalLM(cfg, w, dtype=torch.float32, device="cpu", engine=OracleEngine(cfg, w))
    return _models[(arch, seed)]


def golden_video(nrow, arch="vidi15"):
    d = np.load(os.path.join(HERE, "golden", "reference_dattn_7b.npz" if arch == "vidi7b" else "reference_dattn.npz"))
    px, mel = torch.from_numpy(d["A_images"]), torch.from_numpy(d["A_audios"])
    return dict(images=px.repeat(nrow, 1, 1, 1, 1), audios=mel.repeat(nrow, 1, 1, 1), audio_sizes=[100] * nrow)


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_beam_search_reproduces_the_reference_generate(case):
    model = golden_model(case["seed"], case["arch"])
    ids = torch.tensor(case["input_ids"], dtype=torch.int64)
    eos = case["eos_token_id"]
    g = model.generate(ids, do_sample=False, use_cache=True, pad_token_id=0, eos_token_id=eos if len(eos) > 1 else eos[0],
                       output_scores=True, return_dict_in_generate=True, **golden_video(len(case["input_ids"]), case["arch"]), **case["kwargs"])
    assert g.sequences.tolist() == case["sequences"]
    np.testing.assert_allclose(g.sequences_scores.numpy(), np.array(case["sequences_scores"]), rtol=0, atol=2e-4)
    # without return_dict_in_generate: the token tensor alone
    t = model.generate(ids, do_sample=False, pad_token_id=0, eos_token_id=eos if len(eos) > 1 else eos[0],
                       **golden_video(len(case["input_ids"]), case["arch"]), **case["kwargs"])
    assert torch.is_tensor(t) and t.tolist() == case["sequences"]


def test_one_beam_is_the_greedy_path_and_the_best_beam_never_scores_below_it():
    """the search with the greedy sequence's log-probability as a floor: beam 1 of num_beams = 3 scores >= greedy's score"""
    case = GOLD["cases"][0]
    model = golden_model(case["seed"])
    ids = torch.tensor(case["input_ids"], dtype=torch.int64)
    mm = model.encode_mm_state(**{k: v for k, v in golden_video(1).items()})
    greedy = model.generate(ids, mm_state=mm, do_sample=False, max_new_tokens=8, pad_token_id=0, eos_token_id=7)
    # greedy's own score, recomputed through the logits-processor hook (sees the fp32 scores of every step)
    seen = []

    def spy(input_ids, scores):
        seen.append(torch.log_softmax(scores.float(), -1))
        return scores

    again = model.generate(ids, mm_state=mm, do_sample=False, max_new_tokens=8, pad_token_id=0, eos_token_id=7, logits_processor=[spy])
    assert again.tolist() == greedy.tolist()
    lp = sum(float(seen[i][0, int(greedy[0, i])]) for i in range(greedy.shape[1])) / greedy.shape[1]
    g = model.generate(ids, mm_state=mm, do_sample=False, num_beams=3, max_new_tokens=8, pad_token_id=0, eos_token_id=7,
                       return_dict_in_generate=True, output_scores=True)
    assert float(g.sequences_scores[0]) >= lp - 1e-5
    assert g.sequences.tolist() == case["sequences"]                   # resident video state: the same answer as from raw frames


def test_beam_search_argument_rules():
    model = golden_model(6)
    ids = torch.tensor(GOLD["cases"][0]["input_ids"], dtype=torch.int64)
    with pytest.raises(ValueError, match="num_return_sequences"):
        model.generate(ids, num_beams=2, num_return_sequences=3, max_new_tokens=2, **golden_video(1))
    with pytest.raises(ValueError, match="streamer"):
        model.generate(ids, num_beams=2, streamer=object(), max_new_tokens=2, **golden_video(1))


@pytest.mark.parametrize("case", [c for c in GOLD["cases"] if len(c["input_ids"]) == 1], ids=lambda c: c["name"])
def test_forced_rescoring_gives_the_reference_score(case):
    """the checker of tests/test_gpu_beams.py, checked: scoring the reference's sequences token by token through the oracle (log-softmax,
    then the kwargs' processors; length penalty over the new tokens) gives the reference's `sequences_scores`"""
    model = golden_model(case["seed"], case["arch"])
    ids = torch.tensor(case["input_ids"], dtype=torch.int64)
    mm = model.encode_mm_state(**golden_video(1, case["arch"]))
    for seq, want in zip(case["sequences"], case["sequences_scores"]):
        seq = trim_at_eos(seq, case["eos_token_id"])
        lps = forced_log_probs(model, ids, mm, seq, case["kwargs"], case["eos_token_id"])
        assert abs(hypothesis_score(lps, seq, float(case["kwargs"].get("length_penalty", 1.0))) - want) < 2e-4


SAMP = json.load(open(os.path.join(HERE, "golden", "reference_sampling.json")))


@pytest.mark.parametrize("case", SAMP["cases"], ids=lambda c: f"{c['arch']}-{'_'.join(f'{k}{v}' for k, v in c['kwargs'].items()) or 'defaults'}")
def test_sampling_replays_the_reference_generate_draw_for_draw(case):
    """`do_sample=True`: the reference's own `generate()` under `torch.manual_seed` (tests/golden/make_golden_sampling.py) against the
    product class under the same seed — same warpers in the same order with HF's defaults where a knob is absent (top_k = 50; an
    explicit `top_k=None` / 0 switches it off), one multinomial draw per step: token for token, EOS stops included"""
    if torch.__version__ != SAMP["torch"]:
        pytest.skip("the draws are those of the torch build that wrote the golden")
    model = golden_model(case["seed"], case["arch"])
    ids = torch.tensor(case["input_ids"], dtype=torch.int64)
    torch.manual_seed(case["torch_seed"])
    got = model.generate(ids, do_sample=True, max_new_tokens=10, pad_token_id=0, **golden_video(1, case["arch"]), **case["kwargs"])
    assert got.tolist() == case["tokens"]


def test_return_dict_scores_and_max_length_like_the_reference():
    """golden case E (make_golden_dattn.py: the reference's greedy `generate(output_scores=True, return_dict_in_generate=True)`): the same
    object shape here — `.sequences`, `.scores` per step; and `max_length` counts the embedded prompt (HF under `inputs_embeds`)"""
    d = np.load(os.path.join(HERE, "golden", "reference_dattn.npz"))
    model = golden_model(6)
    ids = torch.from_numpy(d["E_input_ids"])
    g = model.generate(ids, do_sample=False, max_new_tokens=8, use_cache=True, pad_token_id=0, output_scores=True, output_logits=True,
                       return_dict_in_generate=True, **golden_video(1))
    assert g.sequences.tolist() == d["E_tokens"].tolist()
    assert len(g.scores) == g.sequences.shape[1] == len(g.logits)
    np.testing.assert_allclose(torch.stack(g.scores, dim=1).numpy(), d["E_scores"], rtol=0, atol=2e-4)
    assert all(torch.equal(a, b) for a, b in zip(g.scores, g.logits))          # no processors: the scores are the raw logits
    n_prompt = ids.shape[1] - 1                                                 # the <image> placeholder is not an embedded position
    t = model.generate(ids, do_sample=False, max_length=n_prompt + 5, pad_token_id=0, **golden_video(1))
    assert t.tolist() == [d["E_tokens"][0, :5].tolist()]
    t = model.generate(ids, do_sample=False, max_length=n_prompt + 5, max_new_tokens=3, pad_token_id=0, **golden_video(1))
    assert t.shape[1] == 3                                                      # max_new_tokens wins
    with pytest.raises(ValueError, match="max_length"):
        model.generate(ids, do_sample=False, max_length=n_prompt, **golden_video(1))


@pytest.mark.parametrize("arch", ["vidi15", "vidi7b"])
def test_forward_call_shapes_like_the_reference(arch):
    """`forward()` as a user of the reference can call it (tests/golden/make_golden_forward.py ran the reference's own forward): logits
    for `logits_to_keep` 0 / 1 / 3, one modality alone, a right-padded two-video batch (valid positions; the reference's pad slots are
    unspecified), and `labels` -> the reference's loss AND its flattened fp32 `.logits` (Vidi-7B: no logits with labels, no
    `logits_to_keep`)"""
    import make_golden_forward as MF
    gold = np.load(os.path.join(HERE, "golden", "reference_forward.npz"))
    tag = "7b_" if arch == "vidi7b" else ""
    model = golden_model(6, arch)
    v = golden_video(1, arch)
    seen = 0
    for name, case in MF.cases().items():
        if tag + name + "_logits" not in gold and tag + name + "_loss" not in gold:
            continue
        seen += 1
        kw = MF.call_kwargs(case, v["images"], v["audios"])
        if tag:
            kw.pop("logits_to_keep")
            kw["input_ids"] = torch.where(kw["input_ids"] == 2, torch.ones_like(kw["input_ids"]), kw["input_ids"])
        o = model.forward(**kw)
        if tag + name + "_logits" not in gold:
            assert o.logits is None, name
        else:
            want = torch.from_numpy(gold[tag + name + "_logits"])
            got = o.logits.float()
            assert got.shape == want.shape, name
            if "batch2" in name:                                       # row 1 has 3 embedded positions; beyond them the reference holds pad-slot values
                L = want.shape[-2] // 2 if want.dim() == 2 else want.shape[1]
                got, want = got.reshape(2, L, -1), want.reshape(2, L, -1)
                got, want = torch.cat((got[0], got[1, :3])), torch.cat((want[0], want[1, :3]))
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=5e-6, err_msg=name)
        if tag + name + "_loss" in gold:
            assert abs(float(o.loss) - float(gold[tag + name + "_loss"][0])) < 1e-5, name
        else:
            assert o.loss is None
    assert seen == (2 if tag else 8)


def test_num_return_sequences_without_beams_follows_hf():
    model = golden_model(6)
    ids = torch.tensor(GOLD["cases"][0]["input_ids"], dtype=torch.int64)
    with pytest.raises(ValueError, match="num_return_sequences"):
        model.generate(ids, do_sample=False, num_return_sequences=2, max_new_tokens=2, **golden_video(1))


@pytest.mark.parametrize("arch", ["vidi15", "vidi7b"])
def test_inner_seams_like_the_reference(arch):
    """`encode_videos` and `prepare_inputs_labels_for_multimodal` (SURVEY 8b's inner seam) on a two-video batch with different audio lengths,
    an all-zero frame, a (right-padded, Vidi1.5) prompt batch, labels and position ids: every returned tensor against the reference's own"""
    import make_golden_forward as MF
    gold = np.load(os.path.join(HERE, "golden", "reference_forward.npz"))
    tag = "7b_" if arch == "vidi7b" else ""
    model = golden_model(6, arch)
    s = MF.seam_inputs(model.config, right_pad=not tag)
    ev = model.encode_videos(s["images"], s["audios"], s["audio_sizes"])
    for n, t in zip(("img", "imask", "aud", "amask"), ev):
        want = gold[tag + "seam_encode_" + n]
        assert tuple(t.shape) == want.shape and (t.dtype == torch.bool) == (want.dtype == np.bool_), n
        np.testing.assert_allclose(t.float().numpy(), want.astype(np.float32), rtol=0, atol=2e-6, err_msg=n)
    pr = model.prepare_inputs_labels_for_multimodal(s["input_ids"], s["position_ids"], s["attention_mask"], None, s["labels"], s["images"], None,
                                                    s["audios"], s["audio_sizes"])
    for n, t in zip(MF.SEAM_NAMES, pr):
        if tag + "seam_prepare_" + n not in gold:
            assert t is None, n
            continue
        want = gold[tag + "seam_prepare_" + n]
        assert tuple(t.shape) == want.shape, n
        if n in ("position_ids", "attention_mask", "labels", "image_attention_mask", "audio_attention_mask"):
            if n == "position_ids":                                     # pad slots hold unspecified positions in the reference (zeros): compare attended ones
                m = torch.from_numpy(gold[tag + "seam_prepare_attention_mask"]).bool()
                assert torch.equal(t[m].long(), torch.from_numpy(want)[m].long())
            else:
                assert np.array_equal(t.numpy().astype(want.dtype), want), n
        else:
            np.testing.assert_allclose(t.float().numpy(), want.astype(np.float32), rtol=0, atol=2e-6, err_msg=n)


@pytest.mark.parametrize("kw", [dict(min_p=0.1, do_sample=True), dict(typical_p=0.5, do_sample=True), dict(num_beam_groups=2, num_beams=4, diversity_penalty=0.5),
                                dict(penalty_alpha=0.6, top_k=4), dict(begin_suppress_tokens=[5]), dict(forced_eos_token_id=7), dict(renormalize_logits=True),
                                dict(sequence_bias={(5,): -1.0}), dict(stop_strings=["."])], ids=lambda k: next(iter(k)))
def test_generation_arguments_without_a_counterpart_are_refused_not_ignored(kw):
    model = golden_model(6)
    ids = torch.tensor(GOLD["cases"][0]["input_ids"], dtype=torch.int64)
    with pytest.raises(NotImplementedError, match=next(iter(kw))):
        model.generate(ids, max_new_tokens=2, **golden_video(1), **kw)
    # the neutral values pass
    t = model.generate(ids, max_new_tokens=2, typical_p=1.0, num_beam_groups=1, renormalize_logits=False, min_p=None, **golden_video(1))
    assert t.shape == (1, 2)
