"""The reference's CLI EXECUTED against the product's model object (north_star "drops into inference.py unchanged").

`/root/reference/Vidi1.5_9B/vidi/eval/inference.py` and `/root/reference/Vidi_7B/inference.py` are imported UNMODIFIED with
vidi_amd/compat (resp. compat_7b) on sys.path — so every `vidi.*` / `model.*` import resolves to this package — and their `ask()`
runs end to end against `vidi_amd.model.VidiForCausalLM` built by `load_pretrained_model(..., synthetic=...)`.  Only the media
decoders (decord / ffmpeg / ffprobe: `load_video`, `load_audio`, `get_length`) are stubbed.  There is no GPU in the build container,
so the model's engine is the CPU oracle behind VidiEngine's interface (tests/oracle_engine.py): everything `ask()` touches — the model
object protocol, `generate()`'s loop / EOS / padding / kwargs, the processors, the tokenizer protocol, the timestamp formatting — is the
product's or the reference's own code.  The same call sequence runs on the HIP engine in tests/test_gpu_cli.py (the GPU box has no
/root/reference, so there it goes through vidi_amd/inference.py, which this file holds to the reference's `ask()` string for string)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

import vidi_oracle as O
from oracle_engine import OracleEngine, oracle_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF15 = "/root/reference/Vidi1.5_9B/vidi/eval/inference.py"
REF7B = "/root/reference/Vidi_7B/inference.py"


class Tok:
    """small deterministic tokenizer object with the protocol `ask()` uses: __call__().input_ids, apply_chat_template,
    bos_token(_id), pad_token_id, batch_decode(skip_special_tokens=)"""
    bos_token_id, bos_token, pad_token_id = 2, "<bos>", 0

    def __init__(self, mistral=False):
        self.mistral = mistral

    def __call__(self, text):
        r = type("R", (), {})()
        r.input_ids = [2] + [10 + (ord(c) % 50) for c in text]
        return r

    def apply_chat_template(self, messages, tokenize=False):
        if self.mistral:
            return "<bos>" + "".join(f"[INST] {m['content']} [/INST]" if m["role"] == "user" else m["content"] for m in messages)
        return "<bos>" + "".join(f"<start_of_turn>{m['role']}\n{m['content']}<end_of_turn>\n" for m in messages)

    def batch_decode(self, ids, skip_special_tokens=True):
        # every token prints one character of the template "0.dd-0.dd, " (d = a digit taken from the token id and position), so
        # whatever a tiny random model generates decodes to well-formed time ranges (Vidi-7B's `float()` rejects anything else)
        tpl = "0.dd-0.dd, "
        out = []
        for row in ids:
            toks = [int(t) for t in row if not (skip_special_tokens and int(t) in (0, 7))]
            out.append("".join(str((t + i) % 10) if tpl[i % len(tpl)] == "d" else tpl[i % len(tpl)] for i, t in enumerate(toks)))
        return out


def processors(cfg):
    from transformers import SiglipImageProcessor, WhisperFeatureExtractor
    S = cfg.vis_image_size
    ip = SiglipImageProcessor(size={"height": S, "width": S}, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    ip.output_size = S
    # 1-s windows of 100 mel frames: hop 160 @ 16 kHz, matching cfg.aud_nb_max_frames = 100
    ap = WhisperFeatureExtractor(feature_size=cfg.aud_num_mel_bins, sampling_rate=16000, hop_length=160, chunk_length=1, n_fft=400)
    return ip, ap


def media(seed, n_frames=3, seconds=1.6):
    from PIL import Image
    rng = np.random.default_rng(seed)
    frames = [Image.fromarray(rng.integers(0, 256, size=(60, 80, 3), dtype=np.uint8)) for _ in range(n_frames)]
    audio = (rng.standard_normal(int(16000 * seconds)) * 0.1).astype(np.float32)
    return frames, audio


def import_unmodified(path, compat, drop):
    sys.path.insert(0, os.path.join(ROOT, "vidi_amd", compat))
    for k in [k for k in sys.modules if k == drop or k.startswith(drop + ".")]:
        del sys.modules[k]
    spec = importlib.util.spec_from_file_location("ref_inference_" + compat, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_model(preset, seed):
    from vidi_amd.model import load_pretrained_model
    with pytest.warns(UserWarning, match="processors are None"):
        model, tok, ip, ap = load_pretrained_model("/nonexistent", synthetic=preset, seed=seed, torch_dtype=torch.float32, device="cpu",
                                                   engine_factory=lambda cfg, w, dt: OracleEngine(cfg, w))
    assert tok is None and ip is None and ap is None
    return model


@pytest.mark.parametrize("arch,path,compat,drop,preset", [("vidi15", REF15, "compat", "vidi", "tiny"), ("vidi7b", REF7B, "compat_7b", "model", "tiny_7b")])
def test_reference_ask_runs_unmodified_against_the_product_model(arch, path, compat, drop, preset, monkeypatch):
    if not os.path.exists(path):
        pytest.skip("needs the reference checkout (build container only)")
    from vidi_amd import inference as OURS
    from vidi_amd.weights import init_random_weights
    from vidi_amd import config as C
    INF = import_unmodified(path, compat, drop)
    try:
        cfg = getattr(C, preset)()
        model = build_model(preset, seed=5)
        model.config.mm_splits = 32                                    # inference.py:87
        tok = Tok(mistral=arch == "vidi7b")
        ip, ap = processors(cfg)
        frames, audio = media(11)
        length = 3723.4
        monkeypatch.setattr(INF, "load_video", lambda p: frames)
        monkeypatch.setattr(INF, "load_audio", lambda p, sr: audio)
        monkeypatch.setattr(INF, "get_length", lambda p: length)
        monkeypatch.setattr(INF.os.path, "exists", lambda p: True)
        monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)      # `.half().cuda()` in ask(): no GPU here
        calls = {}
        gen = model.generate
        def spy(*a, **k):
            calls["kwargs"] = sorted(k)
            out = gen(*a, **k)
            calls["out"] = out
            return out
        monkeypatch.setattr(model, "generate", spy)
        got = INF.ask("a dog running.", "video.mp4", model, tok, ip, ap)          # <- the reference's own function, unmodified

        # expected: the oracle's own greedy loop on the tensors ask() built, decoded and formatted by the oracle's restatement
        video = OURS.process_images(frames, ip, model.config).unsqueeze(0).half().float()
        mel, audio_size = OURS.process_audio(audio, ap)
        mel = mel.unsqueeze(0).half().float()
        ids = OURS.build_prompt("a dog running.", length, tok, arch)
        w32 = {k: v.float() for k, v in init_random_weights(cfg, seed=5, dtype=torch.float32, device="cpu").items()}
        ref_ids = O.generate_greedy(ids, [video[0]], [mel[0]], [audio_size], w32, oracle_config(cfg), 1024)
        n = ref_ids.shape[1]
        assert calls["out"].shape[1] in (n, n - 1) or calls["out"].shape[1] == n          # new tokens only, stops at EOS
        assert calls["out"][0, :n].tolist() == ref_ids[0, : calls["out"].shape[1]].tolist()
        text = tok.batch_decode(ref_ids)[0].strip()
        assert len(text) > 0
        expected = OURS.format_time_ranges(text, length, arch)
        assert got == expected
        if arch == "vidi15":
            assert got == O.format_time_ranges(text, length)
            assert "disable_compile" in calls["kwargs"]
        assert {"audio_sizes", "audios", "do_sample", "images", "max_new_tokens", "pad_token_id", "use_cache"} <= set(calls["kwargs"])
        # and our own CLI (what the GPU box runs) returns the reference's string for the same inputs
        monkeypatch.setattr(OURS, "load_video", lambda p: frames)
        monkeypatch.setattr(OURS, "load_audio", lambda p, sr: audio)
        monkeypatch.setattr(OURS, "get_media_length", lambda p: length)
        monkeypatch.setattr(OURS.os.path, "exists", lambda p: True)
        assert OURS.ask("a dog running.", "video.mp4", model, tok, ip, ap, arch=arch, device="cpu") == got
        assert len(got) > 0, "the synthetic answer should contain at least one time range"
    finally:
        sys.path.pop(0)
        for k in [k for k in sys.modules if k == drop or k.startswith(drop + ".")]:
            del sys.modules[k]


def test_format_time_ranges_matches_reference_golden():
    """our CLI's post-processing against strings the reference's own ask() produced (tests/golden/reference_host.json)"""
    import json
    from vidi_amd import inference as OURS
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_host.json")))
    for row in G["ask"]:
        assert OURS.format_time_ranges(row["answer"], row["length"]) == row["result"]
        ids = OURS.build_prompt(row["question"], row["length"], Tok())
        assert ids.tolist() == row["input_ids"]


def test_generate_eos_list_and_pad(monkeypatch):
    """HF-style eos list ([1, 107] in Gemma2's generation_config) and pad filling in the product's generate()"""
    model = build_model("tiny", seed=3)
    frames, audio = media(12)
    from vidi_amd import inference as OURS
    cfg = model.config
    ip, ap = processors(cfg)
    video = OURS.process_images(frames, ip, cfg).unsqueeze(0)
    mel, audio_size = OURS.process_audio(audio, ap)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24]])
    base = model.generate(ids, images=video, audios=mel.unsqueeze(0), audio_sizes=[audio_size], max_new_tokens=12, eos_token_id=999999)
    assert base.shape == (1, 12)
    stop = int(base[0, 4])
    first = base[0].tolist().index(stop)
    out = model.generate(ids, images=video, audios=mel.unsqueeze(0), audio_sizes=[audio_size], max_new_tokens=12, eos_token_id=[999999, stop])
    assert out[0].tolist() == base[0, : first + 1].tolist()                        # stops right after the first listed eos


def test_generate_hooks_logits_processor_stopping_criteria_streamer():
    """HF generation hooks on the product's generate(): a logits processor that bans the greedy token changes the output from that step
    on, a stopping criterion ends the rows it flags, a streamer receives every step's tokens; with transformers' own
    LogitsProcessorList / StoppingCriteriaList objects (the types a caller of the reference model would pass)."""
    from transformers import LogitsProcessorList, MaxLengthCriteria, StoppingCriteriaList
    from transformers.generation.logits_process import SuppressTokensLogitsProcessor
    model = build_model("tiny", seed=3)
    frames, audio = media(12)
    from vidi_amd import inference as OURS
    cfg = model.config
    ip, ap = processors(cfg)
    video = OURS.process_images(frames, ip, cfg).unsqueeze(0)
    mel, audio_size = OURS.process_audio(audio, ap)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24]])
    kw = dict(images=video, audios=mel.unsqueeze(0), audio_sizes=[audio_size], max_new_tokens=8, eos_token_id=999999)
    base = model.generate(ids, **kw)
    assert base.shape == (1, 8)
    banned = int(base[0, 2])
    procs = LogitsProcessorList([SuppressTokensLogitsProcessor([banned], device="cpu")])
    out = model.generate(ids, logits_processor=procs, **kw)
    assert banned not in out[0].tolist()
    first = base[0].tolist().index(banned)
    assert out[0, :first].tolist() == base[0, :first].tolist() and int(out[0, first]) != banned
    # HF's MaxLengthCriteria sees the NEW tokens only (the reference generates from inputs_embeds): stop after 3 of them
    out = model.generate(ids, stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=3)]), **kw)
    assert out.tolist() == base[:, :3].tolist()

    class Collect:
        def __init__(self):
            self.toks, self.ended = [], False

        def put(self, t):
            self.toks.append(t.tolist())

        def end(self):
            self.ended = True

    st = Collect()
    out = model.generate(ids, streamer=st, **kw)
    # HF order: the (empty, inputs_embeds-driven) prompt first, then one call per new token
    assert st.ended and st.toks[0] == [[]] and [t[0] for t in st.toks[1:]] == out[0].tolist() == base[0].tolist()

    # transformers' own TextStreamer(skip_prompt=True) drops the FIRST put (the prompt): every generated token must still be printed
    from transformers import TextStreamer

    class CharTok:
        def decode(self, toks, **kw):
            return "".join(chr(97 + int(t) % 26) + " " for t in toks)

    class Capture(TextStreamer):
        def __init__(self):
            super().__init__(CharTok(), skip_prompt=True)
            self.text = ""

        def on_finalized_text(self, text, stream_end=False):
            self.text += text

    cap = Capture()
    out = model.generate(ids, streamer=cap, **kw)
    assert cap.text.split() == [chr(97 + int(t) % 26) for t in out[0].tolist()]


def test_batch_of_videos_shares_the_token_budget_like_the_reference():
    """multimodal.py:157-180: the reference concatenates the frames of ALL samples of a batch before the token-budget rule, so two videos
    that are each under the budget are pooled down when their SUM is over it.  The product answers a batch row by row (rows never
    interact) but must hand every row the batch's frame count: `generate()` / `encode_videos()` on a 3 + 4-frame batch whose 7 frames
    cross the (lowered) budget against the oracle's own batched evaluation — and against the per-video evaluation, which differs."""
    import dataclasses
    from vidi_amd.model import VidiForCausalLM
    from vidi_amd.config import tiny
    from vidi_amd.weights import init_random_weights
    cfg = tiny(mm_max_tokens_base=75)                      # budget 300 tokens: 4 frames x 64 = 256 stay, 7 x 64 = 448 take the resize branch
    w = init_random_weights(cfg, seed=4, dtype=torch.float32, device="cpu")
    eng = OracleEngine(cfg, w)
    model = VidiForCausalLM(cfg, w, dtype=torch.float32, device="cpu", engine=eng)
    g = torch.Generator().manual_seed(9)
    S = cfg.vis_image_size
    vids = [(torch.randn((n, 3, S, S), generator=g) * 0.5).clamp(-1, 1) for n in (3, 4)]
    mels = [torch.randn((1, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), generator=g) * 0.3 for _ in range(2)]
    sizes = [100, 83]
    w32 = {k: v.float() for k, v in w.items()}
    ocfg = oracle_config(cfg)
    fb, mb = O.encode_video_images(vids, w32, ocfg)                                    # the reference's batch semantics
    f_alone, _ = O.encode_video_images(vids[:1], w32, ocfg)
    # (tiny dims: the rule's floor of 10 x 10 gives 25 tokens per frame inside the batch against 4 x 4 = 16 for the video alone)
    assert fb.shape[1] // 4 != f_alone.shape[1] // 3, "the batch must be pooled differently from a video on its own for this test to bite"
    fi, mi, fa, ma = model.encode_videos(vids, mels, sizes)
    assert fi.shape == fb.shape and torch.equal(mi, mb)
    assert torch.allclose(fi.float(), fb, rtol=1e-5, atol=1e-6)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24], [2, 31, -200, 32, 33, 34]])
    ref = O.generate_greedy(ids, vids, mels, sizes, w32, ocfg, 5)
    out = model.generate(ids, images=vids, audios=mels, audio_sizes=sizes, max_new_tokens=5, eos_token_id=999999)
    assert out.tolist() == ref.tolist()
    logits = model.forward(ids, images=vids, audios=mels, audio_sizes=sizes, logits_to_keep=1).logits if hasattr(eng, "lm_head") else None
    assert logits is None or logits.shape[0] == 2


def test_generate_plain_generation_kwargs_like_hf():
    """`repetition_penalty`, `min_new_tokens`, `no_repeat_ngram_size`, `suppress_tokens` as PLAIN keyword arguments — the reference hands its
    `**kwargs` to HF's `generate()`, which builds the processors itself (gemma.py:646-655): same tokens as passing transformers' processor
    objects explicitly, EOS held back for `min_new_tokens` steps, no repeated bigram, suppressed tokens never emitted."""
    from transformers import LogitsProcessorList
    from transformers.generation.logits_process import (MinNewTokensLengthLogitsProcessor, NoRepeatNGramLogitsProcessor,
                                                        RepetitionPenaltyLogitsProcessor, SuppressTokensLogitsProcessor)
    model = build_model("tiny", seed=3)
    frames, audio = media(12)
    from vidi_amd import inference as OURS
    cfg = model.config
    ip, ap = processors(cfg)
    video = OURS.process_images(frames, ip, cfg).unsqueeze(0)
    mel, audio_size = OURS.process_audio(audio, ap)
    ids = torch.tensor([[2, 21, 22, -200, 23, 24]])
    kw = dict(images=video, audios=mel.unsqueeze(0), audio_sizes=[audio_size], max_new_tokens=10)
    base = model.generate(ids, eos_token_id=999999, **kw)[0].tolist()
    assert len(base) == 10 and len(set(base)) < 10, "the tiny random model repeats tokens: the penalties below have something to act on"
    # repetition_penalty == the explicit processor
    a = model.generate(ids, eos_token_id=999999, repetition_penalty=1.8, **kw)
    b = model.generate(ids, eos_token_id=999999, logits_processor=LogitsProcessorList([RepetitionPenaltyLogitsProcessor(1.8)]), **kw)
    assert a.tolist() == b.tolist() and a[0].tolist() != base
    # no_repeat_ngram_size = 2: no bigram twice
    c = model.generate(ids, eos_token_id=999999, no_repeat_ngram_size=2, **kw)[0].tolist()
    grams = list(zip(c, c[1:]))
    assert len(grams) == len(set(grams))
    assert c == model.generate(ids, eos_token_id=999999, logits_processor=LogitsProcessorList([NoRepeatNGramLogitsProcessor(2)]), **kw)[0].tolist()
    # min_new_tokens holds the EOS back: with the greedy second token as EOS the plain run stops after 2 tokens, min_new_tokens = 5 after >= 5
    eos = base[1]
    short = model.generate(ids, eos_token_id=eos, **kw)[0].tolist()
    assert short == base[: base.index(eos) + 1]
    d = model.generate(ids, eos_token_id=eos, min_new_tokens=5, **kw)[0].tolist()
    e = model.generate(ids, eos_token_id=eos, logits_processor=LogitsProcessorList([MinNewTokensLengthLogitsProcessor(0, 5, [eos], device="cpu")]), **kw)[0].tolist()
    assert d == e and len(d) >= 5 and eos not in d[:5]
    # suppress_tokens
    f = model.generate(ids, eos_token_id=999999, suppress_tokens=[base[0], base[2]], **kw)[0].tolist()
    assert base[0] not in f and base[2] not in f
    assert f == model.generate(ids, eos_token_id=999999, logits_processor=LogitsProcessorList([SuppressTokensLogitsProcessor([base[0], base[2]], device="cpu")]), **kw)[0].tolist()
