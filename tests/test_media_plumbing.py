"""BASELINE `configs[0]` plumbing — `inference.py` on dummy.mp4 up to the tensors `generate()` receives — with the decoders the image
lacks (decord, ffmpeg, ffprobe) replaced by tests/fakes/ (a synthetic clip with dummy.mp4's parameters: 394 frames at 16 fps, 24.625 s).

vidi_amd/processors.py `load_video / load_audio / get_media_length / process_audio` are held to tests/golden/reference_media.json, which
was produced by EXECUTING the reference's own loaders under the same fakes (tests/golden/make_golden_media.py): identical frame indices
(25 frames for the default 1 fps: stride round(16 / 1)), identical decoded frames, PCM floats, log-mel features and `audio_size`, both
variants (Vidi1.5 clamps a `time_range` to the clip, Vidi-7B does not), and the container duration through both ffprobe command lines."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FAKES = os.path.join(HERE, "fakes")
G = json.load(open(os.path.join(HERE, "golden", "reference_media.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture()
def fake_media(tmp_path, monkeypatch):
    """the fakes on sys.path / PATH and a synthetic dummy.mp4; -> path of the clip"""
    monkeypatch.syspath_prepend(FAKES)
    for k in [k for k in sys.modules if k == "decord" or k.startswith("decord.")]:
        monkeypatch.delitem(sys.modules, k)
    import fake_clip
    monkeypatch.setenv("PATH", fake_clip.install_executables(str(tmp_path / "bin")) + os.pathsep + os.environ["PATH"])
    clip = str(tmp_path / "dummy.mp4")
    assert fake_clip.write_clip(clip) == G["clip"]
    yield clip
    sys.modules.pop("decord", None)


def extractor():
    from transformers import WhisperFeatureExtractor
    return WhisperFeatureExtractor(feature_size=128, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400)


@pytest.mark.parametrize("arch", ["vidi15", "vidi7b"])
def test_loaders_reproduce_the_reference_loaders_on_the_synthetic_dummy_clip(arch, fake_media):
    import fake_clip
    from vidi_amd import processors as P
    load_video = P.load_video if arch == "vidi15" else P.load_video_7b
    ref = G[arch]
    assert [c["kw"] for c in ref["video"]] == G["cases"]["video"]
    for case in ref["video"]:
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in case["kw"].items()}
        if "raises" in case:                                           # Vidi-7B: a range that ends past the clip is not clamped -> the decoder raises
            with pytest.raises(Exception):
                load_video(fake_media, **kw)
            continue
        frames = load_video(fake_media, **kw)
        assert [fake_clip.frame_index(f) for f in frames] == case["indices"], kw
        assert len(frames) == case["n"] and frames[0].mode == case["mode"] == "RGB"
        assert sha(np.stack([np.asarray(f) for f in frames])) == case["sha256"]
    # configs[0]: dummy.mp4 at the default 1 fps -> frames 0, 16, ..., 384 (25 frames)
    assert ref["video"][0]["indices"] == list(range(0, 394, 16)) and ref["video"][0]["n"] == 25
    ext = extractor()
    for case in ref["audio"]:
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in case["kw"].items()}
        sr = kw.pop("sample_rate", 16000)
        pcm = P.load_audio(fake_media, sr, **kw)
        assert pcm.dtype == np.float32 and len(pcm) == case["n"] and sha(pcm) == case["sha256"]
        if "audio_size" in case:
            feats, size = P.process_audio(pcm, ext)
            assert size == case["audio_size"] and list(feats.shape) == case["features_shape"]
            assert sha(feats.numpy()) == case["features_sha256"]
    assert ref["audio"][0]["audio_size"] == 2462                       # 394 000 samples // 160 (one 30-s window, cut)
    assert P.get_media_length(fake_media) == G["vidi15"]["media_length"] == G["vidi7b"]["get_length"] == 24.625


def test_own_cli_runs_dummy_clip_end_to_end_without_stubbing_the_loaders(fake_media):
    """vidi_amd/inference.py `ask()` from the clip's PATH to the answer string with NOTHING monkeypatched: decord / ffmpeg / ffprobe are
    the fakes, the model's engine is the CPU oracle behind VidiEngine's interface (no GPU here; tests/test_gpu_cli.py runs the same on the
    HIP engine).  25 frames, audio_size 2462 -> 246 audio tokens reach generate(), and the length in the prompt is ffprobe's."""
    import torch
    from oracle_engine import OracleEngine
    from test_reference_cli import Tok
    from vidi_amd import inference as INF
    from vidi_amd.model import load_pretrained_model
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, _, _, _ = load_pretrained_model("/nonexistent", synthetic="tiny", seed=5, torch_dtype=torch.float32, device="cpu",
                                               engine_factory=lambda cfg, w, dt: OracleEngine(cfg, w))
    cfg = model.config
    from transformers import SiglipImageProcessor, WhisperFeatureExtractor
    S = cfg.vis_image_size
    ip = SiglipImageProcessor(size={"height": S, "width": S}, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    ip.output_size = S
    ap = WhisperFeatureExtractor(feature_size=cfg.aud_num_mel_bins, sampling_rate=16000, hop_length=160, chunk_length=1, n_fft=400)   # 1-s windows (tiny tower)
    seen = {}
    gen = model.generate

    def spy(*a, **k):
        seen.update(images=k["images"], audios=k["audios"], audio_sizes=k["audio_sizes"], ids=a[0])
        return gen(*a, **dict(k, max_new_tokens=12))
    model.generate = spy
    got = INF.ask("a dog running.", fake_media, model, Tok(), ip, ap, device="cpu")
    assert seen["images"].shape[:2] == (1, 25) and seen["audios"].shape[:2] == (1, 25)         # 25 frames; 24.625 s -> 25 one-second windows
    assert seen["audio_sizes"] == [2462]
    assert isinstance(got, str) and len(got) > 0
    assert torch.equal(seen["ids"].cpu(), INF.build_prompt("a dog running.", 24.625, Tok(), "vidi15"))   # (Vidi1.5's prompt does not quote the length; it scales the answer's percentages)
