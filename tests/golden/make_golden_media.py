"""Golden vectors for the media plumbing of BASELINE `configs[0]` (inference.py on dummy.mp4), produced by EXECUTING the reference's own
loaders — `Vidi_7B/model/vid_utils.py:10-64`, `Vidi1.5_9B/vidi/dataset/vid_utils.py:10-80` and `get_length` of `Vidi_7B/inference.py:68-73` —
with the decoders the image lacks replaced by tests/fakes/ (a `decord` package and `ffmpeg` / `ffprobe` executables serving a synthetic clip with
dummy.mp4's parameters: 394 frames at 16 fps, 24.625 s).  Build container only (needs /root/reference):

    python tests/golden/make_golden_media.py        -> tests/golden/reference_media.json

Third-party stand-ins (nothing of the reference is modified): the fakes above, and a `WhisperFeatureExtractor` subclass that accepts
`return_token_timestamps=True` and reports `num_frames = len(chunk) // hop_length` per chunk, as transformers 4.50 (the reference's pin)
does; the installed 5.x dropped the argument.  Stored: the frame indices each call decoded (read back from the frames themselves), a
SHA-256 of the decoded frames, of the PCM floats and of the log-mel features, the lengths / durations / `audio_size`."""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FAKES = os.path.join(ROOT, "tests", "fakes")
OUT = os.path.join(HERE, "reference_media.json")
CASES = {"video": [dict(), dict(fps=2.0), dict(fps=0.5), dict(time_range=(3.2, 11.7)), dict(fps=2.0, time_range=(0.0, 24.0)), dict(time_range=(20.0, 24.6))],
         "audio": [dict(), dict(time_range=(3.2, 11.7)), dict(sample_rate=8000)]}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def whisper_extractor_4_50():
    """WhisperFeatureExtractor with the `return_token_timestamps` / `num_frames` behaviour of transformers 4.50"""
    import torch
    from transformers import WhisperFeatureExtractor

    class Extractor(WhisperFeatureExtractor):
        def __call__(self, raw_speech, *a, return_token_timestamps=None, **k):
            out = super().__call__(raw_speech, *a, **k)
            if return_token_timestamps:
                out["num_frames"] = torch.tensor([len(x) // self.hop_length for x in raw_speech])
            return out
    return Extractor(feature_size=128, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400)


def run(mod, clip, which, with_length):
    import fake_clip
    res = {"video": [], "audio": []}
    for kw in CASES["video"]:
        try:
            frames = mod.load_video(clip, **kw)
            arr = np.stack([np.asarray(f) for f in frames])
            res["video"].append({"kw": kw, "n": len(frames), "indices": [fake_clip.frame_index(f) for f in frames], "mode": frames[0].mode,
                                 "sha256": sha(arr)})
        except Exception as e:                                     # Vidi-7B's loader does not clamp the last index: decord raises past the end
            res["video"].append({"kw": kw, "raises": type(e).__name__})
    ext = whisper_extractor_4_50()
    for kw in CASES["audio"]:
        args = dict(kw)
        sr = args.pop("sample_rate", 16000)
        pcm = mod.load_audio(clip, sr, **args)
        rec = {"kw": kw, "n": int(len(pcm)), "dtype": str(pcm.dtype), "sha256": sha(pcm), "abs_max": float(np.abs(pcm).max())}
        if sr == 16000:
            feats, length = mod.process_audio(pcm, ext)
            rec.update(audio_size=int(length), features_shape=list(feats.shape), features_sha256=sha(feats.numpy()))
        res["audio"].append(rec)
    if with_length:
        res["media_length"] = mod.get_media_length(clip)
    return res


def main():
    sys.path.insert(0, FAKES)
    import fake_clip
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.environ["PATH"] = fake_clip.install_executables(os.path.join(d, "bin")) + os.pathsep + os.environ["PATH"]
        clip = os.path.join(d, "dummy.mp4")
        meta = fake_clip.write_clip(clip)
        out = {"clip": meta, "cases": {k: [dict(c) for c in v] for k, v in CASES.items()}}
        out["vidi15"] = run(load_by_path("ref_vid_utils_15", "/root/reference/Vidi1.5_9B/vidi/dataset/vid_utils.py"), clip, "vidi15", True)
        out["vidi7b"] = run(load_by_path("ref_vid_utils_7b", "/root/reference/Vidi_7B/model/vid_utils.py"), clip, "vidi7b", False)
        # Vidi-7B's duration query lives in its inference.py (`get_length`): the script imported unmodified over the compat shim
        for q in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
            sys.path.insert(0, q)
        from test_reference_cli import REF7B, import_unmodified
        INF = import_unmodified(REF7B, "compat_7b", "model")
        out["vidi7b"]["get_length"] = INF.get_length(clip)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)
    for arch in ("vidi15", "vidi7b"):
        print(arch, [(v.get("n"), v.get("raises")) for v in out[arch]["video"]], [(a["n"], a.get("audio_size")) for a in out[arch]["audio"]],
              out[arch].get("media_length"), out[arch].get("get_length"))


if __name__ == "__main__":
    main()
