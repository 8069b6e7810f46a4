"""Synthetic media behind stand-ins for the decoders the image lacks (TEST INFRASTRUCTURE).

BASELINE `configs[0]` is "inference.py on dummy.mp4": the reference decodes it with decord (frames) and the ffmpeg / ffprobe executables
(PCM, duration) — none of which exist in the build container or on the GPU box (no network to install them).  What CAN be executed and
compared is everything around the decoders: which frame indices are asked for (stride `round(avg_fps / fps)`, the `time_range` linspace),
the ffmpeg / ffprobe command lines and the parsing of their output (s16le -> float32 / 32768, the duration float), `process_audio`'s
30-s chunking and `audio_size`.  So the decoders are replaced by fakes that serve a SYNTHETIC clip with dummy.mp4's parameters
(394 frames at ~16 fps, 24.625 s):

    tests/fakes/decord/           a `decord` package: VideoReader(len, get_avg_fps, get_batch(idx).asnumpy()), cpu()
    tests/fakes/ffmpeg_fake.py    an `ffmpeg` executable: -i FILE [-ss S -t T] -f s16le -ac 1 -acodec pcm_s16le -ar RATE -  -> PCM on stdout
    tests/fakes/ffprobe_fake.py   an `ffprobe` executable: -show_entries format=duration in both output forms the two CLIs use

A clip is a small JSON file with a media extension; frame i and PCM sample n are pure functions of (seed, i) / (seed, n), so every
process that opens the file sees the same video.  Frame i carries its own index in pixel (0, 0) = (i >> 8, i & 255, 77)."""
import json
import os

import numpy as np

DUMMY = {"vidi_fake_clip": 1, "frames": 394, "fps": 16.0, "height": 120, "width": 160, "duration": 24.625, "seed": 2025}


def write_clip(path: str, **over) -> dict:
    meta = dict(DUMMY, **over)
    with open(path, "w") as f:
        json.dump(meta, f)
    return meta


def read_clip(path: str) -> dict:
    with open(str(path)) as f:
        meta = json.load(f)
    if meta.get("vidi_fake_clip") != 1:
        raise ValueError(f"{path}: not a synthetic clip")
    return meta


def frame(meta: dict, i: int) -> np.ndarray:
    """decoded RGB frame i, uint8 [H, W, 3]"""
    if not 0 <= i < meta["frames"]:
        raise IndexError(f"Out of bound indices: {i}")                 # decord raises DECORDError for these; any exception ends the call
    rng = np.random.default_rng(meta["seed"] * 100003 + i)
    h, w = meta["height"], meta["width"]
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    img[0, 0] = (i >> 8, i & 255, 77)
    return img


def frame_index(img) -> int:
    px = np.asarray(img)[0, 0]
    assert int(px[2]) == 77
    return (int(px[0]) << 8) | int(px[1])


def pcm(meta: dict, rate: int, start: float = 0.0, dur: float = None) -> np.ndarray:
    """int16 mono samples [start, start + dur) seconds at `rate`: a function of the absolute sample index"""
    total = int(round(meta["duration"] * rate))
    n0 = min(total, int(round(start * rate)))
    n1 = total if dur is None else min(total, n0 + int(round(dur * rate)))
    n = np.arange(n0, n1, dtype=np.int64)
    t = n / float(rate)
    x = 0.35 * np.sin(2 * np.pi * (220.0 + 30.0 * t) * t) + 0.1 * np.sin(2 * np.pi * 1234.5 * t + meta["seed"])
    h = (n * 2654435761 + meta["seed"]) % 4093                                  # cheap per-sample hash noise
    x = x + (h / 4093.0 - 0.5) * 0.05
    return np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2")


def install_executables(bindir: str) -> str:
    """`ffmpeg` / `ffprobe` launchers in `bindir` (created at test time: file modes do not have to survive the snapshot); -> bindir"""
    import stat
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    os.makedirs(bindir, exist_ok=True)
    for name in ("ffmpeg", "ffprobe"):
        p = os.path.join(bindir, name)
        with open(p, "w") as f:
            f.write(f"#!/bin/sh\nexec {sys.executable} {os.path.join(here, name + '_fake.py')} \"$@\"\n")
        os.chmod(p, os.stat(p).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    return bindir
