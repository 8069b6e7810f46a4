"""Time the GPU preprocessing kernels against their HBM roofline and against the reference's host path (PIL + HF) on a sample.
    python tools/bench_preproc.py [--frames 3600] [--hw 480 854]"""
import argparse
import json
import time

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3600)
    ap.add_argument("--hw", type=int, nargs=2, default=[480, 854])
    ap.add_argument("--minutes", type=int, default=60)
    a = ap.parse_args()
    from vidi_amd import hip
    from vidi_amd.preproc import FramePreprocessor, LogMelExtractor
    H0, W0 = a.hw
    T = a.frames
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    frames = torch.randint(0, 256, (T, H0, W0, 3), dtype=torch.uint8, device=dev, generator=g)
    pre = FramePreprocessor(384, dtype=torch.bfloat16, frames_per_chunk=512)
    pre(frames[:8]); torch.cuda.synchronize()
    timer = hip.KernelTimer(); hip.TIMER = timer
    t0 = time.perf_counter(); out = pre(frames); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    hip.TIMER = None
    fam = timer.summary()
    h, v = fam["resize_h"], fam["resize_v_norm"]
    res = {"frames": T, "hw": [H0, W0], "wall_s": t_all,
           "resize_h": {"ms": h["ms"], "launches": h["launches"], "GB/s": h["work"] / h["ms"] / 1e6},
           "resize_v_norm": {"ms": v["ms"], "launches": v["launches"], "GB/s": v["work"] / v["ms"] / 1e6},
           "frames_per_s_gpu": T / ((h["ms"] + v["ms"]) * 1e-3)}
    # host reference on a sample: PIL resize + SiglipImageProcessor (what img_utils.py:181-185 costs per frame)
    from PIL import Image
    from transformers import SiglipImageProcessor
    proc = SiglipImageProcessor(size={"height": 384, "width": 384}, image_mean=[0.5] * 3, image_std=[0.5] * 3)
    sample = frames[:32].cpu().numpy()
    t0 = time.perf_counter()
    for f in sample:
        proc.preprocess(Image.fromarray(f).resize((384, 384), resample=Image.BICUBIC), return_tensors="pt")
    res["frames_per_s_cpu_1core"] = len(sample) / (time.perf_counter() - t0)
    # audio
    n = 16000 * 60 * a.minutes
    audio = (torch.randn(n, generator=torch.Generator().manual_seed(1)) * 0.1).numpy()
    ext = LogMelExtractor(dtype=torch.bfloat16)
    ext(audio[: 480000 * 2]); torch.cuda.synchronize()
    t0 = time.perf_counter(); mel, length = ext(audio); torch.cuda.synchronize(); t_mel = time.perf_counter() - t0
    res["logmel"] = {"windows": int(mel.shape[0]), "wall_ms": t_mel * 1e3, "audio_seconds_per_s": n / 16000 / t_mel}
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=128)
    chunks = [audio[i: i + fe.n_samples] for i in range(0, 480000 * 4, fe.n_samples)]
    t0 = time.perf_counter(); fe(chunks, sampling_rate=16000, return_tensors="pt"); t_cpu = time.perf_counter() - t0
    res["logmel"]["cpu_audio_seconds_per_s"] = 120.0 / t_cpu
    print(json.dumps(res))


if __name__ == "__main__":
    main()
