"""CPU oracle for the host preprocessing that feeds the Vidi hot path.  TEST INFRASTRUCTURE ONLY:
only tests/ import it; the product path (vidi_amd/preproc.py + csrc/preproc.hip) never does.

What the reference runs per video on the host CPU (Vidi1.5_9B/vidi/dataset/):
  * frames:  `image.resize((384, 384), resample=Image.BICUBIC)` (img_utils.py:181-184) then
             `image_processor.preprocess(image)` = SigLIP rescale 1/255 + normalise mean 0.5 / std 0.5, CHW float32
             (img_utils.py:185), cast to the model dtype by the caller (eval/inference.py `.to(dtype)`)
  * audio:   `WhisperFeatureExtractor(chunks, return_token_timestamps=True)` (vid_utils.py:53-64): 30-s windows padded with
             zeros, log-mel spectrogram, `length = sum(len(chunk) // hop_length)`

The arithmetic lives in third-party packages that are not under /root/reference, restated here from their published
algorithms:
  * Pillow (requirements.txt: `pillow`, unpinned; 12.2.0 installed) — `ImagingResample` (src/libImaging/Resample.c): two
    passes (horizontal, then vertical) over 8-bit channels with fixed-point coefficients, PRECISION_BITS = 32 - 8 - 2,
    antialiasing filter support scaled by the downscale factor, bicubic a = -0.5;
  * transformers==4.50.0 — `image_transforms.rescale/normalize` (float64 multiply, cast to float32, then (x - mean) / std
    in float32) and `WhisperFeatureExtractor._torch_extract_fbank_features` (torch.stft with a periodic Hann window,
    center/reflect padding, power spectrum without the last frame, Slaney mel filter bank, log10 clamp 1e-10, floor at
    (per-window max - 8), (x + 4) / 4).
Pinned (tests/test_preproc_oracle.py): bit-exact against `PIL.Image.resize` itself on seeded images (down/up-scaling, odd
sizes), bit-exact against the installed SiglipImageProcessor, and within 1e-4 of the installed WhisperFeatureExtractor
(it computes in float32 via FFT, this file in float64 via the DFT definition).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c


def bicubic_filter(x: float) -> float:
    """Resample.c `bicubic_filter`, a = -0.5, support 2."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int, support: float = 2.0, filt=bicubic_filter) -> Tuple[int, np.ndarray, np.ndarray]:
    """Resample.c `precompute_coeffs` (full-image box) + `normalize_coeffs_8bpc`.
    Returns (ksize, bounds[out,2] = (xmin, n), kk[out,ksize] int32 fixed point)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - sup + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(n):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, n)
    return ksize, bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_bicubic_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """`Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)` for img uint8 [H, W, C]; integer arithmetic throughout."""
    H, W, C = img.shape
    cur = img
    if out_w != W:                                                      # horizontal pass first (Resample.c: ImagingResample)
        _, b, kk = precompute_coeffs(W, out_w)
        tmp = np.empty((H, out_w, C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for xx in range(out_w):
            x0, n = b[xx]
            acc = (src[:, x0:x0 + n, :] * kk[xx, :n].astype(np.int64)[None, :, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    if out_h != H:
        _, b, kk = precompute_coeffs(H, out_h)
        out = np.empty((out_h, cur.shape[1], C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for yy in range(out_h):
            y0, n = b[yy]
            acc = (src[y0:y0 + n] * kk[yy, :n].astype(np.int64)[:, None, None]).sum(axis=0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur


def siglip_rescale_normalize(img_u8: np.ndarray, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), rescale_factor: float = 1 / 255) -> np.ndarray:
    """transformers 4.50 `image_transforms.rescale` + `normalize` + to channels-first: uint8 [H,W,3] -> float32 [3,H,W]."""
    x = (img_u8.astype(np.float64) * rescale_factor).astype(np.float32)
    x = (x - np.array(mean, dtype=np.float32)) / np.array(std, dtype=np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def process_frame(img_u8: np.ndarray, size: int = 384) -> np.ndarray:
    """img_utils.py:181-185 ('resize' aspect mode) for one RGB frame."""
    return siglip_rescale_normalize(pil_resize_bicubic_u8(img_u8, size, size))


# ---------------------------------------------------------------------------------------------------------------------
# Whisper log-mel
# ---------------------------------------------------------------------------------------------------------------------
def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), 200.0 * m / 3.0)


def mel_filter_bank(n_freq: int = 201, n_mels: int = 128, fmin: float = 0.0, fmax: float = 8000.0, sr: int = 16000) -> np.ndarray:
    """transformers `audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` -> [n_freq, n_mels] float64."""
    mel_pts = np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2)
    hz_pts = mel_to_hz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    fdiff = np.diff(hz_pts)
    slopes = hz_pts[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    return fb * enorm[None, :]


def whisper_logmel_window(chunk: np.ndarray, n_fft: int = 400, hop: int = 160, n_samples: int = 480000,
                          mel_filters: np.ndarray = None) -> np.ndarray:
    """one 30-s window (zero-padded to n_samples) -> log-mel [n_mels, n_samples // hop] float64"""
    if mel_filters is None:
        mel_filters = mel_filter_bank()
    x = np.zeros(n_samples, dtype=np.float64)
    x[: len(chunk)] = chunk.astype(np.float32)                           # the reference feeds float32 PCM
    xp = np.pad(x, n_fft // 2, mode="reflect")                           # torch.stft(center=True, pad_mode="reflect")
    n = np.arange(n_fft)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)                 # torch.hann_window(periodic=True)
    n_frames = n_samples // hop                                          # stft has n_frames + 1 columns; the last is dropped
    idx = np.arange(n_frames)[:, None] * hop + n[None, :]
    frames = xp[idx] * window[None, :]
    k = np.arange(n_fft // 2 + 1)
    ang = 2.0 * np.pi * np.outer(n, k) / n_fft
    re = frames @ np.cos(ang)
    im = -(frames @ np.sin(ang))
    power = re * re + im * im                                            # [frames, 201]
    mel = power @ mel_filters                                            # [frames, n_mels]
    log_spec = np.log10(np.maximum(mel, 1e-10))
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return ((log_spec + 4.0) / 4.0).T


def process_audio(audio: np.ndarray, n_samples: int = 480000, hop: int = 160, n_fft: int = 400, mel_filters=None) -> Tuple[np.ndarray, int]:
    """vid_utils.py:53-64: -> (input_features [C, n_mels, 3000] float32, length = sum(len(chunk) // hop))"""
    chunks: List[np.ndarray] = [audio[i: i + n_samples] for i in range(0, len(audio), n_samples)]
    feats = np.stack([whisper_logmel_window(c, n_fft, hop, n_samples, mel_filters) for c in chunks]).astype(np.float32)
    length = int(sum(len(c) // hop for c in chunks))
    return feats, length
