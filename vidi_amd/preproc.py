"""Frame and audio preprocessing on the GPU — the device-side counterpart of `processors.process_images`
('resize' mode) and `processors.process_audio`, for callers that hand over decoded RGB frames / PCM instead of tensors:

    frames uint8 [T,H,W,3]  ->  pixel_values [T,3,S,S]      == PIL resize(BICUBIC) + SiglipImageProcessor, BIT-EXACT
    pcm float32 [n]         ->  input_features [C,128,3000] == WhisperFeatureExtractor (within fp32 FFT noise), + length

The kernels (csrc/preproc.hip) do the per-pixel / per-sample work; this file only builds the small parameter tables they
consume, exactly as the third-party code builds them on the host (double precision, same rounding):
  * Pillow's resampling coefficients (src/libImaging/Resample.c `precompute_coeffs` + `normalize_coeffs_8bpc`)
  * the 256-entry per-channel value table of `rescale` + `normalize` (transformers image_transforms)
  * the Hann-windowed real-DFT matrix and the Slaney mel filter bank (feature_extraction_whisper.py)
Reference call sites replaced: Vidi1.5_9B/vidi/dataset/img_utils.py:181-185, vid_utils.py:53-64.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip

_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def pillow_bicubic_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """(bounds [out,2] int32 = (first source index, taps), kk [out,ksize] int32) of Pillow's antialiased bicubic filter"""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    lo = np.maximum((centers - support + 0.5).astype(np.int64), 0)          # C `(int)` truncation of a non-negative value
    lo = np.where(centers - support + 0.5 < 0, 0, lo)
    hi = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    n = hi - lo
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    w = _bicubic((taps + lo[:, None] - centers[:, None] + 0.5) / fscale)
    w = np.where(taps < n[:, None], w, 0.0)
    tot = np.zeros(out_size)
    for j in range(ksize):                                                   # left-to-right accumulation, as the C loop
        tot = tot + w[:, j]
    w = np.where(tot[:, None] != 0.0, w / np.where(tot == 0.0, 1.0, tot)[:, None], w)
    fixed = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64)   # trunc toward 0
    fixed = np.where(taps < n[:, None], fixed, 0)
    return np.stack([lo, n], axis=1).astype(np.int32), fixed.astype(np.int32)


def normalize_table(mean: Sequence[float], std: Sequence[float], rescale_factor: float, dtype: torch.dtype) -> torch.Tensor:
    """value of every byte after `rescale` (float64 multiply, float32 result) and `normalize` ((x - mean) / std in float32),
    cast to the model dtype the way the caller's `.to(dtype)` does -> [3, 256]"""
    v = (np.arange(256, dtype=np.float64) * rescale_factor).astype(np.float32)
    t = (v[None, :] - np.asarray(mean, dtype=np.float32)[:, None]) / np.asarray(std, dtype=np.float32)[:, None]
    return torch.from_numpy(np.ascontiguousarray(t.astype(np.float32))).to(dtype)


class FramePreprocessor:
    """`process_images(frames, image_processor, model_cfg)` for mm_image_aspect_ratio == "resize" on the GPU."""

    def __init__(self, size: int = 384, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), rescale_factor: float = 1 / 255,
                 dtype: torch.dtype = torch.bfloat16, device: str = "cuda", frames_per_chunk: int = 256):
        hip.load_library()
        self.size, self.dtype, self.dev, self.chunk = size, dtype, torch.device(device), frames_per_chunk
        self.lut = normalize_table(mean, std, rescale_factor, dtype).to(self.dev)
        self._tables: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}

    @classmethod
    def from_image_processor(cls, image_processor, dtype=torch.bfloat16, device="cuda"):
        size = getattr(image_processor, "output_size", None) or image_processor.size["height"]
        return cls(size, image_processor.image_mean, image_processor.image_std, image_processor.rescale_factor, dtype, device)

    def tables(self, n_in: int):
        key = (n_in, self.size)
        if key not in self._tables:
            b, k = pillow_bicubic_tables(n_in, self.size)
            self._tables[key] = (torch.from_numpy(b).to(self.dev), torch.from_numpy(k).to(self.dev))
        return self._tables[key]

    def __call__(self, frames) -> torch.Tensor:
        """frames: uint8 [T,H,W,3] (torch tensor on host or device, or numpy) -> [T,3,S,S] in `dtype` on the device"""
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(frames)
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3
        T, H0, W0, _ = frames.shape
        S = self.size
        out = torch.empty((T, 3, S, S), dtype=self.dtype, device=self.dev)
        bh, kh = self.tables(W0)
        bv, kv = self.tables(H0)
        for t0 in range(0, T, self.chunk):                                   # bounds the uint8 intermediate (chunk*H0*S*3 bytes)
            fr = frames[t0: t0 + self.chunk].to(self.dev, non_blocking=True).contiguous()
            tmp = torch.empty((fr.shape[0], H0, (S * 3 + 3) // 4 * 4), dtype=torch.uint8, device=self.dev)   # dword row pitch
            hip.resize_h_u8(fr, tmp, bh, kh)
            hip.resize_v_u8_norm(tmp, out[t0: t0 + fr.shape[0]], bv, kv, self.lut)
        return out


# ---------------------------------------------------------------------------------------------------------------------
def _slaney_mel_filters(n_freq: int, n_mels: int, sr: int) -> np.ndarray:
    """audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney") -> [n_freq, n_mels] float64"""
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)

    hz = mel2hz(np.linspace(hz2mel(0.0), hz2mel(sr / 2), n_mels + 2))
    freqs = np.linspace(0, sr // 2, n_freq)
    slopes = hz[None, :] - freqs[:, None]
    d = np.diff(hz)
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / d[:-1], slopes[:, 2:] / d[1:]))
    return fb * (2.0 / (hz[2:] - hz[:-2]))[None, :]


class LogMelExtractor:
    """`process_audio(audio, audio_processor)` on the GPU: 30-s windows -> [C, n_mels, 3000] + the reference's `length`."""

    def __init__(self, n_mels: int = 128, n_fft: int = 400, hop: int = 160, n_samples: int = 480000, sampling_rate: int = 16000,
                 mel_filters: Optional[np.ndarray] = None, dtype: torch.dtype = torch.bfloat16, device: str = "cuda"):
        hip.load_library()
        assert n_fft % 16 == 0 and hop % 4 == 0 and n_samples % hop == 0
        self.n_mels, self.n_fft, self.hop, self.n_samples = n_mels, n_fft, hop, n_samples
        self.dtype, self.dev = dtype, torch.device(device)
        self.nf = n_fft // 2 + 1
        self.frames = n_samples // hop                                        # 3000 (the STFT's last column is dropped)
        # rows per window in the frame matrix: stride of the padded waveform / hop, so all windows form ONE row view
        self.R = -(-(n_samples + n_fft) // hop)                                # 3003
        self.stride = self.R * hop
        n = np.arange(n_fft, dtype=np.float64)
        window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)                  # torch.hann_window(periodic=True)
        ang = 2.0 * np.pi * np.outer(np.arange(self.nf, dtype=np.float64), n) / n_fft
        dft = np.concatenate([np.cos(ang) * window[None, :], -np.sin(ang) * window[None, :]], axis=0)   # [2 nf, n_fft]
        self.dft = torch.from_numpy(dft.astype(np.float32)).to(self.dev)
        self.ldy = (2 * self.nf + 3) // 4 * 4
        fb = _slaney_mel_filters(self.nf, n_mels, sampling_rate) if mel_filters is None else np.asarray(mel_filters, dtype=np.float64)
        assert fb.shape == (self.nf, n_mels)
        self.kp = (self.nf + 15) // 16 * 16                                    # K of the mel GEMM, zero padded
        melw = np.zeros((n_mels, self.kp), dtype=np.float32)
        melw[:, : self.nf] = fb.T.astype(np.float32)
        self.melw = torch.from_numpy(melw).to(self.dev)

    @classmethod
    def from_feature_extractor(cls, fe, dtype=torch.bfloat16, device="cuda"):
        return cls(fe.feature_size, fe.n_fft, fe.hop_length, fe.n_samples, fe.sampling_rate, fe.mel_filters, dtype, device)

    def num_frames(self, n: int) -> int:
        """sum over windows of len(window) // hop — `audios.num_frames.sum()` (vid_utils.py:62)"""
        return sum(min(self.n_samples, n - s) // self.hop for s in range(0, n, self.n_samples))

    def __call__(self, audio, windows_per_batch: int = 32) -> Tuple[torch.Tensor, int]:
        if isinstance(audio, np.ndarray):
            audio = torch.from_numpy(audio)
        audio = audio.to(torch.float32).reshape(-1)
        n = int(audio.shape[0])
        C = -(-n // self.n_samples)
        wave = torch.zeros((C, self.n_samples), dtype=torch.float32, device=self.dev)       # zero padding of the last window
        wave.view(-1)[:n] = audio.to(self.dev, non_blocking=True)
        out = torch.empty((C, self.n_mels, self.frames), dtype=self.dtype, device=self.dev)
        for c0 in range(0, C, windows_per_batch):
            cb = min(windows_per_batch, C - c0)
            padded = torch.empty((cb, self.stride), dtype=torch.float32, device=self.dev)
            hip.reflect_pad_f32(wave[c0: c0 + cb], padded, self.n_fft // 2)
            # frame f of window c = padded[c, f*hop : f*hop + n_fft] = row c*R + f of a [M, n_fft] view with ldx = hop over
            # the whole buffer (stride = R*hop); M = every row that fits, which covers rows c*R + f, f < frames, of all windows
            M = (cb * self.stride - self.n_fft) // self.hop + 1
            X = torch.as_strided(padded, (M, self.n_fft), (self.hop, 1))
            Y = hip.gemm_f32(X, self.dft, None)                                             # [M, 2 nf] re | im
            P = torch.empty((M, self.kp), dtype=torch.float32, device=self.dev)
            hip.power_spectrum_f32(Y, P, self.nf)
            mel = hip.gemm_f32(P, self.melw, None)                                          # [M, n_mels]
            cmax = torch.empty((cb,), dtype=torch.float32, device=self.dev)
            hip.logmel_finish(mel, cmax, out[c0: c0 + cb], C=cb, R=self.R, F=self.frames)
        return out, self.num_frames(n)
