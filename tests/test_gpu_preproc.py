"""GPU preprocessing (csrc/preproc.hip via vidi_amd/preproc.py) against the oracle AND against the third-party code the
reference actually calls (PIL.Image.resize + SiglipImageProcessor; WhisperFeatureExtractor).  Frames: bit-exact.
Audio: |err| <= 2e-4 on the O(1) log-mel scale (fp32 DFT vs the oracle's float64; HF's own fp32 FFT differs from the
oracle by 5e-5), `length` bit-exact."""
import numpy as np
import pytest
import torch

import preproc_oracle as P
from test_preproc_oracle import rand_img, synth_audio

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(480, 854), (360, 640), (384, 384), (200, 300), (1080, 1920), (97, 131)])
def test_frames_bit_exact_vs_oracle_f32(hw):
    from vidi_amd.preproc import FramePreprocessor
    T = 3
    frames = np.stack([rand_img(*hw, seed=11 * i + hw[0]) for i in range(T)])
    pre = FramePreprocessor(384, dtype=torch.float32, frames_per_chunk=2)            # chunking exercised (2 + 1)
    got = pre(frames).cpu().numpy()
    ref = np.stack([P.process_frame(f) for f in frames])
    assert got.shape == ref.shape == (T, 3, 384, 384)
    assert np.array_equal(got, ref), f"{np.count_nonzero(got != ref)} of {ref.size} values differ, max {np.abs(got - ref).max()}"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_frames_bit_exact_vs_pil_and_hf(dt):
    """the reference's own call sequence: PIL resize -> SiglipImageProcessor.preprocess -> .to(dtype)"""
    from PIL import Image
    from transformers import SiglipImageProcessor
    from vidi_amd.preproc import FramePreprocessor
    proc = SiglipImageProcessor(size={"height": 384, "width": 384}, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    proc.output_size = 384
    frames = np.stack([rand_img(480, 854, seed=70 + i) for i in range(4)])
    ref = []
    for f in frames:
        im = Image.fromarray(f).convert("RGB").resize((384, 384), resample=Image.BICUBIC)
        ref.append(proc.preprocess(im, return_tensors="pt")["pixel_values"][0])
    ref = torch.stack(ref).to(dt)
    got = FramePreprocessor.from_image_processor(proc, dtype=dt)(torch.from_numpy(frames).cuda()).cpu()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


def test_frames_small_target_and_misaligned_view():
    """tiny tower size (98) and a frame tensor that starts at an odd byte offset are rejected/handled explicitly"""
    from vidi_amd.preproc import FramePreprocessor
    frames = np.stack([rand_img(123, 77, seed=5)])
    got = FramePreprocessor(98, dtype=torch.float32)(frames).cpu().numpy()          # 98*3 = 294: not a multiple of 4
    assert np.array_equal(got[0], P.siglip_rescale_normalize(P.pil_resize_bicubic_u8(frames[0], 98, 98)))


def test_logmel_vs_oracle_and_hf():
    from transformers import WhisperFeatureExtractor
    from vidi_amd.preproc import LogMelExtractor
    fe = WhisperFeatureExtractor(feature_size=128)
    audio = synth_audio(16000 * 70 + 1234, 3)
    ext = LogMelExtractor.from_feature_extractor(fe, dtype=torch.float32)
    got, length = ext(audio, windows_per_batch=2)                                     # batches of 2 + 1 windows
    ref, ref_len = P.process_audio(audio, mel_filters=fe.mel_filters)
    assert length == ref_len and tuple(got.shape) == ref.shape == (3, 128, 3000)
    err = np.abs(got.cpu().numpy() - ref)
    assert err.max() <= 2e-4, f"max |err| {err.max():.3g}, {np.count_nonzero(err > 2e-4)} over"
    chunks = [audio[i: i + fe.n_samples] for i in range(0, len(audio), fe.n_samples)]
    hf = fe(chunks, sampling_rate=16000, return_tensors="pt").input_features.numpy()
    assert np.abs(got.cpu().numpy() - hf).max() <= 2e-4
    # model dtype output == rounding of the fp32 output
    got16, _ = LogMelExtractor.from_feature_extractor(fe, dtype=torch.bfloat16)(audio)
    assert torch.equal(got16.cpu(), got.cpu().to(torch.bfloat16))


def test_logmel_silence_and_short_clip():
    """all-zero audio: every bin clamps to log10(1e-10) = -10 -> (-10 + 4) / 4 = -1.5; a clip shorter than one hop"""
    from vidi_amd.preproc import LogMelExtractor
    ext = LogMelExtractor(dtype=torch.float32)
    got, length = ext(np.zeros(16000 * 3, dtype=np.float32))
    assert length == 300 and float((got + 1.5).abs().max()) <= 1e-6          # log10f(1e-10f) is one ulp off -10
    got, length = ext(synth_audio(16000, 9)[:100])
    assert length == 0 and got.shape == (1, 128, 3000) and torch.isfinite(got).all()


def test_frames_gpu_matches_reference_process_images():
    """GPU frame path == the reference's own process_images output (tests/golden/reference_process_images.npz), bit for bit"""
    import os
    from vidi_amd.preproc import FramePreprocessor
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_process_images.npz"))
    got = FramePreprocessor(98, dtype=torch.float32)(g["frames"]).cpu().numpy()
    assert np.array_equal(got, g["pixel_values"])
