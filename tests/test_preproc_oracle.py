"""The preprocessing oracle (oracle/preproc_oracle.py) pinned on the real third-party implementations the reference calls:
PIL.Image.resize (bit-exact), the installed SiglipImageProcessor (bit-exact) and WhisperFeatureExtractor (1e-4)."""
import numpy as np
import pytest
import torch

import preproc_oracle as P


def rand_img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    base[: h // 3, : w // 2] = rng.integers(0, 256, size=3, dtype=np.uint8)          # flat patch: saturation / rounding paths
    base[h // 2:, w // 2:] = np.where(rng.random((h - h // 2, w - w // 2, 1)) > 0.5, 255, 0)   # hard edges: overshoot -> clip8
    return base


@pytest.mark.parametrize("hw", [(480, 854), (360, 640), (384, 384), (200, 300), (1080, 1920), (97, 131), (720, 405)])
def test_resize_bit_exact_vs_pil(hw):
    from PIL import Image
    img = rand_img(*hw, seed=hw[0] * 7 + hw[1])
    ref = np.asarray(Image.fromarray(img).convert("RGB").resize((384, 384), resample=Image.BICUBIC))
    got = P.pil_resize_bicubic_u8(img, 384, 384)
    assert np.array_equal(got, ref), f"{(got.astype(int) - ref).__abs__().max()} max diff, {np.count_nonzero(got != ref)} pixels"


def test_resize_other_target_sizes():
    from PIL import Image
    img = rand_img(123, 77, 5)
    for (w, h) in [(98, 98), (224, 224), (50, 200)]:
        ref = np.asarray(Image.fromarray(img).resize((w, h), resample=Image.BICUBIC))
        assert np.array_equal(P.pil_resize_bicubic_u8(img, w, h), ref)


def test_normalize_bit_exact_vs_hf_processor():
    from transformers import SiglipImageProcessor
    from PIL import Image
    proc = SiglipImageProcessor(size={"height": 384, "width": 384}, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    img = rand_img(384, 384, 9)
    img[0, :256, 0] = np.arange(256)                                                 # every byte value
    ref = proc.preprocess(Image.fromarray(img), return_tensors="pt")["pixel_values"][0].numpy()
    got = P.siglip_rescale_normalize(img)
    assert got.dtype == np.float32 and np.array_equal(got, ref)
    # whole frame path of img_utils.py:181-185
    big = rand_img(480, 854, 10)
    pil = Image.fromarray(big).resize((384, 384), resample=Image.BICUBIC)
    ref = proc.preprocess(pil, return_tensors="pt")["pixel_values"][0].numpy()
    assert np.array_equal(P.process_frame(big), ref)


def test_mel_filter_bank_matches_hf():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=128)
    np.testing.assert_allclose(P.mel_filter_bank(201, 128), fe.mel_filters, rtol=1e-12, atol=1e-15)
    fe80 = WhisperFeatureExtractor(feature_size=80)
    np.testing.assert_allclose(P.mel_filter_bank(201, 80), fe80.mel_filters, rtol=1e-12, atol=1e-15)


def synth_audio(n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t * (1 + 0.1 * t / t[-1])) + 0.02 * rng.standard_normal(n)
    x[n // 3: n // 3 + 8000] = 0.0                                                  # silence: the (max - 8) floor
    return (np.round(x * 32768).clip(-32768, 32767).astype(np.int16).astype(np.float32) / 32768.0)   # load_audio's s16le -> f32


def test_logmel_vs_hf_feature_extractor():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=128)
    audio = synth_audio(16000 * 70 + 1234, 3)                                        # 2 full windows + a ragged third
    chunks = [audio[i: i + fe.n_samples] for i in range(0, len(audio), fe.n_samples)]
    ref = fe(chunks, sampling_rate=16000, return_tensors="pt").input_features.numpy()
    got, length = P.process_audio(audio, mel_filters=fe.mel_filters)
    assert got.shape == ref.shape == (3, 128, 3000)
    assert length == sum(len(c) // 160 for c in chunks) == 2 * 3000 + (len(audio) - 2 * 480000) // 160
    # float32 FFT (HF) vs float64 DFT (here); scale of the features is O(1)
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=0)        # observed 5.4e-5 on 58 of 1.15 M low-power bins (HF's fp32 FFT noise)
