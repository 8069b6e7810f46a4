"""bench.py's verification leg restricts the oracle's encode pipelines to a few frames / windows of a longer video (global frame count
and indices for the token-budget rule and pos_t).  Here: those restrictions reproduce `oracle/vidi_oracle.py`'s own
encode_video_images / encode_video_audios (which are pinned on the reference-executed goldens) on the frames / windows they keep."""
import dataclasses
import os
import sys

import pytest
import torch

import vidi_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _setup(base=60000):
    from vidi_amd.config import tiny
    from vidi_amd.weights import init_random_weights
    cfg = dataclasses.replace(tiny(), mm_max_tokens_base=base)
    names = {f.name for f in dataclasses.fields(O.OracleConfig)}
    d = {k: v for k, v in cfg.to_dict().items() if k in names}
    d["vis_select_layer"] = cfg.mm_vision_select_layer
    d["arch"] = cfg.arch
    w = {k: v.float() for k, v in init_random_weights(cfg, seed=3, dtype=torch.float32).items()}
    return cfg, O.OracleConfig(**d), w


@pytest.mark.parametrize("base", [60000, 50])
def test_frame_subset_equals_full_encode(base):
    import bench
    cfg, ocfg, w = _setup(base)
    T = 7
    g = torch.Generator().manual_seed(5)
    px = (torch.randn((T, 3, cfg.vis_image_size, cfg.vis_image_size), generator=g) * 0.5).clamp(-1, 1)
    full, mask = O.encode_video_images([px], w, ocfg)
    assert bool(mask.all())
    tpf = full.shape[1] // T
    full = full[0].reshape(T, tpf, -1) * torch.tensor(cfg.hidden_size ** 0.5)
    pick = [0, 3, 6]
    sub = bench._oracle_frame_embeds(O, px[pick], pick, T, w, ocfg, torch.float32)
    assert sub.shape == (3, tpf, cfg.hidden_size)
    assert torch.allclose(sub, full[pick], rtol=1e-5, atol=1e-5)


def test_window_subset_equals_full_encode():
    import bench
    cfg, ocfg, w = _setup()
    C = 3
    g = torch.Generator().manual_seed(6)
    mel = torch.randn((C, cfg.aud_num_mel_bins, cfg.aud_nb_max_frames), generator=g) * 0.3
    audio_size = C * cfg.aud_nb_max_frames - 17                     # the last window is clipped by the global floors; windows 0, 1 are whole
    full, mask = O.encode_video_audios([mel], [audio_size], w, ocfg)
    per = cfg.aud_max_source_positions // cfg.mm_audio_pool_size
    full = full[0] * torch.tensor(cfg.hidden_size ** 0.5)
    sub = bench._oracle_window_embeds(O, mel[[0, 1]], [0, 1], audio_size, w, ocfg, torch.float32)
    assert sub.shape == (2, per, cfg.hidden_size)
    assert torch.allclose(sub.reshape(2 * per, -1), full[: 2 * per], rtol=1e-5, atol=1e-5)


def test_cache_row_unpacking_matches_the_test_packer():
    """bench._cache_rows reads K / V rows back out of the tiled cache layout (tests/util.py:pack_kv_cache writes it)"""
    import bench
    from util import pack_kv_cache
    N, nkv, hd = 200, 2, 16
    k = torch.randn(N, nkv, hd); v = torch.randn(N, nkv, hd)
    kc, vtc = pack_kv_cache(k, v, (N + 63) // 64)
    mm = type("MM", (), {})()
    mm.kc, mm.vtc = kc[None], vtc[None]
    rows = [0, 1, 15, 16, 31, 32, 63, 64, 100, 199]
    kg, vg = bench._cache_rows(mm, 0, rows, nkv, hd)
    assert torch.equal(kg, k[rows].reshape(len(rows), -1)) and torch.equal(vg, v[rows].reshape(len(rows), -1))


def test_synthetic_video_is_a_function_of_the_global_index():
    """bench.py's synthetic frames / mel: any rank's shard (a slice of the global element range) equals the same slice of the one-rank
    video, the values are standard normal and the two streams (frames, mel) are independent"""
    import bench
    whole = bench.synth_normal(0, 300_000, 0x51A1, "cpu")
    for a, b in ((0, 1000), (123_457, 200_001), (299_000, 300_000)):
        assert torch.equal(bench.synth_normal(a, b - a, 0x51A1, "cpu"), whole[a:b])
    assert abs(float(whole.mean())) < 0.01 and abs(float(whole.std()) - 1.0) < 0.01 and bool(torch.isfinite(whole).all())
    other = bench.synth_normal(0, 300_000, 0xA0D1, "cpu")
    assert abs(float(torch.corrcoef(torch.stack([whole, other]))[0, 1])) < 0.01
    assert abs(float(torch.corrcoef(torch.stack([whole[:-1], whole[1:]]))[0, 1])) < 0.01
