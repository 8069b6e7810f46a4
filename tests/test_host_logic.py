"""Integer / index host logic: bit-exact against the oracle restatement of the reference lines."""
import numpy as np
import torch

import vidi_oracle as O
from vidi_amd.config import tiny, vidi15_9b
from vidi_amd.engine import audio_token_counts, token_budget_hw
from vidi_amd.model import strip_image_token


def test_token_budget_rule():
    for T in [1, 25, 299, 300, 306, 307, 308, 400, 600, 1199, 1200, 1800, 3600, 7200, 10000]:
        assert token_budget_hw(T, 27, 2, 60000) == O.token_budget_hw(T, 27, 2, 60000)
    assert token_budget_hw(3600, 27, 2, 60000) == (10, 10)         # 25 tokens/frame -> 90 000 (BASELINE 60-min config)
    assert token_budget_hw(300, 27, 2, 60000) == (28, 28)          # 196 tokens/frame -> 58 800 (5-min config)


def test_audio_floors():
    cfg = vidi15_9b()
    ocfg = O.OracleConfig()
    for size in [0, 7, 199, 200, 2460, 2999, 3000, 3001, 30000, 360000, 359999]:
        s1, s2 = O.audio_token_counts([size], ocfg)
        assert audio_token_counts(size, cfg) == (int(s1[0]), int(s2[0]))
    assert audio_token_counts(360000, cfg) == (180000, 36000)      # 10 audio tokens / second


def test_strip_image_token_matches_oracle():
    ids = torch.tensor([[2, 5, -200, 9, 9, 0, 0], [2, 7, 8, -200, 6, 4, 3]])
    am = torch.tensor([[1, 1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1, 1]])
    for side in ("right", "left"):
        got, mask, pos = strip_image_token(ids, am, side)
        rows, m2, p2 = O.strip_image_token(ids, am, side)
        assert torch.equal(mask, m2) and torch.equal(pos, p2)
        for i, r in enumerate(rows):
            assert torch.equal(got[i][mask[i]], r)
    got, mask, pos = strip_image_token(torch.tensor([[2, -200, 11, 12]]))
    assert got.tolist() == [[2, 11, 12]] and pos.tolist() == [[0, 1, 2]]


def test_timestamp_formatting():
    """eval/inference.py:52-66"""
    assert O.format_time_ranges("0.10-0.25, 0.50-0.75", 3600.0) == "00:06:00-00:15:00, 00:30:00-00:45:00"
    assert O.format_time_ranges("garbage", 10.0) == ""
    assert O.format_time_ranges("0.999-1.000", 3661.5) == "01:00:57-01:01:01"


def test_tensor_split_bounds():
    for n, parts in [(10, 3), (3600, 32), (7, 8), (120, 8)]:
        ref = [int(x.shape[0]) for x in torch.tensor_split(torch.zeros(n), parts)]
        got = [e - s for s, e in O.tensor_split_bounds(n, parts)]
        assert got == ref


def test_weight_shapes_cover_engine_needs():
    from vidi_amd.weights import init_random_weights, weight_shapes
    cfg = tiny()
    w = init_random_weights(cfg, dtype=torch.float32)
    assert set(w) == set(weight_shapes(cfg))
    assert w["model.mm_rand_pos_t.mlp.0.weight"].dtype == torch.float32
    assert abs(float(w["model.mm_rand_llm_norm.weight"][0]) - cfg.mm_std) < 1e-7
