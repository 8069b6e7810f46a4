"""vidi_amd/sampling.py against HF's own logits warpers (what `GenerationMixin` applies for do_sample=True)."""
import pytest
import torch

from vidi_amd.sampling import sample, warp_logits


@pytest.mark.parametrize("temperature,top_k,top_p", [(0.7, None, None), (None, 5, None), (None, None, 0.9), (1.3, 50, 0.8), (0.2, 1, 0.5)])
def test_warpers_match_hf(temperature, top_k, top_p):
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(0)
    logits = torch.randn((3, 1000), generator=g) * 3
    logits[1, :10] = logits[1, 0]                                                    # ties at the top-k boundary
    ref = logits.clone()
    ids = torch.zeros((3, 1), dtype=torch.long)
    if temperature is not None:
        ref = TemperatureLogitsWarper(temperature)(ids, ref)
    if top_k is not None:
        ref = TopKLogitsWarper(top_k)(ids, ref)
    if top_p is not None:
        ref = TopPLogitsWarper(top_p)(ids, ref)
    got = warp_logits(logits, temperature, top_k, top_p)
    assert torch.equal(torch.isinf(got), torch.isinf(ref))
    torch.testing.assert_close(got[~torch.isinf(got)], ref[~torch.isinf(ref)])


def test_sample_respects_filter_and_seed():
    logits = torch.full((2, 50), float("-inf"))
    logits[0, 7] = 0.0
    logits[1, [3, 4]] = torch.tensor([0.0, 0.0])
    a = sample(logits, torch.Generator().manual_seed(5))
    b = sample(logits, torch.Generator().manual_seed(5))
    assert int(a[0]) == 7 and int(a[1]) in (3, 4) and torch.equal(a, b)
