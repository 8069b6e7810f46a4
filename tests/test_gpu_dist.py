"""Multi-rank path on real hardware: two ranks share the one GPU of the test box (gloo transport for the
collectives, every kernel on the GPU).  Frame/window-sharded encode + sharded diagonal stream + per-layer
partial cross-attention all-gather/merge must reproduce the single-rank text hidden states."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

from util import report

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(world, out):
    worker = os.path.join(HERE, "dist_worker.py")
    if world == 1:
        cmd = [sys.executable, worker, out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29533", worker, out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return torch.load(out)


def test_two_rank_sharded_prefill_matches_single_rank():
    with tempfile.TemporaryDirectory() as d:
        a = _launch(1, os.path.join(d, "w1.pt"))
        b = _launch(2, os.path.join(d, "w2.pt"))
    assert (a["g_img"], a["g_aud"]) == (b["g_img"], b["g_aud"]) == (4 * 16, 17)
    assert b["n_img_local"] == 2 * 16 and b["n_aud_local"] == 10            # rank 0 holds frames 0-1 and window 0
    report("sharded prefill hidden", b["prefill"], a["prefill"], 5e-2 * a["prefill"].std().item(), 3e-2)   # merge order differs in fp32 -> 1 bf16 ulp at the attention output
    report("sharded decode hidden", b["decode"], a["decode"], 5e-2 * a["decode"].std().item(), 3e-2)
    # ONE packed all-gather per decoder layer per forward (numerator + (m, l) of both modalities), none on a single rank
    assert a["collectives_per_forward"] == 0 and b["collectives_per_forward"] == b["layers"]
    # the public API: generate() over the sharded video gives the single-rank tokens; so does a query against the resident shards
    assert torch.equal(b["tokens_cached"], b["tokens"])
    assert torch.equal(a["tokens"], a["tokens_cached"])
    assert torch.equal(a["tokens"][:, :1], b["tokens"][:, :1])                # first token: margin-independent in this seeded case
    print("greedy tokens single-rank", a["tokens"].tolist(), "two ranks", b["tokens"].tolist())   # later tokens may flip on a 1-ulp tie
