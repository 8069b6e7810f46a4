"""Same-box A/B of the SigLIP tower under two engine configurations (environment switches read by VidiEngine at construction:
VIDI_LN_FOLD, VIDI_ATTN_RM): two engines in one process, timed alternately (ABAB...) so clocks / box differences cancel.
usage: python tools/ab_ln_fold.py [frames] [rounds] [armA] [armB]     arm = comma-separated NAME=VALUE pairs
defaults: armA = shipped path (LayerNorm folded, head-major q|k|v + transpose-read attention), armB = VIDI_ATTN_RM=0 (round-1 Vt path)"""
import dataclasses
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from vidi_amd import config as C
    from vidi_amd.engine import VidiEngine
    from vidi_amd.weights import init_random_weights
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dt = torch.bfloat16
    cfg = dataclasses.replace(C.vidi15_9b(), num_hidden_layers=1, aud_num_layers=1, vocab_size=1024)
    engs = {}
    arm_a = sys.argv[3] if len(sys.argv) > 3 else "VIDI_LN_FOLD=1,VIDI_ATTN_RM=1"
    arm_b = sys.argv[4] if len(sys.argv) > 4 else "VIDI_LN_FOLD=1,VIDI_ATTN_RM=0"
    for name, arm in (("fold", arm_a), ("plain", arm_b)):
        for kv in arm.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
        engs[name] = VidiEngine(cfg, init_random_weights(cfg, seed=3, dtype=dt, device="cuda"), dtype=dt, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    S = cfg.vis_image_size
    px = (torch.randn((T, 3, S, S), generator=g, device="cuda") * 0.5).clamp_(-1, 1).to(dt)
    outs = {}
    for name, e in engs.items():
        outs[name] = e.siglip_forward(px).float()
    torch.cuda.synchronize()
    d = (outs["fold"] - outs["plain"]).abs()
    print(json.dumps({"max_abs_diff": float(d.max()), "rms_diff": float(d.pow(2).mean().sqrt()), "rms_ref": float(outs["plain"].pow(2).mean().sqrt())}), flush=True)
    del outs
    tot = {k: 0.0 for k in engs}
    for r in range(rounds):
        for name, e in engs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e.siglip_forward(px); e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            tot[name] += ms
            print(json.dumps({"round": r, "arm": name, "frames": T, "siglip_ms": ms}), flush=True)
    print(json.dumps({"frames": T, "fold_ms": tot["fold"] / rounds, "plain_ms": tot["plain"] / rounds,
                      "speedup": tot["plain"] / tot["fold"]}), flush=True)


if __name__ == "__main__":
    main()
