"""Same-box A/B of the multimodal stream with its norm pairs fused (vidi_resid_norm2, many-row form) or as two launches each: one engine,
the switch flipped between timed runs, cache contents compared (the fused form is bit-identical).
usage: python tools/ab_stream_norm.py [layers] [rounds] [switch]   switch: stream_norm2 (default) | fold_repkv (o_proj over repeat_kv(V) with the
repeated column blocks of the weight summed: half the K; caches compared within tolerance, not bit for bit)"""
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
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    switch = sys.argv[3] if len(sys.argv) > 3 else "stream_norm2"
    dt = torch.bfloat16
    cfg = dataclasses.replace(C.vidi15_9b(), num_hidden_layers=layers, vis_num_layers=2, aud_num_layers=1, vocab_size=1024)
    eng = VidiEngine(cfg, init_random_weights(cfg, seed=3, dtype=dt, device="cuda"), dtype=dt, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    img = (torch.randn((90000, cfg.hidden_size), generator=g, device="cuda") * cfg.mm_std * eng.normalizer).to(dt)
    aud = (torch.randn((36000, cfg.hidden_size), generator=g, device="cuda") * cfg.mm_std * eng.normalizer).to(dt)
    ref = None
    tot = {True: 0.0, False: 0.0}
    for r in range(rounds + 1):
        for flag in (True, False):
            setattr(eng, switch, flag)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            mm = eng.mm_stream_prefill(img, None, aud, None, pre_normalized=True, check_masks=False)
            e1.record(); torch.cuda.synchronize()
            if r == 0:
                if ref is None:
                    ref = (mm.kc.clone(), mm.vtc.clone())
                else:
                    dk = (ref[0].float() - mm.kc.float()).abs().max().item() / ref[0].float().std().item()
                    dv = (ref[1].float() - mm.vtc.float()).abs().max().item() / ref[1].float().std().item()
                    print(json.dumps({"switch": switch, "bit_identical_caches": bool(torch.equal(ref[0], mm.kc) and torch.equal(ref[1], mm.vtc)),
                                      "max_abs_diff_over_std_K": dk, "max_abs_diff_over_std_V": dv}), flush=True)
            else:
                tot[flag] += e0.elapsed_time(e1)
            del mm
    print(json.dumps({"switch": switch, "layers": layers, "fused_ms": tot[True] / rounds, "two_launch_ms": tot[False] / rounds,
                      "speedup": tot[False] / tot[True]}), flush=True)


if __name__ == "__main__":
    main()
