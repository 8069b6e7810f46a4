"""SigLIP tower time vs activation chunk size (cfg.vis_frames_per_chunk): the persistent GEMM walks ceil(M/256) x ceil(N/256) tiles with
256 blocks, so the tile count of a chunk's GEMMs modulo 256 decides how full the last round is (N = 1152 gives 5 n-tiles: 360 frames ->
5130 tiles = 20.04 rounds).  usage: python tools/bench_vis_chunk.py [frames] [chunk ...]"""
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
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
    chunks = [int(x) for x in sys.argv[2:]] or [300, 360, 400, 450, 600, 900, 1200, 1800, 3600]
    dt = torch.bfloat16
    cfg = dataclasses.replace(C.vidi15_9b(), num_hidden_layers=1, aud_num_layers=1, vocab_size=1024)
    eng = VidiEngine(cfg, init_random_weights(cfg, seed=3, dtype=dt, device="cuda"), dtype=dt, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    S = cfg.vis_image_size
    px = (torch.randn((T, 3, S, S), generator=g, device="cuda") * 0.5).clamp_(-1, 1).to(dt)
    for fc in chunks:
        cfg.vis_frames_per_chunk = fc
        eng._ws.clear()
        torch.cuda.empty_cache()
        eng.siglip_forward(px)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            eng.siglip_forward(px)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 2
        print(json.dumps({"frames": T, "vis_frames_per_chunk": fc, "siglip_ms": ms, "frames_per_s": T / ms * 1e3}), flush=True)


if __name__ == "__main__":
    main()
