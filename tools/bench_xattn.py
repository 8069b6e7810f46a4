"""Cross-attention decode micro-benchmark (Lq=1 over the 60-min video's keys): ms and algorithmic GB/s per zsplit.
    PYTHONPATH=. python tools/bench_xattn.py [--keys 126000] [--iters 20]"""
import argparse
import json

import torch

from vidi_amd import hip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keys", type=int, default=126000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--lq", type=int, default=1)
    ap.add_argument("--zsplit", type=int, nargs="*", default=[16, 32, 64])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--softcap", type=float, default=50.0, help="0: none (Vidi-7B)")
    ap.add_argument("--masked", type=int, default=1, help="1 (default): pass a key-padding mask, as EVERY launch of the product does (engine._cross_local "
                    "hands over mm.img_mask / mm.aud_mask whenever a key is invalid, and the kernels' mask path is what round 5's figures missed); 0: mask=None")
    a = ap.parse_args()
    hip.load_library()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    cap = a.softcap if a.softcap > 0 else None
    nkv, G, HD, Nk = 8, 2, 256, a.keys
    ntile = (Nk + 63) // 64
    g = torch.Generator(device="cuda").manual_seed(0)
    kc = (torch.randn((nkv, ntile, 64, HD), device="cuda", generator=g)).to(dt)
    vtc = (torch.randn((nkv, 2 * ntile, HD, 32), device="cuda", generator=g)).to(dt)
    # several independent caches so that consecutive iterations do not hit in the 256 MB infinity cache (42 layers in the model)
    caches = [(kc, vtc)] + [(torch.randn_like(kc.float()).to(dt), torch.randn_like(vtc.float()).to(dt)) for _ in range(5)]
    q = torch.randn((a.lq, nkv * G * HD), device="cuda", generator=g).to(dt)
    mask = None
    if a.masked:
        mask = torch.ones((Nk + 63) // 64 * 64, dtype=torch.uint8, device="cuda")
        mask[::97] = 0
        mask[Nk:] = 0
    R = a.lq * G
    Rpad = (R + 31) // 32 * 32
    for zs in a.zsplit:
        if zs == 0:          # the engine's rule (VidiEngine._cross_local): fill the 256 CUs with (kv head) x (row block) x (key slice) blocks
            row_blocks = -(-(Rpad // 32) // hip.attn_cross_row_tiles_per_block(Rpad, cap, dt))
            zs = max(1, min(256 // max(1, nkv * row_blocks), ((Nk + 31) // 32 + 7) // 8))
        opart, ml = hip.attn_cross_workspace(zs, nkv, Rpad, HD, "cuda")
        o = torch.empty((a.lq, nkv * G * HD), dtype=dt, device="cuda")

        def f(i):
            k, v = caches[i % len(caches)]
            hip.attn_cross(q, k, v, mask, opart, ml, R=R, Rpad=Rpad, G=G, nkv=nkv, HD=HD, ntile64=ntile, key_start=0,
                           n_keys=Nk, scale=HD ** -0.5, softcap=cap, zsplit=zs)
            hip.attn_merge(opart, ml, o, W=zs, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
        for i in range(3):
            f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.iters):
            f(i)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        print(json.dumps({"kernel": "attn_cross+merge", "Lq": a.lq, "keys": Nk, "zsplit": zs, "masked": bool(a.masked), "ms": ms,
                          "GBps": Nk * 2 * nkv * HD * 2 / ms / 1e6, "TFLOPs": 4.0 * a.lq * Nk * nkv * G * HD / ms / 1e9,
                          "row_tiles_per_block": hip.attn_cross_row_tiles_per_block(Rpad, cap, dt), "dtype": a.dtype, "softcap": cap}), flush=True)


if __name__ == "__main__":
    main()
