"""Micro-benchmarks of the hot kernels at BASELINE shapes (run on the GPU box; prints one JSON line per case).
Times with torch.cuda.Event on the current stream (the kernels are launched on torch's current stream)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from vidi_amd import hip  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def rnd(shape, dt=torch.bfloat16, s=1.0):
    return (torch.randn(shape, device="cuda") * s).to(dt)


def main():
    hip.load_library()
    out = []
    # ---- GEMM shapes: (name, M, N, K)
    shapes = [("siglip_qkv", 96 * 729, 3456, 1152), ("siglip_fc1", 96 * 729, 4352, 1152), ("siglip_fc2", 96 * 729, 1152, 4352),
              ("siglip_out", 96 * 729, 1152, 1152), ("llm_kv", 65536, 4096, 3584), ("llm_down", 65536, 3584, 14336),
              ("llm_gateup(geglu)", 65536, 28672, 3584)]
    for name, M, N, K in shapes:
        x, w = rnd((M, K)), rnd((N, K), s=0.02)
        for cfg in (0, 1, 2):
            if "geglu" in name:
                y = torch.empty((M, N // 2), dtype=torch.bfloat16, device="cuda")
                f = lambda: hip.gemm_geglu(x, w, y, tile_cfg=cfg)
            else:
                y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
                f = lambda: hip.gemm(x, w, None, y, tile_cfg=cfg)
            try:
                ms = timeit(f, iters=5, warm=2)
                out.append({"kernel": "gemm", "shape": name, "M": M, "N": N, "K": K, "cfg": cfg, "ms": ms,
                            "tflops": 2.0 * M * N * K / ms / 1e9})
            except Exception as e:  # noqa
                out.append({"kernel": "gemm", "shape": name, "cfg": cfg, "error": str(e)})
            print(json.dumps(out[-1]), flush=True)
        del x, w
    # ---- SigLIP attention
    B, N, H, D = 96, 729, 16, 72
    Npad = 768
    qk = rnd((B * N, 2 * H * D)); vt = rnd((B, H, D, Npad)); o = torch.empty((B * N, H * D), dtype=torch.bfloat16, device="cuda")
    ms = timeit(lambda: hip.attn_self(qk, vt, o, B=B, N=N, Npad=Npad, H=H, D=D, koff=H * D, scale=D ** -0.5))
    out.append({"kernel": "attn_self", "B": B, "N": N, "H": H, "D": D, "ms": ms, "tflops": 4.0 * N * N * D * H * B / ms / 1e9})
    print(json.dumps(out[-1]), flush=True)
    # ---- cross attention decode (60-min video keys)
    nkv, G, HD, Nk = 8, 2, 256, 90000
    ntile = (Nk + 63) // 64
    kc = rnd((nkv, ntile, 64, HD)); vtc = rnd((nkv, 2 * ntile, HD, 32))
    for Lq in (1, 40):
        q = rnd((Lq, nkv * G * HD))
        R = Lq * G; Rpad = (R + 31) // 32 * 32
        for zs in (8, 32, 64):
            if zs * (Rpad // 32) > 64 * 4:
                continue
            opart, ml = hip.attn_cross_workspace(zs, nkv, Rpad, HD, "cuda")
            o = torch.empty((Lq, nkv * G * HD), dtype=torch.bfloat16, device="cuda")

            def f():
                hip.attn_cross(q, kc, vtc, None, opart, ml, R=R, Rpad=Rpad, G=G, nkv=nkv, HD=HD, ntile64=ntile, key_start=0,
                               n_keys=Nk, scale=HD ** -0.5, softcap=50.0, zsplit=zs)
                hip.attn_merge(opart, ml, o, W=zs, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
            ms = timeit(f)
            out.append({"kernel": "attn_cross+merge", "Lq": Lq, "keys": Nk, "zsplit": zs, "ms": ms,
                        "GBps": Nk * 2 * nkv * HD * 2 / ms / 1e6})
            print(json.dumps(out[-1]), flush=True)
    # ---- norms (HBM)
    x = rnd((126000, 3584)); w = rnd((3584,))
    y = torch.empty_like(x)
    ms = timeit(lambda: hip.norm(hip.NORM_GEMMA, x, w, eps=1e-6, out=y))
    out.append({"kernel": "rmsnorm_gemma", "rows": 126000, "H": 3584, "ms": ms, "GBps": 2 * x.numel() * 2 / ms / 1e6})
    print(json.dumps(out[-1]), flush=True)
    # ---- gemv (decode weight streaming)
    x1 = rnd((1, 3584)); wv = rnd((28672, 3584), s=0.02)
    ms = timeit(lambda: hip.gemv(x1, wv))
    out.append({"kernel": "gemv", "N": 28672, "K": 3584, "ms": ms, "GBps": wv.numel() * 2 / ms / 1e6})
    print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
