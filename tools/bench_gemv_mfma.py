"""Decode-batch projections, one box, one process: vidi_gemv (VALU), vidi_gemv_mfma (matrix pipe) and vidi_gemm_skinny (split-K) on the
decoder's shapes at M rows — microseconds per launch and TB/s of weight streaming (HIP events around `reps` back-to-back launches, the
weights of a shape rotated over `nbuf` copies so that no launch finds its weight in the Infinity Cache).
    python tools/bench_gemv_mfma.py [M=8] [reps=40] [arms=mfma,valu,skinny]     (VIDI_GEMVM_KS / _TARGET_WAVES / _RESIDENT_WAVES: launch shape)"""
import json
import sys

import torch

sys.path.insert(0, ".")
from vidi_amd import hip  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None      # arms to run (default: all)
dt = torch.bfloat16
hip.load_library()
SHAPES = [("qkv", 8192, 3584, False), ("o", 3584, 4096, False), ("gate_up", 14336, 3584, True), ("down", 3584, 14336, False), ("lm_head", 256000, 3584, False)]


def timed(fn, n):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K, glu in SHAPES:
    rows = 2 * N if glu else N
    wbytes = rows * K * 2
    nbuf = max(2, min(8, int(1.2e9 // wbytes)))
    ws = [torch.randn((rows, K), device="cuda", dtype=torch.float32).mul_(0.05).to(dt) for _ in range(nbuf)]
    for m in sorted({M, 3 * M} if name == "o" else {M}):
        x = torch.randn((m, K), device="cuda").to(dt)
        out = torch.empty((m, N), device="cuda", dtype=dt)
        import os
        rec = {"shape": name, "M": m, "N": N, "K": K, "weight_MB": wbytes / 1e6, "env": {k: v for k, v in os.environ.items() if k.startswith("VIDI_GEMVM")}}
        if glu:
            arms = {"mfma": lambda i: hip.gemv_mfma(x, ws[i % nbuf], out, glu_act=hip.ACT_GELU_TANH)}
            if m <= 8:
                arms["valu"] = lambda i: hip.gemv_glu(x, ws[i % nbuf], out, hip.ACT_GELU_TANH)
        else:
            arms = {"mfma": lambda i: hip.gemv_mfma(x, ws[i % nbuf], out)}
            if m <= 8:
                arms["valu"] = lambda i: hip.gemv(x, ws[i % nbuf], out)
            need = hip.gemm_skinny_workspace_bytes(m, N, K)
            if need:
                wsp = torch.empty(need // 4, device="cuda", dtype=torch.float32)
                arms["skinny"] = lambda i: hip.gemm_skinny(x, ws[i % nbuf], wsp, out)
        for k, fn in arms.items():
            if only and k not in only:
                continue
            us = min(timed(fn, reps) for _ in range(3))
            rec[k + "_us"] = round(us, 1)
            rec[k + "_TB/s"] = round(wbytes / us / 1e6, 2)
        print(json.dumps(rec), flush=True)
    del ws
