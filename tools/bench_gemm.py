"""GEMM micro-benchmark: python tools/bench_gemm.py M N K cfg [geglu]"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidi_amd import hip
from tools.bench_kernels import timeit, rnd
hip.load_library()
M, N, K, cfg = [int(x) for x in sys.argv[1:5]]
geglu = len(sys.argv) > 5
x, w = rnd((M, K)), rnd((N, K), s=0.02)
if geglu:
    y = torch.empty((M, N // 2), dtype=torch.bfloat16, device="cuda")
    f = lambda: hip.gemm_geglu(x, w, y, tile_cfg=cfg)
else:
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    f = lambda: hip.gemm(x, w, None, y, tile_cfg=cfg)
ms = timeit(f, iters=int(os.environ.get("ITERS", "5")), warm=2)
row = {"M": M, "N": N, "K": K, "cfg": cfg, "ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9}
if os.environ.get("CHECK") and not geglu:          # experimental schedules: compare with the default tile's result
    y2 = torch.empty_like(y)
    hip.gemm(x, w, None, y2, tile_cfg=4)
    row["max_abs_diff_vs_cfg4"] = (y.float() - y2.float()).abs().max().item()
print(json.dumps(row))
