"""GEMM micro-benchmark: python tools/bench_gemm.py M N K cfg [geglu]"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidi_amd import hip
from tools.bench_kernels import timeit, rnd
hip.load_library(os.environ.get("VIDI_LIB"))       # VIDI_LIB: A/B against another build of the library
if os.environ.get("VIDI_LIB"): hip._lib = hip.load_library(os.environ["VIDI_LIB"])
M, N, K, cfg = [int(x) for x in sys.argv[1:5]]
geglu = len(sys.argv) > 5
x, w = rnd((M, K)), rnd((N, K), s=0.02)
if geglu:
    y = torch.empty((M, N // 2), dtype=torch.bfloat16, device="cuda")
    f = lambda: hip.gemm_geglu(x, w, y, tile_cfg=cfg)
else:
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    act = {"": hip.ACT_NONE, "tanh": hip.ACT_GELU_TANH, "erf": hip.ACT_GELU_ERF}[os.environ.get("ACT", "")]
    res = rnd((M, N)) if os.environ.get("RES") else None
    bias = rnd((N,)) if os.environ.get("BIAS") else None
    f = lambda: hip.gemm(x, w, bias, y, tile_cfg=cfg, act=act, residual=res)
ms = timeit(f, iters=int(os.environ.get("ITERS", "5")), warm=2)
row = {"M": M, "N": N, "K": K, "cfg": cfg, "act": os.environ.get("ACT", ""), "res": bool(os.environ.get("RES")), "ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9}
if os.environ.get("CHECK") and not geglu:          # experimental schedules: compare with the default tile's result
    y2 = torch.empty_like(y)
    hip.gemm(x, w, None, y2, tile_cfg=4)
    row["max_abs_diff_vs_cfg4"] = (y.float() - y2.float()).abs().max().item()
print(json.dumps(row))
