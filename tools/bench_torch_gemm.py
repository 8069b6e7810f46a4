"""Calibration only (NOT used by the product): what the vendor GEMM (hipBLASLt via torch.matmul) reaches on the same
shapes and random data, to judge how much headroom the hand-written kernel has on this chip."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_kernels import timeit, rnd
for (M, N, K) in [(69984, 4352, 1152), (69984, 1152, 4352), (69984, 3456, 1152), (65536, 3584, 14336), (65536, 4096, 3584), (65536, 28672, 3584)]:
    x, w = rnd((M, K)), rnd((N, K), s=0.02)
    ms = timeit(lambda: torch.nn.functional.linear(x, w), iters=5, warm=2)
    print(json.dumps({"lib": "torch/hipBLASLt", "M": M, "N": N, "K": K, "ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9}))
