"""Per-shape GEMM table of the 60-min workload: this library's kernel (auto tile) and, as calibration only (NOT used by
the product), the vendor GEMM behind torch.matmul on the same random data.  usage: bench_shapes.py [ours|vendor|both]"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidi_amd import hip
from tools.bench_kernels import timeit, rnd
hip.load_library()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
SHAPES = [("siglip qkv", 262440, 3456, 1152), ("siglip o", 262440, 1152, 1152), ("siglip fc1", 262440, 4352, 1152),
          ("siglip fc2", 262440, 1152, 4352), ("mm kv", 126000, 4096, 3584), ("mm o", 126000, 3584, 4096),
          ("mm gate/up", 126000, 28672, 3584), ("mm down", 126000, 3584, 14336),
          ("whisper qkv", 180000, 3840, 1280), ("whisper fc1", 180000, 5120, 1280), ("whisper fc2", 180000, 1280, 5120)]
for name, M, N, K in SHAPES:
    x, w = rnd((M, K)), rnd((N, K), s=0.02)
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    row = {"shape": name, "M": M, "N": N, "K": K}
    if which in ("ours", "both"):
        ms = timeit(lambda: hip.gemm(x, w, None, y), iters=5, warm=2)
        row["ours_ms"], row["ours_tflops"] = ms, 2.0 * M * N * K / ms / 1e9
    if which in ("vendor", "both"):
        ms = timeit(lambda: torch.nn.functional.linear(x, w), iters=5, warm=2)
        row["vendor_ms"], row["vendor_tflops"] = ms, 2.0 * M * N * K / ms / 1e9
    print(json.dumps(row), flush=True)
    del x, w, y
