import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vidi_amd import hip
from tools.bench_kernels import timeit, rnd
hip.load_library()
shapes = [(96, 729, 16, 72), (24, 729, 16, 72), (96, 128, 16, 72), (96, 729, 16, 64)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (B, N, H, D) in shapes:
    Npad = (N + 63) // 64 * 64
    qk = rnd((B * N, 2 * H * D)); vt = rnd((B, H, D, Npad)); o = torch.empty((B * N, H * D), dtype=torch.bfloat16, device="cuda")
    ms = timeit(lambda: hip.attn_self(qk, vt, o, B=B, N=N, Npad=Npad, H=H, D=D, koff=H * D, scale=D ** -0.5))
    print(json.dumps({"abl": os.environ.get("VIDI_ATTN_ABL", "0"), "B": B, "N": N, "D": D, "ms": ms, "tflops": 4.0 * N * N * D * H * B / ms / 1e9}))
