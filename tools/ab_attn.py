"""Same-box A/B of encoder-attention schedule variants: several builds of libvidi_hip (same ABI) loaded side by side with ctypes and
timed alternately on the SigLIP shape (N = 729, d = 72, 16 heads).  usage: python tools/ab_attn.py lib_a.so lib_b.so ... [--frames 360]"""
import ctypes
import json
import sys

import torch


def main():
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    def opt(name, default):
        return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
    B, N, H, D = opt("--frames", 360), opt("--N", 729), opt("--H", 16), opt("--D", 72)
    Npad = (N + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qk = (torch.randn((B * N, 2 * H * D), generator=g, device="cuda")).to(torch.bfloat16)
    vt = (torch.randn((B, H, D, Npad), generator=g, device="cuda")).to(torch.bfloat16)
    outs, fns = {}, {}
    for path in libs:
        lib = ctypes.CDLL(path)
        f = lib.vidi_attn_self
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 8 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        o = torch.empty((B * N, H * D), dtype=torch.bfloat16, device="cuda")

        def run(f=f, o=o):
            rc = f(qk.data_ptr(), vt.data_ptr(), o.data_ptr(), B, N, Npad, H, D, 2 * H * D, H * D, H * D, D ** -0.5, 0,
                   torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        run(); torch.cuda.synchronize()
        outs[path], fns[path] = o, run
    # the row-major-V kernel of every library that has it (vidi_attn_self_rm), on the same Q/K/V
    qkv = torch.cat([qk, vt.new_zeros((B * N, H * D))], dim=1)
    Vrm = torch.randn((B, N, H, D), generator=g, device="cuda").to(torch.bfloat16)
    qkv[:, 2 * H * D:] = Vrm.reshape(B * N, H * D)
    hm = torch.randn((3 * B * H * N * D,), generator=g, device="cuda").to(torch.bfloat16)      # ONE head-major input for every library
    hm_ps = hm.clone()                                                                          # the same input with scale * log2(e) folded into Q
    hm_ps[: B * H * N * D] = (hm[: B * H * N * D].float() * (D ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
    for path in list(libs):
        lib = ctypes.CDLL(path)
        if not hasattr(lib, "vidi_attn_self_rm"):
            continue
        f = lib.vidi_attn_self_rm
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_longlong] * 4 + [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        o = torch.empty((B * N, H * D), dtype=torch.bfloat16, device="cuda")

        def run_rm(f=f, o=o):
            rc = f(qkv.data_ptr(), o.data_ptr(), B, N, H, D, 3 * H * D, H * D, 2 * H * D, 0, 0, H * D, D ** -0.5, 0, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        run_rm(); torch.cuda.synchronize()
        name = path + ":rm"
        libs.append(name); outs[name], fns[name] = o, run_rm
        # the same kernel on a HEAD-MAJOR input ([3][B][H][N][D]: every head's key rows contiguous)
        o2 = torch.empty((B * N, H * D), dtype=torch.bfloat16, device="cuda")

        def run_hm(f=f, o2=o2, hm=hm):
            rc = f(hm.data_ptr(), o2.data_ptr(), B, N, H, D, D, B * H * N * D, 2 * B * H * N * D, H * N * D, N * D, H * D, D ** -0.5, 0,
                   torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        run_hm(); torch.cuda.synchronize()
        libs.append(path + ":rm_headmajor"); outs[path + ":rm_headmajor"], fns[path + ":rm_headmajor"] = o2, run_hm
        # head-major with the scale folded into Q by the caller (scale = 0): d = 72 takes the max inside the contraction
        o3 = torch.empty((B * N, H * D), dtype=torch.bfloat16, device="cuda")

        def run_ps(f=f, o3=o3, hm=hm_ps):
            rc = f(hm.data_ptr(), o3.data_ptr(), B, N, H, D, D, B * H * N * D, 2 * B * H * N * D, H * N * D, N * D, H * D, 0.0, 0,
                   torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        run_ps(); torch.cuda.synchronize()
        libs.append(path + ":rm_prescaled"); outs[path + ":rm_prescaled"], fns[path + ":rm_prescaled"] = o3, run_ps
    # every library's result per mode (Vt kernel, rm on row-major input, rm on head-major input) against the FIRST library's
    for mode in ("", ":rm", ":rm_headmajor", ":rm_prescaled"):
        same = [x for x in libs if (x.endswith(mode) if mode else ":rm" not in x)]
        for path in same[1:]:
            print(json.dumps({"lib": path, "bit_identical_to_first": bool(torch.equal(outs[path], outs[same[0]])),
                              "max_abs_diff": float((outs[path].float() - outs[same[0]].float()).abs().max())}), flush=True)
    tot = {p: 0.0 for p in libs}
    rounds = 4
    for r in range(rounds):
        for path in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fns[path]()
            e1.record(); torch.cuda.synchronize()
            tot[path] += e0.elapsed_time(e1) / 5
    for path in libs:
        ms = tot[path] / rounds
        print(json.dumps({"lib": path, "frames": B, "ms": ms, "useful_tflops": 4.0 * N * N * D * H * B / ms / 1e9}), flush=True)


if __name__ == "__main__":
    main()
