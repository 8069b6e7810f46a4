"""ctypes binding of libvidi_hip.so (include/vidi_hip.h).

PyTorch is used only as the owner of device memory and streams: every function here takes torch
tensors, checks device/dtype/contiguity, and passes `data_ptr()` + sizes + the current HIP stream
through the C ABI.  There is NO fallback: if the library is missing or a kernel returns an error
the call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIDI_HIP_LIB") or os.path.join(_HERE, "libvidi_hip.so")      # VIDI_HIP_LIB: A/B a second build of the same ABI

DT_BF16, DT_F16, DT_F32 = 0, 1, 2
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU = 0, 1, 2, 3
NORM_GEMMA, NORM_GEMMA_ADD, NORM_MM, NORM_MM_NOW, NORM_LLM, NORM_LAYER = range(6)

_c_int, _c_ll, _c_f, _c_vp = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p

# name -> argtypes (mirrors include/vidi_hip.h; tests/test_abi.py checks every symbol exists)
SIGNATURES = {
    "vidi_gemm": [_c_vp] * 5 + [_c_int] * 8 + [_c_ll] * 3 + [_c_int] * 6 + [_c_vp],
    "vidi_gemm_geglu": [_c_vp] * 3 + [_c_int] * 8 + [_c_vp],
    "vidi_gemm_glu": [_c_vp] * 3 + [_c_int] * 9 + [_c_vp],
    "vidi_glu_unpack": [_c_vp] * 2 + [_c_int] * 4 + [_c_vp],
    "vidi_im2col_nhwc": [_c_vp] * 2 + [_c_int] * 5 + [_c_vp],
    "vidi_resize_bilinear_ac": [_c_vp] * 2 + [_c_int] * 5 + [_c_vp],
    "vidi_gemm_qkv_vt": [_c_vp] * 5 + [_c_int] * 13 + [_c_vp],
    "vidi_gemm_kv_cache": [_c_vp] * 5 + [_c_int] * 10 + [_c_vp],
    "vidi_row_stats": [_c_vp, _c_vp, _c_ll, _c_int, _c_ll, _c_f, _c_int, _c_vp],
    "vidi_gemm_ln": [_c_vp] * 6 + [_c_int] * 9 + [_c_vp],
    "vidi_gemm_res_stats": [_c_vp] * 6 + [_c_int] * 9 + [_c_vp],
    "vidi_ln_finalize": [_c_vp, _c_vp, _c_ll, _c_int, _c_f, _c_vp],
    "vidi_gemm_qkv_vt_ln": [_c_vp] * 7 + [_c_int] * 13 + [_c_vp],
    "vidi_gemv": [_c_vp] * 3 + [_c_int] * 7 + [_c_vp],
    "vidi_gemm_skinny": [_c_vp] * 5 + [_c_int] * 7 + [_c_vp],
    "vidi_gemv_glu": [_c_vp] * 3 + [_c_int] * 8 + [_c_vp],
    "vidi_gemv_mfma": [_c_vp] * 3 + [_c_int] * 8 + [_c_vp],
    "vidi_probe_mfma": [_c_vp, _c_vp, _c_int, _c_vp],
    "vidi_probe_hbm_read": [_c_vp, _c_vp, _c_ll, _c_vp],
    "vidi_gemv_norm2": [_c_vp] * 7 + [_c_ll, _c_f, _c_vp, _c_vp] + [_c_int] * 6 + [_c_vp],
    "vidi_gemv_glu_norm2": [_c_vp] * 7 + [_c_ll, _c_f, _c_vp, _c_vp] + [_c_int] * 7 + [_c_vp],
    "vidi_gemm_f32": [_c_vp] * 4 + [_c_int] * 7 + [_c_vp],
    "vidi_attn_self": [_c_vp] * 3 + [_c_int] * 8 + [_c_f, _c_int, _c_vp],
    "vidi_attn_self_rm": [_c_vp] * 2 + [_c_int] * 5 + [_c_ll] * 4 + [_c_int, _c_f, _c_int, _c_vp],
    "vidi_gemm_ln_heads": [_c_vp] * 6 + [_c_int] * 9 + [_c_vp],
    "vidi_attn_cross": [_c_vp] * 6 + [_c_int] * 9 + [_c_f, _c_f, _c_int, _c_int, _c_vp],
    "vidi_attn_merge": [_c_vp] * 5 + [_c_int] * 9 + [_c_vp],
    "vidi_attn_merge2": [_c_vp] * 3 + [_c_int] * 2 + [_c_vp] * 3 + [_c_int] * 2 + [_c_int] * 7 + [_c_vp],
    "vidi_attn_merge2_sharded": ([_c_vp] * 2 + [_c_ll] * 2 + [_c_vp] * 3 + [_c_int] * 2) * 2 + [_c_int] * 8 + [_c_vp],
    "vidi_attn_cross2": [_c_vp] * 3 + ([_c_vp] * 3 + [_c_int] * 3) * 2 + [_c_int] * 7 + [_c_f, _c_f, _c_int, _c_vp],
    "vidi_attn_text_decode": [_c_vp, _c_int] + [_c_vp] * 6 + [_c_int] * 6 + [_c_vp, _c_int, _c_f, _c_f, _c_int, _c_vp],
    "vidi_attn_text_decode_merge2": [_c_vp, _c_int] + [_c_vp] * 6 + [_c_int] * 6 + [_c_vp, _c_int, _c_f, _c_f] +
                                    ([_c_vp] * 3 + [_c_int] * 2) * 2 + [_c_int] * 4 + [_c_vp],
    "vidi_attn_text": [_c_vp] * 5 + [_c_int] * 8 + [_c_f, _c_f, _c_int, _c_vp],
    "vidi_attn_text_dyn": [_c_vp] * 5 + [_c_int] * 6 + [_c_vp, _c_int, _c_f, _c_f, _c_int, _c_vp],
    "vidi_rope": [_c_vp] * 4 + [_c_int] * 5 + [_c_vp],
    "vidi_rope_cache": [_c_vp, _c_int] + [_c_vp] * 5 + [_c_int] * 7 + [_c_vp, _c_int, _c_vp],
    "vidi_norm": [_c_int] + [_c_vp] * 7 + [_c_int] * 2 + [_c_ll] * 3 + [_c_f, _c_f, _c_vp, _c_int, _c_vp],
    "vidi_resid_norm2": [_c_vp] * 8 + [_c_int, _c_int, _c_ll, _c_f, _c_int, _c_vp],
    "vidi_scale": [_c_vp] * 2 + [_c_ll, _c_f, _c_int, _c_vp],
    "vidi_any_nonzero": [_c_vp, _c_ll, _c_vp, _c_int, _c_vp],
    "vidi_im2col_patch": [_c_vp] * 2 + [_c_int] * 5 + [_c_vp],
    "vidi_patch_embed": [_c_vp] * 5 + [_c_int] * 9 + [_c_vp],
    "vidi_conv_window": [_c_vp] * 3 + [_c_int] * 8 + [_c_vp],
    "vidi_pool_s2d": [_c_vp] * 2 + [_c_int] * 8 + [_c_vp],
    "vidi_add_pos": [_c_vp] * 4 + [_c_int] * 5 + [_c_vp],
    "vidi_add3": [_c_vp] * 4 + [_c_ll, _c_int, _c_vp],
    "vidi_embed": [_c_vp] * 3 + [_c_int, _c_int, _c_ll, _c_f, _c_int, _c_vp],
    "vidi_geglu_unpack": [_c_vp] * 2 + [_c_int] * 3 + [_c_vp],
    "vidi_softcap_argmax": [_c_vp] * 2 + [_c_int, _c_int, _c_ll, _c_f, _c_int, _c_vp, _c_vp],
    "vidi_mel_transpose_pad": [_c_vp] * 2 + [_c_int] * 4 + [_c_vp],
    "vidi_sinusoid": [_c_vp] * 2 + [_c_int] * 5 + [_c_vp],
    "vidi_resize_h_u8": [_c_vp] * 4 + [_c_ll, _c_int, _c_int, _c_int, _c_int, _c_vp],
    "vidi_resize_v_u8_norm": [_c_vp] * 5 + [_c_int] * 7 + [_c_vp],
    "vidi_reflect_pad_f32": [_c_vp] * 2 + [_c_int] * 4 + [_c_vp],
    "vidi_power_spectrum_f32": [_c_vp] * 2 + [_c_ll, _c_int, _c_int, _c_int, _c_vp],
    "vidi_logmel_finish": [_c_vp] * 3 + [_c_int] * 5 + [_c_vp],
}

_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen the C-ABI library and declare prototypes.  Raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: build it with `python -m vidi_amd.build` (hipcc --offload-arch=gfx950). "
            "vidi_amd has no CPU or PyTorch fallback for its kernels.")
    lib = ctypes.CDLL(p)
    lib.vidi_abi_version.restype = _c_int
    lib.vidi_build_info.restype = ctypes.c_char_p
    lib.vidi_attn_cross_workspace_bytes.restype = ctypes.c_size_t
    lib.vidi_attn_cross_workspace_bytes.argtypes = [_c_int] * 4
    lib.vidi_softcap_argmax_workspace_bytes.restype = ctypes.c_size_t
    lib.vidi_softcap_argmax_workspace_bytes.argtypes = [_c_int]
    lib.vidi_gemm_skinny_workspace_bytes.restype = ctypes.c_size_t
    lib.vidi_gemm_skinny_workspace_bytes.argtypes = [_c_int] * 3
    lib.vidi_attn_cross_row_tiles_per_block.restype = _c_int
    lib.vidi_attn_cross_row_tiles_per_block.argtypes = [_c_int, _c_f, _c_int]
    lib.vidi_gemv_mfma_fits.restype = _c_int
    lib.vidi_gemv_mfma_fits.argtypes = [_c_int] * 4
    lib.vidi_stat_strips.restype = _c_int
    lib.vidi_stat_strips.argtypes = [_c_int]
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _c_int
        setattr(lib, name, _Timed(fn, name))
    if path is None:
        _lib = lib
    return lib


# ---------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream
# ---------------------------------------------------------------------------------------------
TIMER = None      # set to a KernelTimer to time every C-ABI launch
AFTER_CALL = None  # test hook (tests/conftest.py under VIDI_CANARY=2): callable(name) run after every C-ABI launch — the guard-zone check


def _work(name, a):
    """(family, algorithmic work, unit) of one launch — DESIGN.md 'algorithmic work per launch'."""
    if name == "vidi_gemm":
        return "gemm", 2.0 * a[5] * a[6] * a[7] * a[16], "flop"
    if name == "vidi_gemm_geglu":
        return "gemm", 2.0 * a[3] * (2 * a[4]) * a[5], "flop"
    if name == "vidi_gemm_skinny":
        return "gemm", 2.0 * a[5] * a[6] * a[7], "flop"
    if name == "vidi_patch_embed":                      # the convolution's products: T (S/P)^2 patches x N x 3 P^2 (not the loader's padded K)
        return "gemm", 2.0 * a[5] * (a[6] // a[7]) ** 2 * a[8] * 3 * a[7] * a[7], "flop"
    if name == "vidi_conv_window":                      # T (side-k+1)^2 outputs x N x k^2 C
        return "gemm", 2.0 * a[3] * (a[4] - a[6] + 1) ** 2 * a[7] * a[6] * a[6] * a[5], "flop"
    if name == "vidi_gemm_qkv_vt":
        return "gemm", 2.0 * a[5] * a[6] * a[7], "flop"
    if name == "vidi_gemm_qkv_vt_ln":
        return "gemm", 2.0 * a[7] * a[8] * a[9], "flop"
    if name in ("vidi_gemm_ln", "vidi_gemm_res_stats"):
        return "gemm", 2.0 * a[6] * a[7] * a[8], "flop"
    if name == "vidi_row_stats":
        return "norm", float(a[2]) * a[3] * 2, "byte"
    if name == "vidi_gemm_kv_cache":
        return "gemm", 2.0 * a[5] * (2 * a[6]) * a[7], "flop"
    if name == "vidi_attn_self":
        return "attn_self", 4.0 * a[4] * a[4] * a[7] * a[6] * a[3], "flop"
    if name == "vidi_attn_self_rm":
        return "attn_self", 4.0 * a[3] * a[3] * a[5] * a[4] * a[2], "flop"
    if name == "vidi_gemm_ln_heads":
        return "gemm", 2.0 * a[6] * a[7] * a[8], "flop"
    if name == "vidi_attn_cross":
        return "attn_cross", float(a[14]) * 2 * a[9] * a[10] * 2, "byte"
    if name == "vidi_attn_cross2":
        return "attn_cross", float(a[7] + a[13]) * 2 * a[18] * a[19] * 2, "byte"
    if name == "vidi_norm":
        return "norm", float(a[8]) * a[9] * 2 * 2, "byte"
    if name == "vidi_resid_norm2":
        return "norm", float(a[8]) * a[9] * 2 * (2 + (a[1] is not None) + (a[2] is not None) + 2), "byte"
    if name == "vidi_gemv":
        return "gemv", float(a[4]) * a[5] * 2, "byte"
    if name == "vidi_gemv_glu":
        return "gemv", 2.0 * a[4] * a[5] * 2, "byte"
    if name == "vidi_gemv_mfma":
        return "gemv", (2.0 if a[9] >= 0 else 1.0) * a[4] * a[5] * 2, "byte"
    if name == "vidi_gemv_norm2":
        return "gemv", float(a[12]) * a[13] * 2, "byte"
    if name == "vidi_gemv_glu_norm2":
        return "gemv", 2.0 * a[12] * a[13] * 2, "byte"
    if name == "vidi_gemm_f32":
        return "gemm_f32", 2.0 * a[4] * a[5] * a[6], "flop"
    if name == "vidi_resize_h_u8":                      # read every source byte once, write the uint8 intermediate once
        return "resize_h", float(a[4]) * (a[5] * 3 + a[7]), "byte"
    if name == "vidi_resize_v_u8_norm":                 # read the intermediate once, write the planar output once
        return "resize_v_norm", float(a[5]) * (a[6] * a[9] + 3.0 * a[8] * a[7] * a[11]), "byte"
    return "other", 0.0, "none"


def _alg_bytes(name, a):
    """Algorithmic HBM bytes of one GEMM-family launch: X + W + Y once each (2-byte elements)."""
    if name == "vidi_gemm":
        return 2.0 * (a[5] * a[7] + a[6] * a[7] + a[5] * a[6]) * a[16]
    if name == "vidi_gemm_geglu":
        return 2.0 * (a[3] * a[5] + 2 * a[4] * a[5] + a[3] * a[4])
    if name == "vidi_gemm_skinny":
        return 2.0 * (a[5] * a[7] + a[6] * a[7] + a[5] * a[6])
    if name == "vidi_conv_window":                      # features + weight + output
        return 2.0 * (a[3] * a[4] * a[4] * a[5] + a[7] * a[6] * a[6] * a[5] + a[3] * (a[4] - a[6] + 1) ** 2 * a[7])
    if name == "vidi_patch_embed":                      # pixels + weight + output
        return 2.0 * (a[5] * 3 * a[6] * a[6] + a[8] * a[9] + a[5] * (a[6] // a[7]) ** 2 * a[8])
    if name == "vidi_gemm_qkv_vt":
        return 2.0 * (a[5] * a[7] + a[6] * a[7] + a[5] * a[6])
    if name == "vidi_gemm_qkv_vt_ln":
        return 2.0 * (a[7] * a[9] + a[8] * a[9] + a[7] * a[8])
    if name == "vidi_gemm_ln" or name == "vidi_gemm_ln_heads":
        return 2.0 * (a[6] * a[8] + a[7] * a[8] + a[6] * a[7])
    if name == "vidi_gemm_res_stats":
        return 2.0 * (a[6] * a[8] + a[7] * a[8] + 2 * a[6] * a[7])
    if name == "vidi_gemm_kv_cache":
        return 2.0 * (a[5] * a[7] + 2 * a[6] * a[7] + 2 * a[5] * a[6])
    return 0.0


class _Timed:
    def __init__(self, fn, name):
        self.fn, self.name = fn, name

    def __call__(self, *a):
        t = TIMER
        rc = self.fn(*a) if t is None else t.run(self.name, a, self.fn)
        if AFTER_CALL is not None:
            AFTER_CALL(self.name)
        return rc


class KernelTimer:
    """Brackets every launch with torch.cuda.Event pairs recorded on the current stream (the stream the
    kernels are enqueued on).  summary() synchronises once and aggregates per kernel family."""

    def __init__(self):
        self.rec = []

    def run(self, name, a, fn):
        fam, work, unit = _work(name, a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*a)
        e1.record()
        self.rec.append((fam, work, unit, e0, e1, _alg_bytes(name, a)))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, work, unit, e0, e1, nbytes in self.rec:
            d = out.setdefault(fam, {"launches": 0, "ms": 0.0, "work": 0.0, "unit": unit, "bytes": 0.0})
            d["bytes"] += nbytes
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["work"] += work
        return out


class VidiHipError(RuntimeError):
    pass


_ERR = {-1: "bad shape", -2: "bad dtype", -3: "bad alignment", -4: "bad argument"}


def _check(rc: int, op: str):
    if rc != 0:
        raise VidiHipError(f"{op} failed: {_ERR.get(rc, f'hipError {rc}')}")


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float16:
        return DT_F16
    raise VidiHipError(f"unsupported dtype {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise VidiHipError("vidi_amd kernels need device tensors (no CPU path)")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rowmajor(t: torch.Tensor, name: str):
    if t.stride(-1) != 1:
        raise VidiHipError(f"{name}: last dimension must be contiguous")


# ---------------------------------------------------------------------------------------------
# projections
# ---------------------------------------------------------------------------------------------
def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, *,
         act: int = ACT_NONE, residual: Optional[torch.Tensor] = None, rmod: int = 0,
         repkv: Optional[tuple] = None, tile_cfg: int = -1, K: Optional[int] = None,
         M: Optional[int] = None, ldx: Optional[int] = None,
         batch: int = 1, bsX: int = 0, bsY: int = 0, bsR: int = 0) -> torch.Tensor:
    """out[m,n] = epi(x[m,:] . w[n,:] + bias[n]) — 2-D row-major views (arbitrary leading stride).
    `K`, `M`, `ldx` override the logical shape for overlapping-row (conv-as-GEMM) views."""
    lib = load_library()
    _rowmajor(x, "x"); _rowmajor(w, "w")
    Mv = M if M is not None else x.shape[0]
    N = w.shape[0]
    Kv = K if K is not None else (w.shape[1] if repkv is None else w.shape[1])
    ldxv = ldx if ldx is not None else x.stride(0)
    if out is None:
        out = torch.empty((Mv, N), dtype=x.dtype, device=x.device) if batch == 1 else None
    if out is None:
        raise VidiHipError("batched gemm needs an explicit out")
    ldy = out.stride(-2)
    ldr = residual.stride(-2) if residual is not None else 0
    rc = lib.vidi_gemm(_p(x), _p(w), _p(bias), _p(out), _p(residual), Mv, N, Kv, ldxv, w.stride(0), ldy, ldr, rmod,
                       bsX, bsY, bsR, batch, act, repkv[0] if repkv else 0, repkv[1] if repkv else 0,
                       tile_cfg, _dt(x), _stream())
    _check(rc, "vidi_gemm")
    return out


def gemm_geglu(x: torch.Tensor, wgu: torch.Tensor, out: Optional[torch.Tensor] = None, tile_cfg: int = -1) -> torch.Tensor:
    lib = load_library()
    M, K = x.shape
    I = wgu.shape[0] // 2
    if out is None:
        out = torch.empty((M, I), dtype=x.dtype, device=x.device)
    _check(lib.vidi_gemm_geglu(_p(x), _p(wgu), _p(out), M, I, K, x.stride(0), wgu.stride(0), out.stride(0),
                               tile_cfg, _dt(x), _stream()), "vidi_gemm_geglu")
    return out


def gemm_glu(x: torch.Tensor, wgu: torch.Tensor, out: Optional[torch.Tensor] = None, act: int = ACT_GELU_TANH,
             tile_cfg: int = -1) -> torch.Tensor:
    """Gated MLP front half, act(gate) * up fused in the epilogue: ACT_GELU_TANH (Gemma2) or ACT_SILU (Mistral)."""
    lib = load_library()
    M, K = x.shape
    I = wgu.shape[0] // 2
    if out is None:
        out = torch.empty((M, I), dtype=x.dtype, device=x.device)
    _check(lib.vidi_gemm_glu(_p(x), _p(wgu), _p(out), M, I, K, x.stride(0), wgu.stride(0), out.stride(0),
                             act, tile_cfg, _dt(x), _stream()), "vidi_gemm_glu")
    return out


def gemm_qkv_vt(x, w, bias, yqk, vt, *, vstart, hd, seq, seqpad, nheads, tile_cfg=-1):
    lib = load_library()
    M, K = x.shape
    _check(lib.vidi_gemm_qkv_vt(_p(x), _p(w), _p(bias), _p(yqk), _p(vt), M, w.shape[0], K, x.stride(0), w.stride(0),
                                yqk.stride(0), vstart, hd, seq, seqpad, nheads, tile_cfg, _dt(x), _stream()), "vidi_gemm_qkv_vt")


def row_stats(x: torch.Tensor, stats: torch.Tensor, eps: float) -> torch.Tensor:
    """stats[m] = (mean, rstd) of LayerNorm over row m of x — consumed by gemm_ln / gemm_qkv_vt_ln (LayerNorm folded into the projection)"""
    _rowmajor(x, "x")
    rows, H = x.shape
    if stats.dtype != torch.float32 or stats.numel() < 2 * rows or not stats.is_contiguous():
        raise VidiHipError("row_stats: stats must be a contiguous fp32 buffer of at least 2*rows elements")
    _check(load_library().vidi_row_stats(_p(x), _p(stats), rows, H, x.stride(0), float(eps), _dt(x), _stream()), "vidi_row_stats")
    return stats


def gemm_res_stats(x, w, bias, out, residual, part, *, tile_cfg: int = -1):
    """out = x w^T + bias + residual, plus part[m][entry] = (sum, sum of squares) of the stored row values per column group — stat_strips(N)
    entries per row (the next LayerNorm's statistics without a pass over `out`: ln_finalize)"""
    _rowmajor(x, "x"); _rowmajor(w, "w")
    M, K = x.shape
    N = w.shape[0]
    if part.dtype != torch.float32 or not part.is_contiguous() or part.numel() < 2 * M * stat_strips(N):
        raise VidiHipError("gemm_res_stats: `part` must be a contiguous fp32 buffer of 2 * M * stat_strips(N) elements")
    _check(load_library().vidi_gemm_res_stats(_p(x), _p(w), _p(bias), _p(out), _p(residual), _p(part), M, N, K, x.stride(0), w.stride(0),
                                              out.stride(0), residual.stride(0), tile_cfg, _dt(x), _stream()), "vidi_gemm_res_stats")
    return out


def stat_strips(N: int) -> int:
    """(sum, sum of squares) entries per row of the `part` buffer of gemm_res_stats / ln_finalize for an output width N"""
    return int(load_library().vidi_stat_strips(int(N)))


def ln_finalize(part, stats, rows: int, N: int, eps: float):
    _check(load_library().vidi_ln_finalize(_p(part), _p(stats), rows, N, float(eps), _stream()), "vidi_ln_finalize")
    return stats


def _ln_vecs(stats, colsum, shift, M, N):
    for t, n, name in ((stats, 2 * M, "stats"), (colsum, N, "colsum"), (shift, N, "shift")):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < n:
            raise VidiHipError(f"folded LayerNorm: `{name}` must be a contiguous fp32 tensor of at least {n} elements")


def gemm_ln(x, wf, stats, colsum, shift, out=None, *, act: int = ACT_NONE, tile_cfg: int = -1):
    """out = act(Linear(LayerNorm(x))) with the LayerNorm folded: wf = W * gamma, colsum[n] = sum_k wf[n,k], shift = W beta + bias"""
    lib = load_library()
    _rowmajor(x, "x"); _rowmajor(wf, "wf")
    M, K = x.shape
    N = wf.shape[0]
    _ln_vecs(stats, colsum, shift, M, N)
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _check(lib.vidi_gemm_ln(_p(x), _p(wf), _p(stats), _p(colsum), _p(shift), _p(out), M, N, K, x.stride(0), wf.stride(0), out.stride(0),
                            act, tile_cfg, _dt(x), _stream()), "vidi_gemm_ln")
    return out


def gemm_qkv_vt_ln(x, wf, stats, colsum, shift, yqk, vt, *, vstart, hd, seq, seqpad, nheads, tile_cfg=-1):
    lib = load_library()
    M, K = x.shape
    _ln_vecs(stats, colsum, shift, M, wf.shape[0])
    _check(lib.vidi_gemm_qkv_vt_ln(_p(x), _p(wf), _p(stats), _p(colsum), _p(shift), _p(yqk), _p(vt), M, wf.shape[0], K, x.stride(0),
                                   wf.stride(0), yqk.stride(0), vstart, hd, seq, seqpad, nheads, tile_cfg, _dt(x), _stream()),
           "vidi_gemm_qkv_vt_ln")


def gemm_kv_cache(x, wkv, kc, vtc, vrow, *, kvd, hd, ntile64, tok0, tile_cfg=-1):
    lib = load_library()
    M, K = x.shape
    _check(lib.vidi_gemm_kv_cache(_p(x), _p(wkv), _p(kc), _p(vtc), _p(vrow), M, kvd, K, x.stride(0), wkv.stride(0),
                                  hd, ntile64, tok0, tile_cfg, _dt(x), _stream()), "vidi_gemm_kv_cache")


def gemv(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = load_library()
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _check(lib.vidi_gemv(_p(x), _p(w), _p(out), M, N, K, x.stride(0), w.stride(0), out.stride(0), _dt(x), _stream()), "vidi_gemv")
    return out


def probe_box(buf: torch.Tensor, mfma_ms: float = 250.0) -> dict:
    """Box-speed reference (csrc/probe.hip, frozen): sustained bf16 MFMA rate of a register-operand loop and the HBM read rate of one sweep
    over `buf` (a resident tensor of a few GB), each the best of 3 after a warm-up.  Synchronises the device."""
    lib = load_library()
    dev = buf.device
    g = torch.Generator(device="cpu").manual_seed(11)
    ops = (torch.randn(4096 * 8, generator=g) * 2.0 ** -6).to(torch.bfloat16).to(dev)
    out = torch.empty(2048 * 512, dtype=torch.float32, device=dev)
    flop_per_iter = 2048 * 8 * 32 * 16384.0

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        return e0.elapsed_time(e1)
    run = lambda it: _check(lib.vidi_probe_mfma(_p(ops), _p(out), int(it), _stream()), "vidi_probe_mfma")          # noqa: E731
    ms1 = timed(lambda: run(2000))
    iters = max(2000, int(2000 * mfma_ms / max(ms1, 1e-3)))
    mf = max(flop_per_iter * iters / (timed(lambda: run(iters)) * 1e-3) / 1e12 for _ in range(3))
    flat = buf.contiguous().view(-1)
    nbytes = (flat.numel() * flat.element_size()) // 16 * 16
    sweep = lambda: _check(lib.vidi_probe_hbm_read(_p(flat), _p(out), nbytes, _stream()), "vidi_probe_hbm_read")    # noqa: E731
    timed(sweep)
    bw = max(nbytes / (timed(sweep) * 1e-3) / 1e9 for _ in range(3))
    return {"mfma_bf16_16x16x32_TFLOP/s": mf, "hbm_read_GB/s": bw, "hbm_read_bytes": nbytes, "mfma_ms": mfma_ms,
            "note": "frozen reference kernels (vidi_amd/csrc/probe.hip): divide figures of different boxes by these to compare builds"}


def attn_cross_row_tiles_per_block(Rpad: int, softcap: Optional[float], dtype: torch.dtype) -> int:
    """32-row tiles one block of vidi_attn_cross / vidi_attn_cross2 covers at Rpad rows per kv head (1 or 4): callers size `zsplit` with it"""
    return int(load_library().vidi_attn_cross_row_tiles_per_block(int(Rpad), float(softcap or 0.0), DT_BF16 if dtype == torch.bfloat16 else DT_F16))


def gemv_mfma_fits(M: int, N: int, K: int, glu: bool = False, x: Optional[torch.Tensor] = None, w: Optional[torch.Tensor] = None) -> bool:
    """True iff vidi_gemv_mfma WILL take the call: the library's shape predicate and — when the operands are given — the leading-dimension
    (multiples of 8 elements) and 16-byte base-alignment rules its dispatcher enforces (engine.proj falls back to gemv / gemm_skinny otherwise)"""
    if not bool(load_library().vidi_gemv_mfma_fits(int(M), int(N), int(K), int(glu))):
        return False
    for t in (x, w):
        if t is not None and (t.stride(0) % 8 or t.data_ptr() % 16 or t.stride(-1) != 1):
            return False
    return True


def gemv_mfma(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, glu_act: int = -1) -> torch.Tensor:
    """a batch of decode rows on the matrix pipe (csrc/gemv_mfma.hip): out = x @ w.T for 1 <= M <= 32 rows, or with `glu_act` the gated pair
    act(x Wg^T) * (x Wu^T) on the interleaved gate/up weight (M <= 16)"""
    lib = load_library()
    M, K = x.shape
    N = w.shape[0] // 2 if glu_act >= 0 else w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _check(lib.vidi_gemv_mfma(_p(x), _p(w), _p(out), M, N, K, x.stride(0), w.stride(0), out.stride(0), int(glu_act), _dt(x), _stream()), "vidi_gemv_mfma")
    return out


def gemm_skinny_workspace_bytes(M: int, N: int, K: int) -> int:
    """bytes of fp32 split-K workspace vidi_gemm_skinny needs for (M, N, K); 0: the shape is not taken (use gemm / gemv)"""
    lib = load_library()
    return int(lib.vidi_gemm_skinny_workspace_bytes(int(M), int(N), int(K)))


def gemm_skinny(x: torch.Tensor, w: torch.Tensor, workspace: torch.Tensor, out: Optional[torch.Tensor] = None,
                bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = x @ w.T (+ bias) for 1 <= M <= 128 rows: split-K weight streaming (csrc/gemm_skinny.h); `workspace`: a tensor of at least
    gemm_skinny_workspace_bytes(M, N, K) bytes owned by the caller (the library keeps no state)"""
    lib = load_library()
    M, K = x.shape
    N = w.shape[0]
    need = gemm_skinny_workspace_bytes(M, N, K)
    if need == 0:
        raise VidiHipError(f"vidi_gemm_skinny does not take M={M}, N={N}, K={K}")
    if workspace.numel() * workspace.element_size() < need:
        raise VidiHipError(f"vidi_gemm_skinny: workspace of {workspace.numel() * workspace.element_size()} bytes, {need} needed")
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _check(lib.vidi_gemm_skinny(_p(x), _p(w), _p(bias), _p(out), _p(workspace), M, N, K, x.stride(0), w.stride(0), out.stride(0), _dt(x), _stream()),
           "vidi_gemm_skinny")
    return out


def gemv_norm2_fits(M: int, K: int) -> bool:
    return M <= 4 and K <= 4096 and K % 8 == 0


def gemv_norm2(a, b, c, res, w1, w2, y1, w, out, *, eps: float):
    """resid_norm2(a, b, c, res, w1, w2) -> (y1, x) and out = x @ w.T in one launch (decode; y1 must not alias res)"""
    lib = load_library()
    M, K = a.shape
    N = w.shape[0]
    for t in (a, b, c, res, y1):
        if t is not None and (t.stride(0) != a.stride(0) or t.stride(1) != 1):
            raise VidiHipError("gemv_norm2: all row tensors must share the row stride")
    _check(lib.vidi_gemv_norm2(_p(a), _p(b), _p(c), _p(res), _p(w1), _p(w2), _p(y1), a.stride(0), float(eps), _p(w), _p(out), M, N, K,
                               w.stride(0), out.stride(0), _dt(a), _stream()), "vidi_gemv_norm2")
    return out


def gemv_glu_norm2(a, b, c, res, w1, w2, y1, wgu, out, *, eps: float, act: int = ACT_GELU_TANH):
    """resid_norm2(...) -> (y1, x) and out = act(x Wg^T) * (x Wu^T) on the interleaved gate/up weight in one launch"""
    lib = load_library()
    M, K = a.shape
    I = wgu.shape[0] // 2
    for t in (a, b, c, res, y1):
        if t is not None and (t.stride(0) != a.stride(0) or t.stride(1) != 1):
            raise VidiHipError("gemv_glu_norm2: all row tensors must share the row stride")
    _check(lib.vidi_gemv_glu_norm2(_p(a), _p(b), _p(c), _p(res), _p(w1), _p(w2), _p(y1), a.stride(0), float(eps), _p(wgu), _p(out), M, I, K,
                                   wgu.stride(0), out.stride(0), act, _dt(a), _stream()), "vidi_gemv_glu_norm2")
    return out


def gemv_glu(x: torch.Tensor, wgu: torch.Tensor, out: torch.Tensor, act: int = ACT_GELU_TANH) -> torch.Tensor:
    """out[m, i] = act(gate_i . x_m) * (up_i . x_m) for M <= 8 rows on the interleaved gate/up weight (gemv + unpack in one launch)"""
    lib = load_library()
    M, K = x.shape
    I = wgu.shape[0] // 2
    _check(lib.vidi_gemv_glu(_p(x), _p(wgu), _p(out), M, I, K, x.stride(0), wgu.stride(0), out.stride(0), act, _dt(x), _stream()),
           "vidi_gemv_glu")
    return out


def gemm_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE) -> torch.Tensor:
    lib = load_library()
    assert x.dtype == torch.float32 and w.dtype == torch.float32
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _check(lib.vidi_gemm_f32(_p(x), _p(w), _p(bias), _p(out), M, N, K, x.stride(0), w.stride(0), out.stride(0), act, _stream()),
           "vidi_gemm_f32")
    return out


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
def attn_self(qk, vt, out, *, B, N, Npad, H, D, koff, scale):
    lib = load_library()
    _check(lib.vidi_attn_self(_p(qk), _p(vt), _p(out), B, N, Npad, H, D, qk.stride(0), koff, out.stride(0), float(scale),
                              _dt(qk), _stream()), "vidi_attn_self")


def attn_self_rm(qkv, out, *, B, N, H, D, scale, koff=None, voff=None, head_major=False):
    """encoder self-attention reading Q | K | V (V in natural order, transposed on the fly by the LDS transpose read) from one projection
    output: row-major [B*N, ld] with column offsets koff / voff, or head-major [3][B][H][N][D] (gemm_ln_heads).  scale <= 0: Q already
    carries scale * log2(e) (include/vidi_hip.h)"""
    lib = load_library()
    if head_major:
        if not qkv.is_contiguous() or qkv.numel() < 3 * B * H * N * D:
            raise VidiHipError("attn_self_rm: head-major input must be a contiguous [3][B][H][N][D] buffer")
        per = B * H * N * D
        args = (D, per, 2 * per, H * N * D, N * D)
    else:
        _rowmajor(qkv, "qkv")
        args = (qkv.stride(0), koff, voff, 0, 0)
    _check(lib.vidi_attn_self_rm(_p(qkv), _p(out), B, N, H, D, *args, out.stride(0), float(scale), _dt(qkv), _stream()), "vidi_attn_self_rm")


def gemm_ln_heads(x, wf, stats, colsum, shift, out, *, seq: int, hd: int, tile_cfg: int = -1):
    """gemm_ln (no activation) writing the q | k | v projection head-major: out[which][frame][head][token][d]"""
    lib = load_library()
    _rowmajor(x, "x"); _rowmajor(wf, "wf")
    M, K = x.shape
    N = wf.shape[0]
    _ln_vecs(stats, colsum, shift, M, N)
    if not out.is_contiguous() or out.numel() < M * N:
        raise VidiHipError("gemm_ln_heads: `out` must be a contiguous buffer of M * N elements")
    _check(lib.vidi_gemm_ln_heads(_p(x), _p(wf), _p(stats), _p(colsum), _p(shift), _p(out), M, N, K, x.stride(0), wf.stride(0), seq, hd,
                                  tile_cfg, _dt(x), _stream()), "vidi_gemm_ln_heads")
    return out


def attn_cross_workspace(zsplit: int, nkv: int, Rpad: int, HD: int, device) -> tuple:
    W = zsplit
    opart = torch.empty((W, nkv, Rpad, HD), dtype=torch.float32, device=device)
    ml = torch.empty((W, nkv, Rpad, 2), dtype=torch.float32, device=device)
    return opart, ml


def attn_cross(q, kc, vtc, mask, opart, ml, *, R, Rpad, G, nkv, HD, ntile64, key_start, n_keys, scale, softcap, zsplit):
    lib = load_library()
    _check(lib.vidi_attn_cross(_p(q), _p(kc), _p(vtc), _p(mask), _p(opart), _p(ml), R, Rpad, G, nkv, HD, q.stride(0),
                               ntile64, key_start, n_keys, float(scale), float(softcap or 0.0), zsplit, _dt(q), _stream()),
           "vidi_attn_cross")


def attn_cross2(q, kc, vtc, set_a, set_b, *, R, Rpad, G, nkv, HD, ntile64, scale, softcap):
    """T2V and T2A of one layer in one launch; set_x = dict(mask, opart, ml, key_start, n_keys, zsplit)"""
    lib = load_library()
    sa, sb = set_a, set_b
    _check(lib.vidi_attn_cross2(_p(q), _p(kc), _p(vtc),
                                _p(sa["mask"]), _p(sa["opart"]), _p(sa["ml"]), sa["key_start"], sa["n_keys"], sa["zsplit"],
                                _p(sb["mask"]), _p(sb["opart"]), _p(sb["ml"]), sb["key_start"], sb["n_keys"], sb["zsplit"],
                                R, Rpad, G, nkv, HD, q.stride(0), ntile64, float(scale), float(softcap or 0.0), _dt(q), _stream()),
           "vidi_attn_cross2")


def attn_merge(opart, ml, out, *, W, nkv, R, Rpad, G, HD, zero_out=False, out_f32=None, out_ml=None, dtype=None):
    lib = load_library()
    # out_f32 [nkv,Rpad,HD] / out_ml [nkv,Rpad,2]: merged result in partial form (for the cross-GPU merge)
    ldo = out.stride(0) if out is not None else 0
    dt = _dt(out) if out is not None else (dtype if dtype is not None else DT_BF16)
    _check(lib.vidi_attn_merge(_p(opart), _p(ml), _p(out), _p(out_f32), _p(out_ml), W, nkv, R, Rpad, G, HD, ldo,
                               1 if zero_out else 0, dt, _stream()), "vidi_attn_merge")


def attn_merge2(opart_a, ml_a, out_a, W_a, zero_a, opart_b, ml_b, out_b, W_b, zero_b, *, nkv, R, Rpad, G, HD):
    """two attn_merge calls (image and audio partials of one layer) in one launch"""
    lib = load_library()
    assert out_a.stride(0) == out_b.stride(0)
    _check(lib.vidi_attn_merge2(_p(opart_a), _p(ml_a), _p(out_a), W_a, 1 if zero_a else 0, _p(opart_b), _p(ml_b), _p(out_b), W_b,
                                1 if zero_b else 0, nkv, R, Rpad, G, HD, out_a.stride(0), _dt(out_a), _stream()), "vidi_attn_merge2")


def attn_merge2_sharded(sets, *, nkv, R, Rpad, rpo, G, HD, ldo, dtype):
    """Both modalities' merges of one layer in one launch, with explicit partial strides (frame-sharded keys).
    `sets`: two entries (T2V, T2A), each None (modality absent) or a dict with
       opart, ml     fp32 partial buffers (may be None when W == 0: the neutral partial of a rank without keys)
       ws_o, ws_ml   floats between successive partials;  W: number of partials
       out           [tokens, ldo] model-dtype output or None;  out_f32 / out_ml: partial-form outputs (row stride rpo) or None
       zero          sample has no valid key (gemma.py:180-192)"""
    lib = load_library()
    args = []
    for s_ in sets:
        if s_ is None:
            args += [None, None, 0, 0, None, None, None, 0, 0]
        else:
            args += [_p(s_.get("opart")), _p(s_.get("ml")), int(s_["ws_o"]), int(s_["ws_ml"]), _p(s_.get("out")), _p(s_.get("out_f32")),
                     _p(s_.get("out_ml")), int(s_["W"]), 1 if s_.get("zero") else 0]
    _check(lib.vidi_attn_merge2_sharded(*args, nkv, R, Rpad, rpo, G, HD, ldo, dtype, _stream()), "vidi_attn_merge2_sharded")


def attn_text(q, kc, vc, kmask, out, *, B, Lq, Lmax, nq, nkv, HD, past_len, window, scale, softcap):
    lib = load_library()
    _check(lib.vidi_attn_text(_p(q), _p(kc), _p(vc), _p(kmask), _p(out), B, Lq, Lmax, nq, nkv, HD, past_len, window,
                              float(scale), float(softcap or 0.0), _dt(q), _stream()), "vidi_attn_text")


def attn_text_dyn(q, kc, vc, kmask, out, *, B, Lq, Lmax, nq, nkv, HD, past_len_dev, window, scale, softcap):
    """attn_text with the cache length in device memory (int32[1]): capturable in a hipGraph."""
    if past_len_dev.dtype != torch.int32 or not past_len_dev.is_cuda:
        raise VidiHipError("past_len_dev must be a device int32 tensor")
    lib = load_library()
    _check(lib.vidi_attn_text_dyn(_p(q), _p(kc), _p(vc), _p(kmask), _p(out), B, Lq, Lmax, nq, nkv, HD, _p(past_len_dev), window,
                                  float(scale), float(softcap or 0.0), _dt(q), _stream()), "vidi_attn_text_dyn")


def attn_text_decode_fits(*, nq, nkv, HD, Lmax, window, pos0=None) -> bool:
    """whether vidi_attn_text_decode's LDS plan (query rows, scores of the visible keys, PV partials) fits its 64 KB"""
    G = nq // nkv
    nvis = Lmax if pos0 is None else pos0 + 1
    if window and window > 0:
        nvis = min(nvis, window + 1)
    lcap = (nvis + 3) & ~3
    return HD in (64, 128, 256) and G <= 8 and (G * HD + 8 + G * lcap + (256 // (HD // 8)) * G * HD) * 4 <= 64 * 1024


def attn_text_decode(qkv, kc, vc, kmask, cos, sin, out, *, B, Lmax, nq, nkv, HD, window, scale, softcap, pos0=0, pos_dev=None):
    """Lq = 1: rope(q), rope(k), KV-cache append and the T2T attention in one launch (rope_cache + attn_text[_dyn])"""
    lib = load_library()
    if pos_dev is not None and (pos_dev.dtype != torch.int32 or not pos_dev.is_cuda):
        raise VidiHipError("attn_text_decode: pos_dev must be a CUDA int32 tensor")
    _check(lib.vidi_attn_text_decode(_p(qkv), qkv.stride(0), _p(kc), _p(vc), _p(kmask), _p(cos), _p(sin), _p(out), B, Lmax, nq, nkv, HD,
                                     int(pos0), _p(pos_dev), int(window), float(scale), float(softcap or 0.0), _dt(qkv), _stream()),
           "vidi_attn_text_decode")


def attn_text_decode_merge2(qkv, kc, vc, kmask, cos, sin, out, merge_a, merge_b, *, B, Lmax, nq, nkv, HD, window, scale, softcap, R, Rpad,
                            pos0=0, pos_dev=None):
    """attn_text_decode + attn_merge2 in one launch; merge_x = (opart, ml, out, W, zero_out)"""
    lib = load_library()
    if pos_dev is not None and (pos_dev.dtype != torch.int32 or not pos_dev.is_cuda):
        raise VidiHipError("attn_text_decode_merge2: pos_dev must be a CUDA int32 tensor")
    (oa, mla, outa, wa, za), (ob, mlb, outb, wb, zb) = merge_a, merge_b
    if outa.stride(0) != outb.stride(0):
        raise VidiHipError("attn_text_decode_merge2: the two merge outputs must share the row stride")
    _check(lib.vidi_attn_text_decode_merge2(_p(qkv), qkv.stride(0), _p(kc), _p(vc), _p(kmask), _p(cos), _p(sin), _p(out), B, Lmax, nq, nkv, HD,
                                            int(pos0), _p(pos_dev), int(window), float(scale), float(softcap or 0.0),
                                            _p(oa), _p(mla), _p(outa), wa, 1 if za else 0, _p(ob), _p(mlb), _p(outb), wb, 1 if zb else 0,
                                            R, Rpad, outa.stride(0), _dt(qkv), _stream()), "vidi_attn_text_decode_merge2")


def rope_cache(qkv, qr, kc, vc, cos, sin, *, B, Lq, Lmax, nq, nkv, HD, pos0=0, pos_dev=None):
    """rope(q) -> qr, rope(k) / v -> cache slots pos0.. (or *pos_dev..) of kc/vc [B, Lmax, nkv*HD], in one launch"""
    lib = load_library()
    if pos_dev is not None and (pos_dev.dtype != torch.int32 or not pos_dev.is_cuda):
        raise VidiHipError("pos_dev must be a device int32 tensor")
    _rowmajor(qkv, "qkv")
    _check(lib.vidi_rope_cache(_p(qkv), qkv.stride(0), _p(qr), _p(kc), _p(vc), _p(cos), _p(sin), B, Lq, Lmax, nq, nkv, HD, int(pos0),
                               _p(pos_dev), _dt(qkv), _stream()), "vidi_rope_cache")


def rope(q, k, cos, sin, *, rows, nq, nkv, HD):
    lib = load_library()
    _check(lib.vidi_rope(_p(q), _p(k), _p(cos), _p(sin), rows, nq, nkv, HD, _dt(q), _stream()), "vidi_rope")


# ---------------------------------------------------------------------------------------------
# norms / elementwise
# ---------------------------------------------------------------------------------------------
def norm(mode: int, x: Optional[torch.Tensor], weight: Optional[torch.Tensor], *, eps: float, out: Optional[torch.Tensor] = None,
         bias=None, residual=None, mask_out=None, x_f32=None, normalizer: float = 1.0, sample_flag: Optional[torch.Tensor] = None,
         dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    lib = load_library()
    src = x if x is not None else x_f32
    H = src.shape[-1]
    rows = src.numel() // H
    if out is None:
        out = torch.empty(src.shape, dtype=(x.dtype if x is not None else dtype), device=src.device)
    ldx = src.stride(-2) if src.dim() > 1 else H
    ldy = out.stride(-2) if out.dim() > 1 else H
    ldr = (residual.stride(-2) if residual.dim() > 1 else H) if residual is not None else 0
    _check(lib.vidi_norm(mode, _p(x), _p(x_f32), _p(weight), _p(bias), _p(residual), _p(out), _p(mask_out), rows, H,
                         ldx, ldy, ldr, float(eps), float(normalizer), _p(sample_flag), _dt(out), _stream()), "vidi_norm")
    return out


def resid_norm2(a, b, c, res, w1, w2, y1, y2, *, eps: float):
    """y1 = T(res + T(gemma(T(T(a+b)+c); w1))), y2 = T(gemma(y1; w2)) — add3 + norm_add + norm in one launch (y1 may alias res)"""
    lib = load_library()
    rows, H = a.shape
    for t in (a, b, c, res, y1, y2):
        if t is not None and (t.stride(0) != a.stride(0) or t.stride(1) != 1):
            raise VidiHipError("resid_norm2: all row tensors must share the row stride")
    _check(lib.vidi_resid_norm2(_p(a), _p(b), _p(c), _p(res), _p(w1), _p(w2), _p(y1), _p(y2), rows, H, a.stride(0), float(eps), _dt(a),
                                _stream()), "vidi_resid_norm2")
    return y1, y2


def patch_embed_weight(w: torch.Tensor, P: int) -> torch.Tensor:
    """conv weight [N, 3, P, P] -> the GEMM weight vidi_patch_embed reads: [N, K], k = (c*P + dy)*16 + dx, zero elsewhere, K % 64 == 0"""
    N = w.shape[0]
    K = (3 * P * 16 + 63) // 64 * 64
    out = torch.zeros((N, K), dtype=w.dtype, device=w.device)
    out[:, : 3 * P * 16].view(N, 3 * P, 16)[:, :, :P] = w.reshape(N, 3 * P, P)
    return out


def patch_embed(px, w, bias, pos, out, *, T, S, P):
    """SigLIP patch embedding (conv + bias + position table) straight from the NCHW pixels; `w` from patch_embed_weight()"""
    _rowmajor(w, "w")
    _check(load_library().vidi_patch_embed(_p(px), _p(w), _p(bias), _p(pos), _p(out), T, S, P, w.shape[0], w.shape[1], w.stride(0),
                                           out.stride(0), pos.stride(0), _dt(px), _stream()), "vidi_patch_embed")
    return out


def conv_window(f, w, out, *, T, side, C, k):
    """Conv2d(C, N, k, stride 1, valid) over token-major features f [T, side*side, C] -> out [T*(side-k+1)^2, N]; w [N, k*k*C] in (dy, dx, c) order"""
    _rowmajor(w, "w")
    _check(load_library().vidi_conv_window(_p(f), _p(w), _p(out), T, side, C, k, w.shape[0], w.stride(0), out.stride(0), _dt(f), _stream()),
           "vidi_conv_window")
    return out


def im2col_patch(px, out, *, T, S, P, Kpad):
    _check(load_library().vidi_im2col_patch(_p(px), _p(out), T, S, P, Kpad, _dt(px), _stream()), "vidi_im2col_patch")


def pool_s2d(f, out, *, T, side, C, h, w, m, resize):
    _check(load_library().vidi_pool_s2d(_p(f), _p(out), T, side, C, h, w, m, 1 if resize else 0, _dt(f), _stream()), "vidi_pool_s2d")


def add_pos(f, ph, pw, pt, *, T, oh, ow, H):
    _check(load_library().vidi_add_pos(_p(f), _p(ph), _p(pw), _p(pt), T, oh, ow, H, _dt(f), _stream()), "vidi_add_pos")


def add3(a, b, c, out):
    _check(load_library().vidi_add3(_p(a), _p(b), _p(c), _p(out), a.numel(), _dt(a), _stream()), "vidi_add3")
    return out


def embed(ids, E, out, *, normalizer):
    n = ids.numel()
    _check(load_library().vidi_embed(_p(ids), _p(E), _p(out), n, E.shape[1], E.shape[0], float(normalizer), _dt(E), _stream()),
           "vidi_embed")
    return out


def geglu_unpack(yp, out):
    M, I2 = yp.shape
    _check(load_library().vidi_geglu_unpack(_p(yp), _p(out), M, I2 // 2, _dt(yp), _stream()), "vidi_geglu_unpack")
    return out


def glu_unpack(yp, out, act=ACT_GELU_TANH):
    M, I2 = yp.shape
    _check(load_library().vidi_glu_unpack(_p(yp), _p(out), M, I2 // 2, act, _dt(yp), _stream()), "vidi_glu_unpack")
    return out


def im2col_nhwc(x, out, *, T, side, C, k):
    _check(load_library().vidi_im2col_nhwc(_p(x), _p(out), T, side, C, k, _dt(x), _stream()), "vidi_im2col_nhwc")
    return out


def resize_bilinear_ac(x, out, *, T, s_in, s_out, C):
    _check(load_library().vidi_resize_bilinear_ac(_p(x), _p(out), T, s_in, s_out, C, _dt(x), _stream()), "vidi_resize_bilinear_ac")
    return out


def softcap_argmax_workspace(B: int, device) -> torch.Tensor:
    """zeroed caller-owned scratch of vidi_softcap_argmax for up to B rows (every call leaves it zeroed)"""
    n = load_library().vidi_softcap_argmax_workspace_bytes(int(B))
    return torch.zeros((n // 8,), dtype=torch.int64, device=device)


_am_ws = {}


def softcap_argmax(logits, idx, cap, workspace: Optional[torch.Tensor] = None):
    """workspace: from softcap_argmax_workspace (one per stream that may run concurrently).  Without one, a zeroed scratch cached per
    (device, stream) is used — allocated eagerly on first use, so pass an explicit workspace to calls that are graph-captured."""
    B, V = logits.shape
    if workspace is None:
        key = (logits.device.index, _stream())
        workspace = _am_ws.get(key)
        if workspace is None or workspace.numel() * 8 < 16 * B:
            if torch.cuda.is_current_stream_capturing():
                raise VidiHipError("vidi_softcap_argmax: pass a pre-allocated workspace when capturing a graph")
            workspace = _am_ws[key] = softcap_argmax_workspace(max(B, 64), logits.device)
    elif workspace.numel() * workspace.element_size() < 16 * B:
        raise VidiHipError("vidi_softcap_argmax: workspace too small")
    _check(load_library().vidi_softcap_argmax(_p(logits), _p(idx), B, V, logits.stride(0), float(cap or 0.0), _dt(logits), _p(workspace),
                                              _stream()), "vidi_softcap_argmax")
    return idx


def mel_transpose_pad(mel, out):
    C, nmel, L = mel.shape
    _check(load_library().vidi_mel_transpose_pad(_p(mel), _p(out), C, nmel, L, _dt(mel), _stream()), "vidi_mel_transpose_pad")
    return out


def scale(x, out, s: float):
    _check(load_library().vidi_scale(_p(x), _p(out), x.numel(), float(s), _dt(x), _stream()), "vidi_scale")
    return out


def any_nonzero(x, flag):
    """flag (int32[1], zeroed by the caller) |= any(x != 0)"""
    _check(load_library().vidi_any_nonzero(_p(x), x.numel(), _p(flag), _dt(x), _stream()), "vidi_any_nonzero")
    return flag


def sinusoid(pe, div, *, rows, i0, l, N, d):
    _check(load_library().vidi_sinusoid(_p(pe), _p(div), rows, i0, l, N, d, _stream()), "vidi_sinusoid")
    return pe


# ---------------------------------------------------------------------------------------------
# preprocessing (csrc/preproc.hip): PIL-exact frame resize + normalise, Whisper log-mel pieces
# ---------------------------------------------------------------------------------------------
def _out_dt(dtype: torch.dtype) -> int:
    return {torch.bfloat16: DT_BF16, torch.float16: DT_F16, torch.float32: DT_F32}[dtype]


def resize_h_u8(frames: torch.Tensor, tmp: torch.Tensor, bounds: torch.Tensor, kk: torch.Tensor):
    """frames [T,H0,W0,3] u8 -> tmp [T,H0,pitch] u8 holding OW x 3 bytes per row (Pillow horizontal pass)"""
    lib = load_library()
    T, H0, W0, C = frames.shape
    assert C == 3 and frames.dtype == torch.uint8 and tmp.dtype == torch.uint8 and frames.is_contiguous() and tmp.is_contiguous()
    OW, pitch = bounds.shape[0], tmp.shape[2]
    assert bounds.dtype == torch.int32 and kk.dtype == torch.int32 and bounds.shape == (OW, 2) and kk.shape[0] == OW
    _check(lib.vidi_resize_h_u8(_p(frames), _p(tmp), _p(bounds), _p(kk), T * H0, W0, OW, pitch, kk.shape[1], _stream()), "vidi_resize_h_u8")


def resize_v_u8_norm(tmp: torch.Tensor, out: torch.Tensor, bounds: torch.Tensor, kk: torch.Tensor, lut: torch.Tensor):
    """tmp [T,H0,pitch] u8 -> out [T,3,OH,OW] (Pillow vertical pass + per-channel value table + HWC->CHW)"""
    lib = load_library()
    T, H0, pitch = tmp.shape
    OH, OW = out.shape[2], out.shape[3]
    assert out.shape == (T, 3, OH, OW) and out.is_contiguous() and lut.shape == (3, 256) and lut.dtype == out.dtype
    assert bounds.shape == (OH, 2) and kk.shape[0] == OH
    _check(lib.vidi_resize_v_u8_norm(_p(tmp), _p(out), _p(bounds), _p(kk), _p(lut), T, H0, OW, OH, pitch, kk.shape[1], out.element_size(),
                                     _stream()), "vidi_resize_v_u8_norm")


def reflect_pad_f32(wave: torch.Tensor, out: torch.Tensor, pad: int):
    lib = load_library()
    C, n = wave.shape
    assert wave.dtype == torch.float32 and out.dtype == torch.float32 and wave.is_contiguous() and out.is_contiguous()
    _check(lib.vidi_reflect_pad_f32(_p(wave), _p(out), C, n, pad, out.shape[1], _stream()), "vidi_reflect_pad_f32")


def power_spectrum_f32(Y: torch.Tensor, P: torch.Tensor, nf: int):
    lib = load_library()
    _check(lib.vidi_power_spectrum_f32(_p(Y), _p(P), Y.shape[0], nf, Y.stride(0), P.stride(0), _stream()), "vidi_power_spectrum_f32")


def logmel_finish(mel: torch.Tensor, cmax: torch.Tensor, out: torch.Tensor, *, C: int, R: int, F: int):
    lib = load_library()
    nmel = mel.shape[-1]
    assert mel.dtype == torch.float32 and cmax.dtype == torch.float32 and out.shape == (C, nmel, F) and out.is_contiguous()
    _check(lib.vidi_logmel_finish(_p(mel), _p(cmax), _p(out), C, R, F, nmel, _out_dt(out.dtype), _stream()), "vidi_logmel_finish")
