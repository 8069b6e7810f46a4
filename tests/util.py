"""Shared helpers for the parity tests (layout packers mirror the kernels' documented layouts)."""
import os

import numpy as np
import torch


def perm16(x):
    return 8 * ((x >> 2) & 1) + (x & 3) + 4 * (x >> 3)


def perm_positions(n):
    """storage position of key k inside a perm16-permuted axis of length n (n % 16 == 0)"""
    k = np.arange(n)
    return (k & ~15) | perm16(k & 15)


def seeded(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def logit_tol(dt, ref):
    """Absolute tolerance on logits, a fraction of their spread: 5 % (bf16) / 1 % (fp16) of std(ref), no relative part.  Sized from the audit of
    what the kernels use (profiles/r4_tolerance_audit.jsonl: worst logit error 4.7 % of the spread for bf16 — Vidi-7B's text-only prompt — and
    3.3 % typically; 0.6 % for fp16); round 4 allowed 7 % / 1.2 %.  |err| <= tol on every logit implies the same argmax wherever the
    reference's top-2 margin exceeds 2 x tol.  tests/test_gpu_reference_golden.py::test_a_two_percent_kernel_error_is_caught shows what an
    error of 2 % in one kernel's output does to the checks built on this bound."""
    return (5e-2 if dt == torch.bfloat16 else 1e-2) * float(ref.float().std())


def report(name, got, ref, atol, rtol):
    """assert closeness with a diagnostic that says WHERE and HOW the mismatch looks."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    if not torch.isfinite(got).all():
        bad = (~torch.isfinite(got)).nonzero()
        raise AssertionError(f"{name}: {bad.shape[0]} non-finite values, first at {bad[0].tolist()}")
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    nbad = int((err > tol).sum())
    log = os.environ.get("VIDI_TEST_REPORT")
    if log:                     # slack audit: how much of each tolerance the kernels actually use (tolerances are kept <= ~2x observed)
        import json
        used = float((err / tol.clamp_min(1e-30)).max()) if got.numel() else 0.0
        with open(log, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "name": name, "max_err": float(err.max()) if got.numel() else 0.0,
                                "atol": atol, "rtol": rtol, "tol_used": used, "ref_rms": float(ref.pow(2).mean().sqrt()) if got.numel() else 0.0}) + "\n")
    if nbad:
        idx = (err - tol).argmax()
        pos = np.unravel_index(int(idx), got.shape)
        corr = torch.corrcoef(torch.stack([got.flatten(), ref.flatten()]))[0, 1].item() if got.numel() > 2 else float("nan")
        raise AssertionError(
            f"{name}: {nbad}/{got.numel()} outside tol (atol={atol}, rtol={rtol}); max|err|={err.max():.4g} at {pos} "
            f"got={got[pos]:.5g} ref={ref[pos]:.5g}; ref rms={ref.pow(2).mean().sqrt():.4g}; corr={corr:.4f}")
    return float(err.max())


def pack_vt(v, npad):
    """v:[B,N,H,D] -> Vt:[B,H,D,npad] with the perm16 key order (what vidi_gemm_qkv_vt writes)."""
    B, N, H, D = v.shape
    out = torch.zeros((B, H, D, npad), dtype=v.dtype)
    pos = torch.from_numpy(perm_positions(npad))[:N]
    out[:, :, :, pos] = v.permute(0, 2, 3, 1)
    return out


def unpack_vt(vt, n):
    pos = torch.from_numpy(perm_positions(vt.shape[-1]))[:n]
    return vt[:, :, :, pos].permute(0, 3, 1, 2)          # [B,N,H,D]


def pack_kv_cache(k, v, ntile64, tok0=0):
    """k,v:[N,nkv,hd] -> Kc[nkv,ntile64,64,hd], Vtc[nkv,2*ntile64,hd,32(perm16)]."""
    N, nkv, hd = k.shape
    kc = torch.zeros((nkv, ntile64 * 64, hd), dtype=k.dtype)
    vt = torch.zeros((nkv, 2 * ntile64, hd, 32), dtype=v.dtype)
    kc[:, tok0: tok0 + N] = k.permute(1, 0, 2)
    tok = np.arange(tok0, tok0 + N)
    tile = torch.from_numpy(tok >> 5)
    pos = torch.from_numpy(perm_positions(32)[tok & 31])
    vt[:, tile, :, pos] = v            # advanced indices (tile,pos) separated by a slice -> result [N,nkv,hd]
    return kc.view(nkv, ntile64, 64, hd), vt
