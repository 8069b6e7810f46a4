"""Per-kernel parity: every C-ABI entry point against the CPU oracle (oracle/vidi_oracle.py) on seeded
inputs.  Inputs are rounded to the storage dtype first, the oracle then runs in fp32 on those values.

Tolerances (stated per test): a kernel output is one rounding (bf16: 2^-8 rel, fp16: 2^-11 rel) away
from the fp32 result plus accumulation-order noise, so  atol = 1e-2*rms(ref)-ish, rtol = 1e-2 (bf16) /
2e-3 (fp16) unless noted (`tol`).  Integer / index outputs are bit-exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vidi_oracle as O
from util import pack_kv_cache, pack_vt, perm_positions, report, seeded, unpack_vt

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]


def tol(dt, scale=1.0, k=1.0):
    """(atol, rtol): 1 % of `scale` + 1 % relative for bf16 (0.2 % / 0.2 % fp16) — about 2x what the kernels use (audited through
    VIDI_TEST_REPORT, tests/util.py: 0.25-0.5 of this bound); k=2 for the kernels with a second rounding inside (fused activations,
    softmax probabilities rounded to the dtype before PV)"""
    return (1e-2 * scale * k, 1e-2 * k) if dt == torch.bfloat16 else (2e-3 * scale * k, 2e-3 * k)


@pytest.fixture(scope="module")
def hip():
    from vidi_amd import hip as h
    h.load_library()
    return h


def dev(t):
    return t.cuda()


# ---------------------------------------------------------------------------------------------
# GEMM family
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 11])
def test_gemm_plain_tiles(hip, dt, cfg):
    """asymmetric operands, M/N not multiples of the tile, bias"""
    M, N, K = 300, 352, 192
    x = seeded((M, K), 1, dtype=dt); w = seeded((N, K), 2, 0.1, dtype=dt); b = seeded((N,), 3, dtype=dt)
    ref = F.linear(x.float(), w.float(), b.float())
    y = hip.gemm(dev(x), dev(w), dev(b), tile_cfg=cfg)
    report(f"gemm cfg{cfg}", y, ref, *tol(dt, ref.std().item()))


@pytest.mark.parametrize("cfg", [0, 1, 2, 4, 5, 11])
def test_gemm_large_k_and_auto(hip, cfg):
    dt = torch.bfloat16
    M, N, K = 1000, 1152, 4352
    x = seeded((M, K), 4, dtype=dt); w = seeded((N, K), 5, 0.02, dtype=dt)
    ref = x.float() @ w.float().T
    y = hip.gemm(dev(x), dev(w), None, tile_cfg=cfg)
    report(f"gemm largeK cfg{cfg}", y, ref, *tol(dt, ref.std().item()))
    y2 = hip.gemm(dev(x), dev(w), None, tile_cfg=-1)
    report("gemm auto", y2, ref, *tol(dt, ref.std().item()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("act", ["tanh", "erf"])
def test_gemm_act_residual(hip, dt, act):
    from vidi_amd import hip as H
    M, N, K, rmod = 200, 128, 128, 50
    x = seeded((M, K), 6, dtype=dt); w = seeded((N, K), 7, 0.1, dtype=dt); b = seeded((N,), 8, dtype=dt)
    r = seeded((rmod, N), 9, dtype=dt)
    lin = F.linear(x.float(), w.float(), b.float())
    a = O.gelu_tanh(lin) if act == "tanh" else O.gelu_erf(lin)
    ref = a + r.float().repeat(M // rmod, 1)
    y = hip.gemm(dev(x), dev(w), dev(b), act=H.ACT_GELU_TANH if act == "tanh" else H.ACT_GELU_ERF, residual=dev(r), rmod=rmod)
    report("gemm act+res", y, ref, *tol(dt, ref.std().item(), k=2))
    # in-place residual (out aliases residual), as the encoder layers use it
    res = dev(seeded((M, N), 10, dtype=dt))
    ref2 = lin + res.float().cpu()
    hip.gemm(dev(x), dev(w), dev(b), res, residual=res)
    report("gemm inplace residual", res, ref2, *tol(dt, ref2.std().item()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("bias,act,res", [(0, "", 0), (1, "", 0), (1, "", 1), (1, "", 2), (1, "tanh", 0), (1, "erf", 0), (1, "erf", 2), (0, "", 1)])
def test_gemm_w4_epilogues(hip, dt, bias, act, res):
    """persistent 4-wave kernel (tile_cfg 5): every instantiated epilogue combination, ragged M / N edges, 4 K slices
    (main loop + both tail bodies); `res` 1 = row residual (in place), 2 = position table (rows repeat every rmod)"""
    from vidi_amd import hip as H
    M, N, K, rmod = 600, 416, 256, 100
    x = seeded((M, K), 31, dtype=dt); w = seeded((N, K), 32, 0.1, dtype=dt)
    b = seeded((N,), 33, dtype=dt) if bias else None
    lin = F.linear(x.float(), w.float(), None if b is None else b.float())
    lin = lin.to(dt).float()                                                  # module output rounds before the activation
    a = O.gelu_tanh(lin) if act == "tanh" else O.gelu_erf(lin) if act == "erf" else lin
    a = a.to(dt).float()
    kw = {"act": {"": H.ACT_NONE, "tanh": H.ACT_GELU_TANH, "erf": H.ACT_GELU_ERF}[act], "tile_cfg": 5}
    if res == 1:
        r = seeded((M, N), 34, dtype=dt)
        out = dev(r).clone()
        hip.gemm(dev(x), dev(w), None if b is None else dev(b), out, residual=out, **kw)
        ref = a + r.float()
    elif res == 2:
        r = seeded((rmod, N), 35, dtype=dt)
        out = hip.gemm(dev(x), dev(w), None if b is None else dev(b), residual=dev(r), rmod=rmod, **kw)
        ref = a + r.float().repeat(M // rmod, 1)
    else:
        out = hip.gemm(dev(x), dev(w), None if b is None else dev(b), **kw)
        ref = a
    report(f"w4 bias={bias} act={act} res={res}", out, ref, *tol(dt, ref.std().item()))


def test_gemm_w4_tile_loop_matches_8wave_bitwise(hip):
    """more tiles than CUs, so every persistent block walks several tiles (next-tile DMA issued before the epilogue, first
    k32 step with C = 0): the output must equal the 8-wave kernel's bit for bit (same accumulation order), both tile orders"""
    import os
    dt = torch.bfloat16
    M, N, K = 256 * 21 + 40, 256 * 15 + 96, 192
    x = seeded((M, K), 36, dtype=dt); w = seeded((N, K), 37, 0.1, dtype=dt)
    y4 = hip.gemm(dev(x), dev(w), None, tile_cfg=4)
    y5 = hip.gemm(dev(x), dev(w), None, tile_cfg=5)
    assert torch.equal(y4, y5)
    ref = x.float() @ w.float().T
    report("w4 tile loop", y5, ref, *tol(dt, ref.std().item()))


def test_gemm_w4_batched_overlapping_rows(hip):
    """conv-as-GEMM view through the persistent kernel: ldx < K (rows overlap), batch strides, GELU(erf) + position table"""
    from vidi_amd import hip as H
    dt = torch.bfloat16
    C, L, nm, Da = 5, 300, 64, 320
    buf = seeded((C, L + 2, nm), 38, dtype=dt)
    w = seeded((Da, 3 * nm), 39, 0.1, dtype=dt); b = seeded((Da,), 40, dtype=dt); pos = seeded((L, Da), 41, dtype=dt)
    out = torch.zeros((C, L, Da), dtype=dt).cuda()
    hip.gemm(dev(buf)[0], dev(w), dev(b), out, act=H.ACT_GELU_ERF, residual=dev(pos), rmod=L, M=L, K=3 * nm, ldx=nm,
             batch=C, bsX=(L + 2) * nm, bsY=L * Da, bsR=0, tile_cfg=5)
    rows = torch.stack([buf[:, t: t + 3].reshape(C, -1) for t in range(L)], dim=1).float()
    lin = (rows @ w.float().T + b.float()).to(dt).float()
    ref = O.gelu_erf(lin).to(dt).float() + pos.float()
    report("w4 batched overlap", out, ref, *tol(dt, ref.std().item()))


def test_gemm_batched_overlapping_rows(hip):
    """conv-as-GEMM view: ldx < K (rows overlap), batch strides — the Whisper stem pattern"""
    dt = torch.bfloat16
    C, L, nm, Da = 3, 40, 64, 64
    buf = seeded((C, L + 2, nm), 11, dtype=dt)
    w = seeded((Da, 3 * nm), 12, 0.1, dtype=dt); b = seeded((Da,), 13, dtype=dt)
    out = torch.zeros((C, L, Da), dtype=dt).cuda()
    hip.gemm(dev(buf)[0], dev(w), dev(b), out, M=L, K=3 * nm, ldx=nm, batch=C, bsX=(L + 2) * nm, bsY=L * Da)
    rows = torch.stack([buf[:, t: t + 3].reshape(C, -1) for t in range(L)], dim=1).float()
    ref = rows @ w.float().T + b.float()
    report("gemm batched overlap", out, ref, *tol(dt, ref.std().item()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [-1, 2, 4, 5])
def test_gemm_repkv(hip, dt, cfg):
    nkv, G, hd, H, M = 2, 2, 128, 256, 150
    v = seeded((M, nkv * hd), 14, dtype=dt); wo = seeded((H, nkv * G * hd), 15, 0.05, dtype=dt)
    vrep = O.repeat_kv(v.float().view(1, M, nkv, hd).transpose(1, 2), G).transpose(1, 2).reshape(M, -1)
    ref = vrep @ wo.float().T
    y = hip.gemm(dev(v), dev(wo), None, repkv=(hd, G), K=G * nkv * hd, tile_cfg=cfg)
    report("gemm repkv", y, ref, *tol(dt, ref.std().item()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [0, 1, 2, 4, 5])
def test_gemm_geglu(hip, dt, cfg):
    M, I, K = 200, 256, (192 if cfg == 5 else 128)
    x = seeded((M, K), 16, dtype=dt); g = seeded((I, K), 17, 0.1, dtype=dt); u = seeded((I, K), 18, 0.1, dtype=dt)
    wgu = torch.stack([g.view(I // 32, 32, K), u.view(I // 32, 32, K)], dim=1).reshape(2 * I, K).contiguous()
    ref = O.gelu_tanh(x.float() @ g.float().T) * (x.float() @ u.float().T)
    y = hip.gemm_geglu(dev(x), dev(wgu), tile_cfg=cfg)
    report("gemm geglu", y, ref, *tol(dt, ref.std().item(), k=2))


@pytest.mark.parametrize("cfg", [0, 4, 5])
def test_gemm_glu_gelu_equals_geglu(hip, cfg):
    """vidi_gemm_glu(act = GELU_TANH) is vidi_gemm_geglu: p.act selects the gate function only (the 8-wave kernel once
    applied it a second time in its copy-out pass)"""
    from vidi_amd import hip as H
    dt = torch.bfloat16
    M, I, K = 200, 256, 192
    x = seeded((M, K), 16, dtype=dt); g = seeded((I, K), 17, 0.1, dtype=dt); u = seeded((I, K), 18, 0.1, dtype=dt)
    wgu = torch.stack([g.view(I // 32, 32, K), u.view(I // 32, 32, K)], dim=1).reshape(2 * I, K).contiguous()
    y0 = hip.gemm_geglu(dev(x), dev(wgu), tile_cfg=cfg)
    y1 = hip.gemm_glu(dev(x), dev(wgu), act=H.ACT_GELU_TANH, tile_cfg=cfg)
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("cfg", [-1, 2, 4, 5])
@pytest.mark.parametrize("hd,N,nh", [(72, 729, 4), (16, 49, 4), (64, 50, 2)])
def test_gemm_qkv_vt(hip, hd, N, nh, cfg):
    dt = torch.bfloat16
    B, Hd = 2, nh * hd
    K = 64 if (Hd <= 64 and cfg != 5) else 192           # the persistent 4-wave kernel needs K >= 192
    Npad = (N + 63) // 64 * 64
    x = seeded((B * N, K), 19, dtype=dt); w = seeded((3 * Hd, K), 20, 0.1, dtype=dt); b = seeded((3 * Hd,), 21, dtype=dt)
    if (3 * Hd) % 32:
        pytest.skip("N must be a multiple of 32")
    ref = F.linear(x.float(), w.float(), b.float())
    yqk = torch.zeros((B * N, 2 * Hd), dtype=dt).cuda()
    vt = torch.zeros((B, nh, hd, Npad), dtype=dt).cuda()
    hip.gemm_qkv_vt(dev(x), dev(w), dev(b), yqk, vt, vstart=2 * Hd, hd=hd, seq=N, seqpad=Npad, nheads=nh, tile_cfg=cfg)
    report("qkv_vt QK", yqk, ref[:, : 2 * Hd], *tol(dt, ref.std().item()))
    v = unpack_vt(vt.cpu(), N).reshape(B * N, Hd)
    report("qkv_vt V", v, ref[:, 2 * Hd:], *tol(dt, ref.std().item()))


def _fold_ln(w, b, gamma, beta, dt):
    """host-side folding as vidi_amd/engine.py does it (include/vidi_hip.h: vidi_gemm_ln)"""
    wf = (w.float() * gamma.float()[None, :]).to(dt)
    return wf, wf.float().sum(1).contiguous(), (w.float() @ beta.float() + b.float()).contiguous()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("rows,H", [(7, 64), (300, 1152), (130, 1280), (5, 3584)])
def test_row_stats(hip, dt, rows, H):
    x = (seeded((rows, H), 60, 2.0) + seeded((rows, 1), 61, 3.0)).to(dt)          # rows with a non-zero mean
    st = torch.zeros(2 * rows, dtype=torch.float32).cuda()
    hip.row_stats(dev(x), st, 1e-6)
    xf = x.float()
    mean = xf.mean(1)
    rstd = torch.rsqrt(xf.var(1, unbiased=False) + 1e-6)
    got = st.cpu().view(rows, 2)
    report("row_stats mean", got[:, 0], mean, 1e-5, 1e-5)
    report("row_stats rstd", got[:, 1], rstd, 1e-6, 1e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("act", ["none", "tanh", "erf"])
@pytest.mark.parametrize("M,N,K,cfg", [(300, 352, 192, 0), (12800, 1152, 1152, -1), (9000, 4352, 1152, 5), (70, 192, 64, -1)])
def test_gemm_ln_equals_layernorm_then_linear(hip, dt, act, M, N, K, cfg):
    """LayerNorm folded into the projection (vidi_row_stats + vidi_gemm_ln) against LayerNorm -> Linear -> act evaluated in fp32 on the
    same rounded inputs; rows carry a mean of the order of their spread and a few outlier channels (what the towers' residual streams
    look like), gamma/beta are non-trivial.  Both the small-problem 128x128 tile and the persistent 4-wave kernel."""
    from vidi_amd import hip as H
    x = seeded((M, K), 62, 1.0) + seeded((M, 1), 63, 1.5)
    x[:, 5] *= 20.0; x[:, K // 2] *= -12.0
    x = x.to(dt)
    w = seeded((N, K), 64, 0.05, dtype=dt); b = seeded((N,), 65, 0.3, dtype=dt)
    gamma = (1.0 + seeded((K,), 66, 0.2)).to(dt); beta = seeded((K,), 67, 0.2, dtype=dt)
    h = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-6)
    ref = F.linear(h, w.float(), b.float())
    ref = {"none": ref, "tanh": O.gelu_tanh(ref), "erf": O.gelu_erf(ref)}[act]
    wf, cs, sh = _fold_ln(w, b, gamma, beta, dt)
    st = torch.zeros(2 * M, dtype=torch.float32).cuda()
    hip.row_stats(dev(x), st, 1e-6)
    y = hip.gemm_ln(dev(x), dev(wf), st, dev(cs), dev(sh), act={"none": H.ACT_NONE, "tanh": H.ACT_GELU_TANH, "erf": H.ACT_GELU_ERF}[act], tile_cfg=cfg)
    # TWO independent roundings — the output and the folded weight Wf = T(W * gamma) (the reference rounds LayerNorm(x) instead) — on rows
    # with outlier channels: k = 1.5 (sqrt(2) rounded up) of the single-kernel bound, 2 with an activation behind it
    report(f"gemm_ln act={act}", y, ref, *tol(dt, ref.std().item(), k=2 if act != "none" else 1.5))




@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K,cfg", [(12900, 1152, 1152, -1), (12900, 1152, 4352, 5), (10001, 1280, 1280, -1), (300, 352, 192, -1), (70, 64, 64, -1),
                                       (13217, 1152, 1152, -1), (200, 1152, 192, -1), (12900, 576, 1152, -1)])
def test_gemm_res_stats_and_finalize(hip, dt, M, N, K, cfg):
    """out_proj / fc2 + residual with the fused emission of the stored rows' partial sums (persistent kernels: 256-wide tiles, or the
    288 x 224 geometry for widths that are multiples of 288 and not of 256 — N = 1 152, 576; M = 13 217 ends in a 1-row tile) or the one-pass
    fallback (small problems), in place (Y aliases R, as the towers call it): Y equals the plain bias + residual GEMM bit for bit,
    and vidi_ln_finalize of the partials equals the two-pass row statistics of Y (row_stats) to fp32 round-off."""
    x = seeded((M, K), 80, dtype=dt); w = seeded((N, K), 81, 0.03, dtype=dt); b = seeded((N,), 82, dtype=dt)
    r = (seeded((M, N), 83, 2.0) + seeded((M, 1), 84, 1.5)).to(dt)
    y_ref = r.clone().cuda()
    hip.gemm(dev(x), dev(w), dev(b), y_ref, residual=y_ref, tile_cfg=cfg)
    y = r.clone().cuda()
    nstr = hip.stat_strips(N)
    part = torch.full((2 * M * nstr,), float("nan"), dtype=torch.float32).cuda()
    hip.gemm_res_stats(dev(x), dev(w), dev(b), y, y, part, tile_cfg=cfg)
    assert torch.equal(y, y_ref)
    p = part.view(M, nstr, 2).cpu()
    assert torch.isfinite(p).all()
    yf = y.float().cpu()
    # the column group behind every entry of a row: 128-column strips — or, for the widths of the 288-wide tile geometry run by its fused
    # kernel, per tile the two 128-column main parts and the two 16-column tails; the one-pass fallback writes strips and pads with zeros
    fused = cfg in (-1, 5) and ((N + 255) // 256) * ((M + 255) // 256) >= 192 and K % 64 == 0 and K >= 192
    if nstr != (N + 127) // 128 and fused:
        assert nstr == 4 * (N // 288)
        groups = []
        for t in range(N // 288):
            groups += [range(288 * t, 288 * t + 128), range(288 * t + 128, 288 * t + 256), range(288 * t + 256, 288 * t + 272), range(288 * t + 272, 288 * t + 288)]
    else:
        groups = [range(128 * i, min(N, 128 * i + 128)) for i in range((N + 127) // 128)] + [range(0)] * (nstr - (N + 127) // 128)
    strips = torch.zeros((M, nstr, 128))
    for i, g in enumerate(groups):
        strips[:, i, : len(g)] = yf[:, list(g)]
    # the fused emission sums the fp32 values before their rounding to the storage dtype: against sums of the STORED values that is
    # 128 independent half-ulp roundings per strip (6-sigma bound below); the fallback pass reads the stored values (exact)
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11           # spacing of the storage dtype relative to the value
    rms, ymax = float(yf.pow(2).mean().sqrt()), float(yf.abs().max())
    six_sigma = 6 * (128 / 12) ** 0.5 * ulp                            # 128 independent roundings, each uniform in +-ulp*|y|/2
    report("partial sums", p[..., 0], strips.sum(-1), six_sigma * ymax + 1e-4 * rms * 128 ** 0.5, 1e-5)
    report("partial sums of squares", p[..., 1], (strips * strips).sum(-1), six_sigma * 2 * ymax * ymax + 1e-4 * rms * rms * 128 ** 0.5, 1e-5)
    st = torch.zeros(2 * M, dtype=torch.float32).cuda(); st2 = torch.zeros(2 * M, dtype=torch.float32).cuda()
    hip.ln_finalize(part, st, M, N, 1e-6)
    hip.row_stats(y, st2, 1e-6)
    a, bb = st.view(M, 2).cpu(), st2.view(M, 2).cpu()
    # (mean, rstd) against the two-pass statistics of the STORED rows: the mean of N roundings moves by 6 sigma = 6 ulp |y|max / sqrt(12 N)
    # (3e-4 of the spread here, bf16), rstd by the same relative amount — an order of magnitude below what the consumer's bf16 output resolves
    mean_tol = 6 * ulp * ymax / (12 * N) ** 0.5
    report("finalize mean", a[:, 0], bb[:, 0], mean_tol + 1e-6, 1e-5)
    report("finalize rstd", a[:, 1], bb[:, 1], 0.0, 1e-3 if dt == torch.bfloat16 else 2e-4)

@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(39, 4096, 3584), (117, 3584, 4096), (39, 3584, 14336), (9, 64, 256), (16, 128, 512), (17, 192, 768), (64, 3584, 4096),
                                   (65, 256, 1024), (128, 1024, 2048), (1, 64, 256), (33, 8192, 3584)])
def test_gemm_skinny(hip, dt, M, N, K):
    """vidi_gemm_skinny (a prompt's 9..128 rows against a weight matrix, split-K, fp32 partial sums in the caller's workspace): against
    the fp32 product of the same bf16 / fp16 operands to the output rounding, and against vidi_gemm on the same operands (both round one
    fp32 sum per element: they may differ by the summation order only)."""
    x = seeded((M, K), 92, dtype=dt); w = seeded((N, K), 93, 0.05, dtype=dt); b = seeded((N,), 94, dtype=dt)
    need = hip.gemm_skinny_workspace_bytes(M, N, K)
    assert need > 0 and need % (4 * M * N) == 0
    ws = torch.full((need // 4 + 16,), float("nan"), dtype=torch.float32).cuda()
    ref = x.float() @ w.float().T
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    for bias in (None, b):
        want = ref if bias is None else ref + bias.float()
        y = hip.gemm_skinny(dev(x), dev(w), ws, bias=None if bias is None else dev(bias))
        assert torch.isnan(ws[need // 4:]).all()                       # nothing written past the stated workspace
        report("skinny vs fp32", y, want, 1e-3 * float(want.abs().max()), ulp)
        y2 = hip.gemm(dev(x), dev(w), None if bias is None else dev(bias))
        report("skinny vs vidi_gemm", y, y2.float(), 1e-3 * float(want.abs().max()), 2 * ulp)
    # a view with a row stride (the engine projects slices of wider buffers) and an output slice
    xw = torch.zeros((M, K + 64), dtype=dt); xw[:, :K] = x
    yw = torch.full((M, N + 32), 7.0, dtype=dt).cuda()
    hip.gemm_skinny(dev(xw)[:, :K], dev(w), ws, out=yw[:, :N])
    report("strided", yw[:, :N], ref, 1e-3 * float(ref.abs().max()), ulp)
    assert bool((yw[:, N:] == 7.0).all())


def test_gemm_skinny_shapes_it_does_not_take(hip):
    assert hip.gemm_skinny_workspace_bytes(129, 4096, 3584) == 0       # rows
    assert hip.gemm_skinny_workspace_bytes(39, 4096 + 32, 3584) == 0   # N % 64
    assert hip.gemm_skinny_workspace_bytes(39, 4096, 3584 + 64) == 0   # K % 256
    x = torch.zeros((39, 320), dtype=torch.bfloat16).cuda(); w = torch.zeros((64, 320), dtype=torch.bfloat16).cuda()
    with pytest.raises(Exception, match="does not take"):
        hip.gemm_skinny(x, w, torch.zeros(1 << 20, dtype=torch.float32).cuda())


@pytest.mark.parametrize("N", [64, 352, 1152, 1280, 3968, 4096, 8192])
@pytest.mark.parametrize("rows", [1, 255, 256, 257, 5000])
def test_ln_finalize_rows_and_widths(hip, rows, N):
    """vidi_ln_finalize alone: block edges (a block finalizes 256 rows), every entry count of vidi_stat_strips (1 .. 16 staged through
    LDS with consecutive lanes on consecutive entries, >= 32 on the row-per-thread kernel), against float64 sums of the same partials"""
    nstr = hip.stat_strips(N)
    part = torch.stack((seeded((rows, nstr), 90, 40.0) + 3.0, seeded((rows, nstr), 91, 30.0).abs() * N / nstr + 50.0), dim=-1).float()
    st = torch.full((2 * rows + 2,), float("nan"), dtype=torch.float32).cuda()
    hip.ln_finalize(part.reshape(-1).cuda(), st, rows, N, 1e-6)
    got = st.cpu()
    assert torch.isnan(got[2 * rows:]).all()                              # nothing written past the last row
    s = part.double().sum(1)
    mean = s[:, 0] / N
    var = (s[:, 1] / N - mean * mean).clamp_min(0)
    report("mean", got[: 2 * rows].view(rows, 2)[:, 0], mean.float(), 1e-6 * float(mean.abs().max()), 1e-6)
    report("rstd", got[: 2 * rows].view(rows, 2)[:, 1], (var + 1e-6).rsqrt().float(), 0.0, 2e-5)


@pytest.mark.parametrize("cfg", [-1, 0, 5])
@pytest.mark.parametrize("hd,N,nh,B", [(72, 729, 4, 2), (16, 49, 4, 2), (64, 1500, 4, 12)])
def test_gemm_qkv_vt_ln(hip, hd, N, nh, B, cfg):
    dt = torch.bfloat16
    Hd = nh * hd
    K = 192
    Npad = (N + 63) // 64 * 64
    if (3 * Hd) % 32:
        pytest.skip("N must be a multiple of 32")
    x = (seeded((B * N, K), 68, 1.0) + seeded((B * N, 1), 69, 1.0)).to(dt)
    w = seeded((3 * Hd, K), 70, 0.1, dtype=dt); b = seeded((3 * Hd,), 71, dtype=dt)
    gamma = (1.0 + seeded((K,), 72, 0.2)).to(dt); beta = seeded((K,), 73, 0.2, dtype=dt)
    ref = F.linear(F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5), w.float(), b.float())
    wf, cs, sh = _fold_ln(w, b, gamma, beta, dt)
    st = torch.zeros(2 * B * N, dtype=torch.float32).cuda()
    hip.row_stats(dev(x), st, 1e-5)
    yqk = torch.zeros((B * N, 2 * Hd), dtype=dt).cuda()
    vt = torch.zeros((B, nh, hd, Npad), dtype=dt).cuda()
    hip.gemm_qkv_vt_ln(dev(x), dev(wf), st, dev(cs), dev(sh), yqk, vt, vstart=2 * Hd, hd=hd, seq=N, seqpad=Npad, nheads=nh, tile_cfg=cfg)
    report("qkv_vt_ln QK", yqk, ref[:, : 2 * Hd], *tol(dt, ref.std().item()))
    v = unpack_vt(vt.cpu(), N).reshape(B * N, Hd)
    report("qkv_vt_ln V", v, ref[:, 2 * Hd:], *tol(dt, ref.std().item()))


@pytest.mark.parametrize("cfg", [-1, 2, 4, 5])
def test_gemm_kv_cache(hip, cfg):
    dt = torch.bfloat16
    nkv, hd, K, M, tok0 = 2, 128, (192 if cfg == 5 else 128), 170, 64
    kvd = nkv * hd
    ntile = (tok0 + M + 63) // 64
    x = seeded((M, K), 22, dtype=dt); w = seeded((2 * kvd, K), 23, 0.1, dtype=dt)
    ref = x.float() @ w.float().T
    kc = torch.zeros((nkv, ntile, 64, hd), dtype=dt).cuda(); vtc = torch.zeros((nkv, 2 * ntile, hd, 32), dtype=dt).cuda()
    vrow = torch.zeros((M, kvd), dtype=dt).cuda()
    hip.gemm_kv_cache(dev(x), dev(w), kc, vtc, vrow, kvd=kvd, hd=hd, ntile64=ntile, tok0=tok0, tile_cfg=cfg)
    kref, vtref = pack_kv_cache(ref[:, :kvd].view(M, nkv, hd), ref[:, kvd:].view(M, nkv, hd), ntile, tok0)
    a, r = tol(dt, ref.std().item())
    report("kv_cache Vrow", vrow, ref[:, kvd:], a, r)
    report("kv_cache Kc", kc, kref, a, r)
    report("kv_cache Vtc", vtc, vtref, a, r)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M", [1, 2, 3, 8])
def test_gemv(hip, dt, M):
    N, K = 1000, 3584
    x = seeded((M, K), 24, dtype=dt); w = seeded((N, K), 25, 0.02, dtype=dt)
    ref = x.float() @ w.float().T
    y = hip.gemv(dev(x), dev(w))
    report("gemv", y, ref, *tol(dt, ref.std().item()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(8, 8192, 3584), (24, 3584, 4096), (8, 3584, 14336), (1, 16, 64), (2, 48, 128), (5, 1008, 3584), (16, 256, 512),
                                   (17, 64, 192), (32, 3584, 4096), (8, 4000, 3584), (3, 32, 320), (8, 6144, 4096),
                                   (8, 67200, 256), (3, 40016, 128), (20, 36000, 192)])      # more feature groups than resident blocks: blocks walk groups
def test_gemv_mfma(hip, dt, M, N, K):
    """vidi_gemv_mfma (a batch of decode rows on the matrix pipe; every K split 1 / 2 / 4 / 8, one and two 16-row groups, blocks that walk
    several feature groups): against the fp32 product of the same operands to the output rounding, and against vidi_gemv (M <= 8: both
    round one fp32 sum per element, they differ by the summation order only); a strided input / output slice leaves its surroundings alone."""
    assert hip.gemv_mfma_fits(M, N, K)
    x = seeded((M, K), 95, dtype=dt); w = seeded((N, K), 96, 0.05, dtype=dt)
    ref = x.float() @ w.float().T
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    y = hip.gemv_mfma(dev(x), dev(w))
    report("gemv_mfma vs fp32", y, ref, 1e-3 * float(ref.abs().max()), ulp)
    if M <= 8:
        report("gemv_mfma vs vidi_gemv", y, hip.gemv(dev(x), dev(w)).float(), 1e-3 * float(ref.abs().max()), 2 * ulp)
    xw = torch.zeros((M, K + 64), dtype=dt); xw[:, :K] = x
    yw = torch.full((M + 1, N + 32), 7.0, dtype=dt).cuda()
    hip.gemv_mfma(dev(xw)[:, :K], dev(w), out=yw[:M, :N])
    report("strided", yw[:M, :N], ref, 1e-3 * float(ref.abs().max()), ulp)
    assert bool((yw[:M, N:] == 7.0).all()) and bool((yw[M] == 7.0).all())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,I,K,act", [(8, 14336, 3584, "gelu"), (5, 512, 256, "gelu"), (1, 96, 64, "silu"), (8, 64, 128, "silu"), (16, 1024, 4096, "silu"),
                                       (2, 14336, 4096, "silu"), (8, 40000, 128, "gelu")])
def test_gemv_mfma_glu(hip, dt, M, I, K, act):
    """the gated pair on the interleaved gate/up weight: against the oracle's rounding points (T(act(T(g))) * T(u), gemma.py:116-123 via HF
    Gemma2MLP / mistral.py:131-137) from fp32 dot products, and against vidi_gemv_glu for M <= 8 (same rounding points, other summation
    order: an element may move by one rounding of g or u)"""
    code = hip.ACT_GELU_TANH if act == "gelu" else hip.ACT_SILU
    assert hip.gemv_mfma_fits(M, I, K, True)
    x = seeded((M, K), 97, dtype=dt); w = seeded((2 * I, K), 98, 0.05, dtype=dt)
    full = (x.float() @ w.float().T).view(M, I // 32, 2, 32)
    g, u = full[:, :, 0].reshape(M, I).to(dt).float(), full[:, :, 1].reshape(M, I).to(dt).float()
    a = F.gelu(g, approximate="tanh") if act == "gelu" else F.silu(g)
    ref = a.to(dt).float() * u
    out = torch.full((M, I), 7.0, dtype=dt, device="cuda")
    hip.gemv_mfma(dev(x), dev(w), out, glu_act=code)
    report("gemv_mfma glu", out, ref, *tol(dt, ref.std().item(), k=2))
    if M <= 8:
        o2 = torch.empty_like(out)
        hip.gemv_glu(dev(x), dev(w), o2, code)
        report("gemv_mfma glu vs vidi_gemv_glu", out, o2.float(), *tol(dt, ref.std().item(), k=2))


def test_gemv_mfma_shapes_it_does_not_take(hip):
    assert not hip.gemv_mfma_fits(33, 4096, 3584) and not hip.gemv_mfma_fits(17, 4096, 3584, True)
    assert not hip.gemv_mfma_fits(8, 4096 + 8, 3584) and not hip.gemv_mfma_fits(8, 4096, 3584 + 32)
    assert hip.gemv_mfma_fits(8, 4096 + 16, 3584) and not hip.gemv_mfma_fits(8, 14336 + 16, 3584, True)      # the gated pair needs I % 32: fits() is the dispatcher's exact predicate
    x = torch.zeros((8, 96), dtype=torch.bfloat16).cuda(); w = torch.zeros((64, 96), dtype=torch.bfloat16).cuda()
    with pytest.raises(Exception, match="vidi_gemv_mfma"):
        hip.gemv_mfma(x, w)



def test_gemm_f32(hip):
    from vidi_amd import hip as H
    M, N, K = 150, 256, 256
    x = seeded((M, K), 26); w = seeded((N, K), 27, 0.1); b = seeded((N,), 28)
    ref = F.gelu(F.linear(x, w, b))
    y = hip.gemm_f32(dev(x), dev(w), dev(b), H.ACT_GELU_ERF)
    report("gemm_f32", y, ref, 2e-4, 2e-5)        # exact-fp32 MFMA: fp32 accumulation-order noise only


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("D,N,H,B", [(72, 729, 3, 2), (64, 1500, 2, 1), (16, 49, 4, 3), (16, 50, 4, 2), (32, 130, 2, 2)])
def test_attn_self(hip, dt, D, N, H, B):
    """SigLIP (N=729,d=72), Whisper (N=1500,d=64) and tiny shapes; asymmetric q/k/v"""
    Hd = H * D
    Npad = (N + 63) // 64 * 64
    q = seeded((B, N, H, D), 30, dtype=dt); k = seeded((B, N, H, D), 31, dtype=dt); v = seeded((B, N, H, D), 32, dtype=dt)
    scale = D ** -0.5
    ref = O.sdpa_reference(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2), scale)
    ref = ref.transpose(1, 2).reshape(B * N, Hd)
    qk = torch.cat([q.reshape(B * N, Hd), k.reshape(B * N, Hd)], dim=1).contiguous()
    vt = pack_vt(v, Npad)
    out = torch.zeros((B * N, Hd), dtype=dt).cuda()
    hip.attn_self(dev(qk), dev(vt), out, B=B, N=N, Npad=Npad, H=H, D=D, koff=Hd, scale=scale)
    report(f"attn_self D{D} N{N}", out, ref, *tol(dt, max(0.05, ref.std().item()), k=2))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("D,N,H,B", [(72, 729, 3, 2), (72, 729, 16, 9), (64, 1500, 2, 1), (64, 1500, 20, 3), (16, 49, 4, 3), (16, 50, 4, 2), (32, 130, 2, 2), (72, 64, 2, 8), (64, 1, 2, 2)])
def test_attn_self_rm(hip, dt, D, N, H, B):
    """row-major-V variant (V transposed on the fly by ds_read_b64_tr_b16): same shapes as test_attn_self plus full head counts (the
    XCD-aware block order), a one-tile and a one-key case; Q, K, V are column ranges of ONE [B*N, 3*H*D + pad] buffer with a row stride
    that is not the packed width; checked against the oracle and against the Vt kernel (same math, same rounding points)."""
    Hd = H * D
    Npad = (N + 63) // 64 * 64
    q = seeded((B, N, H, D), 30, dtype=dt); k = seeded((B, N, H, D), 31, dtype=dt); v = seeded((B, N, H, D), 32, 1.0, dtype=dt) + 0.25
    scale = D ** -0.5
    ref = O.sdpa_reference(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2), scale)
    ref = ref.transpose(1, 2).reshape(B * N, Hd)
    ld = 3 * Hd + 8
    qkv = torch.full((B * N, ld), float("nan"), dtype=dt)
    qkv[:, :Hd] = q.reshape(B * N, Hd); qkv[:, Hd: 2 * Hd] = k.reshape(B * N, Hd); qkv[:, 2 * Hd: 3 * Hd] = v.reshape(B * N, Hd)
    qkv[:, 3 * Hd:] = 0
    out = torch.zeros((B * N, Hd), dtype=dt).cuda()
    hip.attn_self_rm(dev(qkv), out, B=B, N=N, H=H, D=D, koff=Hd, voff=2 * Hd, scale=scale)
    report(f"attn_self_rm D{D} N{N}", out, ref, *tol(dt, ref.std().item(), k=2))
    out2 = torch.zeros((B * N, Hd), dtype=dt).cuda()
    hip.attn_self(dev(qkv[:, : 2 * Hd].contiguous()), dev(pack_vt(v, Npad)), out2, B=B, N=N, Npad=Npad, H=H, D=D, koff=Hd, scale=scale)
    report(f"attn_self_rm vs Vt kernel D{D} N{N}", out, out2.float().cpu(), *tol(dt, ref.std().item(), k=1))
    assert torch.equal(out, out2)                                   # same MFMA operands in the same slots: bit-identical
    # head-major input [3][B][H][N][D] (what vidi_gemm_ln_heads writes): same result bit for bit
    hm = torch.stack([q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)]).contiguous()
    out3 = torch.zeros((B * N, Hd), dtype=dt).cuda()
    hip.attn_self_rm(dev(hm), out3, B=B, N=N, H=H, D=D, scale=scale, head_major=True)
    assert torch.equal(out3, out)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("D,N,H,B", [(72, 729, 3, 2), (72, 729, 16, 9), (72, 64, 2, 8), (72, 50, 2, 3), (72, 300, 4, 2), (72, 96, 2, 3), (72, 20, 2, 3),
                                     (64, 1500, 2, 1), (16, 49, 4, 3)])
def test_attn_self_rm_prescaled_q(hip, dt, D, N, H, B):
    """scale <= 0: Q already carries scale * log2(e) (the towers fold it into the q projection's weights).  d = 72 then takes the running
    maximum inside the QK^T contraction (the kernel's spare contraction chunk) and the exponent needs no FMA; every other head dim runs
    the ordinary body with a unit scale.  Checked against the oracle on the SAME rounded Q (softmax of q'.k * ln 2), with planted score
    outliers late in the key range so that the running maximum moves by more than the lazy-rescale threshold after the first tile, and
    with large negative scores everywhere (the first sub-tile must establish the maximum, not assume 0).  N = 729, 96 and 20 end in a
    tile whose second 32 keys are all past N (skipped by the d = 72 loop: twelve, two and one tile); 300, 64 and 50 do not."""
    import math
    Hd = H * D
    q = seeded((B, N, H, D), 40, dtype=dt); k = seeded((B, N, H, D), 41, dtype=dt); v = seeded((B, N, H, D), 42, 1.0, dtype=dt) + 0.25
    if N >= 300:
        k[:, N - 7] = (q[:, 5].float() * 6.0).to(dt)              # a key aligned with query 5 of every head: a late, large maximum
    sc2 = D ** -0.5 * math.log2(math.e)
    for shift in (0.0, -40.0):                                     # -40: every score far below zero (constant offset through one q/k column pair)
        qq, kk = q.clone(), k.clone()
        if shift:
            qq[..., 0] = 4.0; kk[..., 0] = shift / 4.0 / sc2
        qs = (qq.float() * sc2).to(dt)
        ref = O.sdpa_reference(qs.float().transpose(1, 2), kk.float().transpose(1, 2), v.float().transpose(1, 2), math.log(2.0))
        ref = ref.transpose(1, 2).reshape(B * N, Hd)
        hm = torch.stack([qs.permute(0, 2, 1, 3), kk.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)]).contiguous()
        out = torch.zeros((B * N, Hd), dtype=dt).cuda()
        hip.attn_self_rm(dev(hm), out, B=B, N=N, H=H, D=D, scale=0.0, head_major=True)
        report(f"attn_self_rm prescaled D{D} N{N} shift{shift}", out, ref, *tol(dt, max(0.05, ref.std().item()), k=2))
        # the ordinary form on the same operands (scale = ln 2 -> unit factor in base 2): same math, different max rounding
        out2 = torch.zeros((B * N, Hd), dtype=dt).cuda()
        hip.attn_self_rm(dev(hm), out2, B=B, N=N, H=H, D=D, scale=math.log(2.0), head_major=True)
        report(f"attn_self_rm prescaled vs scaled form D{D} N{N} shift{shift}", out, out2.float().cpu(), *tol(dt, max(0.05, ref.std().item()), k=2))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,early_key,late_key", [(729, 3, 722), (729, 40, 300), (300, 3, 250)])
def test_attn_self_rm_prescaled_q_early_outlier_then_drop(hip, dt, N, early_key, late_key):
    """The d = 72 prescaled-Q body enters its rescale branch WAVE-WIDE when any query of a 32-query set sees a score above its lazy
    threshold.  Query 9 gets one early key ≈ +150 base-2 units above everything after it; query 5 of the SAME set gets a late key that
    triggers the branch while query 9's own sub-tile maximum sits ≈150 below its running maximum: the lane that did not trigger must keep
    its maximum (alpha <= 1) — with the maximum allowed to move down, alpha = 2^150 turns query 9's accumulator into inf / NaN."""
    import math
    D, H, B = 72, 2, 2
    q = seeded((B, N, H, D), 50, dtype=dt); k = seeded((B, N, H, D), 51, dtype=dt); v = seeded((B, N, H, D), 52, 1.0, dtype=dt) + 0.25
    sc2 = D ** -0.5 * math.log2(math.e)
    q9 = q[:, 9].float()
    k[:, early_key] = (q9 * (150.0 / (sc2 * q9.pow(2).sum(-1, keepdim=True)))).to(dt)        # score(query 9, early key) ≈ +150 in base-2 units
    k[:, late_key] = (q[:, 5].float() * 6.0).to(dt)                                            # query 5: a late maximum far above its threshold
    qs = (q.float() * sc2).to(dt)
    ref = O.sdpa_reference(qs.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2), math.log(2.0))
    ref = ref.transpose(1, 2).reshape(B * N, H * D)
    s9 = (qs[:, 9].float() * k[:, early_key].float()).sum(-1)
    assert float(s9.min()) > 120.0, "the planted outlier is not large enough to exercise the case"
    hm = torch.stack([qs.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)]).contiguous()
    out = torch.zeros((B * N, H * D), dtype=dt).cuda()
    hip.attn_self_rm(dev(hm), out, B=B, N=N, H=H, D=D, scale=0.0, head_major=True)
    report(f"attn_self_rm prescaled early outlier N{N} keys {early_key}/{late_key}", out, ref, *tol(dt, max(0.05, ref.std().item()), k=2))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("seq,frames,heads,hd,K,cfg", [(729, 18, 16, 72, 1152, -1), (49, 3, 4, 16, 192, -1), (49, 3, 4, 16, 192, 0), (1500, 9, 20, 64, 1280, 5)])
def test_gemm_ln_heads_is_gemm_ln_rearranged(hip, dt, seq, frames, heads, hd, K, cfg):
    """the head-major q/k/v projection equals the row-major one with its columns / rows regrouped: Y[which][frame][head][token][d]"""
    M, N = frames * seq, 3 * heads * hd
    x = (seeded((M, K), 90, 1.0) + seeded((M, 1), 91, 1.0)).to(dt)
    w = seeded((N, K), 92, 0.05, dtype=dt); b = seeded((N,), 93, 0.3, dtype=dt)
    gamma = (1.0 + seeded((K,), 94, 0.2)).to(dt); beta = seeded((K,), 95, 0.2, dtype=dt)
    wf, cs, sh = _fold_ln(w, b, gamma, beta, dt)
    st = torch.zeros(2 * M, dtype=torch.float32).cuda()
    hip.row_stats(dev(x), st, 1e-6)
    y = hip.gemm_ln(dev(x), dev(wf), st, dev(cs), dev(sh), tile_cfg=cfg)
    yh = torch.full((M * N,), float("nan"), dtype=dt).cuda()
    hip.gemm_ln_heads(dev(x), dev(wf), st, dev(cs), dev(sh), yh, seq=seq, hd=hd, tile_cfg=cfg)
    want = y.view(frames, seq, 3, heads, hd).permute(2, 0, 3, 1, 4).contiguous().view(-1)
    assert torch.equal(yh, want)


def _cross_ref(q, k, v, mask, scale, softcap, G):
    """q:[Lq,nq,hd], k,v:[N,nkv,hd], mask:[N] bool -> [Lq,nq*hd] (fp32)"""
    qh = q.float().permute(1, 0, 2)[None]
    kh = O.repeat_kv(k.float().permute(1, 0, 2)[None], G)
    vh = O.repeat_kv(v.float().permute(1, 0, 2)[None], G)
    add = torch.zeros(k.shape[0]); add[~mask] = float("-inf")
    o = O.sdpa_reference(qh, kh, vh, scale, softcap, add[None, None, None, :])
    return o[0].permute(1, 0, 2).reshape(q.shape[0], -1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("HD,nkv,G,Lq,N,start,softcap,masked,zsplit", [
    (256, 8, 2, 1, 5000, 0, 50.0, False, 8),       # decode, Gemma dims
    (256, 8, 2, 40, 1000, 128, 50.0, False, 2),    # prefill Lq=40 (3 row tiles), region offset
    (256, 2, 2, 5, 333, 64, 50.0, True, 3),        # key mask + ragged tail
    (128, 2, 2, 3, 200, 0, None, True, 1),         # Mistral-style: no softcap, hd=128
    (256, 2, 2, 2, 17, 0, 50.0, False, 4),         # fewer sub-tiles than waves
    # two row tiles and more: the blocks' four waves own four row tiles and share one K / V stream (attn_cross_rows_kernel)
    (256, 8, 2, 304, 2000, 64, 50.0, True, 3),     # 8 prompts of 38 tokens (608 rows = 19 row tiles: 5 row blocks, the last with one idle wave), key mask
    (128, 2, 4, 50, 700, 0, None, False, 2),       # hd = 128, no softcap, 7 row tiles
    (256, 2, 2, 17, 100, 0, 50.0, False, 5),       # 2 row tiles (two idle waves), more key slices than sub-tiles (an empty slice emits the neutral partial)
    (256, 8, 2, 39, 90000 // 8, 0, 50.0, False, 16),   # the single-prompt shape at an eighth of the 60-min key count
])
def test_attn_cross(hip, dt, HD, nkv, G, Lq, N, start, softcap, masked, zsplit):
    nq = nkv * G
    q = seeded((Lq, nq, HD), 40, dtype=dt); k = seeded((N, nkv, HD), 41, dtype=dt); v = seeded((N, nkv, HD), 42, dtype=dt)
    mask = torch.ones(N, dtype=torch.bool)
    if masked:
        mask[torch.randperm(N, generator=torch.Generator().manual_seed(43))[: N // 3]] = False
    scale = HD ** -0.5
    ref = _cross_ref(q, k, v, mask, scale, softcap, G)
    ntile = (start + N + 63) // 64
    kc, vtc = pack_kv_cache(k, v, ntile, start)
    R = Lq * G
    Rpad = (R + 31) // 32 * 32
    opart, ml = hip.attn_cross_workspace(zsplit, nkv, Rpad, HD, "cuda")
    mpad = torch.zeros((N + 63) // 64 * 64, dtype=torch.uint8); mpad[:N] = mask.to(torch.uint8)
    qd = dev(q.reshape(Lq, nq * HD).contiguous())
    hip.attn_cross(qd, dev(kc), dev(vtc), dev(mpad) if masked else None, opart, ml, R=R, Rpad=Rpad, G=G, nkv=nkv, HD=HD,
                   ntile64=ntile, key_start=start, n_keys=N, scale=scale, softcap=softcap, zsplit=zsplit)
    out = torch.zeros((Lq, nq * HD), dtype=dt).cuda()
    hip.attn_merge(opart, ml, out, W=zsplit, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
    report("attn_cross", out, ref, *tol(dt, max(0.05, ref.std().item()), k=2))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("softcap", [50.0, 66.0, 67.0, 80.0, 200.0, None])
def test_attn_cross_many_rows_softmax_forms_and_saturated_rows(hip, dt, softcap):
    """The many-row kernel has a softmax WITHOUT a running maximum (bf16 + a cap with softcap log2 e <= 96, i.e. <= 66.5: Gemma2's 50) and
    forms with a per-row running reference (fp16; larger caps; no cap).  The launcher must pick by the cap's VALUE: round 5 tested
    `softcap > 0`, and a cap of 80 in the fixed form underflows rows whose logits all sit at the low end to l = 0 (zeros out of the merge).
    The inputs PIN rows at both ends — queries 20 x and -20 x a common key direction saturate every logit at +cap / -cap — next to ordinary
    rows; a third of the keys is masked, the first 70 keys entirely (a slice that starts with nothing to reference)."""
    HD, nkv, G, Lq, N, zsplit = 256, 2, 2, 48, 1500, 3            # 96 rows = 3 row tiles
    nq = nkv * G
    assert hip.attn_cross_row_tiles_per_block(96, softcap, dt) == 4
    g = torch.Generator().manual_seed(77)
    base = torch.randn(HD, generator=g)
    base = base / base.norm() * HD ** 0.5
    k = (base[None, None, :] + 0.1 * torch.randn((N, nkv, HD), generator=g)).to(dt)                   # every key close to one direction
    v = torch.randn((N, nkv, HD), generator=g).to(dt)
    q = torch.randn((Lq, nq, HD), generator=g)
    amp = 20.0 if softcap else 0.05                                 # (without a cap the logits are unbounded: keep exp() finite in the reference too)
    q[:12] = amp * base             # all logits of these rows at +cap
    q[12:24] = -amp * base          # ... at -cap: the rows a too-large cap underflows in the fixed form
    q = q.to(dt)
    mask = torch.ones(N, dtype=torch.bool)
    mask[::3] = False
    mask[:70] = False
    scale = HD ** -0.5
    ref = _cross_ref(q, k, v, mask, scale, softcap, G)
    ntile = (N + 63) // 64
    kc, vtc = pack_kv_cache(k, v, ntile, 0)
    R, Rpad = Lq * G, 96
    opart, ml = hip.attn_cross_workspace(zsplit, nkv, Rpad, HD, "cuda")
    mpad = torch.zeros(ntile * 64, dtype=torch.uint8); mpad[:N] = mask.to(torch.uint8)
    hip.attn_cross(dev(q.reshape(Lq, nq * HD).contiguous()), dev(kc), dev(vtc), dev(mpad), opart, ml, R=R, Rpad=Rpad, G=G, nkv=nkv, HD=HD,
                   ntile64=ntile, key_start=0, n_keys=N, scale=scale, softcap=softcap, zsplit=zsplit)
    out = torch.zeros((Lq, nq * HD), dtype=dt).cuda()
    hip.attn_merge(opart, ml, out, W=zsplit, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
    assert float(out[12:24].float().abs().max()) > 0, "saturated-low rows came back as zeros (l underflowed)"
    report(f"attn_cross softcap {softcap}", out, ref, *tol(dt, max(0.05, ref.std().item()), k=2))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("softcap", [50.0, None])
def test_attn_cross_running_reference_climbs(hip, dt, softcap):
    """the running-reference forms re-reference a row whenever its maximum outgrows what the dtype can hold against the old reference: here
    every row's logits CLIMB along the key axis (keys = ramp x direction, queries along it), over ~100 (capped) / ~60 (uncapped) binades from
    the first sub-tile to the last, so a wave takes the cold path many times per slice; rows 0-15 climb, rows 16-31 fall (never re-referenced
    after the first step: their early keys dominate and the late ones vanish), the rest are ordinary.  One slice and four, hd 128 and 256."""
    for HD, zsplit in ((256, 1), (128, 4)):
        nkv, G, Lq, N = 2, 2, 40, 2048
        nq = nkv * G
        g = torch.Generator().manual_seed(78)
        base = torch.randn(HD, generator=g)
        base = base / base.norm()
        ramp = torch.linspace(-1.0, 1.0, N)[:, None, None]
        k = (ramp * base[None, None, :] * HD ** 0.5 + 0.05 * torch.randn((N, nkv, HD), generator=g)).to(dt)
        v = torch.randn((N, nkv, HD), generator=g).to(dt)
        q = 0.3 * torch.randn((Lq, nq, HD), generator=g)
        amp = (3.0 if softcap else 40.0) * HD ** 0.5                       # |q . k| scale up to amp along the ramp (capped: tanh saturates both ends)
        q[:16] += amp * base
        q[16:32] -= amp * base
        q = q.to(dt)
        mask = torch.ones(N, dtype=torch.bool)
        mask[5::11] = False
        scale = HD ** -0.5
        ref = _cross_ref(q, k, v, mask, scale, softcap, G)
        ntile = N // 64
        kc, vtc = pack_kv_cache(k, v, ntile, 0)
        R = Lq * G
        Rpad = (R + 31) // 32 * 32
        opart, ml = hip.attn_cross_workspace(zsplit, nkv, Rpad, HD, "cuda")
        mpad = mask.to(torch.uint8)
        hip.attn_cross(dev(q.reshape(Lq, nq * HD).contiguous()), dev(kc), dev(vtc), dev(mpad), opart, ml, R=R, Rpad=Rpad, G=G, nkv=nkv, HD=HD,
                       ntile64=ntile, key_start=0, n_keys=N, scale=scale, softcap=softcap, zsplit=zsplit)
        out = torch.zeros((Lq, nq * HD), dtype=dt).cuda()
        hip.attn_merge(opart, ml, out, W=zsplit, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
        assert bool(torch.isfinite(out.float()).all())
        report(f"attn_cross climbing reference HD {HD} softcap {softcap}", out, ref, *tol(dt, max(0.05, ref.std().item()), k=2))


def test_attn_cross_split_invariance(hip):
    """property at a BASELINE-scale key count: the merged result does not depend on the KV split"""
    dt = torch.bfloat16
    HD, nkv, G, Lq, N = 256, 8, 2, 1, 90000
    nq = nkv * G
    g = torch.Generator(device="cuda").manual_seed(5)
    ntile = (N + 63) // 64
    kc = (torch.randn((nkv, ntile, 64, HD), generator=g, device="cuda")).to(dt)
    vtc = (torch.randn((nkv, 2 * ntile, HD, 32), generator=g, device="cuda")).to(dt)
    q = torch.randn((Lq, nq * HD), generator=g, device="cuda").to(dt)
    outs = []
    for zs in (4, 32):
        opart, ml = hip.attn_cross_workspace(zs, nkv, 32, HD, "cuda")
        hip.attn_cross(q, kc, vtc, None, opart, ml, R=Lq * G, Rpad=32, G=G, nkv=nkv, HD=HD, ntile64=ntile, key_start=0,
                       n_keys=N, scale=HD ** -0.5, softcap=50.0, zsplit=zs)
        po = torch.zeros((nkv, 32, HD), dtype=torch.float32, device="cuda")
        pml = torch.zeros((nkv, 32, 2), dtype=torch.float32, device="cuda")
        hip.attn_merge(opart, ml, None, W=zs, nkv=nkv, R=Lq * G, Rpad=32, G=G, HD=HD, out_f32=po, out_ml=pml, dtype=0)
        # second-level merge of the partial form (what the multi-GPU path does with all-gathered partials)
        o = torch.zeros((Lq, nq * HD), dtype=dt, device="cuda")
        hip.attn_merge(po[None].contiguous(), pml[None].contiguous(), o, W=1, nkv=nkv, R=Lq * G, Rpad=32, G=G, HD=HD)
        outs.append((po[:, : Lq * G] / pml[:, : Lq * G, 1:2]).clone())
        o1 = torch.zeros((Lq, nq * HD), dtype=dt, device="cuda")
        hip.attn_merge(opart, ml, o1, W=zs, nkv=nkv, R=Lq * G, Rpad=32, G=G, HD=HD)
        assert torch.equal(o.cpu(), o1.cpu()), "two-level merge must equal the direct merge"
    report("split invariance", outs[0], outs[1], 2e-4, 1e-3)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("HD", [256, 64])
def test_rope_and_attn_text(hip, dt, HD):
    B, Lq, past, nq, nkv, W = 2, 5, 9, 4, 2, 6
    Lmax = 32
    G = nq // nkv
    q = seeded((B, Lq, nq, HD), 50, dtype=dt); kn = seeded((B, Lq, nkv, HD), 51, dtype=dt)
    kc = seeded((B, Lmax, nkv, HD), 52, dtype=dt); vc = seeded((B, Lmax, nkv, HD), 53, dtype=dt)
    pos = torch.arange(past, past + Lq)[None].repeat(B, 1)
    cos, sin = O.rope_cos_sin(pos, HD, 10000.0, dt)
    qr, kr = O.apply_rope(q.float().transpose(1, 2), kn.float().transpose(1, 2), cos.float(), sin.float())
    qd = dev(q.reshape(B * Lq, nq * HD).clone()); kd = dev(kn.reshape(B * Lq, nkv * HD).clone())
    hip.rope(qd, kd, dev(cos.reshape(B * Lq, HD).contiguous()), dev(sin.reshape(B * Lq, HD).contiguous()), rows=B * Lq, nq=nq, nkv=nkv, HD=HD)
    a, r = tol(dt, 1.0)
    report("rope q", qd, qr.transpose(1, 2).reshape(B * Lq, -1), a, r)
    report("rope k", kd, kr.transpose(1, 2).reshape(B * Lq, -1), a, r)
    # attention over the cache (keys 0..past+i), right-padding mask on row 1, sliding window W
    kmask = torch.ones((B, Lmax), dtype=torch.uint8); kmask[1, 3:6] = 0
    for window in (0, W):
        kk = kc.float().transpose(1, 2)[:, :, : past + Lq]; vv = vc.float().transpose(1, 2)[:, :, : past + Lq]
        qi = torch.arange(past, past + Lq)[:, None]; kj = torch.arange(past + Lq)[None, :]
        allowed = kj <= qi
        if window:
            allowed = allowed & (kj >= qi - window)
        allowed = allowed[None, None] & kmask[:, None, None, : past + Lq].bool()
        add = torch.zeros(allowed.shape); add[~allowed] = float("-inf")
        ref = O.sdpa_reference(q.float().transpose(1, 2), O.repeat_kv(kk, G), O.repeat_kv(vv, G), HD ** -0.5, 50.0, add)
        ref = ref.transpose(1, 2).reshape(B * Lq, nq * HD)
        out = torch.zeros((B * Lq, nq * HD), dtype=dt).cuda()
        hip.attn_text(dev(q.reshape(B * Lq, nq * HD).contiguous()), dev(kc.reshape(B, Lmax, -1).contiguous()),
                      dev(vc.reshape(B, Lmax, -1).contiguous()), dev(kmask), out, B=B, Lq=Lq, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD,
                      past_len=past, window=window, scale=HD ** -0.5, softcap=50.0)
        report(f"attn_text window={window}", out, ref, *tol(dt, 0.3))


# ---------------------------------------------------------------------------------------------
# norms
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H", [64, 1152, 3584])
def test_norms(hip, dt, H):
    from vidi_amd import hip as Hh
    rows = 37
    x = seeded((rows, H), 60, 2.0, dtype=dt); w = seeded((H,), 61, 0.2, dtype=dt); b = seeded((H,), 62, 0.2, dtype=dt)
    res = seeded((rows, H), 63, dtype=dt)
    a, r = tol(dt, 1.0)
    report("gemma", hip.norm(Hh.NORM_GEMMA, dev(x), dev(w), eps=1e-6), O.gemma_rmsnorm(x.float(), w.float(), 1e-6), a, r)
    report("gemma_add", hip.norm(Hh.NORM_GEMMA_ADD, dev(x), dev(w), eps=1e-6, residual=dev(res)),
           res.float() + O.gemma_rmsnorm(x.float(), w.float(), 1e-6), a, r)
    report("mm", hip.norm(Hh.NORM_MM, dev(x), dev(w), eps=1e-5), O.mm_RMSNorm(x.float(), w.float()), a, r)
    report("mm_now", hip.norm(Hh.NORM_MM_NOW, dev(x), None, eps=1e-5), O.mm_rms_norm(x.float()), a, r)
    xf = seeded((rows, H), 64, 3.0)
    report("mm_now f32 in", hip.norm(Hh.NORM_MM_NOW, None, None, eps=1e-5, x_f32=dev(xf), dtype=dt), O.mm_rms_norm(xf.to(dt).float()), a, r)
    report("layer", hip.norm(Hh.NORM_LAYER, dev(x), dev(w), eps=1e-6, bias=dev(b)), O.layer_norm(x.float(), w.float(), b.float(), 1e-6), a, r)
    # LLM norm: mask derivation is bit-exact (zero rows -> mask 0), features scaled by the normalizer
    x2 = x.clone(); x2[5] = 0; x2[11] = 0
    mask = torch.zeros(rows, dtype=torch.uint8).cuda()
    flag = torch.ones(1, dtype=torch.int32).cuda()
    nz = float(torch.tensor(H ** 0.5, dtype=dt).float())
    y = hip.norm(Hh.NORM_LLM, dev(x2), dev(w), eps=1e-5, mask_out=mask, sample_flag=flag, normalizer=nz)
    mref = x2.float().abs().sum(-1) != 0
    assert torch.equal(mask.cpu().bool(), mref)
    ref = O.mm_RMSNorm(x2.float(), w.float()) * mref[:, None] * nz
    report("llm", y, ref, a * nz, r)
    flag.zero_()
    hip.norm(Hh.NORM_LLM, dev(x2), dev(w), eps=1e-5, mask_out=mask, sample_flag=flag, normalizer=nz)
    assert int(mask.sum()) == 0


# ---------------------------------------------------------------------------------------------
# data movement / elementwise
# ---------------------------------------------------------------------------------------------
def test_im2col_matches_conv(hip):
    dt = torch.bfloat16
    T, S, P, Hv = 3, 98, 14, 32
    px = seeded((T, 3, S, S), 70, dtype=dt); w = seeded((Hv, 3, P, P), 71, 0.05, dtype=dt)
    kp = 640
    A = torch.zeros((T * 49, kp), dtype=dt).cuda()
    hip.im2col_patch(dev(px), A, T=T, S=S, P=P, Kpad=kp)
    ref = F.conv2d(px.float(), w.float(), stride=P).flatten(2).transpose(1, 2).reshape(T * 49, Hv)
    got = A.float().cpu()[:, : 3 * P * P] @ w.float().reshape(Hv, -1).T
    report("im2col", got, ref, 1e-4, 1e-4)                     # pure data movement: exact up to fp32 sum order
    assert float(A[:, 3 * P * P:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,S,P,Hv", [(16, 384, 14, 1152), (3, 384, 14, 1152), (1, 98, 14, 64), (5, 98, 14, 96), (2, 64, 16, 32), (3, 40, 8, 64)])
def test_patch_embed_reads_pixels_directly(hip, dt, T, S, P, Hv):
    """vidi_patch_embed — SiglipVisionEmbeddings' Conv2d(kernel = stride = P, valid) + bias + position table with the GEMM's loader
    gathering the patches from the NCHW pixels (4-byte-aligned 16-byte DMA pieces, 16-pixel runs whose surplus columns meet zero
    weights) — against F.conv2d in fp32 on the same rounded inputs, and against the im2col + GEMM form it replaces.  SigLIP-so400m dims
    (384 px, 14-px patches -> 27 x 27, 6 unused pixel columns / rows), the tiny tower, a patch as wide as the run (P = 16) and P = 8."""
    side = S // P
    n = side * side
    px = seeded((T, 3, S, S), 170, dtype=dt); w = seeded((Hv, 3, P, P), 171, 0.05, dtype=dt)
    b = seeded((Hv,), 172, 0.1, dtype=dt); pos = seeded((n, Hv), 173, 0.1, dtype=dt)
    conv = F.conv2d(px.float(), w.float(), stride=P).flatten(2).transpose(1, 2).reshape(T * n, Hv) + b.float()
    ref = conv.to(dt).float() + pos.float().repeat(T, 1)          # the conv output is a dtype tensor before the table is added (siglip:178)
    w16 = hip.patch_embed_weight(dev(w), P)
    assert w16.shape == (Hv, (3 * P * 16 + 63) // 64 * 64) and float(w16.view(Hv, -1)[:, 3 * P * 16:].abs().max() if w16.shape[1] > 3 * P * 16 else 0) == 0.0
    out = torch.full((T * n, Hv), float("nan"), dtype=dt, device="cuda")
    hip.patch_embed(dev(px), w16, dev(b), dev(pos), out, T=T, S=S, P=P)
    report("patch_embed vs conv2d", out, ref, *tol(dt, ref.std().item()))
    # the form it replaces: im2col buffer + GEMM with the bias / position-table epilogue (same products, another summation order)
    kp = (3 * P * P + 63) // 64 * 64
    A = torch.zeros((T * n, kp), dtype=dt, device="cuda")
    hip.im2col_patch(dev(px), A, T=T, S=S, P=P, Kpad=kp)
    wp = torch.zeros((Hv, kp), dtype=dt, device="cuda"); wp[:, : 3 * P * P] = dev(w).reshape(Hv, -1)
    old = hip.gemm(A, wp, dev(b), residual=dev(pos), rmod=n)
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    # two roundings (conv + bias, then + position): a 1-ulp flip at the first can leave the results 2 ulps apart
    report("patch_embed vs im2col + gemm", out, old.float(), 4 * ulp * ref.std().item(), 4 * ulp)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,side,C,k,N", [(12, 27, 1152, 14, 1152), (2, 27, 1152, 14, 1152), (3, 7, 64, 4, 64), (5, 9, 128, 3, 96), (3, 6, 64, 6, 32), (2, 5, 256, 1, 64)])
def test_conv_window_gathers_in_the_loader(hip, dt, T, side, C, k, N):
    """vidi_conv_window — Vidi-7B's learned Conv2DPool conv (Vidi_7B/model/mm_vision/pool.py:19-26: Conv2d(C, C, k, stride 1, no bias)) with
    the k x k window gathered from the token-major features by the GEMM's loader — against F.conv2d in fp32 on the same rounded inputs and
    bit for bit against the im2col + GEMM form it replaces (same products, same order).  SigLIP-so400m dims (27 x 27, k = 14 -> 14 x 14
    outputs, K = 225 792), the tiny tower (one K slice per window position), a window as large as the map (one output per frame), a 1 x 1 window."""
    oc = side - k + 1
    f = seeded((T, side * side, C), 180, dtype=dt); w = seeded((N, C, k, k), 181, (k * k * C) ** -0.5, dtype=dt)
    ref = F.conv2d(f.float().reshape(T, side, side, C).permute(0, 3, 1, 2), w.float()).permute(0, 2, 3, 1).reshape(T * oc * oc, N)
    wg = dev(w.permute(0, 2, 3, 1).reshape(N, -1).contiguous())
    out = torch.full((T * oc * oc, N), float("nan"), dtype=dt, device="cuda")
    hip.conv_window(dev(f), wg, out, T=T, side=side, C=C, k=k)
    report("conv_window vs conv2d", out, ref, *tol(dt, ref.std().item()))
    col = torch.empty((T * oc * oc, k * k * C), dtype=dt, device="cuda")
    hip.im2col_nhwc(dev(f), col, T=T, side=side, C=C, k=k)
    old = hip.gemm(col, wg, None, tile_cfg=5 if T * oc * oc >= 256 and N >= 64 else -1)
    if T * oc * oc >= 256 and N >= 64:
        assert torch.equal(out.view(torch.int16), old.view(torch.int16)), "same kernel body, same K order: bit-identical"
    else:
        report("conv_window vs im2col + gemm (another tile kernel)", out, old.float(), *tol(dt, ref.std().item()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("side,hw", [(27, (28, 28)), (27, (10, 10)), (27, (26, 26)), (7, (28, 28)), (7, (10, 10))])
def test_pool_s2d(hip, dt, side, hw):
    T, C, m = 3, 16, 2
    f = seeded((T, side * side, C), 72, dtype=dt)
    x = f.float().reshape(T, side, side, C).permute(0, 3, 1, 2)
    ref = O.conv2d_pool(x, hw, m).permute(0, 2, 3, 1)
    resize = hw[0] != 28
    h, w = hw if resize else (side + 1, side + 1)
    out = torch.zeros((T, h // m, w // m, C * m * m), dtype=dt).cuda()
    hip.pool_s2d(dev(f), out, T=T, side=side, C=C, h=h, w=w, m=m, resize=resize)
    report("pool_s2d", out, ref, *tol(dt, 1.0))


def test_elementwise_misc(hip):
    dt = torch.bfloat16
    T, oh, ow, H = 3, 4, 5, 64
    f = seeded((T, oh, ow, H), 73, dtype=dt); ph = seeded((oh, H), 74, dtype=dt); pw = seeded((ow, H), 75, dtype=dt); pt = seeded((T, H), 76, dtype=dt)
    ref = ((f + ph[None, :, None, :]) + pw[None, None, :, :]) + pt[:, None, None, :]      # bf16 eager = same rounding chain
    fd = dev(f.clone())
    hip.add_pos(fd, dev(ph), dev(pw), dev(pt), T=T, oh=oh, ow=ow, H=H)
    assert torch.equal(fd.cpu(), ref), "add_pos must reproduce the eager rounding chain bit-exactly"
    a, b, c = seeded((40, H), 77, dtype=dt), seeded((40, H), 78, dtype=dt), seeded((40, H), 79, dtype=dt)
    out = torch.zeros((40, H), dtype=dt).cuda()
    hip.add3(dev(a), dev(b), dev(c), out)
    assert torch.equal(out.cpu(), (a + b) + c)
    # embed gather * normalizer; negative id -> zero row
    E = seeded((50, H), 80, dtype=dt); ids = torch.tensor([3, 49, -200, 0, 7], dtype=torch.int64)
    nz = float(torch.tensor(H ** 0.5, dtype=dt).float())
    eo = torch.zeros((5, H), dtype=dt).cuda()
    hip.embed(dev(ids), dev(E), eo, normalizer=nz)
    ref = E[ids.clamp(min=0)] * torch.tensor(H ** 0.5, dtype=dt); ref[2] = 0
    assert torch.equal(eo.cpu(), ref)
    # geglu unpack
    M, I = 3, 64
    yp = seeded((M, 2 * I), 81, dtype=dt)
    g = yp.view(M, I // 32, 2, 32)[:, :, 0].reshape(M, I); u = yp.view(M, I // 32, 2, 32)[:, :, 1].reshape(M, I)
    go = torch.zeros((M, I), dtype=dt).cuda()
    hip.geglu_unpack(dev(yp), go)
    report("geglu_unpack", go, O.gelu_tanh(g.float()) * u.float(), *tol(dt, 1.0))
    # softcap + argmax (index is bit-exact vs the same rounding chain)
    lg = seeded((3, 1000), 82, 20.0, dtype=dt)
    ref = (torch.tanh(lg / 30.0) * 30.0)
    ld = dev(lg.clone()); idx = torch.zeros(3, dtype=torch.int64).cuda()
    hip.softcap_argmax(ld, idx, 30.0)
    report("softcap", ld, ref.float(), *tol(dt, 10.0))
    assert torch.equal(idx.cpu(), torch.argmax(ld.float().cpu(), dim=-1))
    # mel transpose/pad
    mel = seeded((2, 16, 20), 83, dtype=dt)
    mo = torch.ones((2, 22, 16), dtype=dt).cuda()
    hip.mel_transpose_pad(dev(mel), mo)
    refm = F.pad(mel.permute(0, 2, 1), (0, 0, 1, 1))
    assert torch.equal(mo.cpu(), refm)
    # scale, any_nonzero
    so = torch.zeros((40, H), dtype=dt).cuda()
    hip.scale(dev(a), so, nz)
    assert torch.equal(so.cpu(), a * torch.tensor(H ** 0.5, dtype=dt))
    flag = torch.zeros(1, dtype=torch.int32).cuda()
    z = torch.zeros(1003, dtype=dt).cuda()
    hip.any_nonzero(z, flag); assert int(flag) == 0
    z[1001] = 1.0
    hip.any_nonzero(z, flag); assert int(flag) == 1


def test_sinusoid_and_pos_table(hip):
    l, N, d, i0, rows = 37, 100, 64, 5, 20
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float) * -(math.log(10000.0) / d))
    pe = torch.zeros((rows, d), dtype=torch.float32).cuda()
    hip.sinusoid(pe, dev(div), rows=rows, i0=i0, l=l, N=N, d=d)
    p = torch.arange(l, dtype=torch.float) / (l - 1) * (N - 1)
    ref = O.fractional_sinusoid(p, d)[i0: i0 + rows]
    report("sinusoid", pe, ref, 2e-5, 0.0)                     # fp32 sin/cos of identical fp32 arguments


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("dyn", [False, True], ids=["host-pos", "device-pos"])
def test_rope_cache_fused(hip, dt, dyn):
    """rope(q) + rope(k) + cache append in one launch == vidi_rope followed by slot copies, bit for bit; other slots untouched"""
    B, Lq, past, nq, nkv, HD, Lmax = 2, (1 if dyn else 5), 9, 4, 2, 128, 32
    kvd = nkv * HD
    qkv = seeded((B * Lq, nq * HD + 2 * kvd + 16), 60, dtype=dt)[:, : nq * HD + 2 * kvd]        # row stride != width
    pos = torch.arange(past, past + Lq)[None].repeat(B, 1)
    cos, sin = O.rope_cos_sin(pos, HD, 10000.0, dt)
    cs, sn = dev(cos.reshape(B * Lq, HD).contiguous()), dev(sin.reshape(B * Lq, HD).contiguous())
    qd = dev(qkv[:, : nq * HD].contiguous()); kd = dev(qkv[:, nq * HD: nq * HD + kvd].contiguous())
    hip.rope(qd, kd, cs, sn, rows=B * Lq, nq=nq, nkv=nkv, HD=HD)
    kc0 = seeded((B, Lmax, kvd), 61, dtype=dt); vc0 = seeded((B, Lmax, kvd), 62, dtype=dt)
    kref, vref = kc0.clone(), vc0.clone()
    kref[:, past: past + Lq] = kd.cpu().view(B, Lq, kvd)
    vref[:, past: past + Lq] = qkv[:, nq * HD + kvd:].reshape(B, Lq, kvd)
    kc, vc = dev(kc0), dev(vc0)
    qr = torch.empty((B * Lq, nq * HD), dtype=dt, device="cuda")
    pos_dev = torch.tensor([past], dtype=torch.int32, device="cuda") if dyn else None
    hip.rope_cache(dev(qkv) if False else qkv.cuda(), qr, kc, vc, cs, sn, B=B, Lq=Lq, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD,
                   pos0=(0 if dyn else past), pos_dev=pos_dev)
    assert torch.equal(qr.cpu().view(torch.int16), qd.cpu().view(torch.int16))
    assert torch.equal(kc.cpu().view(torch.int16), kref.view(torch.int16))
    assert torch.equal(vc.cpu().view(torch.int16), vref.view(torch.int16))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,H,nsrc", [(1, 3584, 3), (5, 3584, 1), (39, 256, 2), (3, 1536, 3)])
def test_resid_norm2_equals_unfused(hip, dt, M, H, nsrc):
    """add3 -> norm(GEMMA_ADD) -> norm(GEMMA) in one launch: bit-identical to the three launches"""
    a = seeded((M, H), 70, dtype=dt).cuda(); b = seeded((M, H), 71, dtype=dt).cuda(); c = seeded((M, H), 72, dtype=dt).cuda()
    res = seeded((M, H), 73, dtype=dt).cuda()
    w1 = seeded((H,), 74, 0.1, dtype=dt).cuda(); w2 = seeded((H,), 75, 0.1, dtype=dt).cuda()
    bb, cc = (b if nsrc >= 2 else None), (c if nsrc >= 3 else None)
    if nsrc == 1:
        s = a
    else:
        s = torch.empty_like(a)
        hip.add3(a, bb, cc, s)
    y1_ref = hip.norm(hip.NORM_GEMMA_ADD, s, w1, eps=1e-6, residual=res)
    y2_ref = hip.norm(hip.NORM_GEMMA, y1_ref, w2, eps=1e-6)
    y1 = res.clone(); y2 = torch.empty_like(a)
    hip.resid_norm2(a, bb, cc, y1, w1, w2, y1, y2, eps=1e-6)                      # Y1 aliases Res, as the engine uses it
    assert torch.equal(y1.view(torch.int16), y1_ref.view(torch.int16))
    assert torch.equal(y2.view(torch.int16), y2_ref.view(torch.int16))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,I,K,act", [(1, 14336, 3584, "gelu"), (3, 512, 256, "gelu"), (1, 96, 64, "silu"), (8, 64, 128, "silu")])
def test_gemv_glu_equals_gemv_plus_unpack(hip, dt, M, I, K, act):
    """gate/up GEMV with the gated activation fused: bit-identical to vidi_gemv + (ge)glu_unpack"""
    x = seeded((M, K), 80, dtype=dt).cuda(); w = seeded((2 * I, K), 81, 0.05, dtype=dt).cuda()
    yp = hip.gemv(x, w)
    ref = torch.empty((M, I), dtype=dt, device="cuda")
    code = hip.ACT_GELU_TANH if act == "gelu" else hip.ACT_SILU
    hip.glu_unpack(yp, ref, code)
    out = torch.empty_like(ref)
    hip.gemv_glu(x, w, out, code)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


# ---------------------------------------------------------------------------------------------
# decode-step launches (round 2): fused T2T, dual cross-attention, multi-block argmax
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nq,nkv,HD", [(16, 8, 256), (8, 2, 128), (4, 2, 64)])
@pytest.mark.parametrize("pos,window,dyn", [(0, 0, False), (5, 0, True), (70, 0, False), (200, 0, True), (200, 16, False), (131, 64, True)])
def test_attn_text_decode_equals_rope_cache_then_attn_text(hip, dt, nq, nkv, HD, pos, window, dyn):
    """vidi_attn_text_decode (rope + cache append + T2T in one launch) against vidi_rope_cache + vidi_attn_text[_dyn] at Lq = 1:
    the caches bit for bit; the output within accumulation-order noise of the SAME scores and probabilities (fp32 dot products summed
    in a different order, then one rounding to the dtype): 2 ulp of the dtype relative + 0.2 % of the output spread."""
    B, Lmax = 2, 256
    kvd = nkv * HD
    qkv = seeded((B, nq * HD + 2 * kvd + 8), 90, dtype=dt)[:, : nq * HD + 2 * kvd].cuda()
    posn = torch.tensor([[pos], [max(pos - 3, 0)]])                                  # per-row rope positions (right-padded batch)
    cos, sin = O.rope_cos_sin(posn, HD, 10000.0, dt)
    cs, sn = dev(cos.reshape(B, HD).contiguous()), dev(sin.reshape(B, HD).contiguous())
    kc0 = seeded((B, Lmax, kvd), 91, dtype=dt); vc0 = seeded((B, Lmax, kvd), 92, dtype=dt)
    kmask = torch.ones((B, Lmax), dtype=torch.uint8)
    kmask[1, 1: pos: 3] = 0                                                           # some padded keys in row 1
    kmask = kmask.cuda()
    pos_dev = torch.tensor([pos], dtype=torch.int32, device="cuda") if dyn else None
    sc, cap = HD ** -0.5, 50.0
    kc_a, vc_a = dev(kc0.clone()), dev(vc0.clone())
    qr = torch.empty((B, nq * HD), dtype=dt, device="cuda")
    ref = torch.zeros((B, nq * HD), dtype=dt, device="cuda")
    hip.rope_cache(qkv, qr, kc_a, vc_a, cs, sn, B=B, Lq=1, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD, pos0=pos, pos_dev=pos_dev)
    if dyn:
        hip.attn_text_dyn(qr, kc_a, vc_a, kmask, ref, B=B, Lq=1, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD, past_len_dev=pos_dev, window=window,
                          scale=sc, softcap=cap)
    else:
        hip.attn_text(qr, kc_a, vc_a, kmask, ref, B=B, Lq=1, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD, past_len=pos, window=window, scale=sc,
                      softcap=cap)
    kc_b, vc_b = dev(kc0.clone()), dev(vc0.clone())
    out = torch.zeros((B, nq * HD), dtype=dt, device="cuda")
    assert hip.attn_text_decode_fits(nq=nq, nkv=nkv, HD=HD, Lmax=Lmax, window=window, pos0=None if dyn else pos)
    hip.attn_text_decode(qkv, kc_b, vc_b, kmask, cs, sn, out, B=B, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD, window=window, scale=sc, softcap=cap,
                         pos0=pos, pos_dev=pos_dev)
    assert torch.equal(kc_a.view(torch.int16), kc_b.view(torch.int16)) and torch.equal(vc_a.view(torch.int16), vc_b.view(torch.int16))
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    report("attn_text_decode vs two launches", out, ref.float(), 2e-3 * ref.float().std().item(), 2 * ulp)


def test_attn_text_decode_rejects_what_does_not_fit(hip):
    """a cache whose scores do not fit the kernel's LDS plan is refused (the engine then uses the two-launch form)"""
    dt = torch.bfloat16
    nq, nkv, HD, Lmax = 16, 8, 256, 8192
    assert not hip.attn_text_decode_fits(nq=nq, nkv=nkv, HD=HD, Lmax=Lmax, window=0, pos0=None)
    qkv = torch.zeros((1, (nq + 2 * nkv) * HD), dtype=dt, device="cuda")
    kc = torch.zeros((1, Lmax, nkv * HD), dtype=dt, device="cuda")
    cs = torch.zeros((1, HD), dtype=dt, device="cuda")
    out = torch.zeros((1, nq * HD), dtype=dt, device="cuda")
    pos_dev = torch.tensor([3], dtype=torch.int32, device="cuda")
    with pytest.raises(hip.VidiHipError):
        hip.attn_text_decode(qkv, kc, kc.clone(), None, cs, cs, out, B=1, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD, window=0, scale=1.0, softcap=0.0,
                             pos_dev=pos_dev)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("HD,nkv,G,Lq,Na,Nb,za,zb,masked", [(256, 8, 2, 1, 3000, 1100, 5, 2, True), (128, 2, 4, 3, 700, 200, 3, 1, False),
                                                            (256, 2, 2, 39, 500, 333, 2, 2, True)])
def test_attn_cross2_equals_two_launches(hip, dt, HD, nkv, G, Lq, Na, Nb, za, zb, masked):
    """T2V + T2A in one launch: the partials (and so the merged outputs) are bit-identical to two vidi_attn_cross launches"""
    nq = nkv * G
    start_b = (Na + 63) // 64 * 64
    ntile = (start_b + Nb + 63) // 64
    g = torch.Generator(device="cuda").manual_seed(17)
    kc = torch.randn((nkv, ntile, 64, HD), generator=g, device="cuda").to(dt)
    vtc = torch.randn((nkv, 2 * ntile, HD, 32), generator=g, device="cuda").to(dt)
    q = torch.randn((Lq, nq * HD), generator=g, device="cuda").to(dt)
    ma = None
    if masked:
        ma = torch.ones((Na + 63) // 64 * 64, dtype=torch.uint8, device="cuda")
        ma[5: Na: 7] = 0
    R = Lq * G
    Rpad = (R + 31) // 32 * 32
    outs = []
    for dual in (False, True):
        wa = hip.attn_cross_workspace(za, nkv, Rpad, HD, "cuda"); wb = hip.attn_cross_workspace(zb, nkv, Rpad, HD, "cuda")
        for t in (*wa, *wb):
            t.fill_(float("nan"))
        kw = dict(R=R, Rpad=Rpad, G=G, nkv=nkv, HD=HD, ntile64=ntile, scale=HD ** -0.5, softcap=50.0)
        if dual:
            hip.attn_cross2(q, kc, vtc, dict(mask=ma, opart=wa[0], ml=wa[1], key_start=0, n_keys=Na, zsplit=za),
                            dict(mask=None, opart=wb[0], ml=wb[1], key_start=start_b, n_keys=Nb, zsplit=zb), **kw)
        else:
            hip.attn_cross(q, kc, vtc, ma, wa[0], wa[1], key_start=0, n_keys=Na, zsplit=za, **kw)
            hip.attn_cross(q, kc, vtc, None, wb[0], wb[1], key_start=start_b, n_keys=Nb, zsplit=zb, **kw)
        oa = torch.zeros((Lq, nq * HD), dtype=dt, device="cuda"); ob = torch.zeros_like(oa)
        hip.attn_merge2(wa[0], wa[1], oa, za, False, wb[0], wb[1], ob, zb, False, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
        outs.append((oa, ob))
    assert torch.isfinite(outs[0][0].float()).all() and torch.isfinite(outs[0][1].float()).all()
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,V,pad", [(1, 256000, 0), (3, 1000, 0), (2, 1003, 5), (4, 40, 0), (2, 1000, 3), (600, 512, 0)])
def test_softcap_argmax_multi_block(hip, dt, B, V, pad):
    """rows spread over many blocks: softcap values as the elementwise chain, FIRST maximal index (ties planted in different
    blocks' slices), repeated calls (the per-row scratch cleans itself), rows that are not 16-byte aligned"""
    cap = 30.0
    for rep in range(3):
        lg = seeded((B, V + pad), 100 + rep, 20.0, dtype=dt)[:, : V]
        big = lg.float().abs().max().item() * 2 + 1.0
        for b in range(B):                                                 # the same maximum at three places; the first one must win
            for i in sorted(set([(7 * b + 3) % V, V // 2, (V - 1 - b) % V])):
                lg[b, i] = big
        ld = lg.cuda() if pad == 0 else torch.empty((B, V + pad), dtype=dt, device="cuda")[:, : V].copy_(lg)
        idx = torch.full((B,), -1, dtype=torch.int64, device="cuda")
        hip.softcap_argmax(ld, idx, cap)
        x = lg.float() / cap
        ref = (torch.tanh(x.to(dt).float()).to(dt).float() * cap).to(dt)
        report("softcap (multi-block)", ld, ref.float(), *tol(dt, 10.0))
        want = torch.tensor([min((7 * b + 3) % V, V // 2, (V - 1 - b) % V) for b in range(B)])
        assert torch.equal(idx.cpu(), want), (idx.cpu(), want)
        assert torch.equal(idx.cpu(), torch.argmax(ld.float().cpu(), dim=-1))


def test_softcap_argmax_two_streams_are_independent(hip):
    """stateless, re-entrant, workspace passed in explicitly.  Two streams run vidi_softcap_argmax concurrently on different
    logits with their own caller-owned workspaces (many rows x many blocks per row, 20 rounds without any host synchronisation between
    the streams); every index must be the row's first maximum, both workspaces must be left zeroed, and a call without a zeroed
    workspace must be rejected by the binding rather than guessed at."""
    dt = torch.bfloat16
    B, V = 48, 256000
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    wss = [hip.softcap_argmax_workspace(B, "cuda") for _ in streams]
    lgs, idxs, wants = [], [], []
    for s in range(2):
        lg = seeded((B, V), 300 + s, 20.0, dtype=dt)
        big = lg.float().abs().max().item() * 2 + 1.0
        want = torch.tensor([(9973 * (b + 1) * (s + 1)) % V for b in range(B)])
        for b in range(B):
            lg[b, want[b]] = big
            lg[b, min(V - 1, int(want[b]) + 4097)] = big              # the same value later in another block's slice: the first one wins
        lgs.append(lg.cuda()); wants.append(want)
        idxs.append(torch.full((20, B), -1, dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    for rep in range(20):
        for s, st in enumerate(streams):
            with torch.cuda.stream(st):
                hip.softcap_argmax(lgs[s], idxs[s][rep], 0.0, wss[s])        # cap = 0: logits untouched, so every round sees the same data
    torch.cuda.synchronize()
    for s in range(2):
        assert torch.equal(idxs[s].cpu(), wants[s][None].expand(20, B)), f"stream {s}"
        assert int(wss[s].abs().sum()) == 0, "the workspace must be left zeroed"
    with pytest.raises(Exception):
        hip.softcap_argmax(lgs[0], idxs[0][0], 0.0, torch.zeros(1, dtype=torch.int64, device="cuda"))    # too small for 48 rows


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,K,N,nsrc,glu", [(1, 3584, 8192, 1, False), (1, 3584, 14336, 3, True), (3, 3584, 512, 2, False), (4, 256, 96, 3, True),
                                            (2, 4096, 1000, 1, False), (1, 1536, 64, 2, True)])
def test_gemv_norm2_equals_resid_norm2_then_gemv(hip, dt, M, K, N, nsrc, glu):
    """the decode fusion (norm pair inside the projection) against vidi_resid_norm2 followed by vidi_gemv / vidi_gemv_glu: the residual
    output within one ulp of the dtype (the sums of squares are reduced in a different order; same element arithmetic), the projection
    within the GEMV tolerance of the two-launch result"""
    a = seeded((M, K), 110, dtype=dt).cuda(); b = seeded((M, K), 111, dtype=dt).cuda(); c = seeded((M, K), 112, dtype=dt).cuda()
    res = seeded((M, K), 113, dtype=dt).cuda()
    w1 = seeded((K,), 114, 0.1, dtype=dt).cuda(); w2 = seeded((K,), 115, 0.1, dtype=dt).cuda()
    bb, cc = (b if nsrc >= 2 else None), (c if nsrc >= 3 else None)
    rows = 2 * N if glu else N
    w = seeded((rows, K), 116, 0.05, dtype=dt).cuda()
    y1_ref = torch.empty_like(a); x_ref = torch.empty_like(a)
    hip.resid_norm2(a, bb, cc, res, w1, w2, y1_ref, x_ref, eps=1e-6)
    out_ref = torch.zeros((M, N), dtype=dt, device="cuda")
    if glu:
        hip.gemv_glu(x_ref, w, out_ref, hip.ACT_GELU_TANH)
    else:
        hip.gemv(x_ref, w, out_ref)
    y1 = torch.full_like(a, float("nan")); out = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
    if glu:
        hip.gemv_glu_norm2(a, bb, cc, res, w1, w2, y1, w, out, eps=1e-6, act=hip.ACT_GELU_TANH)
    else:
        hip.gemv_norm2(a, bb, cc, res, w1, w2, y1, w, out, eps=1e-6)
    ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    report("gemv_norm2: residual stream", y1, y1_ref.float(), ulp * y1_ref.float().abs().max().item(), ulp)
    report("gemv_norm2: projection", out, out_ref.float(), *tol(dt, out_ref.float().std().item(), k=2 if glu else 1))
    with pytest.raises(hip.VidiHipError):                                  # in-place residual is refused (blocks race on it)
        hip.gemv_norm2(a, bb, cc, res, w1, w2, res, w[:N], out, eps=1e-6)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("nq,nkv,HD,B,pos,dyn", [(16, 8, 256, 1, 70, False), (8, 2, 128, 2, 131, True), (16, 8, 256, 1, 0, True)])
def test_attn_text_decode_merge2_equals_the_two_launches(hip, dt, nq, nkv, HD, B, pos, dyn):
    """the decode step's T2T and the merge of the cross-attention partials as ONE launch: text-attention output, caches and both merged
    outputs bit-identical to vidi_attn_text_decode + vidi_attn_merge2"""
    Lmax, window = 256, 0
    G = nq // nkv
    kvd = nkv * HD
    qkv = seeded((B, nq * HD + 2 * kvd), 130, dtype=dt).cuda()
    posn = torch.full((B, 1), pos)
    cos, sin = O.rope_cos_sin(posn, HD, 10000.0, dt)
    cs, sn = dev(cos.reshape(B, HD).contiguous()), dev(sin.reshape(B, HD).contiguous())
    kc0 = seeded((B, Lmax, kvd), 131, dtype=dt); vc0 = seeded((B, Lmax, kvd), 132, dtype=dt)
    kmask = torch.ones((B, Lmax), dtype=torch.uint8).cuda()
    pos_dev = torch.tensor([pos], dtype=torch.int32, device="cuda") if dyn else None
    R, Rpad, WA, WB = B * G, 32, 5, 3
    g = torch.Generator(device="cuda").manual_seed(133)
    parts = []
    for W in (WA, WB):
        op = torch.randn((W, nkv, Rpad, HD), generator=g, device="cuda")
        ml = torch.stack([torch.randn((W, nkv, Rpad), generator=g, device="cuda"),
                          torch.rand((W, nkv, Rpad), generator=g, device="cuda") + 0.5], dim=-1).contiguous()
        ml[0, :, 0, 0] = float("-inf")                                       # an empty partial (a wave without keys)
        parts.append((op, ml))
    kw = dict(B=B, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD, window=window, scale=HD ** -0.5, softcap=50.0, pos0=pos, pos_dev=pos_dev)
    res = []
    for fused in (False, True):
        kc, vc = dev(kc0.clone()), dev(vc0.clone())
        o_t = torch.full((B, nq * HD), float("nan"), dtype=dt, device="cuda")
        oa = torch.full((B, nq * HD), float("nan"), dtype=dt, device="cuda"); ob = torch.full_like(oa, float("nan"))
        ma = (parts[0][0], parts[0][1], oa, WA, False); mb = (parts[1][0], parts[1][1], ob, WB, False)
        if fused:
            hip.attn_text_decode_merge2(qkv, kc, vc, kmask, cs, sn, o_t, ma, mb, R=R, Rpad=Rpad, **kw)
        else:
            hip.attn_text_decode(qkv, kc, vc, kmask, cs, sn, o_t, **kw)
            hip.attn_merge2(*ma, *mb, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
        res.append((o_t, oa, ob, kc, vc))
    for x, y in zip(*res):
        assert torch.isfinite(x.float()).all()
        assert torch.equal(x.view(torch.int16), y.view(torch.int16))


# ---------------------------------------------------------------------------------------------
# the fused decode entry points against the ORACLE at Gemma2-9B dims (their unfused HIP twins are oracle-checked above and the fused
# forms are held to the twins; these close the loop so that a defect shared by a twin pair cannot pass)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,K,N,nsrc,glu", [(1, 3584, 8192, 1, False), (1, 3584, 14336, 3, True), (4, 3584, 8192, 1, False), (2, 3584, 14336, 3, True)])
def test_gemv_norm2_vs_oracle(hip, dt, M, K, N, nsrc, glu):
    """vidi_gemv_norm2 / vidi_gemv_glu_norm2 = the Gemma2 norm pair of DattnGemma2DecoderLayer (gemma.py:236-237 + :118, resp. :120-121 +
    the next layer's :162) followed by q|k|v resp. gate/up + GeGLU (modeling_gemma2.py:79-82), against the fp32 oracle functions"""
    eps = 1e-6
    srcs = [seeded((M, K), 140 + i, dtype=dt) for i in range(3)]
    res = seeded((M, K), 143, dtype=dt)
    w1 = seeded((K,), 144, 0.1, dtype=dt); w2 = seeded((K,), 145, 0.1, dtype=dt)
    # the reference's rounding points: `text + image + audio` are dtype adds (gemma.py:236), Gemma2RMSNorm computes in fp32 and rounds its
    # output once (modeling_gemma2.py:55-63), the residual add rounds again (gemma.py:237)
    s = srcs[0].float()
    for t in srcs[1:nsrc]:
        s = (s + t.float()).to(dt).float()
    y1_ref = res.float() + O.gemma_rmsnorm(s, w1.float(), eps).to(dt).float()
    x_ref = O.gemma_rmsnorm(y1_ref.to(dt).float(), w2.float(), eps)               # the stream is stored in the dtype between the norms
    if glu:
        gate = seeded((N, K), 146, 0.05, dtype=dt); up = seeded((N, K), 147, 0.05, dtype=dt)
        w = torch.stack([gate.view(N // 32, 32, K), up.view(N // 32, 32, K)], dim=1).reshape(2 * N, K).contiguous()     # engine._pack layout
        xr = x_ref.to(dt).float()
        g_ref, u_ref = O.linear(xr, gate.float()), O.linear(xr, up.float())
        out_ref = O.gelu_tanh(g_ref) * u_ref
        lin_std = g_ref.std().item()
    else:
        w = seeded((N, K), 146, 0.05, dtype=dt)
        out_ref = O.linear(x_ref.to(dt).float(), w.float())
        lin_std = out_ref.std().item()
    y1 = torch.full((M, K), float("nan"), dtype=dt, device="cuda"); out = torch.full((M, N), float("nan"), dtype=dt, device="cuda")
    a, b, c = (dev(t) for t in srcs)
    args = (a, b if nsrc >= 2 else None, c if nsrc >= 3 else None, dev(res), dev(w1), dev(w2), y1, dev(w), out)
    if glu:
        hip.gemv_glu_norm2(*args, eps=eps, act=hip.ACT_GELU_TANH)
    else:
        hip.gemv_norm2(*args, eps=eps)
    # residual stream: one norm + one add, rounded once
    report("gemv_norm2 vs oracle: residual stream", y1, y1_ref, *tol(dt, y1_ref.std().item(), k=1.5))
    # projection: its input row is the kernel's own normalised row rounded to the dtype, which differs from the oracle's rounding of ITS
    # row by an ulp in a fraction of the K = 3 584 elements, on top of the output rounding: a linear output is within 2 % (bf16; 0.4 %
    # fp16) of the outputs' spread + the usual relative part (a 5-sigma bound over 8-57 k outputs)
    lin = (2e-2 if dt == torch.bfloat16 else 4e-3) * lin_std
    if glu:
        # gelu(gate) * up: the error of each factor is scaled by the other one, so the bound is per element:
        #   |err| <= lin * (|gelu(gate)| + 1.13 |up|) + relative part     (1.13 = max |gelu'|)
        scale = (O.gelu_tanh(g_ref).abs() + 1.13 * u_ref.abs() + 1e-3)
        rt = tol(dt, 1.0, k=2)[1]
        report("gemv_norm2 vs oracle: projection (error / per-element scale)", (out.float().cpu() - out_ref) / (lin * scale + rt * out_ref.abs()),
               torch.zeros_like(out_ref), 1.0, 0.0)
    else:
        report("gemv_norm2 vs oracle: projection", out, out_ref, lin, tol(dt, 1.0)[1])


@pytest.mark.parametrize("dt", DTYPES)
def test_attn_cross2_vs_oracle(hip, dt):
    """vidi_attn_cross2 + vidi_attn_merge2 (T2V and T2A of one decode step, Gemma2-9B head geometry: 8 kv heads x 256, G = 2) against
    flash-attn's published definition (`sdpa_reference`, xattn.py:141-263) on both modalities' keys, image keys partly masked"""
    HD, nkv, G, Lq, Na, Nb, za, zb = 256, 8, 2, 1, 3000, 1100, 5, 2
    nq = nkv * G
    start_b = (Na + 63) // 64 * 64
    ntile = (start_b + Nb + 63) // 64
    q = seeded((Lq, nq, HD), 150, dtype=dt)
    ka = seeded((Na, nkv, HD), 151, dtype=dt); va = seeded((Na, nkv, HD), 152, dtype=dt)
    kb = seeded((Nb, nkv, HD), 153, dtype=dt); vb = seeded((Nb, nkv, HD), 154, dtype=dt)
    mask_a = torch.ones(Na, dtype=torch.bool); mask_a[5: Na: 7] = False
    scale, cap = HD ** -0.5, 50.0
    ref_a = _cross_ref(q, ka, va, mask_a, scale, cap, G)
    ref_b = _cross_ref(q, kb, vb, torch.ones(Nb, dtype=torch.bool), scale, cap, G)
    kc_a, vt_a = pack_kv_cache(ka, va, ntile, 0)
    kc_b, vt_b = pack_kv_cache(kb, vb, ntile, start_b)
    kc, vtc = dev(kc_a + kc_b), dev(vt_a + vt_b)                                    # disjoint regions of one cache (zeros elsewhere)
    ma = torch.zeros((Na + 63) // 64 * 64, dtype=torch.uint8); ma[:Na] = mask_a.to(torch.uint8)
    R = Lq * G
    Rpad = 32
    wa = hip.attn_cross_workspace(za, nkv, Rpad, HD, "cuda"); wb = hip.attn_cross_workspace(zb, nkv, Rpad, HD, "cuda")
    hip.attn_cross2(dev(q.reshape(Lq, nq * HD).contiguous()), kc, vtc, dict(mask=dev(ma), opart=wa[0], ml=wa[1], key_start=0, n_keys=Na, zsplit=za),
                    dict(mask=None, opart=wb[0], ml=wb[1], key_start=start_b, n_keys=Nb, zsplit=zb),
                    R=R, Rpad=Rpad, G=G, nkv=nkv, HD=HD, ntile64=ntile, scale=scale, softcap=cap)
    oa = torch.zeros((Lq, nq * HD), dtype=dt, device="cuda"); ob = torch.zeros_like(oa)
    hip.attn_merge2(wa[0], wa[1], oa, za, False, wb[0], wb[1], ob, zb, False, nkv=nkv, R=R, Rpad=Rpad, G=G, HD=HD)
    report("attn_cross2 vs oracle: image keys", oa, ref_a, *tol(dt, max(0.05, ref_a.std().item()), k=2))
    report("attn_cross2 vs oracle: audio keys", ob, ref_b, *tol(dt, max(0.05, ref_b.std().item()), k=2))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("pos,window,dyn", [(70, 0, False), (200, 64, True), (5883, 0, False)])
def test_attn_text_decode_merge2_vs_oracle(hip, dt, pos, window, dyn):
    """vidi_attn_text_decode_merge2 at Gemma2-9B head geometry: (a) the T2T role = RoPE of the new q / k, cache append, causal (windowed)
    softcapped attention over the text cache (modeling_gemma2.py:248-288 under FA2), (b) the merge role = the LSE identity over the
    split-KV partials (`merge_partials`).  pos = 5883 is the largest position whose score plan fills the kernel's 64 KB of LDS exactly
    (lcap 5884): the merge role's staging shares that allocation."""
    nq, nkv, HD, B = 16, 8, 256, 1
    G, kvd = nq // nkv, nkv * HD
    Lmax = max(256, pos + 2) if not dyn else 256
    if pos == 5883:
        assert hip.attn_text_decode_fits(nq=nq, nkv=nkv, HD=HD, Lmax=Lmax, window=window, pos0=pos)
        assert not hip.attn_text_decode_fits(nq=nq, nkv=nkv, HD=HD, Lmax=Lmax + 8, window=window, pos0=pos + 4)
    qkv = seeded((B, nq * HD + 2 * kvd), 160, dtype=dt)
    cos, sin = O.rope_cos_sin(torch.full((B, 1), pos), HD, 10000.0, dt)
    kc0 = seeded((B, Lmax, kvd), 161, dtype=dt); vc0 = seeded((B, Lmax, kvd), 162, dtype=dt)
    kmask = torch.ones((B, Lmax), dtype=torch.uint8); kmask[0, 2: pos: 5] = 0
    sc, cap = HD ** -0.5, 50.0
    # ---- oracle: T2T ----
    qn = qkv[:, : nq * HD].float().view(B, 1, nq, HD).transpose(1, 2)
    kn = qkv[:, nq * HD: nq * HD + kvd].float().view(B, 1, nkv, HD).transpose(1, 2)
    vn = qkv[:, nq * HD + kvd:].float().view(B, 1, nkv, HD).transpose(1, 2)
    qr, kr = O.apply_rope(qn, kn, cos.float(), sin.float())
    kr = kr.to(dt).float()                                                          # the cache stores the dtype
    kk = torch.cat([kc0[:, :pos].float().view(B, pos, nkv, HD).transpose(1, 2), kr], dim=2)
    vv = torch.cat([vc0[:, :pos].float().view(B, pos, nkv, HD).transpose(1, 2), vn], dim=2)
    kj = torch.arange(pos + 1)[None, :]
    allowed = kj <= pos
    if window:
        allowed = allowed & (kj >= pos - window)
    km = kmask[:, : pos + 1].bool().clone(); km[:, pos] = True
    allowed = allowed[None, None] & km[:, None, None, :]
    add = torch.zeros(allowed.shape); add[~allowed] = float("-inf")
    ref_t = O.sdpa_reference(qr, O.repeat_kv(kk, G), O.repeat_kv(vv, G), sc, cap, add).transpose(1, 2).reshape(B, nq * HD)
    # ---- oracle: merge of the partials (running maxima are kept in log2 units by the kernels) ----
    R, Rpad, WA, WB = B * G, 32, 5, 3
    g = torch.Generator().manual_seed(163)
    parts, refs = [], []
    for W in (WA, WB):
        op = torch.randn((W, nkv, Rpad, HD), generator=g)
        m2 = torch.randn((W, nkv, Rpad), generator=g) * 3
        l = torch.rand((W, nkv, Rpad), generator=g) + 0.5
        m2[0, :, 0] = float("-inf")                                                 # an empty partial
        merged = O.merge_partials(op, m2 * math.log(2.0), l)                        # [nkv, Rpad, HD]
        refs.append(merged[:, :R].reshape(nkv, B, G, HD).permute(1, 0, 2, 3).reshape(B, nq * HD))
        parts.append((dev(op), dev(torch.stack([m2, l], dim=-1).contiguous())))
    kc, vc = dev(kc0.clone()), dev(vc0.clone())
    o_t = torch.full((B, nq * HD), float("nan"), dtype=dt, device="cuda")
    oa = torch.full_like(o_t, float("nan")); ob = torch.full_like(o_t, float("nan"))
    pos_dev = torch.tensor([pos], dtype=torch.int32, device="cuda") if dyn else None
    hip.attn_text_decode_merge2(dev(qkv), kc, vc, dev(kmask), dev(cos.reshape(B, HD).contiguous()), dev(sin.reshape(B, HD).contiguous()), o_t,
                                (parts[0][0], parts[0][1], oa, WA, False), (parts[1][0], parts[1][1], ob, WB, False), R=R, Rpad=Rpad,
                                B=B, Lmax=Lmax, nq=nq, nkv=nkv, HD=HD, window=window, scale=sc, softcap=cap, pos0=pos, pos_dev=pos_dev)
    report("attn_text_decode_merge2 vs oracle: T2T", o_t, ref_t, *tol(dt, 0.3, k=2))
    report("attn_text_decode_merge2 vs oracle: appended K", kc[:, pos].float().cpu(), kr.transpose(1, 2).reshape(B, kvd), *tol(dt, 1.0))
    assert torch.equal(vc[:, pos].cpu().view(torch.int16), qkv[:, nq * HD + kvd:].contiguous().view(torch.int16))
    report("attn_text_decode_merge2 vs oracle: image merge", oa, refs[0], *tol(dt, refs[0].std().item()))
    report("attn_text_decode_merge2 vs oracle: audio merge", ob, refs[1], *tol(dt, refs[1].std().item()))
