"""Integer / index host logic: bit-exact against the oracle restatement of the reference lines."""
import numpy as np
import torch

import vidi_oracle as O
from vidi_amd.config import tiny, vidi15_9b
from vidi_amd.engine import audio_token_counts, token_budget_hw
from vidi_amd.model import strip_image_token


def test_token_budget_rule():
    for T in [1, 25, 299, 300, 306, 307, 308, 400, 600, 1199, 1200, 1800, 3600, 7200, 10000]:
        assert token_budget_hw(T, 27, 2, 60000) == O.token_budget_hw(T, 27, 2, 60000)
    assert token_budget_hw(3600, 27, 2, 60000) == (10, 10)         # 25 tokens/frame -> 90 000 (BASELINE 60-min config)
    assert token_budget_hw(300, 27, 2, 60000) == (28, 28)          # 196 tokens/frame -> 58 800 (5-min config)


def test_audio_floors():
    cfg = vidi15_9b()
    ocfg = O.OracleConfig()
    for size in [0, 7, 199, 200, 2460, 2999, 3000, 3001, 30000, 360000, 359999]:
        s1, s2 = O.audio_token_counts([size], ocfg)
        assert audio_token_counts(size, cfg) == (int(s1[0]), int(s2[0]))
    assert audio_token_counts(360000, cfg) == (180000, 36000)      # 10 audio tokens / second


def test_strip_image_token_matches_oracle():
    ids = torch.tensor([[2, 5, -200, 9, 9, 0, 0], [2, 7, 8, -200, 6, 4, 3]])
    am = torch.tensor([[1, 1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1, 1]])
    for side in ("right", "left"):
        got, mask, pos = strip_image_token(ids, am, side)
        rows, m2, p2 = O.strip_image_token(ids, am, side)
        assert torch.equal(mask, m2) and torch.equal(pos, p2)
        for i, r in enumerate(rows):
            assert torch.equal(got[i][mask[i]], r)
    got, mask, pos = strip_image_token(torch.tensor([[2, -200, 11, 12]]))
    assert got.tolist() == [[2, 11, 12]] and pos.tolist() == [[0, 1, 2]]


def test_timestamp_formatting():
    """eval/inference.py:52-66"""
    assert O.format_time_ranges("0.10-0.25, 0.50-0.75", 3600.0) == "00:06:00-00:15:00, 00:30:00-00:45:00"
    assert O.format_time_ranges("garbage", 10.0) == ""
    assert O.format_time_ranges("0.999-1.000", 3661.5) == "01:00:57-01:01:01"


def test_tensor_split_bounds():
    for n, parts in [(10, 3), (3600, 32), (7, 8), (120, 8)]:
        ref = [int(x.shape[0]) for x in torch.tensor_split(torch.zeros(n), parts)]
        got = [e - s for s, e in O.tensor_split_bounds(n, parts)]
        assert got == ref


def test_weight_shapes_cover_engine_needs():
    from vidi_amd.weights import init_random_weights, weight_shapes
    cfg = tiny()
    w = init_random_weights(cfg, dtype=torch.float32)
    assert set(w) == set(weight_shapes(cfg))
    assert w["model.mm_rand_pos_t.mlp.0.weight"].dtype == torch.float32
    assert abs(float(w["model.mm_rand_llm_norm.weight"][0]) - cfg.mm_std) < 1e-7


def _umulhi(a, b):
    return (a * b) >> 32


def test_patch_embed_weight_layout_and_gather_addresses():
    """vidi_patch_embed's contract, emulated on the host: the re-laid weight (`hip.patch_embed_weight`: k = (c*P + dy)*16 + dx, zero
    elsewhere) times the rows the loader gathers — slice ks, chunk cg of output row m = (frame, py, px) reads 8 pixels at
    ((f*3*S + py*P)*S + px*P) + (rr*S + c*(S - P)*S) + 8*(cg & 1), rr = 4 ks + (cg >> 1), c = (rr >= P) + (rr >= 2P) — equals the
    convolution; pixels past the frame / the tensor are finite garbage or zeros that only meet zero weight columns."""
    import torch.nn.functional as F
    from vidi_amd.hip import patch_embed_weight
    for (T, S, P, Hv) in [(2, 98, 14, 8), (1, 384, 14, 4), (2, 64, 16, 4), (3, 40, 8, 4)]:
        side = S // P
        n = side * side
        g = torch.Generator().manual_seed(S + P)
        px = torch.randn((T, 3, S, S), generator=g)
        w = torch.randn((Hv, 3, P, P), generator=g)
        w16 = patch_embed_weight(w, P)
        K = w16.shape[1]
        assert K % 64 == 0 and K >= 3 * P * 16 and float(w16.view(Hv, -1, 16)[:, :, P:].abs().max() if P < 16 else 0.0) == 0.0
        flat = torch.cat([px.reshape(-1), torch.full((4096,), 7.0)])       # what lies behind the tensor must not matter
        rows = torch.zeros((T * n, K))
        for m in range(0, T * n, max(1, (T * n) // 23)):                   # a sample of output rows (first, last, frame boundaries)
            f, rem = divmod(m, n)
            py, pxx = divmod(rem, side)
            base = (f * 3 * S + py * P) * S + pxx * P
            for ks in range(K // 64):
                for cg in range(8):
                    rr = 4 * ks + (cg >> 1)
                    c = int(rr >= P) + int(rr >= 2 * P)
                    off = base + rr * S + c * (S - P) * S + 8 * (cg & 1)
                    rows[m, ks * 64 + cg * 8: ks * 64 + cg * 8 + 8] = flat[off: off + 8]
        ref = F.conv2d(px, w, stride=P).flatten(2).transpose(1, 2).reshape(T * n, Hv)
        sel = list(range(0, T * n, max(1, (T * n) // 23)))
        assert torch.allclose((rows @ w16.T)[sel], ref[sel], atol=2e-4, rtol=1e-4)


def test_conv_window_gather_addresses():
    """vidi_conv_window's contract (Vidi-7B's learned Conv2DPool): slice ks of output row m = (frame, oy, ox) is 64 contiguous channels
    c0 .. of window position (dy, dx) = divmod((ks * 64) // C, k) of the token-major [T, side*side, C] map, divisions by multiply-high
    with magic = floor(2^32 / d) + 1 (d = 1 handled apart)."""
    import torch.nn.functional as F
    udiv = lambda x, d: x if d == 1 else _umulhi(x, (2 ** 32) // d + 1)          # noqa: E731
    for (T, side, C, k, N) in [(2, 7, 64, 4, 8), (1, 27, 128, 14, 4), (3, 5, 192, 2, 8), (3, 6, 64, 6, 4), (2, 5, 256, 1, 4)]:
        oc = side - k + 1
        n = oc * oc
        g = torch.Generator().manual_seed(side * 100 + k)
        f = torch.randn((T, side * side, C), generator=g)
        w = torch.randn((N, C, k, k), generator=g)
        wg = w.permute(0, 2, 3, 1).reshape(N, -1)
        flat = f.reshape(-1)
        K = k * k * C
        out = torch.zeros((T * n, N))
        for m in range(T * n):
            fr = udiv(m, n); rem = m - fr * n
            oy = udiv(rem, oc); ox = rem - oy * oc
            assert (fr, oy, ox) == (m // n, (m % n) // oc, (m % n) % oc)
            base = ((fr * side + oy) * side + ox) * C
            row = torch.zeros(K)
            for ks in range(K // 64):
                spc = C >> 6
                dd = udiv(ks, spc); c0 = (ks - dd * spc) << 6
                dy = udiv(dd, k); dx = dd - dy * k
                koff = (dy * side + dx) * C + c0
                row[ks * 64: ks * 64 + 64] = flat[base + koff: base + koff + 64]
            out[m] = wg @ row
        ref = F.conv2d(f.reshape(T, side, side, C).permute(0, 3, 1, 2), w).permute(0, 2, 3, 1).reshape(T * n, N)
        assert torch.allclose(out, ref, atol=5e-3, rtol=1e-4)


def test_attention_d72_tile_dma_piece_map():
    """The d = 72 encoder-attention kernel's branch-free tile DMA (attn_self_rm.hip, issue_dma_flat), emulated on the host: every wave
    issues exactly five 1-KB pieces (64 lanes x 16 bytes, lane-linear in LDS); together they must fill the K image [64 keys][9 chunks],
    the V main block [64 keys][8 chunks, chunk c of key r at slot c ^ 2 (r & 3)] and the V tail block [64 keys][1 chunk] — every byte
    written, pieces that overlap (K piece 8 by waves 0 and 1, the tail by waves 2 and 3) writing the same source bytes."""
    D, NCH, MCH, MAINC = 72, 9, 8, 64
    KBYTES, VMAIN = 64 * NCH * 16, 64 * MCH * 16
    BUF = KBYTES + VMAIN + 1024
    lds = {}                                                         # LDS byte offset of a 16-byte chunk -> ("K" | "V", key row, first d)

    def put(dst, lane, src):
        off = dst + lane * 16
        assert lds.setdefault(off, src) == src, (off, lds[off], src)

    vswz = lambda r: 2 * (r & 3)                                     # noqa: E731
    for wave in range(4):
        issued = 0
        for lane in range(64):
            tid = wave * 64 + lane
            for j in range(2):                                       # K pieces wave and 4 + wave
                i = j * 256 + tid
                put((j * 4 + wave) * 1024, lane, ("K", i // NCH, (i % NCH) * 8))
            if wave < 2:                                             # slot 2: K piece 8 (chunks 512 .. 575) ...
                i = 512 + lane
                put(8 * 1024, lane, ("K", i // NCH, (i % NCH) * 8))
            else:                                                    # ... or the V tail block: key = lane, d 64 .. 71
                put(KBYTES + VMAIN, lane, ("V", lane, MAINC))
            for j in range(2):                                       # V main pieces wave and 4 + wave: 8 keys per piece
                pc = j * 4 + wave
                row = pc * (64 // MCH) + lane // MCH
                put(KBYTES + pc * 1024, lane, ("V", row, ((lane % MCH) ^ vswz(row)) * 8))
            issued = 5
        assert issued == 5
    assert sorted(lds) == list(range(0, BUF, 16))                    # every chunk of the slot, nothing outside it
    for r in range(64):
        for c in range(NCH):                                         # K image: rows unpadded, no swizzle at 9 chunks per row
            assert lds[(r * NCH + c) * 16] == ("K", r, c * 8)
        for c in range(MCH):                                         # V main block: chunk c of key r sits at slot c ^ 2 (r & 3)
            assert lds[KBYTES + (r * MCH + (c ^ vswz(r))) * 16] == ("V", r, c * 8)
        assert lds[KBYTES + VMAIN + r * 16] == ("V", r, MAINC)
    # the contraction slots the kernel borrows: d = 72 occupies 72 of the 5 x 16 slots of the QK^T k-steps; slot 72 carries (1.0 in Q) x
    # (0 | -inf in K) for keys past N, slot 73 (-max~ in Q) x (1.0 in K): the matrix pipe returns  q.k + bias - max~
    q = torch.randn(D, dtype=torch.float64); k = torch.randn(D, dtype=torch.float64)
    for bias, mt in ((0.0, 3.25), (float("-inf"), -1.5)):
        qx = torch.cat([q, torch.tensor([1.0, -mt]), torch.zeros(6, dtype=torch.float64)])
        kx = torch.cat([k, torch.tensor([bias, 1.0]), torch.zeros(6, dtype=torch.float64)])
        got = float((qx * kx).sum())
        assert got == float("-inf") if bias else abs(got - (float(q @ k) - mt)) < 1e-12
