// Non-causal multi-head self-attention for the encoder towers (SigLIP: N=729, d=72; Whisper:
// N=1500, d=64) — the MFMA-bound attention of the hot path (reference call sites: HF
// SiglipAttention / WhisperAttention under attn_implementation="flash_attention_2",
// Vidi1.5_9B/vidi/model/lmm/dattn/multimodal.py:44-57).
//
// Layout (produced by the QKV GEMM's MODE_QKV_VT epilogue):
//   QK  [B*N, ldqk]  row-major, Q of head h at column h*D, K at column koff + h*D
//   Vt  [B][H][D][Npad]  V transposed, key order permuted inside every 16-key slab (perm16) so
//        that the PV MFMA's contraction slots line up with the registers of the swapped QK^T tile
//   O   [B*N, ldo]   row-major, head h at column h*D
//
// Per block: one (batch item, head, 128-query tile); 4 waves x 32 queries.  Scores are computed
// "swapped" (S^T = K Q^T) so each lane owns one query column: softmax statistics are per lane,
// P feeds the PV MFMA as the B operand straight from registers, and O^T[d][q] gives 8-byte stores.
// K/Vt tiles (64 keys) are register-staged into padded, conflict-free LDS rows with the next
// tile's global loads issued before the current tile's MFMAs.
//
// Algorithmic FLOPs = 4*N*N*D per (batch, head).
#include "kernels.h"
#include <stdlib.h>


template <typename T, int D, int ABL = 0, int NBUF = 2>
__global__ __launch_bounds__(256) void attn_self_kernel(AttnSelfParams p) {
    constexpr int KS = (D + 15) / 16;          // k16 steps of the QK^T contraction
    constexpr int NCH = D / 8;                 // 16-byte chunks per head row
    constexpr int DT = (D + 31) / 32;          // 32-wide output d tiles
    constexpr int KCH = (2 * KS) | 1;          // K-tile row stride in chunks (odd => conflict-free)
    constexpr int KROW = KCH * 16;
    constexpr int VROW = 9 * 16;               // 64 positions = 8 chunks + 1 pad chunk
    constexpr int KL = (64 * NCH + 255) / 256; // staging loads per thread (K tile)
    constexpr int VL = (D * 8 + 255) / 256;    // staging loads per thread (Vt tile)
    constexpr int ORW = (NCH % 2 == 0) ? (NCH + 1) * 16 : (NCH + 2) * 16;      // output staging row: odd number of 16-B chunks
    constexpr int KVBYTES = 64 * KROW + DT * 32 * VROW;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * KVBYTES > 128 * ORW ? NBUF * KVBYTES : 128 * ORW];   // NBUF-deep K/V ring

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nqt = (p.N + 127) / 128, per_b = nqt * p.H;
    int qt, h, b;
    {
        const int L = blockIdx.x, nb = gridDim.x;
        const int nfull = (p.B / 8) * 8;                     // frames that can be dealt round-robin over the 8 XCDs
        // Only when a head's output segment is whole 128-byte lines (D=64): with D=72 the 16 heads write
        // pieces of the SAME lines and co-locating them on one XCD measured 2x slower stores (line contention).
        if ((D * 2) % 128 == 0 && L < nfull * per_b) {
            const int xcd = L & 7, j = L >> 3;
            b = 8 * (j / per_b) + xcd;
            const int w = j % per_b;
            h = w / nqt; qt = w % nqt;
        } else if ((D * 2) % 128 == 0) {                      // remainder frames: plain order
            const int w = L - nfull * per_b;
            b = nfull + w / per_b;
            h = (w % per_b) / nqt; qt = (w % per_b) % nqt;
        } else {                                              // q-tile fastest, then head, then frame
            qt = L % nqt; h = (L / nqt) % p.H; b = L / per_b;
        }
        (void)nb;
    }
    const int q = qt * 128 + wave * 32 + l31;
    const int qc = min(q, p.N - 1);

    // zero the padding that MFMAs read but staging never writes
    for (int bufi = 0; bufi < NBUF; ++bufi) {
        char* zK = smem + bufi * KVBYTES;
        char* zV = zK + 64 * KROW;
        for (int i = tid; i < 64 * KCH; i += 256) {
            const int c = i % KCH;
            if (c >= NCH) *(u32x4*)(zK + (i / KCH) * KROW + c * 16) = u32x4{0, 0, 0, 0};
        }
        for (int i = tid; i < DT * 32 * 9; i += 256) {
            const int d = i / 9, c = i % 9;
            if (d >= D || c == 8) *(u32x4*)(zV + d * VROW + c * 16) = u32x4{0, 0, 0, 0};
        }
    }

    // Q fragments (B operand: column = query, contraction chunk = 2s + hi)
    u32x4 qf[KS];
    {
        const u16* qrow = p.QK + ((size_t)b * p.N + qc) * p.ldqk + h * D;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 2 * s + hi;
            qf[s] = (c < NCH) ? *(const u32x4*)(qrow + c * 8) : u32x4{0, 0, 0, 0};
        }
    }

    const u16* kbase_ptr = p.QK + (size_t)b * p.N * p.ldqk + p.koff + h * D;
    const u16* vbase_ptr = p.Vt + ((size_t)b * p.H + h) * D * p.Npad;

    u32x4 kreg[KL], vreg[VL];
    auto issue_loads = [&](int kb) {
#pragma unroll
        for (int j = 0; j < KL; ++j) {
            const int i = j * 256 + tid;
            if (i < 64 * NCH) {
                const int key = i / NCH, c = i % NCH;
                kreg[j] = *(const u32x4*)(kbase_ptr + (size_t)min(kb + key, p.N - 1) * p.ldqk + c * 8);
            }
        }
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const int i = j * 256 + tid;
            if (i < D * 8) {
                const int d = i >> 3, c = i & 7;
                vreg[j] = *(const u32x4*)(vbase_ptr + (size_t)d * p.Npad + kb + c * 8);
            }
        }
    };
    auto write_lds = [&](int bufi) {
        char* sK = smem + bufi * KVBYTES;
        char* sV = sK + 64 * KROW;
#pragma unroll
        for (int j = 0; j < KL; ++j) {
            const int i = j * 256 + tid;
            if (i < 64 * NCH) *(u32x4*)(sK + (i / NCH) * KROW + (i % NCH) * 16) = kreg[j];
        }
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const int i = j * 256 + tid;
            if (i < D * 8) *(u32x4*)(sV + (i >> 3) * VROW + (i & 7) * 16) = vreg[j];
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;     // fold log2(e): softmax in base 2

    const int ntiles = (p.N + 63) / 64;
    issue_loads(0);
    __syncthreads();
    write_lds(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int kb = t * 64;
        const char* sK = smem + (NBUF == 2 ? (t & 1) : 0) * KVBYTES;
        const char* sV = sK + 64 * KROW;
        if (!(ABL & 2) && t + 1 < ntiles) issue_loads(kb + 64);

        // ---- S^T = K Q^T : two 32-key sub-tiles ------------------------------------------------
        f32x16 s2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s2[u][i] = 0.f;
            if (!(ABL & 8)) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const u32x4 kf = *(const u32x4*)(sK + (u * 32 + l31) * KROW + (2 * s + hi) * 16);
                    s2[u] = T::mfma32(kf, qf[s], s2[u]);
                }
            } else {
                s2[u][0] = __uint_as_float(qf[0][0] & 0x3fffffff);
            }
        }
        // ---- online softmax (per lane = per query; halves hold disjoint keys) ------------------
        float mx = -INFINITY;
        const bool tail = (kb + 64 > p.N);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = s2[u][r] * sc;
                if (tail && (kb + u * 32 + krow32(r, hi) >= p.N)) x = -INFINITY;
                s2[u][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        u32x4 pf[4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[r] = (ABL & 1) ? s2[u][r] : fast_exp2(s2[u][r] - m_new);
                psum += pv[r];
            }
            pf[2 * u] = pack8<T>(pv);
            pf[2 * u + 1] = pack8<T>(pv + 8);
        }
        l_run = l_run * alpha + psum;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        }
        // ---- O^T += Vt P^T ---------------------------------------------------------------------
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                if (ABL & 4) { o[dt][sl] += __uint_as_float(pf[sl][0]); continue; }
                const u32x4 vf = *(const u32x4*)(sV + (dt * 32 + l31) * VROW + (2 * sl + hi) * 16);
                o[dt] = T::mfma32(vf, pf[sl], o[dt]);
            }
        if (!(ABL & 2)) {
            // the other ring slot was last read during tile t-1, which every wave finished before the
            // barrier that ended it -> safe to fill now; ONE barrier per tile
            if constexpr (NBUF == 2) {
                if (t + 1 < ntiles) write_lds((t + 1) & 1);
                __syncthreads();
            } else {
                __syncthreads();
                if (t + 1 < ntiles) write_lds(0);
                __syncthreads();
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if constexpr ((ABL & 16) != 0) {              // ablation: no output stores
        if (o[0][0] * inv == 12345.678f) p.O[0] = 1;
        return;
    }
    // ---- epilogue: O^T registers -> LDS [128 q][D] -> row-contiguous 16-byte global stores.
    //      (per-lane 8-byte stores straight from the MFMA layout touch 32 rows per instruction with
    //      partial 144-byte row segments: measured 10x slower than the whole MFMA loop.)
    __syncthreads();
    {
        char* so = smem + (wave * 32 + l31) * ORW;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = dt * 32 + 8 * j + 4 * hi;
                if (d < D) {
                    const u32x2 ov = {pack2<T>(o[dt][4 * j] * inv, o[dt][4 * j + 1] * inv),
                                      pack2<T>(o[dt][4 * j + 2] * inv, o[dt][4 * j + 3] * inv)};
                    *(u32x2*)(so + d * 2) = ov;
                }
            }
    }
    __syncthreads();
    for (int i = tid; i < 128 * NCH; i += 256) {
        const int row = i / NCH, c = i % NCH;
        const int qq = qt * 128 + row;
        if (qq < p.N)
            *(u32x4*)(p.O + ((size_t)b * p.N + qq) * p.ldo + h * D + c * 8) = *(const u32x4*)(smem + row * ORW + c * 16);
    }
}

int vidi_attn_self_dispatch(const AttnSelfParams& p, int D, int dtype, hipStream_t st) {
    if (p.B <= 0 || p.N <= 0 || p.H <= 0) return VIDI_ERR_SHAPE;
    if (p.Npad % 64 != 0 || p.Npad < ((p.N + 63) / 64) * 64) return VIDI_ERR_SHAPE;
    if ((p.ldqk % 8) || (p.koff % 8) || (p.ldo % 8)) return VIDI_ERR_ALIGN;
    if (((uintptr_t)p.QK & 15) || ((uintptr_t)p.Vt & 15) || ((uintptr_t)p.O & 15)) return VIDI_ERR_ALIGN;
    const dim3 grid(((p.N + 127) / 128) * p.H * p.B);
// K/V ring depth, measured on MI355X (tools/bench_attn.py): D=72 -> 1 buffer (25 KB LDS, 2 blocks/CU:
// 0.76 ms vs 1.12 ms with 2 buffers at B=96,N=729,H=16); D=64 -> 2 buffers (0.48 ms vs 0.53 ms).
#define LAUNCH(TT, DD) hipLaunchKernelGGL((attn_self_kernel<TT, DD, 0, ((DD) == 72 ? 1 : 2)>), grid, dim3(256), 0, st, p)
    if (dtype == VIDI_DT_BF16) {
        if (D == 72) LAUNCH(BF16, 72); else if (D == 64) LAUNCH(BF16, 64); else if (D == 16) LAUNCH(BF16, 16);
        else if (D == 32) LAUNCH(BF16, 32); else return VIDI_ERR_SHAPE;
    } else if (dtype == VIDI_DT_F16) {
        if (D == 72) LAUNCH(F16, 72); else if (D == 64) LAUNCH(F16, 64); else if (D == 16) LAUNCH(F16, 16);
        else if (D == 32) LAUNCH(F16, 32); else return VIDI_ERR_SHAPE;
    } else {
        return VIDI_ERR_DTYPE;
    }
#undef LAUNCH
    return (int)hipGetLastError();
}
