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
// K/Vt tiles (64 keys) arrive by 16-byte LDS-DMA into a 2-deep ring (one barrier per tile); bank
// conflicts are avoided by permuting the source chunk each lane fetches, not by padding.
// VALU diet (the exp/convert work is co-dominant with the MFMAs at d = 64..72): scale folded into the
// exponent FMA, lazy rescaling (accumulators only touched when a maximum grows by > 2^8), cross-half
// reductions by v_permlane32_swap, softmax denominator accumulated by the PV MFMA through a ones row
// (d = 72), tail masking only in the last tile.  Built with -amdgpu-mfma-vgpr-form so accumulators
// stay in architectural VGPRs (150-160 registers -> 3 waves/SIMD).
//
// Algorithmic FLOPs = 4*N*N*D per (batch, head).
#include "kernels.h"
#include <stdlib.h>


// schedule knobs (tools/build_variant.sh + tools/ab_attn.py build and time alternatives as separate libraries of the same ABI; the
// DIAG variants are timing diagnostics with wrong results and exist only in such lab builds)
#ifndef VIDI_ATTN_PRIO
#define VIDI_ATTN_PRIO 1               // bit 0: raise the wave's priority over its QK^T MFMAs, bit 1: over its PV MFMAs.  Same-box A/B
#endif                                 // (tools/ab_attn.py, N = 729, d = 72): 0 -> 640, 1 -> 658, 2 -> 649, 3 -> 652 useful TFLOP/s, all bit-identical
#ifndef VIDI_ATTN_PRIO_LEVEL
#define VIDI_ATTN_PRIO_LEVEL 1
#endif
#define VIDI_STR2(x) #x
#define VIDI_STR(x) VIDI_STR2(x)
#define VIDI_ATTN_PRIO_HI(bit) do { if (VIDI_ATTN_PRIO & (bit)) asm volatile("s_setprio " VIDI_STR(VIDI_ATTN_PRIO_LEVEL) ::: "memory"); } while (0)
#define VIDI_ATTN_PRIO_LO(bit) do { if (VIDI_ATTN_PRIO & (bit)) asm volatile("s_setprio 0" ::: "memory"); } while (0)

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_self_kernel(AttnSelfParams p) {
    constexpr int KS = (D + 15) / 16;          // k16 steps of the QK^T contraction
    constexpr int NCH = D / 8;                 // 16-byte chunks per head row
    constexpr int DT = (D + 31) / 32;          // 32-wide output d tiles
    // LDS images are written by 16-byte LDS-DMA (lane-linear), so rows are unpadded and bank conflicts are
    // avoided by permuting which global chunk each lane fetches: chunk c of row r sits at slot c ^ swz(r).
    //   K tile  [64 keys][NCH chunks]   (D=72: 144-B rows are conflict-free as they are; slot 9 of the padded
    //                                    5th k16-step reads the next row's first chunk, multiplied by Q zeros)
    //   Vt tile [DT*32 rows d][8 chunks of 8 key positions]   (rows >= D are never written: their products land
    //                                    in output rows that are not stored)
    constexpr int KBYTES = 64 * NCH * 16, VBYTES = DT * 32 * 128, BUF = KBYTES + VBYTES;
    constexpr int KRND = (64 * NCH + 255) / 256, VRND = (D * 8 + 255) / 256;
    constexpr int ORW = (NCH % 2 == 0) ? (NCH + 1) * 16 : (NCH + 2) * 16;      // output staging row: odd number of 16-B chunks
#ifndef VIDI_ATTN_LDS_PAD
#define VIDI_ATTN_LDS_PAD 0            // lab knob: extra LDS per block (fewer co-resident blocks per CU)
#endif
    __shared__ __attribute__((aligned(16))) char smem[(2 * BUF > 128 * ORW ? 2 * BUF : 128 * ORW) + VIDI_ATTN_LDS_PAD];   // 2-deep K/V ring
    auto kswz = [](int r) { return NCH == 8 ? ((r >> 1) & 7) : (NCH == 4 ? ((r >> 2) & 3) : 0); };

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nqt = (p.N + 127) / 128, per_b = nqt * p.H;
    int qt, h, b;
    {
        const int L = blockIdx.x;
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
        } else if ((p.B * p.H) % 8 == 0) {
            // D=72: the q-tiles of one (frame, head) run on ONE XCD (they share its K/V through that L2), while
            // neighbouring heads - whose 144-byte output segments share 128-byte lines - go to different XCDs
            const int xcd = L & 7, j = L >> 3;
            const int unit = (j / nqt) * 8 + xcd;
            qt = j % nqt; b = unit / p.H; h = unit % p.H;
        } else {                                              // q-tile fastest, then head, then frame
            qt = L % nqt; h = (L / nqt) % p.H; b = L / per_b;
        }
    }
    const int q = qt * 128 + wave * 32 + l31;
    const int qc = min(q, p.N - 1);

    // Q fragments (B operand: column = query, contraction chunk = 2s + hi); chunks past D are zero
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

    // per-thread DMA pieces (tile independent): K piece j = (key row, source column), V piece j = source offset
    int krow[KRND], kcol[KRND];
    const u16* vsrc[VRND];
#pragma unroll
    for (int j = 0; j < KRND; ++j) {
        const int i = j * 256 + tid;
        const int row = i / NCH, cs = i % NCH;
        krow[j] = row; kcol[j] = (cs ^ kswz(row)) * 8;
    }
#pragma unroll
    for (int j = 0; j < VRND; ++j) {
        const int i = j * 256 + tid;
        const int d = min(i >> 3, D - 1), cs = i & 7;
        vsrc[j] = vbase_ptr + (size_t)d * p.Npad + (cs ^ ((d >> 1) & 7)) * 8;
    }
    auto issue_dma = [&](int kb, int bufi) {
        char* sK = smem + bufi * BUF;
        char* sV = sK + KBYTES;
#pragma unroll
        for (int j = 0; j < KRND; ++j) {
            const int ib = j * 256 + wave * 64;              // wave-uniform: whole 64-chunk pieces only
            if (ib < 64 * NCH) glds16(kbase_ptr + (size_t)min(kb + krow[j], p.N - 1) * p.ldqk + kcol[j], sK + ib * 16);
        }
#pragma unroll
        for (int j = 0; j < VRND; ++j) {
            const int ib = j * 256 + wave * 64;
            if (ib < D * 8) glds16(vsrc[j] + kb, sV + ib * 16);
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                // m_run in base-2 logit units (score * scale * log2 e)
    const float sc = p.scale * 1.4426950408889634f;      // fold log2(e): softmax in base 2

    const int ntiles = (p.N + 63) / 64;
    const int ksw = kswz(l31), vsw = (l31 >> 1) & 7;
    // When D is not a multiple of 32 the last Vt d-tile has rows that no DMA ever writes.  Row D is set to
    // ones: the PV MFMA then accumulates the softmax denominator (sum of the T-rounded P) in output row D for
    // free, and the 32 VALU adds per tile disappear.  The other spare rows are zeroed.
    constexpr bool kOnesRow = (DT * 32 > D);
    if constexpr (kOnesRow) {
        const unsigned one2 = (unsigned)T::from_f32(1.0f) * 0x10001u;
        for (int i = tid; i < 2 * (DT * 32 - D) * 8; i += 256) {
            const int bufi = i / ((DT * 32 - D) * 8), w = i % ((DT * 32 - D) * 8);
            const int d = D + w / 8, c = w % 8;
            const unsigned v = (d == D) ? one2 : 0u;
            *(u32x4*)(smem + bufi * BUF + KBYTES + d * 128 + c * 16) = u32x4{v, v, v, v};
        }
    }
    issue_dma(0, 0);

    for (int t = 0; t < ntiles; ++t) {
        const int kb = t * 64;
#ifdef VIDI_ATTN_DIAG_NOBAR                               // timing diagnostic only (wrong results): no rendezvous, no refill
        if (t == 0) { wait_vmcnt<0>(); __syncthreads(); }
#else
        wait_vmcnt<0>();                                  // my pieces of tile t have landed ...
        __syncthreads();                                  // ... everyone's have, and tile t-1's buffer is free
        if (t + 1 < ntiles) issue_dma(kb + 64, (t + 1) & 1);
#endif
        const char* sK = smem + (t & 1) * BUF;
        const char* sV = sK + KBYTES;
        const bool tail = (kb + 64 > p.N);

        // ---- S^T = K Q^T (swapped: lane = query column) for both 32-key sub-tiles; the two accumulator
        //      chains are interleaved so no MFMA waits for the previous one's result
        f32x16 s2[2];
        VIDI_ATTN_PRIO_HI(1);
#ifdef VIDI_ATTN_KPRE
        {   // lab variant: all K fragments of the tile requested before the first MFMA (one LDS wait instead of one per pair)
            u32x4 kfa[KS][2];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) kfa[s][u] = *(const u32x4*)(sK + (u * 32 + l31) * (NCH * 16) + (((2 * s + hi) ^ ksw) << 4));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u) s2[u] = T::mfma32(kfa[s][u], qf[s], s == 0 ? zero16 : s2[u]);
        }
#else
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x4 kf = *(const u32x4*)(sK + (u * 32 + l31) * (NCH * 16) + (((2 * s + hi) ^ ksw) << 4));
                s2[u] = T::mfma32(kf, qf[s], s == 0 ? zero16 : s2[u]);
            }
#endif
        VIDI_ATTN_PRIO_LO(1);
        // ---- per sub-tile: online softmax (VALU) then O^T += Vt P^T (MFMA); the softmax of sub-tile 1
        //      runs under the PV MFMAs of sub-tile 0 and under the other waves of this SIMD
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x4 vf[DT][2];                              // issued before the arithmetic so they land under it
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                vf[dt][0] = *(const u32x4*)(sV + (dt * 32 + l31) * 128 + (((4 * u + hi) ^ vsw) << 4));
                vf[dt][1] = *(const u32x4*)(sV + (dt * 32 + l31) * 128 + (((4 * u + 2 + hi) ^ vsw) << 4));
            }
            if (tail) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + u * 32 + krow32(r, hi) >= p.N) s2[u][r] = -INFINITY;
            }
            // online softmax on raw scores (scale folded into the exponent)
            float mx = s2[u][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s2[u][r]);
            mx = xhalf_max(mx);
            // Lazy rescaling: the running reference m_run only moves when some query's maximum outgrows it by
            // more than 2^TAU; until then P = exp2(s - m_run) may exceed 1 (<= 2^TAU: harmless in fp32
            // accumulators and in the 8-bit-exponent P operand) and the 3*16 accumulator registers are left alone.
            // With a 64-lane wave an exact running maximum would rescale on nearly every sub-tile.
            constexpr float TAU = 8.0f;
            const float m_cand = fmaxf(m_run, mx * sc);
            if (__any(m_cand > m_run + TAU)) {                 // also the very first sub-tile (m_run = -inf)
                const float alpha = fast_exp2(m_run - m_cand); // 0 on the first sub-tile (o = l = 0 there)
                m_run = m_cand;
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
            }
            float ps0 = 0.f, ps1 = 0.f;
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
#ifdef VIDI_ATTN_DIAG_NOEXP                               // timing diagnostic only (wrong results): the exponentials dropped
                pv[r] = __builtin_fmaf(s2[u][r], sc, -m_run);
                pv[r + 1] = __builtin_fmaf(s2[u][r + 1], sc, -m_run);
#else
                pv[r] = fast_exp2(__builtin_fmaf(s2[u][r], sc, -m_run));
                pv[r + 1] = fast_exp2(__builtin_fmaf(s2[u][r + 1], sc, -m_run));
#endif
                if constexpr (!kOnesRow) { ps0 += pv[r]; ps1 += pv[r + 1]; }
            }
            const u32x4 pf0 = pack8<T>(pv), pf1 = pack8<T>(pv + 8);
            if constexpr (!kOnesRow) l_run += ps0 + ps1;
            VIDI_ATTN_PRIO_HI(2);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[dt] = T::mfma32(vf[dt][0], pf0, o[dt]);
                o[dt] = T::mfma32(vf[dt][1], pf1, o[dt]);
            }
            VIDI_ATTN_PRIO_LO(2);
        }
    }

    float l_tot;
    if constexpr (kOnesRow) {
        // output row D of the last d-tile: register (D%32 -> j = row/8, e = row%4) of the lanes with hi == (row/4)%2
        constexpr int ROW = D - (DT - 1) * 32, REG = 4 * (ROW / 8) + (ROW % 4), HI = (ROW / 4) % 2;
        const float mine = (hi == HI) ? o[DT - 1][REG] : 0.f;
        l_tot = xhalf_sum(mine);
    } else {
        l_tot = xhalf_sum(l_run);
    }
    const float inv = 1.0f / l_tot;
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
#define LAUNCH(TT, DD) hipLaunchKernelGGL((attn_self_kernel<TT, DD>), grid, dim3(256), 0, st, p)
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
