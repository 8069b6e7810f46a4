// Persistent 4-wave MFMA GEMM, 288(n) x 224(m) x 64 tiles: the geometry for output widths that are multiples of 288 but not of 256
// (SigLIP's N = 1 152 = 4 x 288: out_proj and fc2 were 4.5 tiles of 256 wide, i.e. 10 % of their MFMA work ran on zero columns).
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )        same operands, layouts and K loop as gemm_w4.h (read its header first)
//
// Why this shape costs nothing in the K loop: a wave owns 9 x 7 accumulator tiles of 16 x 16 (252 AGPRs instead of 256) and issues 126
// MFMA 16x16x32 per K slice instead of 128; per 32-wide k step it reads 9 W + 7 X fragments (16, as before); a K slice is 288 W rows + 224
// X rows = 512 LDS rows (64 KB, as before) = 9 + 7 DMA pieces of 32 rows (16, as before).  The iteration's schedule is gemm_w4.h's with the
// piece / fragment indices re-split 7 | 9 instead of 8 | 8 and the two in-loop waits recounted (19 instead of 18 pieces in flight behind
// the next slice's X pieces).
//
// Wave (wn, wm): rows m0 + wm*112 .. +112; columns n0 + wn*128 .. +128 (eight "main" column tiles) PLUS n0 + 256 + wn*16 .. +16 (one
// "tail" tile).  The main part keeps gemm_w4.h's epilogue geometry (16 lanes x 8 columns per row: whole 256-byte row segments, 16-lane
// row reductions for the LayerNorm statistics); the tail is one more read-back / store per 16-row strip on half a wave (2 lanes per row).
// Partial sums for the consumer's LayerNorm: FOUR entries per row and tile — (sum, sum of squares) of the two main parts and of the two
// tails — stat_part[M][4 * N / 288][2]; vidi_ln_finalize sums whatever entries a row has (vidi_gemm_stats_strips tells callers the count).
// Epilogue I/O is gemm_w4.h's form 2 (per-tile buffer descriptors, hardware range check).
//
// Instantiated for the bias + residual(row m) epilogue with and without statistics (MODE_PLAIN).  Y is bit-identical to the 256-wide
// kernel's (same fp32 accumulation order per output element).  Roofline: MFMA-bound; algorithmic FLOPs = 2*M*N*K.
#pragma once
#include "gemm_w4.h"

struct W4NGeom {
    static constexpr int BN = 288, BM = 224, BK = 64, NT = 256, TN = 9, TM = 7, ROWB = 128;
    static constexpr int WROWS = 112;                                         // rows per wave
    static constexpr int STAGE_BYTES = (BN + BM) * ROWB, RING = 2 * STAGE_BYTES;
    static constexpr int SCR_ROW = 144 * 2 + 16, SCR_BYTES = 16 * SCR_ROW;   // per-wave epilogue scratch: 16 rows x (128 + 16 cols + pad)
    static constexpr int CST_OFF = RING + 4 * SCR_BYTES, CST_BYTES = 6144;   // bias (576 B) | folded LayerNorm: colsum 2 KB, shift 2 KB, (mean, rstd) 2 KB
    static constexpr int LDS_BYTES = CST_OFF + 2 * CST_BYTES;
    static_assert(STAGE_BYTES == 65536 && LDS_BYTES <= 163840, "LDS plan");
};

// partial-sum entries per row that the statistics epilogue of this geometry writes for an output width N
__host__ __device__ inline int w4n_stat_strips(int N) { return 4 * ((N + W4NGeom::BN - 1) / W4NGeom::BN); }
// the widths this geometry takes over from the 256-wide kernel
__host__ __device__ inline bool w4n_takes(int N) { return N >= 288 && N % 288 == 0 && N % 256 != 0; }

#ifndef VIDI_W4N_LABTAIL
#define VIDI_W4N_LABTAIL 0              // lab builds only (wrong results): 1 = the tail's global loads / stores skipped, 2 = no tail work in the epilogue
#endif
template <typename T, typename EPI, typename LAB = LabNone>
__global__ __launch_bounds__(256) void gemm_w4n_kernel(GemmParams p) {
    using G = W4NGeom;
    constexpr int BN = G::BN, BM = G::BM, BK = G::BK, NT = G::NT, TN = G::TN, TM = G::TM, ROWB = G::ROWB, STAGE_BYTES = G::STAGE_BYTES;
    constexpr int NPH = TN * TM;                           // 63 MFMAs per phase (k32 step)
    // two epilogue families: bias + residual(row m) [+ statistics] (out_proj / fc2), and the folded LayerNorm with the head-major store
    // (SigLIP's q | k | v projection: N = 3 456 = 12 x 288, i.e. 13.5 tiles of 256 wide)
    constexpr bool LNH = EPI::lnf == 1 && EPI::heads;
    static_assert((EPI::bias && EPI::act == ACT_NONE && EPI::res == 1 && EPI::lnf == 0 && !EPI::heads) ||
                  (LNH && EPI::act == ACT_NONE && EPI::res == 0 && !EPI::stats), "bias + residual(row m) [+ statistics], or LN-fold + head-major");
    auto swz = [](int row) { return (row >> 1) & 7; };
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int l15 = lane & 15, hi = lane >> 4;
    const int sw = swz(l15);                               // (wn*128 + a*16, 256 + wn*16 and wm*112 + b*16 are multiples of 16: swz sees l15 only)
    const int w_row_off = (wn * 128 + l15) * ROWB;
    const int w_tail_off = (256 + wn * 16 + l15) * ROWB;
    const int x_row_off = BN * ROWB + (wm * G::WROWS + l15) * ROWB;
    const int nk = p.K / BK;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_n * ((p.M + BM - 1) / BM);

    unsigned long long t_acc[4] = {0, 0, 0, 0}, t_mark = 0;
    auto stamp = [&](int slot) {
        if constexpr (LAB::stamps) { const unsigned long long t = __builtin_readcyclecounter(); t_acc[slot] += t - t_mark; t_mark = t; }
    };
    if constexpr (LAB::stamps) t_mark = __builtin_readcyclecounter();

    // ---- DMA: piece q < 7 = X rows q*32 + (tid >> 3), piece 7 + j = W rows j*32 + (tid >> 3); tile-invariant lane offsets ----
    const int r0 = tid >> 3, cg0 = (tid & 7) ^ swz(r0);
    const unsigned offW = (unsigned)(r0 * p.ldw + cg0 * 8) * 2u, offX = (unsigned)(r0 * p.ldx + cg0 * 8) * 2u;
    __amdgpu_buffer_rsrc_t srdW, srdX, srdWn, srdXn;
    int m0 = 0, n0 = 0, nm0 = 0, nn0 = 0;
    int pb = 0;
    auto rsrc_of = [](const void* base, unsigned long long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
    };
    auto make_srd = [&](const u16* base, int rows_left, int ld) {
        return rsrc_of(base, ((unsigned long long)(rows_left - 1) * (unsigned)ld + (unsigned)p.K) * 2ull);
    };
    auto locate = [&](int vb, int& tm0, int& tn0) {
        int tile_m, tile_n;
        tile_of_block(p, BN, BM, vb, tiles, tile_m, tile_n);
        tn0 = tile_n * BN; tm0 = tile_m * BM;
    };
    auto piece = [&](char* buf, int ks, int q, auto next_t) {
        if constexpr (LAB::no_dma) return;
        constexpr bool NEXT = decltype(next_t)::value;
        const int k0 = ks * BK;
        if (q < 7) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(NEXT ? srdXn : srdX, (__attribute__((address_space(3))) void*)(buf + BN * ROWB + (q * NT + wave * 64) * 16), 16,
                                                     offX, (unsigned)(k0 + q * 32 * p.ldx) * 2u, 0, 0);
        } else {
            const int j = q - 7;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(NEXT ? srdWn : srdW, (__attribute__((address_space(3))) void*)(buf + (j * NT + wave * 64) * 16), 16, offW,
                                                     (unsigned)(k0 + j * 32 * p.ldw) * 2u, 0, 0);
        }
    };
    auto bar = [&]() { __builtin_amdgcn_s_barrier(); };
    auto aread = [](float x) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(x)); return v; };

    f32x4 acc[TN][TM];
    u32x4 fW[2][TN], fX[2][TM];
    auto rdW = [&](const char* buf, int a, int s) {
        return *(const u32x4*)(buf + (a < 8 ? w_row_off + a * 16 * ROWB : w_tail_off) + (((4 * s + hi) ^ sw) << 4));
    };
    auto rdX = [&](const char* buf, int b, int s) { return *(const u32x4*)(buf + x_row_off + b * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
#define VIDI_PIN __builtin_amdgcn_sched_barrier(0)

    // ---- epilogue constants: bias[n0 .. n0 + 287] (bf16, 576 B) -> LDS by wave 0 during the tile's first iteration ----
    //      folded LayerNorm: wave 0 colsum[n0 ..] (fp32: columns 0..255 to +0, 256..287 to +1024), wave 1 shift[n0 ..] the same way to +2048,
    //      waves 2, 3 (mean, rstd) of rows m0 .. m0 + 255 to +4096 (rows and columns out of range read as zeros)
    __amdgpu_buffer_rsrc_t srdC0;
    if constexpr (LNH) srdC0 = rsrc_of(wave == 0 ? p.ln_s : (wave == 1 ? p.ln_c : p.ln_stats), wave < 2 ? (unsigned long long)p.N * 4 : (unsigned long long)p.M * 8);
    else srdC0 = rsrc_of(p.bias, (unsigned long long)p.N * 2);
    int tpar = 0;
    auto issue_cst = [&]() {
        if constexpr (LAB::no_dma) return;
        char* dst = smem + G::CST_OFF + tpar * G::CST_BYTES;
        if constexpr (LNH) {
            const unsigned voff = wave < 2 ? (unsigned)(n0 + 4 * lane) * 4u : (unsigned)(m0 + (wave - 2) * 128 + 2 * lane) * 8u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdC0, (__attribute__((address_space(3))) void*)(dst + (wave < 2 ? wave * 2048 : 4096 + (wave - 2) * 1024)), 16, voff, 0, 0, 0);
            if (wave < 2)                                           // the 32 tail columns (lanes 8.. read past the tile's columns: never used)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdC0, (__attribute__((address_space(3))) void*)(dst + wave * 2048 + 1024), 16, (unsigned)(n0 + 256 + 4 * lane) * 4u, 0, 0, 0);
        } else {
            if (wave == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdC0, (__attribute__((address_space(3))) void*)dst, 16, (unsigned)(n0 + 8 * lane) * 2u, 0, 0, 0);
        }
    };

    // one K iteration: 2 x 63 MFMAs on slice kt (gemm_w4.h's schedule; pieces 0..6 = X, 7..15 = W; fragments 7 X + 9 W per k32 step)
    auto body = [&](int kt, auto first_t, auto next_t, bool more) {
        constexpr bool FIRST = decltype(first_t)::value, NEXT = decltype(next_t)::value;
        char* bufc = smem + ((pb + kt) & 1) * STAGE_BYTES;
        const char* bufn = smem + ((pb + kt + 1) & 1) * STAGE_BYTES;
        const int ks2 = NEXT ? kt + 2 - nk : kt + 2;
        auto dma = [&](int q) {
            if constexpr (NEXT) { if (more) piece(bufc, ks2, q, next_t); }
            else piece(bufc, ks2, q, next_t);
        };
        VIDI_PIN;
        if constexpr (FIRST) issue_cst();
        VIDI_PIN;
        // ---------------- phase 1: step-0 MFMAs ----------------
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            const int a = i / TM, b = i % TM;
            if constexpr (FIRST) T::mfma16_agpr_first(acc[a][b], fW[0][a], fX[0][b]);
            else T::mfma16_agpr(acc[a][b], fW[0][a], fX[0][b]);
            if (i < 14 && (i & 1) == 0) fX[1][i >> 1] = rdX(bufc, i >> 1, 1);                       // 7 X-fragment reads
            if (i == 19) wait_lgkm0();
            if (i == 20) bar();                                                                  // barrier 1: X part of bufc is dead
            if (i >= 22 && i <= 38 && ((i - 22) & 3) == 0) dma((i - 22) >> 2);                   // X pieces 0..4
            if (i >= 24 && i <= 40 && ((i - 24) & 3) == 0) fW[1][(i - 24) >> 2] = rdW(bufc, (i - 24) >> 2, 1);   // W reads 0..4
            if (i == 42 || i == 44 || i == 46 || i == 48) fW[1][5 + ((i - 42) >> 1)] = rdW(bufc, 5 + ((i - 42) >> 1), 1); // W reads 5..8
            if (i == 51) wait_lgkm0();
            if (i == 52) bar();                                                                  // barrier 2: W part of bufc is dead
            if (i == 53) dma(5);
            if (i == 56) dma(6);
            if (i == 58) dma(7);
            if (i == 61) dma(8);
            VIDI_PIN;
        }
        // ---------------- phase 2: step-1 MFMAs ----------------
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            const int a = i / TM, b = i % TM;
            T::mfma16_agpr(acc[a][b], fW[1][a], fX[1][b]);
            if (i == 1) dma(9);
            if (i == 21) dma(10);
            if (i == 23) dma(11);
            if (i == 25) dma(12);
            if (i == 32) dma(13);
            if (i == 36) dma(14);
            if (i == 60) dma(15);
            // X part of slice kt+1 landed?  behind it in flight: its 9 W pieces + this iteration's 10 pieces
            if (i == 3) { if constexpr (LAB::no_dma) {} else if (!NEXT || more) wait_vm<19>(); else wait_vm<9>(); }
            if (i == 4) bar();                                                                   // barrier 3
            if (i >= 5 && i <= 17 && ((i - 5) & 1) == 0) fX[0][(i - 5) >> 1] = rdX(bufn, (i - 5) >> 1, 0);       // 7 X reads
            if (i == 40) { if constexpr (LAB::no_dma) {} else if (!NEXT || more) wait_vm<15>(); else wait_vm<0>(); }
            if (i == 41) bar();                                                                  // barrier 4: W pieces landed
            if (i >= 42 && i <= 58 && ((i - 42) & 1) == 0) fW[0][(i - 42) >> 1] = rdW(bufn, (i - 42) >> 1, 0);   // 9 W reads
            VIDI_PIN;
        }
    };

    // ---- epilogue: strips of 16 rows through the wave's private scratch; main part (128 columns) + tail (16 columns) ----
    char* scr = smem + G::RING + wave * G::SCR_BYTES;
    constexpr int SROW = G::SCR_ROW;
    constexpr bool stats_on = EPI::stats;
    // ---- folded LayerNorm + head-major store (q | k | v): y = rstd_m * (acc - mean_m * s_n) + c_n, rounded once, stored to
    //      Y[which][frame][head][token][d] (gemm_w4.h's arithmetic and address map; a 288-column tile is four whole heads of d = 72) ----
    auto epilogue_lnh = [&](int em0, int en0) {
        const char* cst = smem + G::CST_OFF + tpar * G::CST_BYTES;
        f32x4 lnS[TN], lnC[TN];
        float lnMu[TM], lnRs[TM];
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int col = (a < 8 ? wn * 128 + a * 16 : 256 + wn * 16) + 4 * hi;          // (columns 0..255 at +0, 256..287 at +1024: contiguous)
            lnS[a] = *(const f32x4*)(cst + col * 4);
            lnC[a] = *(const f32x4*)(cst + 2048 + col * 4);
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const f32x2_t ms = *(const f32x2_t*)(cst + 4096 + (wm * G::WROWS + b * 16 + l15) * 8);
            lnMu[b] = ms[0];
            lnRs[b] = ms[1];
        }
        const int rr = lane >> 4, cc = lane & 15;                    // main read-back: 4 rows x 16 chunks of 8 columns per instruction
        const int tr = (lane >> 1) & 15, tc = lane & 1;              // tail read-back: 16 rows x 2 chunks on lanes 0..31
        const int nmain = en0 + wn * 128 + cc * 8, ntail = en0 + 256 + wn * 16 + tc * 8;
        const bool tl = lane < 32;
        auto head_col = [&](int n) {                                 // (which, head, d) of a column: fixed for the tile
            const int nc = min(n, p.N - 8), hdim = p.hm_heads * p.hm_hd;
            const int which = nc / hdim, nh = nc - which * hdim, hh = nh / p.hm_hd;
            return ((size_t)which * (p.M / p.hm_seq) * p.hm_heads + hh) * p.hm_seq * p.hm_hd + (nh - hh * p.hm_hd);
        };
        const size_t hm_main = head_col(nmain), hm_tail = head_col(ntail);
        const bool ok_main = nmain < p.N && !LAB::no_store, ok_tail = tl && ntail < p.N && !LAB::no_store;
        auto stage = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const float t = -lnRs[b] * lnMu[b];
            const f32x2 tt = {t, t}, rr2 = {lnRs[b], lnRs[b]};
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = aread(acc[a][b][e]);
                const f32x2 u0 = __builtin_elementwise_fma(tt, f32x2{lnS[a][0], lnS[a][1]}, f32x2{lnC[a][0], lnC[a][1]});
                const f32x2 u1 = __builtin_elementwise_fma(tt, f32x2{lnS[a][2], lnS[a][3]}, f32x2{lnC[a][2], lnC[a][3]});
                const f32x2 y0 = __builtin_elementwise_fma(rr2, f32x2{v[0], v[1]}, u0);
                const f32x2 y1 = __builtin_elementwise_fma(rr2, f32x2{v[2], v[3]}, u1);
                const u32x2 o = {pack2<T>(y0[0], y0[1]), pack2<T>(y1[0], y1[1])};
                *(u32x2*)(scr + l15 * SROW + (a < 8 ? a * 32 : 256) + 4 * hi * 2) = o;
            }
        };
        u32x4 val[4], valt;
        auto fetch = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = *(const u32x4*)(scr + (j * 4 + rr) * SROW + cc * 16);
            valt = *(const u32x4*)(scr + tr * SROW + 256 + tc * 16);
        };
        auto store = [&](int b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = em0 + wm * G::WROWS + b * 16 + j * 4 + rr;
                if (ok_main && m < p.M) {
                    const int fr = (int)__umulhi((unsigned)m, p.hm_magic), tok = m - fr * p.hm_seq;      // m / seq, m % seq
                    *(u32x4*)(p.Y + hm_main + ((size_t)fr * p.hm_heads * p.hm_seq + tok) * p.hm_hd) = val[j];
                }
            }
            const int mt = em0 + wm * G::WROWS + b * 16 + tr;
            if (ok_tail && mt < p.M) {
                const int fr = (int)__umulhi((unsigned)mt, p.hm_magic), tok = mt - fr * p.hm_seq;
                *(u32x4*)(p.Y + hm_tail + ((size_t)fr * p.hm_heads * p.hm_seq + tok) * p.hm_hd) = valt;
            }
        };
        auto step = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
            VIDI_PIN;
            fetch();
            VIDI_PIN;
            if constexpr (b + 1 < TM) stage(std::integral_constant<int, b + 1>{});
            VIDI_PIN;
            store(b);
            VIDI_PIN;
        };
        stage(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
    };
    auto epilogue = [&](int em0, int en0) {
        if constexpr (LNH) { epilogue_lnh(em0, en0); return; } else {
        const char* cst = smem + G::CST_OFF + tpar * G::CST_BYTES;
        u32x2 bq[TN];
#pragma unroll
        for (int a = 0; a < TN; ++a) bq[a] = *(const u32x2*)(cst + ((a < 8 ? wn * 128 + a * 16 : 256 + wn * 16) + 4 * hi) * 2);
        const int rr = lane >> 4, cc = lane & 15;                    // main read-back: 4 rows x 16 chunks of 8 columns per instruction
        const int tr = (lane >> 1) & 15, tc = lane & 1;              // tail read-back: 16 rows x 2 chunks on lanes 0..31
        const int nmain = en0 + wn * 128 + cc * 8, ntail = en0 + 256 + wn * 16 + tc * 8;
        const bool tl = lane < 32;
        const unsigned rows_here = (unsigned)min(p.M - em0, BM);
        const unsigned strips = (unsigned)w4n_stat_strips(p.N), tile4 = (unsigned)(en0 / BN) * 4u;
        const __amdgpu_buffer_rsrc_t srdY = rsrc_of(p.Y + (size_t)em0 * p.ldy, (unsigned long long)rows_here * (unsigned)p.ldy * 2ull);
        const __amdgpu_buffer_rsrc_t srdR = rsrc_of(p.R + (size_t)em0 * p.ldr, (unsigned long long)rows_here * (unsigned)p.ldr * 2ull);
        __amdgpu_buffer_rsrc_t srdS;
        (void)srdS;
        constexpr unsigned OOB = 0x80000000u;
        const unsigned vY = (nmain < p.N && !LAB::no_store) ? (unsigned)((wm * G::WROWS + rr) * p.ldy + nmain) * 2u : OOB;
        const unsigned vYt = (tl && ntail < p.N && !LAB::no_store) ? (unsigned)((wm * G::WROWS + tr) * p.ldy + ntail) * 2u : OOB;
        const unsigned vR = (nmain < p.N) ? (unsigned)((wm * G::WROWS + rr) * p.ldr + nmain) * 2u : OOB;
        const unsigned vRt = (tl && ntail < p.N) ? (unsigned)((wm * G::WROWS + tr) * p.ldr + ntail) * 2u : OOB;
        unsigned vS = OOB, vSt = OOB;
        if constexpr (stats_on) {
            srdS = rsrc_of(p.stat_part + (size_t)em0 * strips * 2, (unsigned long long)rows_here * strips * 8ull);
            if (cc == 0 && nmain < p.N) vS = ((unsigned)(wm * G::WROWS + rr) * strips + tile4 + (unsigned)wn) * 8u;
            if (tl && tc == 0 && ntail < p.N) vSt = ((unsigned)(wm * G::WROWS + tr) * strips + tile4 + 2u + (unsigned)wn) * 8u;
        }
        // registers -> scratch (lane: row l15, 4 consecutive columns of each of its 9 column tiles)
        auto stage = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = aread(acc[a][b][e]);
                const u32x2 bv = bq[a];
                v[0] += T::to_f32((u16)(bv[0] & 0xffff)); v[1] += T::to_f32((u16)(bv[0] >> 16));
                v[2] += T::to_f32((u16)(bv[1] & 0xffff)); v[3] += T::to_f32((u16)(bv[1] >> 16));
                const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                *(u32x2*)(scr + l15 * SROW + (a < 8 ? a * 32 : 256) + 4 * hi * 2) = o;
            }
        };
        constexpr int RD = 3;                                           // residual ring: strips b .. b + 1 in flight while strip b is stored
        u32x4 val[4], valt, res[RD][4], rest[RD];
        auto load_res = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
            if constexpr (b < TM) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    res[b % RD][j] = __builtin_amdgcn_raw_buffer_load_b128(srdR, vR + (unsigned)((b * 16 + j * 4) * p.ldr) * 2u, 0, 0);
                if constexpr (VIDI_W4N_LABTAIL == 0) rest[b % RD] = __builtin_amdgcn_raw_buffer_load_b128(srdR, vRt + (unsigned)(b * 16 * p.ldr) * 2u, 0, 0);
                else rest[b % RD] = u32x4{0, 0, 0, 0};
            }
        };
        auto fetch = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = *(const u32x4*)(scr + (j * 4 + rr) * SROW + cc * 16);
            if constexpr (VIDI_W4N_LABTAIL != 2) valt = *(const u32x4*)(scr + tr * SROW + 256 + tc * 16);
            else valt = u32x4{0, 0, 0, 0};
        };
        // x (already T-rounded) + residual; (sum, sum of squares) of the fp32 sums before their rounding (as gemm_w4.h)
        auto add_res = [&](const u32x4& v, const u32x4& r, float& s1, float& s2) {
            float x[8], rv[8];
            unpack8<T>(v, x);
            unpack8<T>(r, rv);
            f32x2_t sa = {0.f, 0.f}, sq = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2_t y = f32x2_t{x[e], x[e + 1]} + f32x2_t{rv[e], rv[e + 1]};
                x[e] = y[0]; x[e + 1] = y[1];
                sa += y;
                sq = __builtin_elementwise_fma(y, y, sq);
            }
            s1 = sa[0] + sa[1]; s2 = sq[0] + sq[1];
            return pack8<T>(x);
        };
        auto store = [&](int b) {
            float sv[8], st1, st2;
            u32x4 outv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) outv[j] = add_res(val[j], res[b % RD][j], sv[2 * j], sv[2 * j + 1]);
            u32x4 outt = {0, 0, 0, 0};
            st1 = st2 = 0.f;
            if constexpr (VIDI_W4N_LABTAIL != 2) outt = add_res(valt, rest[b % RD], st1, st2);
            if constexpr (stats_on && VIDI_W4N_LABTAIL != 2) {
                row16_sum8(sv);
                // the tail's row: lanes 2r, 2r + 1
                asm volatile("s_nop 1\n\t"
                             "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                             "s_nop 0"
                             : "+v"(st1), "+v"(st2));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned rowoff = (unsigned)(b * 16 + j * 4);
                if constexpr (stats_on)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(sv[2 * j]), __float_as_uint(sv[2 * j + 1])}, srdS, vS + rowoff * strips * 8u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(outv[j], srdY, vY + rowoff * (unsigned)p.ldy * 2u, 0, 0);
            }
            if constexpr (VIDI_W4N_LABTAIL == 0) {
                if constexpr (stats_on)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(st1), __float_as_uint(st2)}, srdS, vSt + (unsigned)(b * 16) * strips * 8u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(outt, srdY, vYt + (unsigned)(b * 16) * (unsigned)p.ldy * 2u, 0, 0);
            } else {
                asm volatile("" :: "v"(outt), "v"(st1), "v"(st2));
            }
        };
        load_res(std::integral_constant<int, 0>{});
        load_res(std::integral_constant<int, 1>{});
        auto step = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
            VIDI_PIN;
            fetch();
            load_res(std::integral_constant<int, b + RD - 1>{});
            VIDI_PIN;
            if constexpr (b + 1 < TM) stage(std::integral_constant<int, b + 1>{});
            VIDI_PIN;
            store(b);
            VIDI_PIN;
        };
        stage(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        }
    };

    // =========================================== tile loop ===========================================
    using TT = std::true_type; using FF = std::false_type;
    int vb = blockIdx.x;
    locate(vb, m0, n0);
    srdW = make_srd(p.W + (size_t)n0 * p.ldw, p.N - n0, p.ldw);
    srdX = make_srd(p.X + (size_t)m0 * p.ldx, p.M - m0, p.ldx);
#pragma unroll
    for (int q = 0; q < 16; ++q) piece(smem, 0, q, FF{});
#pragma unroll
    for (int q = 0; q < 16; ++q) piece(smem + STAGE_BYTES, 1, q, FF{});
    if constexpr (!LAB::no_dma) wait_vm<16>();
    bar();
#pragma unroll
    for (int b = 0; b < TM; ++b) fX[0][b] = rdX(smem, b, 0);
#pragma unroll
    for (int a = 0; a < TN; ++a) fW[0][a] = rdW(smem, a, 0);
    stamp(0);
    while (true) {
        const int nvb = vb + gridDim.x;
        const bool has_next = nvb < tiles;
        if (has_next) {
            locate(nvb, nm0, nn0);
            srdWn = make_srd(p.W + (size_t)nn0 * p.ldw, p.N - nn0, p.ldw);
            srdXn = make_srd(p.X + (size_t)nm0 * p.ldx, p.M - nm0, p.ldx);
        }
        body(0, TT{}, FF{}, true);
        int kt = 1;
        for (; kt + 2 < nk; ++kt) body(kt, FF{}, FF{}, true);
        body(kt, FF{}, TT{}, has_next);
        body(kt + 1, FF{}, TT{}, has_next);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // MFMA result -> reader hazard (the MFMAs are asm statements)
        stamp(1);
        if constexpr (LAB::no_epilogue) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) asm volatile("" ::"a"(acc[a][b]));
        } else {
            epilogue(m0, n0);
        }
        stamp(3);
        if (!has_next) break;
        vb = nvb; m0 = nm0; n0 = nn0; srdW = srdWn; srdX = srdXn;
        pb = (pb + nk) & 1;
        tpar ^= 1;
    }
#undef VIDI_PIN
    if constexpr (LAB::stamps) {
        if (p.dbg && tid == 0 && blockIdx.x < 1024) {
            unsigned long long* d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = t_acc[0]; d[1] = t_acc[1]; d[2] = t_acc[2]; d[3] = t_acc[3]; d[4] = 1;
        }
    }
}

template <typename T, typename EPI, typename LAB = LabNone>
static int launch_w4n(const GemmParams& p, hipStream_t st) {
    auto kern = gemm_w4n_kernel<T, EPI, LAB>;
    static bool attr_done = false;
    static int ncu = 0;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W4NGeom::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        attr_done = true;
    }
    const long long tiles = (long long)((p.N + W4NGeom::BN - 1) / W4NGeom::BN) * ((p.M + W4NGeom::BM - 1) / W4NGeom::BM);
    const int grid = (int)(tiles < ncu ? tiles : ncu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), W4NGeom::LDS_BYTES, st, p);
    return (int)hipGetLastError();
}
