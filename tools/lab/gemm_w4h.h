// LAB ONLY (tools/lab/gemm_lab.hip): the persistent 4-wave GEMM in a 256(n) x 128(m) geometry — wave tile 128 x 64 = 8 x 4 accumulator
// tiles = 128 AGPRs, i.e. HALF the accumulator file.  Probe for the "two accumulator sets" design: how fast is the K
// loop when a wave reads 8 W + 4 X fragments per 32 MFMAs (1.5x the LDS bytes per MFMA of the 256 x 256 kernel) and a K slice is 384
// rows = 12 DMA pieces per 64 MFMAs (1.5x the DMA bytes per MFMA)?  Single accumulator set, plain bias + residual (+ statistics) epilogue.
#pragma once
#include "gemm_w4n.h"

struct W4HGeom {
    static constexpr int BN = 256, BM = 128, BK = 64, NT = 256, TN = 8, TM = 4, ROWB = 128;
    static constexpr int WROWS = 64;                                          // rows per wave
    static constexpr int STAGE_BYTES = (BN + BM) * ROWB, RING = 2 * STAGE_BYTES;
    static constexpr int SCR_ROW = 128 * 2 + 16, SCR_BYTES = 16 * SCR_ROW;   // per-wave epilogue scratch: 16 rows x (128 + 16 cols + pad)
    static constexpr int CST_OFF = RING + 4 * SCR_BYTES, CST_BYTES = 4096;
    static constexpr int LDS_BYTES = CST_OFF + 2 * CST_BYTES;
    static_assert(STAGE_BYTES == 49152 && LDS_BYTES <= 163840, "LDS plan");
};

template <typename T, typename EPI, typename LAB = LabNone>
__global__ __launch_bounds__(256) void gemm_w4h_kernel(GemmParams p) {
    using G = W4HGeom;
    constexpr int BN = G::BN, BM = G::BM, BK = G::BK, NT = G::NT, TN = G::TN, TM = G::TM, ROWB = G::ROWB, STAGE_BYTES = G::STAGE_BYTES;
    constexpr int NPH = TN * TM;                           // 32 MFMAs per phase (k32 step)
    static_assert(EPI::bias && EPI::act == ACT_NONE && EPI::res == 1 && !EPI::lnf && !EPI::heads, "bias + residual(row m) [+ statistics]");
    auto swz = [](int row) { return (row >> 1) & 7; };
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int l15 = lane & 15, hi = lane >> 4;
    const int sw = swz(l15);                               // (wn*128 + a*16, 256 + wn*16 and wm*112 + b*16 are multiples of 16: swz sees l15 only)
    const int w_row_off = (wn * 128 + l15) * ROWB;
    const int x_row_off = BN * ROWB + (wm * G::WROWS + l15) * ROWB;
    const int nk = p.K / BK;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles = tiles_n * ((p.M + BM - 1) / BM);

    unsigned long long t_acc[4] = {0, 0, 0, 0}, t_mark = 0;
    auto stamp = [&](int slot) {
        if constexpr (LAB::stamps) { const unsigned long long t = __builtin_readcyclecounter(); t_acc[slot] += t - t_mark; t_mark = t; }
    };
    if constexpr (LAB::stamps) t_mark = __builtin_readcyclecounter();

    // ---- DMA: piece q < 4 = X rows q*32 + (tid >> 3), piece 4 + j = W rows j*32 + (tid >> 3); tile-invariant lane offsets ----
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
        if (q < 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(NEXT ? srdXn : srdX, (__attribute__((address_space(3))) void*)(buf + BN * ROWB + (q * NT + wave * 64) * 16), 16,
                                                     offX, (unsigned)(k0 + q * 32 * p.ldx) * 2u, 0, 0);
        } else {
            const int j = q - 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(NEXT ? srdWn : srdW, (__attribute__((address_space(3))) void*)(buf + (j * NT + wave * 64) * 16), 16, offW,
                                                     (unsigned)(k0 + j * 32 * p.ldw) * 2u, 0, 0);
        }
    };
    auto bar = [&]() { __builtin_amdgcn_s_barrier(); };
    auto aread = [](float x) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(x)); return v; };

    f32x4 acc[TN][TM];
    u32x4 fW[2][TN], fX[2][TM];
    auto rdW = [&](const char* buf, int a, int s) {
        return *(const u32x4*)(buf + w_row_off + a * 16 * ROWB + (((4 * s + hi) ^ sw) << 4));
    };
    auto rdX = [&](const char* buf, int b, int s) { return *(const u32x4*)(buf + x_row_off + b * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
#define VIDI_PIN __builtin_amdgcn_sched_barrier(0)

    // ---- epilogue constants: bias[n0 .. n0 + 255] (bf16, 512 B) -> LDS by wave 0 during the tile's first iteration ----
    __amdgpu_buffer_rsrc_t srdC0 = rsrc_of(p.bias, (unsigned long long)p.N * 2);
    int tpar = 0;
    auto issue_cst = [&]() {
        if constexpr (LAB::no_dma) return;
        char* dst = smem + G::CST_OFF + tpar * G::CST_BYTES;
        if (wave == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srdC0, (__attribute__((address_space(3))) void*)dst, 16, (unsigned)(n0 + 8 * lane) * 2u, 0, 0, 0);
    };

    // one K iteration: 2 x 32 MFMAs on slice kt (gemm_w4.h's schedule compressed; pieces 0..3 = X, 4..11 = W; fragments 4 X + 8 W per k32 step)
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
            if (i < 4) fX[1][i] = rdX(bufc, i, 1);                                               // 4 X-fragment reads
            if (i == 6) wait_lgkm0();
            if (i == 7) bar();                                                                   // barrier 1: X part of bufc is dead
            if (i == 8 || i == 11 || i == 14 || i == 17) dma((i - 8) / 3);                       // X pieces 0..3
            if (i == 9 || i == 12 || i == 15 || i == 18) fW[1][(i - 9) / 3] = rdW(bufc, (i - 9) / 3, 1);   // W reads 0..3
            if (i == 20 || i == 22 || i == 24 || i == 26) fW[1][4 + ((i - 20) >> 1)] = rdW(bufc, 4 + ((i - 20) >> 1), 1); // W reads 4..7
            if (i == 28) wait_lgkm0();
            if (i == 29) bar();                                                                  // barrier 2: W part of bufc is dead
            if (i == 30) dma(4);
            if (i == 31) dma(5);
            VIDI_PIN;
        }
        // ---------------- phase 2: step-1 MFMAs ----------------
#pragma unroll
        for (int i = 0; i < NPH; ++i) {
            const int a = i / TM, b = i % TM;
            T::mfma16_agpr(acc[a][b], fW[1][a], fX[1][b]);
            if (i == 1) dma(6);
            if (i == 9) dma(7);
            if (i == 12) dma(8);
            if (i == 15) dma(9);
            if (i == 18) dma(10);
            if (i == 23) dma(11);
            // X part of slice kt+1 landed?  behind it in flight: its 8 W pieces + this iteration's 7 pieces
            if (i == 3) { if constexpr (LAB::no_dma) {} else if (!NEXT || more) wait_vm<15>(); else wait_vm<8>(); }
            if (i == 4) bar();                                                                   // barrier 3
            if (i >= 5 && i <= 8) fX[0][i - 5] = rdX(bufn, i - 5, 0);                            // 4 X reads
            if (i == 19) { if constexpr (LAB::no_dma) {} else if (!NEXT || more) wait_vm<11>(); else wait_vm<0>(); }
            if (i == 20) bar();                                                                  // barrier 4: W pieces landed
            if (i >= 21 && i <= 28) fW[0][i - 21] = rdW(bufn, i - 21, 0);                        // 8 W reads
            VIDI_PIN;
        }
    };

    // ---- epilogue: strips of 16 rows through the wave's private scratch; main part (128 columns) + tail (16 columns) ----
    char* scr = smem + G::RING + wave * G::SCR_BYTES;
    constexpr int SROW = G::SCR_ROW;
    constexpr bool stats_on = EPI::stats;
    auto epilogue = [&](int em0, int en0) {
        const char* cst = smem + G::CST_OFF + tpar * G::CST_BYTES;
        u32x2 bq[TN];
#pragma unroll
        for (int a = 0; a < TN; ++a) bq[a] = *(const u32x2*)(cst + (wn * 128 + a * 16 + 4 * hi) * 2);
        const int rr = lane >> 4, cc = lane & 15;                    // main read-back: 4 rows x 16 chunks of 8 columns per instruction
        const int nmain = en0 + wn * 128 + cc * 8;
        const unsigned rows_here = (unsigned)min(p.M - em0, BM);
        const unsigned strips = (unsigned)((p.N + 127) >> 7), tile4 = (unsigned)(en0 >> 7);
        const __amdgpu_buffer_rsrc_t srdY = rsrc_of(p.Y + (size_t)em0 * p.ldy, (unsigned long long)rows_here * (unsigned)p.ldy * 2ull);
        const __amdgpu_buffer_rsrc_t srdR = rsrc_of(p.R + (size_t)em0 * p.ldr, (unsigned long long)rows_here * (unsigned)p.ldr * 2ull);
        __amdgpu_buffer_rsrc_t srdS;
        (void)srdS;
        constexpr unsigned OOB = 0x80000000u;
        const unsigned vY = (nmain < p.N && !LAB::no_store) ? (unsigned)((wm * G::WROWS + rr) * p.ldy + nmain) * 2u : OOB;
        const unsigned vR = (nmain < p.N) ? (unsigned)((wm * G::WROWS + rr) * p.ldr + nmain) * 2u : OOB;
        unsigned vS = OOB;
        if constexpr (stats_on) {
            srdS = rsrc_of(p.stat_part + (size_t)em0 * strips * 2, (unsigned long long)rows_here * strips * 8ull);
            if (cc == 0 && nmain < p.N) vS = ((unsigned)(wm * G::WROWS + rr) * strips + tile4 + (unsigned)wn) * 8u;
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
                *(u32x2*)(scr + l15 * SROW + a * 32 + 4 * hi * 2) = o;
            }
        };
        constexpr int RD = 3;                                           // residual ring: strips b .. b + 1 in flight while strip b is stored
        u32x4 val[4], res[RD][4];
        auto load_res = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
            if constexpr (b < TM) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    res[b % RD][j] = __builtin_amdgcn_raw_buffer_load_b128(srdR, vR + (unsigned)((b * 16 + j * 4) * p.ldr) * 2u, 0, 0);
            }
        };
        auto fetch = [&]() {
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = *(const u32x4*)(scr + (j * 4 + rr) * SROW + cc * 16);
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
            float sv[8];
            u32x4 outv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) outv[j] = add_res(val[j], res[b % RD][j], sv[2 * j], sv[2 * j + 1]);
            if constexpr (stats_on) row16_sum8(sv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned rowoff = (unsigned)(b * 16 + j * 4);
                if constexpr (stats_on)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(sv[2 * j]), __float_as_uint(sv[2 * j + 1])}, srdS, vS + rowoff * strips * 8u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(outv[j], srdY, vY + rowoff * (unsigned)p.ldy * 2u, 0, 0);
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
        step(std::integral_constant<int, 3>{});
    };

    // =========================================== tile loop ===========================================
    using TT = std::true_type; using FF = std::false_type;
    int vb = blockIdx.x;
    locate(vb, m0, n0);
    srdW = make_srd(p.W + (size_t)n0 * p.ldw, p.N - n0, p.ldw);
    srdX = make_srd(p.X + (size_t)m0 * p.ldx, p.M - m0, p.ldx);
#pragma unroll
    for (int q = 0; q < 12; ++q) piece(smem, 0, q, FF{});
#pragma unroll
    for (int q = 0; q < 12; ++q) piece(smem + STAGE_BYTES, 1, q, FF{});
    if constexpr (!LAB::no_dma) wait_vm<12>();
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
static int launch_w4h(const GemmParams& p, hipStream_t st) {
    auto kern = gemm_w4h_kernel<T, EPI, LAB>;
    static bool attr_done = false;
    static int ncu = 0;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W4HGeom::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        attr_done = true;
    }
    const long long tiles = (long long)((p.N + W4HGeom::BN - 1) / W4HGeom::BN) * ((p.M + W4HGeom::BM - 1) / W4HGeom::BM);
    const int grid = (int)(tiles < ncu ? tiles : ncu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), W4HGeom::LDS_BYTES, st, p);
    return (int)hipGetLastError();
}
