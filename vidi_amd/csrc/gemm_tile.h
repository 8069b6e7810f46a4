// MFMA GEMM tile kernel for the Vidi hot path (gfx950) — device code shared by the product dispatch (gemm.hip) and the
// bench-only lab (tools/lab/gemm_lab.hip).
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )        X:[M,K] activations, W:[N,K] nn.Linear weight
//
// Both operands are K-contiguous, so both are staged with 16-byte LDS-DMA (`global_load_lds`) into an XOR-swizzled LDS image
// and read back as 8-element MFMA fragments with ds_read_b128.  The MFMA is issued "swapped": the weight tile is the A operand
// and the activation tile the B operand, so the accumulator holds D[row = n][col = m].  Each lane then owns ONE token and four
// consecutive output features per register quad, which makes every epilogue a plain 8-byte LDS store and lets GeGLU / bias /
// residual / KV-cache layouts be applied per lane without shuffles.
//
// Schedules (SCHED):
//   SCHED_RING  (0)  32x32x16 MFMA, k16 steps, the next stage's DMA issued right after the barrier — small problems (128x128 tile)
//   SCHED_LATE  (6)  16x16x32 MFMA, 8 waves x (128x64), 2 waves/SIMD, DMA issued after the first fragment reads (round-1 kernel)
//   (the large projections run on the persistent 4-wave kernel of gemm_w4.h; this template serves small problems, odd tile
//    shapes and the A/B baseline)
//
// LAB is a compile-time policy: the product library instantiates LabNone only, so no environment variable can make
// libvidi_hip.so skip work; the timing diagnostics (skipped DMA / epilogue, phase time stamps) exist only in tools/lab builds.
//
// Roofline: MFMA-bound (2.5 PFLOP/s dense bf16/fp16).  Algorithmic FLOPs = 2*M*N*K.
#pragma once
#include "kernels.h"
#include <type_traits>

enum { SCHED_RING = 0, SCHED_LATE = 6 };

struct LabNone {
    static constexpr bool no_dma = false;        // timing only: K loop without the DMA refills (wrong results)
    static constexpr bool no_epilogue = false;   // timing only: return before the epilogue (no output)
    static constexpr bool no_store = false;      // timing only: staged epilogue without the global stores
    static constexpr bool stamps = false;        // s_memtime stamps of block 0 into p.dbg
};

template <int MI> struct AccOf { typedef f32x16 type; };
template <> struct AccOf<16> { typedef f32x4 type; };

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// block -> tile.  order 0: every XCD walks a contiguous range of the grouped tile order (GROUP_M m-tiles x all n-tiles per
// group, m fastest): 32 CUs of an XCD share 4 X slabs x 8 W slabs in their 4 MB L2.
// order 1: the same per-XCD wave shape, but the 8 XCDs sweep ADJACENT groups (group g -> XCD g % 8) instead of 8 distant
// ranges of M, so at any time the chip works on one compact panel of 8*GROUP_M m-tiles: the W matrix and the panel's X slabs
// stay in the 256 MB Infinity Cache while every XCD re-reads them.
__device__ __forceinline__ void tile_of_block(const GemmParams& p, int BN, int BM, int b, int nb, int& tile_m, int& tile_n) {
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int GROUP_M = p.group_m;
    const int per_group = GROUP_M * tiles_n;
    int rb;
    if (p.order == 1) {
        // blocks b = 8*idx + xcd: XCD `xcd` takes group (8*(idx / per_group) + xcd); the ragged last super-group (fewer than 8
        // full groups) falls back to the contiguous split below so the map stays a bijection
        const int ngroups_full = tiles_m / GROUP_M;                    // groups with GROUP_M m-tiles
        const int nsuper = ngroups_full / 8;                           // super-groups of 8 full groups
        const int covered = nsuper * 8 * per_group;                    // blocks inside full super-groups
        if (b < covered) {
            const int xcd = b & 7, idx = b >> 3;
            const int sg = idx / per_group, in_g = idx - sg * per_group;
            const int g = sg * 8 + xcd;
            tile_m = g * GROUP_M + in_g % GROUP_M;
            tile_n = in_g / GROUP_M;
            return;
        }
        // tail: remaining blocks [covered, nb) over the remaining tiles, XCD-contiguous
        const int nt = nb - covered, bt = b - covered;
        const int q = nt >> 3, r = nt & 7, xcd = bt & 7, idx = bt >> 3;
        rb = covered + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    } else {
        const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
        rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int g = rb / per_group, first_m = g * GROUP_M;
    const int gm = min(tiles_m - first_m, GROUP_M);
    const int in_g = rb - g * per_group;
    tile_m = first_m + in_g % gm;
    tile_n = in_g / gm;
}

template <typename T, int BN, int BM, int WN, int WM, int STAGES, int MODE, bool REPKV, int SCHED, int MI, typename LAB = LabNone>
__global__ __launch_bounds__(WN* WM * 64) void gemm_kernel(GemmParams p) {
    constexpr int BK = 64;
    constexpr int NT = WN * WM * 64;
    constexpr int TN = BN / WN / MI, TM = BM / WM / MI;
    static_assert((MI == 32 && SCHED == SCHED_RING) || (MI == 16 && TM <= TN && SCHED == SCHED_LATE), "MI / schedule");
    constexpr int CPR = BK / 8;                       // 16-byte chunks per LDS row
    constexpr int ROWB = BK * 2;                      // LDS row bytes
    // 16-byte chunk c of LDS row r lives at chunk c ^ swz(r); chosen so that every ds_read_b128 lane group of the
    // fragment reads hits 64 distinct banks (32-row x 2-chunk lanes for MI=32, 16-row x 4-chunk lanes for MI=16)
    auto swz = [](int row) { return (row >> 1) & 7; };
    constexpr int W_LOADS = BN * CPR / NT, X_LOADS = BM * CPR / NT;
    constexpr int LPT = W_LOADS + X_LOADS;
    constexpr int STAGE_BYTES = (BN + BM) * ROWB;
    static_assert(BN * CPR % NT == 0 && BM * CPR % NT == 0, "tile/threads mismatch");
    static_assert(MODE != MODE_GEGLU || ((TN * MI) % 64 == 0), "GeGLU needs whole 32-gate/32-up row blocks per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WM, wm = wave % WM;

    unsigned long long t_start = 0;
    if constexpr (LAB::stamps) t_start = __builtin_readcyclecounter();

    int tile_n, tile_m;
    tile_of_block(p, BN, BM, blockIdx.x, gridDim.x, tile_m, tile_n);
    const int n0 = tile_n * BN, m0 = tile_m * BM;
    const long long bz = blockIdx.y;
    const u16* Xb = p.X + bz * p.bsX;

    // ---- per-thread DMA source pointers (k0 added per stage) ----------------------------------
    const u16* wsrc[W_LOADS];
    const u16* xsrc[X_LOADS];
    int xcol[X_LOADS];
#pragma unroll
    for (int j = 0; j < W_LOADS; ++j) {
        const int pidx = j * NT + tid, row = pidx >> 3, cl = pidx & (CPR - 1), cg = cl ^ swz(row);
        const int n = min(n0 + row, p.N - 1);
        wsrc[j] = p.W + (size_t)n * p.ldw + cg * 8;
    }
#pragma unroll
    for (int j = 0; j < X_LOADS; ++j) {
        const int pidx = j * NT + tid, row = pidx >> 3, cl = pidx & (CPR - 1), cg = cl ^ swz(row);
        const int m = min(m0 + row, p.M - 1);
        xsrc[j] = Xb + (size_t)m * p.ldx;
        xcol[j] = cg * 8;
        if constexpr (!REPKV) xsrc[j] += cg * 8;
    }

    // loads [jw0,jw1) of the W tile and [jx0,jx1) of the X tile of K-slice kt into ring slot `stage`
    auto load_part = [&](int stage, int kt, int jw0, int jw1, int jx0, int jx1) {
        char* sW = smem + stage * STAGE_BYTES;
        char* sX = sW + BN * ROWB;
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < W_LOADS; ++j) if (j >= jw0 && j < jw1) glds16<0>(wsrc[j] + k0, sW + (j * NT + wave * 64) * 16);
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) {
            if (j < jx0 || j >= jx1) continue;
            if constexpr (REPKV) {
                const int k = k0 + xcol[j];
                const int phys = (k / (p.rep_g * p.rep_hd)) * p.rep_hd + (k % p.rep_hd);
                glds16<0>(xsrc[j] + phys, sX + (j * NT + wave * 64) * 16);
            } else {
                glds16<0>(xsrc[j] + k0, sX + (j * NT + wave * 64) * 16);
            }
        }
    };
    auto load_stage = [&](int stage, int kt) { load_part(stage, kt, 0, W_LOADS, 0, X_LOADS); };

    typename AccOf<MI>::type acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int i = 0; i < (MI == 32 ? 16 : 4); ++i) acc[a][b][i] = 0.f;

    // lane -> (row inside an MI-row tile, 8-element k chunk): 32x32x16 = 32 rows x 2 chunks, 16x16x32 = 16 rows x 4 chunks
    const int l31 = (MI == 32) ? (lane & 31) : (lane & 15), hi = (MI == 32) ? (lane >> 5) : (lane >> 4);
    const int sw = swz(l31);                               // row swizzle (tile bases are multiples of 16)
    const int w_row_off = (wn * TN * MI + l31) * ROWB;
    const int x_row_off = (wm * TM * MI + l31) * ROWB;

    const int nk = p.K / BK;
    unsigned long long t_loop = 0;
    if constexpr (LAB::stamps) t_loop = __builtin_readcyclecounter();

    {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nk) load_stage(s, s);

        for (int kt = 0; kt < nk; ++kt) {
            const int rem = min(STAGES - 2, nk - 1 - kt);
            // tile kt must have landed; up to `rem` younger tiles may stay in flight (loads return in order)
            if constexpr (STAGES == 2) {
                wait_vm<0>();
            } else {
                static_assert(STAGES == 3, "ring depth");
                if (rem >= 1) wait_vm<LPT>(); else wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
            const bool do_load = !LAB::no_dma && (kt + STAGES - 1 < nk);
            const int lstage = (kt + STAGES - 1) % STAGES, lkt = kt + STAGES - 1;
            if constexpr (SCHED == SCHED_RING) { if (do_load) load_stage(lstage, lkt); }

            const char* sW = smem + (kt % STAGES) * STAGE_BYTES;
            const char* sX = sW + BN * ROWB;
            if constexpr (MI == 16) {
                // k32 steps; W fragments are refilled in place right after their last MFMA of the step, X fragments
                // are double-buffered (register budget: 2 waves/SIMD = 256 VGPRs, 128 of them accumulators)
                constexpr int NS16 = BK / 32;
                u32x4 wf[TN], xf[2][TM];
                auto rdW = [&](int a, int s) { return *(const u32x4*)(sW + w_row_off + a * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
                auto rdX = [&](int b, int s) { return *(const u32x4*)(sX + x_row_off + b * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
#pragma unroll
                for (int b = 0; b < TM; ++b) xf[0][b] = rdX(b, 0);
#pragma unroll
                for (int a = 0; a < TN; ++a) wf[a] = rdW(a, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (do_load) load_stage(lstage, lkt);
#pragma unroll
                for (int s = 0; s < NS16; ++s) {
#pragma unroll
                    for (int a = 0; a < TN; ++a) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int b = 0; b < TM; ++b) acc[a][b] = T::mfma16(wf[a], xf[s & 1][b], acc[a][b]);
                        if (s + 1 < NS16) {
                            wf[a] = rdW(a, s + 1);
                            if (a < TM) xf[(s + 1) & 1][a] = rdX(a, s + 1);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // fragments of k16-step s+1 are read before the MFMAs of step s (software pipeline in regs)
                u32x4 wf[2][TN], xf[2][TM];
                auto read_frags = [&](int buf, int s) {
                    const int coff = ((2 * s + hi) ^ sw) << 4;
#pragma unroll
                    for (int a = 0; a < TN; ++a) wf[buf][a] = *(const u32x4*)(sW + w_row_off + a * 32 * ROWB + coff);
#pragma unroll
                    for (int b = 0; b < TM; ++b) xf[buf][b] = *(const u32x4*)(sX + x_row_off + b * 32 * ROWB + coff);
                };
                read_frags(0, 0);
#pragma unroll
                for (int s = 0; s < BK / 16; ++s) {
                    if (s < BK / 16 - 1) read_frags((s + 1) & 1, s + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < TN; ++a)
#pragma unroll
                        for (int b = 0; b < TM; ++b) acc[a][b] = T::mfma32(wf[s & 1][a], xf[s & 1][b], acc[a][b]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ---- epilogue -----------------------------------------------------------------------------
    // Row-major outputs are staged through LDS (the stage ring is free now) and leave as whole-row
    // 16-byte stores: per-lane 8-byte stores straight from the MFMA layout touch 32 rows per
    // instruction and were measured to cost ~20 K-iterations per 256x256 tile.  Transposed /
    // scattered destinations (the Vt images) still go straight from registers.
    u16* Yb = p.Y + bz * p.bsY;
    const u16* Rb = p.R ? p.R + bz * p.bsR : nullptr;
    // (copy-out activations belong to MODE_PLAIN only: in MODE_GEGLU p.act selects the gate function and must not run twice)
    const bool act_tanh = (MODE == MODE_PLAIN) && (p.act == ACT_GELU_TANH), act_erf = (MODE == MODE_PLAIN) && (p.act == ACT_GELU_ERF);
    const bool glu_silu = (p.act == ACT_SILU);                     // MODE_GEGLU: SiLU-GLU (Mistral) instead of GELU(tanh)-GLU (Gemma2)
    constexpr int BNO = (MODE == MODE_GEGLU) ? BN / 2 : BN;       // output columns of this tile
    constexpr int CROW = BNO * 2 + 16;                            // padded LDS row (bytes)
    __syncthreads();                                              // every wave is done with the last K slice
    unsigned long long t_epi = 0;
    if constexpr (LAB::stamps) t_epi = __builtin_readcyclecounter();
    if constexpr (LAB::no_epilogue) {
        // keep the accumulators live without storing them
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) asm volatile("" ::"v"(acc[a][b]));
        return;
    }
    char* sC = smem;
    // copy-out geometry (chunk i = it*NT + tid -> row i / CH, 16-byte chunk i % CH of the staged tile)
    constexpr int CH = BNO / 8, ITERS = BM * CH / NT, UNR = 4;
    static_assert((BM * CH) % NT == 0 && ITERS % UNR == 0, "copy-out geometry");
    const int no0 = (MODE == MODE_GEGLU) ? (n0 >> 1) : n0;
    const int Nout = (MODE == MODE_GEGLU) ? (p.N >> 1) : p.N;
    // the residual tile is fetched NOW, so its HBM latency hides under the staging pass below instead of
    // stalling the copy-out (the accumulators are still live)
    // (with 4 waves the tile is 32 chunks per thread: 128 registers on top of 256 accumulators would spill, so that
    //  geometry fetches the residual chunks of each copy-out pass at the top of the pass instead)
    constexpr bool RES_EARLY = (ITERS <= 16);
    u32x4 res[RES_EARLY ? ITERS : UNR];
    const bool wrap = p.rmod < p.M;                               // residual rows repeat every rmod rows (position tables)
    auto load_res = [&](int it) {
        const int i = it * NT + tid;
        const int m = min(m0 + i / CH, p.M - 1), n = min(no0 + (i % CH) * 8, Nout - 8);
        const int mr = wrap ? m % p.rmod : m;
        return *(const u32x4*)(Rb + (size_t)mr * p.ldr + n);
    };
    if constexpr (MODE != MODE_KV_CACHE && RES_EARLY) {
        if (Rb) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) res[it] = load_res(it);
        }
    }
    constexpr int NQ = (MI == 32) ? 4 : 1;                        // 4-row quads per lane per tile (32x32: rows 8j+4hi+e; 16x16: 4hi+e)
    // this lane's bias quads (one 8-byte load per n-quad, shared by all its m-tiles)
    const bool has_bias = (MODE != MODE_GEGLU) && (p.bias != nullptr);
    u32x2 bq[TN][NQ];
    if constexpr (MODE != MODE_GEGLU) {
        if (has_bias) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int n = n0 + wn * TN * MI + a * MI + 8 * j + 4 * hi;
                    bq[a][j] = *(const u32x2*)(p.bias + min(n, p.N - 4));
                }
        }
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int ml = wm * TM * MI + b * MI + l31;               // row inside the tile
        const int m = m0 + ml;
        if constexpr (MODE == MODE_GEGLU) {
            // W rows are interleaved in blocks of 32 gate / 32 up rows: pair every gate quad with its up quad
            constexpr int TPB = 32 / MI;                          // MI-row tiles per 32-row block
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                if ((a / TPB) % 2) continue;                      // up tiles are consumed with their gate tile
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int nol = ((wn * TN * MI) >> 1) + (a / (2 * TPB)) * 32 + (a % TPB) * MI + 8 * j + 4 * hi;   // output column inside the tile
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = rnd<T>(acc[a][b][4 * j + e]);
                        const float u = rnd<T>(acc[a + TPB][b][4 * j + e]);
                        v[e] = rnd<T>(glu_silu ? silu_f(g) : gelu_tanh_f(g)) * u;
                    }
                    const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                    *(u32x2*)(sC + ml * CROW + nol * 2) = o;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int nl = wn * TN * MI + a * MI + 8 * j + 4 * hi;
                    const int n = n0 + nl;
                    if (n >= p.N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * j + e];
                    // folded LayerNorm (see GemmParams): replaces the bias add.  Only the small-problem 128x128 tile carries it (large
                    // problems run on the persistent kernel; the 8-wave 256x256 body has no registers to spare — it spilled with it)
                    if (BN == 128 && BM == 128 && p.ln_stats) {
                        const size_t mr = (size_t)min(m, p.M - 1);
                        const float mu = p.ln_stats[2 * mr], rs = p.ln_stats[2 * mr + 1];
                        const f32x4 sv = *(const f32x4*)(p.ln_s + min(n, p.N - 4)), cv = *(const f32x4*)(p.ln_c + min(n, p.N - 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rs, v[e] - mu * sv[e], cv[e]);
                    } else if (has_bias) {
                        const u32x2 bv = bq[a][j];
                        v[0] += T::to_f32((u16)(bv[0] & 0xffff)); v[1] += T::to_f32((u16)(bv[0] >> 16));
                        v[2] += T::to_f32((u16)(bv[1] & 0xffff)); v[3] += T::to_f32((u16)(bv[1] >> 16));
                    }
                    // (the activation runs in the copy-out pass on the T-rounded staged values — same arithmetic,
                    //  but one small loop body instead of TN*TM*NQ unrolled copies that overflowed the I-cache)
                    const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                    bool staged = true;
                    if constexpr (MODE == MODE_QKV_VT) {
                        if (n >= p.vstart) {
                            staged = false;
                            if (m < p.M) {
                                const int c = n - p.vstart, h = c / p.hd, d = c % p.hd;
                                const int bi = m / p.seq, tok = m % p.seq;
                                const int pos = (tok & ~15) | perm16(tok & 15);
                                u16* dst = p.Vt + (((size_t)bi * p.nheads + h) * p.hd + d) * p.seqpad + pos;
                                dst[0] = (u16)(o[0] & 0xffff); dst[(size_t)p.seqpad] = (u16)(o[0] >> 16);
                                dst[2 * (size_t)p.seqpad] = (u16)(o[1] & 0xffff); dst[3 * (size_t)p.seqpad] = (u16)(o[1] >> 16);
                            }
                        }
                    } else if constexpr (MODE == MODE_KV_CACHE) {
                        if (n >= p.kvd && m < p.M) {               // V: also the transposed, perm16 tile image
                            // Vtc[kvh][tile32][hd][32 positions (perm16)]: a 32-key sub-tile is 64-byte rows back to back, so the
                            // cross-attention fetches whole 128-byte lines (a [hd][64] tile made every sub-tile fetch half-lines
                            // and the other halves were evicted before the next sub-tile came: 1.5x the HBM bytes, measured)
                            const int tok = p.tok0 + m, tile = tok >> 5, tk = tok & 31;
                            const int c = n - p.kvd, kvh = c / p.hd, d = c % p.hd;
                            const int pos = (tk & ~15) | perm16(tk & 15);
                            u16* dst = p.Vtc + (((size_t)kvh * p.ntile64 * 2 + tile) * p.hd + d) * 32 + pos;
                            dst[0] = (u16)(o[0] & 0xffff); dst[32] = (u16)(o[0] >> 16);
                            dst[64] = (u16)(o[1] & 0xffff); dst[96] = (u16)(o[1] >> 16);
                        }
                    }
                    if (staged) *(u32x2*)(sC + ml * CROW + nl * 2) = o;
                }
            }
        }
    }
    __syncthreads();
    unsigned long long t_copy = 0;
    if constexpr (LAB::stamps) t_copy = __builtin_readcyclecounter();
    // ---- copy-out: 16-byte chunks, consecutive lanes = consecutive chunks of one row ------------
    // UNR chunks per pass: their LDS reads are issued together, then processed
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += UNR) {
        u32x4 val[UNR];
        int mm[UNR], nn[UNR];
        bool ok[UNR];
        if constexpr (MODE != MODE_KV_CACHE && !RES_EARLY) {
            if (Rb) {
#pragma unroll
                for (int k = 0; k < UNR; ++k) res[k] = load_res(it0 + k);
            }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int i = (it0 + k) * NT + tid;
            const int ml = i / CH, c = i % CH;
            mm[k] = m0 + ml; nn[k] = no0 + c * 8;
            ok[k] = (mm[k] < p.M) && (nn[k] < Nout);
            if constexpr (MODE == MODE_QKV_VT) ok[k] = ok[k] && (nn[k] < p.vstart);
            if constexpr (LAB::no_store) ok[k] = ok[k] && (p.M < 0);
            val[k] = *(const u32x4*)(sC + ml * CROW + c * 16);
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (!ok[k]) continue;
            const int m = mm[k], n = nn[k];
            if constexpr (MODE == MODE_KV_CACHE) {
                if (n < p.kvd) {
                    const int tok = p.tok0 + m, kvh = n / p.hd, d = n % p.hd;
                    *(u32x4*)(p.Kc + ((size_t)kvh * p.ntile64 * 64 + tok) * p.hd + d) = val[k];
                } else {
                    *(u32x4*)(p.Vrow + (size_t)m * p.kvd + (n - p.kvd)) = val[k];
                }
            } else {
                u32x4 v = val[k];
                if (act_tanh) {
                    float x[8];
                    unpack8<T>(v, x);
#pragma unroll
                    for (int e = 0; e < 8; e += 2) { const f32x2_t y = gelu_tanh_2(f32x2_t{x[e], x[e + 1]}); x[e] = y[0]; x[e + 1] = y[1]; }
                    v = pack8<T>(x);
                } else if (act_erf) {
                    float x[8];
                    unpack8<T>(v, x);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = gelu_erf_f(x[e]);
                    v = pack8<T>(x);
                }
                if (Rb) {
                    float x[8], r[8];
                    unpack8<T>(v, x);
                    unpack8<T>(res[RES_EARLY ? it0 + k : k], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += r[e];          // x is already T-rounded; the sum rounds on pack
                    v = pack8<T>(x);
                }
                if (p.hm_seq) {                                  // head-major output (GemmParams::hm_*)
                    const int hdim = p.hm_heads * p.hm_hd, which = n / hdim, nh = n - which * hdim, hh = nh / p.hm_hd;
                    const int fr = m / p.hm_seq, tok = m - fr * p.hm_seq;
                    *(u32x4*)(Yb + ((((size_t)which * (p.M / p.hm_seq) + fr) * p.hm_heads + hh) * p.hm_seq + tok) * p.hm_hd + (nh - hh * p.hm_hd)) = v;
                } else {
                    *(u32x4*)(Yb + (size_t)m * p.ldy + n) = v;
                }
            }
        }
    }
    if constexpr (LAB::stamps) {
        if (p.dbg && tid == 0 && blockIdx.x < 1024) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t_end = __builtin_readcyclecounter();
            unsigned long long* d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = t_start; d[1] = t_loop; d[2] = t_epi; d[3] = t_copy; d[4] = t_end;
        }
    }
}
