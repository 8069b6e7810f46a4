// Weight-streaming projection of a FEW rows on the matrix pipe (2 <= M <= 32): the decode step of a BATCH of queries sharing one video
// (BASELINE configs[4]: 8 rows; their o_proj over the three attention streams: 24 rows) — the decoder's nn.Linear calls at q_len = 1
// (gemma.py:57-60, :94, Gemma2MLP :119 via HF gemma2; mistral.py:131-137) with a batch dimension.
// Why not gemv.hip: its FMAs are VALU work — per 16-byte weight chunk and lane 8 unpacks + 8 x M FMAs (+ the X unpacks); at M = 8 that is
// ~100 VALU operations per 16 B, i.e. the whole VALU rate of the chip at 6 TB/s: the 8-row step ran at 16.0 ms against an 8.9 ms floor
// (round-4 verdict, item 5).  Why not gemm_skinny.h: split-K through an fp32 workspace + a reduce launch, sized for 39..128 rows.
// Here (HBM-bound, algorithmic bytes = N K 2, one pass over W, no workspace):
//   * a block owns 16 output features (GLU: 16 gate rows + their 16 up rows; wide outputs: 32 features = two 16-row groups against one X
//     fragment) and ALL of K; its KS waves (1 for most shapes: see the launch-shape comment below) take the 64-wide k steps round-robin and
//     their partial accumulators are added in LDS once per feature group;
//   * W goes HBM -> registers, 32 contiguous bytes per lane (lane (row l15, hi): k = 64 s + 16 hi .. + 15), D - 1 steps ahead in a
//     register ring (default cache policy: the non-temporal hint measured 9-19 % slower here); the X rows (<= 32, L2-resident) are loaded
//     the same way by the lanes of real rows — lane (m = l15, hi) takes the same k bytes of its row, which IS the B fragment of MFMA
//     16x16x32 under the k permutation the W fragment uses ({16 hi + 0..7} then {16 hi + 8..15});
//   * the ring runs over the flattened (feature group, step) sequence of a block (grid-stride over the groups), so the pipe never
//     drains between groups; loads are asm with counted vmcnt waits (ordinary loads get a vmcnt(0) at the loop header: gemm_skinny.h).
//   * C[i = 4 hi + r][j = l15] = feature n0 + 4 hi + r, X row 16 g + l15.  Output rounding: T(sum); GLU: T( T(act(T(g))) * T(u) ) —
//     the values of gemv.hip's kernels up to the fp32 summation order.
#include "kernels.h"
#include "gemv_mfma_api.h"
#include <stdlib.h>
#include <utility>

template <int V> struct GmInt { static constexpr int value = V; };
template <int... I, typename F>
__device__ __forceinline__ void gm_unroll(std::integer_sequence<int, I...>, F&& f) { (f(GmInt<I>{}), ...); }

// lab knobs (tools/build_variant.sh): weight loads without the non-temporal hint; no X loads at all (wrong results: what the X path costs);
// ring depth of the plain one-group kernel
#ifndef VIDI_GEMVM_NT
#define VIDI_GEMVM_NT 0         // measured (profiles/r5_gemvm_variants.jsonl): the hint costs 9-19 % on every decode shape in THIS kernel
#endif
#ifndef VIDI_GEMVM_D1
#define VIDI_GEMVM_D1 8
#endif
#ifndef VIDI_GEMVM_DG
#define VIDI_GEMVM_DG 6
#endif
#ifndef VIDI_GEMVM_D2
#define VIDI_GEMVM_D2 6
#endif

__device__ __forceinline__ void gm_load_nt(u32x4& a, u32x4& b, const u16* p) {
#if VIDI_GEMVM_NT
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(a) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16 nt" : "=v"(b) : "v"(p) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(b) : "v"(p) : "memory");
#endif
}
__device__ __forceinline__ void gm_load(u32x4& a, u32x4& b, const u16* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(b) : "v"(p) : "memory");
}
// the same loads for a subset of the lanes (the caller's branch): "+v" — the inactive lanes keep what the registers held
__device__ __forceinline__ void gm_load_keep(u32x4& a, u32x4& b, const u16* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(a) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "+v"(b) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void gm_wait_vm(u32x4& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
__device__ __forceinline__ void gm_pin(u32x4& a) { asm volatile("" : "+v"(a)); }            // keeps a consumer behind the wait above

struct GemvmParams {
    const u16* X; const u16* W; u16* Y;
    int M, N, K, ldx, ldw, ldy, silu;
};

// MODE 0: 16 features per block; 1: gated pair (16 gate + 16 up rows, act(g) * u); 2: 32 features per block (two 16-row groups against the
// same X fragments: the X loads per weight byte halve)
template <typename T, int MT, int KS, int D, int MODE>
__global__ __launch_bounds__(KS * 64) void gemvm_kernel(GemvmParams p) {
    constexpr bool GLU = MODE == 1;
    constexpr int NW = MODE ? 2 : 1;                     // 16-row weight tiles per step
    constexpr int FPG = MODE == 2 ? 32 : 16;             // output features per group
#ifdef VIDI_GEMVM_NOX
    constexpr int OPS = 2 * NW;
#else
    constexpr int OPS = 2 * NW + 2 * MT;                 // memory operations per step and wave (a compile-time constant: counted waits)
#endif
    static_assert((D - 2) * OPS <= 63, "vmcnt is a 6-bit counter");
    __shared__ f32x4 red[2][KS][NW * MT][64];            // partial accumulators, double-buffered by group parity
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, hi = lane >> 4;
    const int spw = p.K / 64 / KS;                       // steps per wave and group
    const int ngroups = p.N / FPG;
    const int mine = blockIdx.x < ngroups ? (ngroups - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int T_steps = mine * spw;

    const u16* xp[MT];
#pragma unroll
    for (int g = 0; g < MT; ++g) xp[g] = p.X + (size_t)min(g * 16 + l15, p.M - 1) * p.ldx + wave * 64 + hi * 16;
    const size_t wlane = (size_t)wave * 64 + hi * 16;
    // only the lanes of real X rows load (l15 < rows of the 16-row group): at M = 8 half of the X requests — which cost 10-15 % of the
    // kernel at full width (profiles/r5_gemvm_variants*.jsonl, the no-X arm) — disappear.  The other lanes keep the zeros they start with
    // (the loads are "+v": their columns of C are never stored, and zeros keep them finite)
    bool xlive[MT];
#pragma unroll
    for (int g = 0; g < MT; ++g) xlive[g] = g * 16 + l15 < p.M;

    u32x4 wr[D][NW][2], xr[D][MT][2];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int g = 0; g < MT; ++g) { xr[d][g][0] = u32x4{0, 0, 0, 0}; xr[d][g][1] = u32x4{0, 0, 0, 0}; }
    f32x4 acc[NW][MT];
#pragma unroll
    for (int a = 0; a < NW; ++a)
#pragma unroll
        for (int g = 0; g < MT; ++g) acc[a][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    int ig = blockIdx.x, is = 0;                         // (group, step) the next issue fetches
    auto issue = [&](auto slot_t) {
        constexpr int slot = decltype(slot_t)::value;
        const int g = min(ig, ngroups - 1);              // past the end: re-load the last group (never consumed; keeps the counts exact)
        const int f = g * FPG + l15;
        const size_t k = (size_t)is * KS * 64;
        if constexpr (MODE == 2) {
            const u16* w0 = p.W + (size_t)f * p.ldw + wlane + k;
            gm_load_nt(wr[slot][0][0], wr[slot][0][1], w0);
            gm_load_nt(wr[slot][1][0], wr[slot][1][1], w0 + (size_t)16 * p.ldw);
        } else if constexpr (GLU) {
            const u16* wg = p.W + ((size_t)(f >> 5) * 64 + (f & 31)) * p.ldw + wlane + k;
            gm_load_nt(wr[slot][0][0], wr[slot][0][1], wg);
            gm_load_nt(wr[slot][1][0], wr[slot][1][1], wg + (size_t)32 * p.ldw);
        } else {
            gm_load_nt(wr[slot][0][0], wr[slot][0][1], p.W + (size_t)f * p.ldw + wlane + k);
        }
#pragma unroll
#ifdef VIDI_GEMVM_NOX
        for (int m = 0; m < MT; ++m) { xr[slot][m][0] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; xr[slot][m][1] = xr[slot][m][0]; }
#else
        for (int m = 0; m < MT; ++m)
            if (xlive[m]) gm_load_keep(xr[slot][m][0], xr[slot][m][1], xp[m] + k);
#endif
        if (++is == spw) { is = 0; ig += gridDim.x; }
    };
    auto land = [&](auto slot_t) {                       // the slot's step landed; the D - 2 younger steps may be in flight
        constexpr int slot = decltype(slot_t)::value;
        gm_wait_vm<(D - 2) * OPS>(wr[slot][0][0]);
        gm_pin(wr[slot][0][1]);
        if constexpr (NW == 2) { gm_pin(wr[slot][1][0]); gm_pin(wr[slot][1][1]); }
#pragma unroll
        for (int m = 0; m < MT; ++m) { gm_pin(xr[slot][m][0]); gm_pin(xr[slot][m][1]); }
    };
    auto compute = [&](auto slot_t) {
        constexpr int slot = decltype(slot_t)::value;
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[a][m] = T::mfma16(wr[slot][a][0], xr[slot][m][0], acc[a][m]);
                acc[a][m] = T::mfma16(wr[slot][a][1], xr[slot][m][1], acc[a][m]);
            }
    };
    // end of a feature group: the KS waves' partial tiles are added in LDS and the 16 x (16 MT) results stored.  Raw s_barrier
    // (__syncthreads() carries a fence for which the compiler drains vmcnt, i.e. the whole ring); one barrier per group: the buffer
    // alternates, and a wave cannot run two groups ahead of another without passing the barrier in between.
    int cs = 0, cg = blockIdx.x, par = 0;
    auto finish_group = [&]() {
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                red[par][wave][a * MT + m][lane] = acc[a][m];
                acc[a][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* rf = (const float*)&red[par][0][0][0];
        constexpr int WSTR = NW * MT * 256;              // floats between two waves' tiles
        for (int idx = tid; idx < MT * 256; idx += KS * 64) {
            const int m16 = idx >> 8, L = (idx >> 2) & 63, r = idx & 3;
            const int n = cg * FPG + 4 * (L >> 4) + r, m = m16 * 16 + (L & 15);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) {
                s0 += rf[w * WSTR + m16 * 256 + (idx & 255)];
                if constexpr (NW == 2) s1 += rf[w * WSTR + (MT + m16) * 256 + (idx & 255)];
            }
            if (m < p.M) {
                if constexpr (GLU) {
                    const float g = rnd<T>(s0), u = rnd<T>(s1);
                    p.Y[(size_t)m * p.ldy + n] = T::from_f32(rnd<T>(p.silu ? silu_f(g) : gelu_tanh_f(g)) * u);
                } else {
                    p.Y[(size_t)m * p.ldy + n] = T::from_f32(s0);
                    if constexpr (MODE == 2) p.Y[(size_t)m * p.ldy + n + 16] = T::from_f32(s1);
                }
            }
        }
        par ^= 1;
        cg += gridDim.x;
    };
    auto step = [&](auto slot_t, auto prev_t, int t) {
        if (t < T_steps) {
            land(slot_t);
            issue(prev_t);                               // refill the slot consumed one step ago with step t + D - 1
            compute(slot_t);
            if (++cs == spw) { cs = 0; finish_group(); }
        }
    };
    // prologue: steps 0 .. D - 2; then D steps per trip, slot = step % D
    if (T_steps > 0) gm_unroll(std::make_integer_sequence<int, D - 1>{}, [&](auto i_t) { issue(i_t); });
    for (int t0 = 0; t0 < T_steps; t0 += D)
        gm_unroll(std::make_integer_sequence<int, D>{}, [&](auto i_t) {
            constexpr int I = decltype(i_t)::value;
            step(i_t, GmInt<(I + D - 1) % D>{}, t0 + I);
        });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the ring's tail (clamped re-loads) must not outlive the wave's registers
}

// Launch shape.  The kernel is a latency-bound stream: what matters is that every wave lives long (many 2 KB steps behind one prologue)
// and that ~1 000-2 000 waves are resident (4-8 per CU x 10-20 KB in flight each covers HBM's latency-bandwidth product several times).
// Round-5 session 1 measured the first form of this launch (8 waves per block, 7 steps per wave, one feature group per block) at the VALU
// GEMV's rate: all prologue, no steady state.  Now: the K split is the SMALLEST of 1 / 2 / 4 / 8 that puts >= TARGET waves on the chip
// (a feature group of 16 rows is one block; fewer waves per block = more steps per wave, fewer partial tiles to add), and at most
// RESIDENT waves are launched — blocks walk the remaining groups with the ring running across group boundaries.
#ifndef VIDI_GEMVM_TARGET_WAVES
#define VIDI_GEMVM_TARGET_WAVES 200          // session-2 sweep: one wave per 16-row group (no K split) was the best form of q|k|v, o and down
#endif
#ifndef VIDI_GEMVM_TARGET_WAVES_GLU
#define VIDI_GEMVM_TARGET_WAVES_GLU 2048     // ... and four waves per group of the gated pair (896 groups)
#endif
#ifndef VIDI_GEMVM_FG2
#define VIDI_GEMVM_FG2 256                   // plain projections of <= 16 rows take 32 features per block (MODE 2) when that leaves at least this many
                                             // groups (0: never).  Session 3c: q|k|v 13.9 -> 12.9 us, lm_head 346 -> 313 us (5.9 TB/s); o and down (112
                                             // groups of 32) lose a third — they keep 16-feature groups
#endif
#ifndef VIDI_GEMVM_RESIDENT_WAVES
#define VIDI_GEMVM_RESIDENT_WAVES 2048
#endif
static inline int gemvm_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static inline int gemvm_ks(int groups, int K, int target) {
    static const int forced = gemvm_env("VIDI_GEMVM_KS", 0);
    const int steps = K / 64;
    if (forced && steps % forced == 0 && (forced == 1 || forced == 2 || forced == 4 || forced == 8)) return forced;
    int best = 1;
    for (int ks : {1, 2, 4, 8}) {
        if (steps % ks) break;
        best = ks;
        if (groups * ks >= target) break;
    }
    return best;
}

// the EXACT shape predicate of vidi_gemv_mfma_dispatch (callers treat "fits" as "will run": engine.proj / proj_glu pick their kernel with it):
// the gated pair walks 32-row blocks of the interleaved gate / up weight, so its N (= I) must be a multiple of 32, plain projections of 16
int vidi_gemvm_fits(int M, int N, int K, int glu) {
    if (M < 1 || M > (glu ? 16 : 32) || N <= 0 || N % (glu ? 32 : 16) || K <= 0 || K % 64) return 0;
    return 1;
}

template <typename T>
static int launch_gemvm(const GemvmParams& p, bool glu, hipStream_t st) {
    static const int resident = gemvm_env("VIDI_GEMVM_RESIDENT_WAVES", VIDI_GEMVM_RESIDENT_WAVES);
    static const int target = gemvm_env("VIDI_GEMVM_TARGET_WAVES", VIDI_GEMVM_TARGET_WAVES);
    static const int target_glu = gemvm_env("VIDI_GEMVM_TARGET_WAVES_GLU", VIDI_GEMVM_TARGET_WAVES_GLU);
    static const int fg2 = gemvm_env("VIDI_GEMVM_FG2", VIDI_GEMVM_FG2);
    const int mt = (p.M + 15) / 16;
    const int mode = glu ? 1 : ((fg2 > 0 && mt == 1 && p.N % 32 == 0 && p.N / 32 >= fg2) ? 2 : 0);
    const int ngroups = p.N / (mode == 2 ? 32 : 16);
    const int ks = gemvm_ks(ngroups, p.K, glu ? target_glu : target);
    const int cap = resident / ks > 0 ? resident / ks : 1;
    const int blocks = ngroups < cap ? ngroups : cap;
#define VIDI_GM(MT_, KS_, D_, MODE_) hipLaunchKernelGGL((gemvm_kernel<T, MT_, KS_, D_, MODE_>), dim3(blocks), dim3(KS_ * 64), 0, st, p)
#define VIDI_GM_KS(MT_, D_, MODE_)                                                   \
    do {                                                                             \
        if (ks == 8) VIDI_GM(MT_, 8, D_, MODE_);                                     \
        else if (ks == 4) VIDI_GM(MT_, 4, D_, MODE_);                                \
        else if (ks == 2) VIDI_GM(MT_, 2, D_, MODE_);                                \
        else VIDI_GM(MT_, 1, D_, MODE_);                                             \
    } while (0)
    if (mode == 1) VIDI_GM_KS(1, VIDI_GEMVM_DG, 1);          // 6 operations per step
    else if (mode == 2) VIDI_GM_KS(1, VIDI_GEMVM_DG, 2);     // 6 operations per step
    else if (mt == 1 && p.K >= 8192) VIDI_GM_KS(1, 12, 0);   // long rows (down_proj): 11 steps in flight (27.1 -> 24.9 us; shorter rows lose with it)
    else if (mt == 1) VIDI_GM_KS(1, VIDI_GEMVM_D1, 0);       // 4 operations per step: 7 steps = 14 KB of W per wave in flight
    else VIDI_GM_KS(2, VIDI_GEMVM_D2, 0);                    // 6 operations per step
#undef VIDI_GM_KS
#undef VIDI_GM
    return (int)hipGetLastError();
}

int vidi_gemv_mfma_dispatch(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int glu_act, int dtype,
                            hipStream_t st) {
    const bool glu = glu_act >= 0;
    if (!vidi_gemvm_fits(M, N, K, glu) || ldw % 8 || ldx % 8) return VIDI_ERR_SHAPE;
    if (glu && glu_act != ACT_GELU_TANH && glu_act != ACT_SILU) return VIDI_ERR_ARG;
    if (((uintptr_t)W & 15) || ((uintptr_t)X & 15)) return VIDI_ERR_ALIGN;
    GemvmParams p{(const u16*)X, (const u16*)W, (u16*)Y, M, N, K, ldx, ldw, ldy, glu_act == ACT_SILU};
    if (dtype == VIDI_DT_BF16) return launch_gemvm<BF16>(p, glu, st);
    if (dtype == VIDI_DT_F16) return launch_gemvm<F16>(p, glu, st);
    return VIDI_ERR_DTYPE;
}
