// Weight-streaming GEMM for 8 < M <= 128 rows: the text prompt (M = 39 rows at prefill) against the decoder's weights — the reference's
// q / k / v / o / down projections of the text stream (HF Gemma2Attention / Gemma2MLP through gemma.py:165-175, 116-123) at Lq > 1.
// The 128 x 128 tile kernel puts N / 128 = 28–32 blocks on 256 CUs for the o / down projections (0.8 TB/s of weight streaming, 15 ms of
// the prefill against a 2.8 ms floor: profiles/r4_notes.md); the M <= 8 GEMV (gemv.hip) does not use the matrix pipe.  Here:
//   * a block owns 64 weight rows (4 waves x 16) and ONE K slice (split-K over blockIdx.y so that every shape launches ~450 blocks);
//   * W goes HBM -> registers, 32 contiguous bytes per lane (lane (row, hi) takes k = 64 s + 16 hi .. + 15 of its row: the four hi-lanes of
//     a row read one whole 128-byte line), D = 4 steps ahead;
//   * the X rows of a step (<= 128 rows x 64 k) go L2 -> LDS once per block by LDS-DMA (chunk-swizzled), also 4 steps ahead in a ring;
//   * MFMA 16x16x32 with the contraction permuted the same way on both operands (k subset {16 hi + 0..7} then {16 hi + 8..15});
//   * fp32 partials part[ks][m][n] in a caller-owned workspace; skinny_reduce_kernel adds the slices (+ bias) and rounds once.
// Measured (tools/lab/skinny_lab.hip, profiles/r4_gemm_lab_skinny.jsonl): 2.2–2.6 TB/s on the 29 MB q / kv / o weights (34–39 -> 13 us),
// 4.0 TB/s on down (128 -> 26 us), results equal to the tile kernel's to the output rounding.
#pragma once
#include "common.h"

template <int V> struct SkInt { static constexpr int value = V; };

// asm memory operations (a generic lambda cannot name a captured variable in an asm operand: plain functions taking references)
__device__ __forceinline__ void sk_load_b128x2(u32x4& a, u32x4& b, const u16* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a) : "v"(p) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(b) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void sk_wait_vm(u32x4& a, u32x4& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ void sk_lds_read(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int MT>
__device__ __forceinline__ void sk_wait_lgkm(u32x4 (&b)[MT][2]) {
    if constexpr (MT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]));
    else if constexpr (MT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
    else if constexpr (MT == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1]));
}

template <typename T, int MT>
__global__ __launch_bounds__(256) void skinny_kernel(const u16* __restrict__ X, const u16* __restrict__ W, float* __restrict__ part,
                                                     int M, int N, int K, int ldx, int ldw, int ksteps) {
    constexpr int D = 4;                                    // prefetch ring depth (steps of 64 k)
    constexpr int PX = (MT * 16 + 31) / 32;                 // LDS-DMA instructions per wave and step: the block stages 32 rows of 128 B per pass
    constexpr int SLAB = PX * 32 * 128;                     // bytes per step.  Every wave issues all PX pieces (rows past M re-read row M - 1 and are
                                                            // never used): the number of memory operations per step is then a compile-time constant,
                                                            // which both the counted wait below and the compiler's own waits for the W registers
                                                            // need — behind a branch the compiler falls back to vmcnt(0), i.e. drains the ring
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [D][PX * 32][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, hi = lane >> 4;
    const int n_row = blockIdx.x * 64 + wave * 16 + l15;
    const int ks = blockIdx.y;
    const size_t k_base = (size_t)ks * ksteps * 64;
    const u16* wp = W + (size_t)n_row * ldw + k_base + hi * 16;
    // X staging: lane -> (row r = 8 wave + (lane >> 3) [+ 32 per pass], chunk c = lane & 7); the chunk is fetched from position c ^ (r & 7)
    // so that the fragment reads below (16 rows x the same chunk) spread over the banks
    const int xr = wave * 8 + (lane >> 3), xc = lane & 7;

    u32x4 wreg[D][2];
    f32x4 acc[MT];
#pragma unroll
    for (int g = 0; g < MT; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int s, auto slot_t) {                  // step s (clamped: the ring's tail re-loads the last step, never consumed)
        constexpr int slot = decltype(slot_t)::value;
        const int sc = s < ksteps ? s : ksteps - 1;
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int r = xr + 32 * p;
            const int rm = r < M ? r : M - 1;
            glds16(X + (size_t)rm * ldx + k_base + (size_t)sc * 64 + ((xc ^ (r & 7)) << 3), smem + slot * SLAB + (wave * 8 + 32 * p) * 128);
        }
        // the weight loads are asm as well: for ordinary loads the compiler adds its own wait in front of their first use and, at the
        // loop header, settles for vmcnt(0).  As asm their completion is covered by the counted wait (which names the registers)
        sk_load_b128x2(wreg[slot][0], wreg[slot][1], wp + (size_t)sc * 64);
    };
    // The X fragments are read with inline-asm LDS reads: through ordinary loads the compiler must assume they may alias the pending
    // LDS-DMA writes of the ring and puts vmcnt(0) in front of them (one full drain of the prefetch ring per 4 steps, seen in the ISA).
    // As asm they carry no memory operand; their completion is waited for explicitly and the values pass through the wait statement so
    // that their consumers stay behind it (the attn_self_rm.hip idiom).
    const __attribute__((address_space(3))) char* lds0 = (const __attribute__((address_space(3))) char*)smem;
    const unsigned xa0 = (unsigned)(uintptr_t)(lds0 + l15 * 128 + (((2 * hi) ^ (l15 & 7)) << 4));       // rows g*16 + l15: (r & 7) == (l15 & 7)
    const unsigned xa1 = (unsigned)(uintptr_t)(lds0 + l15 * 128 + (((2 * hi + 1) ^ (l15 & 7)) << 4));
    auto chunk = [&](auto slot_t, auto g0_t, auto gn_t) {   // groups g0 .. g0 + gn - 1 (<= 4 at a time: 32 fragment registers)
        constexpr int slot = decltype(slot_t)::value, G0 = decltype(g0_t)::value, GN = decltype(gn_t)::value;
        u32x4 b[GN][2];
        sk_lds_read<slot * SLAB + G0 * 2048>(b[0][0], xa0); sk_lds_read<slot * SLAB + G0 * 2048>(b[0][1], xa1);
        if constexpr (GN > 1) { sk_lds_read<slot * SLAB + (G0 + 1) * 2048>(b[1][0], xa0); sk_lds_read<slot * SLAB + (G0 + 1) * 2048>(b[1][1], xa1); }
        if constexpr (GN > 2) { sk_lds_read<slot * SLAB + (G0 + 2) * 2048>(b[2][0], xa0); sk_lds_read<slot * SLAB + (G0 + 2) * 2048>(b[2][1], xa1); }
        if constexpr (GN > 3) { sk_lds_read<slot * SLAB + (G0 + 3) * 2048>(b[3][0], xa0); sk_lds_read<slot * SLAB + (G0 + 3) * 2048>(b[3][1], xa1); }
        sk_wait_lgkm<GN>(b);
#pragma unroll
        for (int g = 0; g < GN; ++g) {
            acc[G0 + g] = T::mfma16(wreg[slot][0], b[g][0], acc[G0 + g]);
            acc[G0 + g] = T::mfma16(wreg[slot][1], b[g][1], acc[G0 + g]);
        }
    };
    auto compute = [&](auto slot_t) {
        chunk(slot_t, SkInt<0>{}, SkInt<(MT < 4 ? MT : 4)>{});
        if constexpr (MT > 4) chunk(slot_t, SkInt<4>{}, SkInt<MT - 4>{});
    };
    auto wait_step = [&](auto slot_t) {                      // step s (ring slot `slot`) landed; steps s + 1 .. s + D - 2 may be in flight
        constexpr int slot = decltype(slot_t)::value;
        sk_wait_vm<(D - 2) * (PX + 2)>(wreg[slot][0], wreg[slot][1]);
    };
    // raw s_barrier: __syncthreads() carries a release fence, for which the compiler drains vmcnt to 0 — i.e. waits for the whole prefetch
    // ring at every step.  The data dependence is covered by the counted wait above (this wave's pieces of step s landed) + the barrier
    // (every wave's did); the empty asm keeps the LDS reads below it.
    auto bar = [&]() { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
    issue(0, SkInt<0>{}); issue(1, SkInt<1>{}); issue(2, SkInt<2>{});
    for (int s0 = 0; s0 < ksteps; s0 += D) {
        wait_step(SkInt<0>{}); bar(); issue(s0 + 3, SkInt<3>{}); compute(SkInt<0>{});
        wait_step(SkInt<1>{}); bar(); issue(s0 + 4, SkInt<0>{}); compute(SkInt<1>{});
        wait_step(SkInt<2>{}); bar(); issue(s0 + 5, SkInt<1>{}); compute(SkInt<2>{});
        wait_step(SkInt<3>{}); bar(); issue(s0 + 6, SkInt<2>{}); compute(SkInt<3>{});
    }
    wait_vmcnt<0>();
    // C[i = 4 hi + r][j = l15]: weight row n = n0 + 4 hi + r, X row m = 16 g + l15
    const int n0 = blockIdx.x * 64 + wave * 16 + hi * 4;
#pragma unroll
    for (int g = 0; g < MT; ++g) {
        const int m = g * 16 + l15;
        if (m < M) *(f32x4*)(part + ((size_t)ks * M + m) * N + n0) = acc[g];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* __restrict__ part, const u16* __restrict__ bias, u16* __restrict__ Y,
                                                            int M, int N, int ldy, int ksplit) {
    const int i = blockIdx.x * 256 + threadIdx.x;           // one thread per 4 consecutive columns
    const int per_row = N / 4;
    if (i >= M * per_row) return;
    const int m = i / per_row, n = (i % per_row) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ksplit; ++k) s += *(const f32x4*)(part + ((size_t)k * M + m) * N + n);
    if (bias) for (int e = 0; e < 4; ++e) s[e] += T::to_f32(bias[n + e]);
    *(u32x2*)(Y + (size_t)m * ldy + n) = u32x2{pack2<T>(s[0], s[1]), pack2<T>(s[2], s[3])};
}

// slices so that ~448 blocks run and every slice is a multiple of 4 steps; 0: the shape does not fit (K % 256, N % 64)
static inline int skinny_ksplit(int N, int K) {
    if (N % 64 || K % 256) return 0;
    const int steps = K / 64, nb = N / 64;
    int best = 1;
    for (int s = 1; s <= 16; ++s)
        if (steps % (4 * s) == 0 && nb * s <= 512) best = s;
    return best;
}

