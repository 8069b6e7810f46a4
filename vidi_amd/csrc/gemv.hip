// Skinny projections of the decode step (M <= 8 rows): HBM-bound weight streaming, one pass over the weight.
// Replaces the decoder's nn.Linear calls at q_len = 1 (gemma.py:57-60, :94, Gemma2MLP :119 via TP gemma2; mistral.py:131-137).
#include "kernels.h"

// rows per wave and chunks in flight per row of the decode GEMVs (lab builds override them: tools/build_variant.sh)
#ifndef VIDI_GEMV_RPW
#define VIDI_GEMV_RPW 2
#endif
#ifndef VIDI_GEMV_UNR
#define VIDI_GEMV_UNR 4
#endif
#ifndef VIDI_GLU_RPW
#define VIDI_GLU_RPW 1
#endif
#ifndef VIDI_GLU_UNR
#define VIDI_GLU_UNR 4
#endif

// ---- skinny GEMM (M <= 8): HBM-bound weight streaming for decode -------------------------------
//   Y[m][n] = sum_k X[m][k] W[n][k].  Every wave owns RPW weight rows per step and streams them
//   with 16-byte non-temporal loads straight into VGPRs (no LDS round trip: each weight byte is
//   used once); the few activation rows are re-read from L1/L2.  Wave-reduce at the end.
//   Algorithmic bytes = N*K*2 (weights); roofline = HBM.
//   Bytes in flight decide the rate (round 2: RPW = 4 with one chunk per row outstanding = 4 KB per wave and 224-512 blocks ran the
//   decoder's projections at 2.2-4.9 TB/s): every lane requests UNR chunks of each of its RPW rows before the first FMA
//   (RPW * UNR * 1 KB per wave) and the grid gives every CU >= 2 blocks.  The per-lane accumulation order is unchanged (chunk
//   lane, lane + 64, ... in sequence), so the results are bit-identical to the one-chunk form.
template <typename T, int MMAX, int RPW, int UNR>
__global__ __launch_bounds__(256) void gemv_kernel(const u16* __restrict__ X, const u16* __restrict__ W, u16* __restrict__ Y,
                                                   int M, int N, int K, int ldx, int ldw, int ldy) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = K / 8;                          // 16-byte chunks per row
    const int waves_total = gridDim.x * 4;
    for (int nb = (blockIdx.x * 4 + wave) * RPW; nb < N; nb += waves_total * RPW) {
        float acc[RPW][MMAX];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MMAX; ++m) acc[r][m] = 0.f;
        const u16* wrow[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) wrow[r] = W + (size_t)min(nb + r, N - 1) * ldw;
        for (int c0 = lane; c0 < nchunk; c0 += 64 * UNR) {
            u32x4 wv[UNR][RPW], xv[UNR][MMAX];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = min(c0 + 64 * u, nchunk - 1);       // clamped: a chunk beyond the row is zeroed below
#pragma unroll
                for (int r = 0; r < RPW; ++r) wv[u][r] = __builtin_nontemporal_load((const u32x4*)(wrow[r] + c * 8));
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = min(c0 + 64 * u, nchunk - 1);
#pragma unroll
                for (int m = 0; m < MMAX; ++m) xv[u][m] = *(const u32x4*)(X + (size_t)min(m, M - 1) * ldx + c * 8);
            }
            __builtin_amdgcn_sched_barrier(0);                     // every request of the step is out before the first FMA
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool live = c0 + 64 * u < nchunk;           // a chunk beyond the row contributes fma(0, x, acc) = acc
                float xf[MMAX][8];
#pragma unroll
                for (int m = 0; m < MMAX; ++m) unpack8<T>(xv[u][m], xf[m]);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    float wf[8];
                    unpack8<T>(live ? wv[u][r] : u32x4{0, 0, 0, 0}, wf);
#pragma unroll
                    for (int m = 0; m < MMAX; ++m)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[r][m] = fmaf(wf[e], xf[m][e], acc[r][m]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MMAX; ++m) {
                const float s = wave_sum(acc[r][m]);
                if (lane == 0 && nb + r < N && m < M) Y[(size_t)m * ldy + nb + r] = T::from_f32(s);
            }
    }
}

// ---- skinny gated-MLP front half: gate/up GEMV + act(gate) * up in one launch ------------------------------------------
//   W: [2I, K] with gate/up rows interleaved in blocks of 32 (the MODE_GEGLU weight layout); a wave owns RPW features i and
//   streams their gate row (i/32)*64 + i%32 and up row (+32), UNR chunks of each in flight.  Y[m][i] = T( T(act(T(g))) * T(u) ) — the
//   values of gemv_kernel followed by geglu_unpack_kernel, bit for bit (same per-lane accumulation order), one launch and no [M, 2I]
//   round trip.
template <typename T, int MMAX, int RPW, int UNR>
__global__ __launch_bounds__(256) void gemv_glu_kernel(const u16* __restrict__ X, const u16* __restrict__ W, u16* __restrict__ Y,
                                                       int M, int I, int K, int ldx, int ldw, int ldy, int silu) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = K / 8;
    const int waves_total = gridDim.x * 4;
    for (int ib = (blockIdx.x * 4 + wave) * RPW; ib < I; ib += waves_total * RPW) {
        float ag[RPW][MMAX], au[RPW][MMAX];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MMAX; ++m) { ag[r][m] = 0.f; au[r][m] = 0.f; }
        const u16* grow[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int i = min(ib + r, I - 1);
            grow[r] = W + ((size_t)(i >> 5) * 64 + (i & 31)) * ldw;
        }
        for (int c0 = lane; c0 < nchunk; c0 += 64 * UNR) {
            u32x4 wg[UNR][RPW], wu[UNR][RPW], xv[UNR][MMAX];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = min(c0 + 64 * u, nchunk - 1);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    wg[u][r] = __builtin_nontemporal_load((const u32x4*)(grow[r] + c * 8));
                    wu[u][r] = __builtin_nontemporal_load((const u32x4*)(grow[r] + (size_t)32 * ldw + c * 8));
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int c = min(c0 + 64 * u, nchunk - 1);
#pragma unroll
                for (int m = 0; m < MMAX; ++m) xv[u][m] = *(const u32x4*)(X + (size_t)min(m, M - 1) * ldx + c * 8);
            }
            __builtin_amdgcn_sched_barrier(0);                     // every request of the step is out before the first FMA
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool live = c0 + 64 * u < nchunk;
                float xf[MMAX][8];
#pragma unroll
                for (int m = 0; m < MMAX; ++m) unpack8<T>(xv[u][m], xf[m]);
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    float gf[8], uf[8];
                    unpack8<T>(live ? wg[u][r] : u32x4{0, 0, 0, 0}, gf);
                    unpack8<T>(live ? wu[u][r] : u32x4{0, 0, 0, 0}, uf);
#pragma unroll
                    for (int m = 0; m < MMAX; ++m) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ag[r][m] = fmaf(gf[e], xf[m][e], ag[r][m]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) au[r][m] = fmaf(uf[e], xf[m][e], au[r][m]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MMAX; ++m) {
                const float g = rnd<T>(wave_sum(ag[r][m])), u = rnd<T>(wave_sum(au[r][m]));
                if (lane == 0 && ib + r < I && m < M) Y[(size_t)m * ldy + ib + r] = T::from_f32(rnd<T>(silu ? silu_f(g) : gelu_tanh_f(g)) * u);
            }
    }
}

int vidi_gemv_glu_dispatch(const void* X, const void* W, void* Y, int M, int I, int K, int ldx, int ldw, int ldy, int act, int dtype,
                           hipStream_t st) {
    if (M <= 0 || M > 8 || I <= 0 || (I % 32) || K <= 0 || K % 8 != 0 || ldw % 8 != 0 || ldx % 8 != 0) return VIDI_ERR_SHAPE;
    if (((uintptr_t)W & 15) || ((uintptr_t)X & 15)) return VIDI_ERR_ALIGN;
    if (act != ACT_GELU_TANH && act != ACT_SILU) return VIDI_ERR_ARG;
    const int silu = act == ACT_SILU;
    auto go = [&](auto kern, int rpw) -> int {
        const int blocks = max(1, min((I + 4 * rpw - 1) / (4 * rpw), 256 * 8));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, (const u16*)X, (const u16*)W, (u16*)Y, M, I, K, ldx, ldw, ldy, silu);
        return (int)hipGetLastError();
    };
    if (dtype == VIDI_DT_BF16) {
        if (M <= 1) return go(gemv_glu_kernel<BF16, 1, VIDI_GLU_RPW, VIDI_GLU_UNR>, VIDI_GLU_RPW);
        if (M <= 2) return go(gemv_glu_kernel<BF16, 2, VIDI_GLU_RPW, VIDI_GLU_UNR>, VIDI_GLU_RPW);
        if (M <= 4) return go(gemv_glu_kernel<BF16, 4, 1, 2>, 1);
        return go(gemv_glu_kernel<BF16, 8, 1, 2>, 1);
    } else if (dtype == VIDI_DT_F16) {
        if (M <= 1) return go(gemv_glu_kernel<F16, 1, VIDI_GLU_RPW, VIDI_GLU_UNR>, VIDI_GLU_RPW);
        if (M <= 2) return go(gemv_glu_kernel<F16, 2, VIDI_GLU_RPW, VIDI_GLU_UNR>, VIDI_GLU_RPW);
        if (M <= 4) return go(gemv_glu_kernel<F16, 4, 1, 2>, 1);
        return go(gemv_glu_kernel<F16, 8, 1, 2>, 1);
    }
    return VIDI_ERR_DTYPE;
}

int vidi_gemv_dispatch(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy,
                       int dtype, hipStream_t st) {
    if (M <= 0 || M > 8 || N <= 0 || K <= 0 || K % 8 != 0 || ldw % 8 != 0 || ldx % 8 != 0) return VIDI_ERR_SHAPE;
    if (((uintptr_t)W & 15) || ((uintptr_t)X & 15)) return VIDI_ERR_ALIGN;
    auto go = [&](auto kern, int rpw) -> int {
        const int blocks = max(1, min((N + 4 * rpw - 1) / (4 * rpw), 256 * 8));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, (const u16*)X, (const u16*)W, (u16*)Y, M, N, K, ldx, ldw, ldy);
        return (int)hipGetLastError();
    };
    if (dtype == VIDI_DT_BF16) {
        if (M <= 1) return go(gemv_kernel<BF16, 1, VIDI_GEMV_RPW, VIDI_GEMV_UNR>, VIDI_GEMV_RPW);
        if (M <= 2) return go(gemv_kernel<BF16, 2, VIDI_GEMV_RPW, VIDI_GEMV_UNR>, VIDI_GEMV_RPW);
        if (M <= 4) return go(gemv_kernel<BF16, 4, VIDI_GEMV_RPW, VIDI_GEMV_UNR>, VIDI_GEMV_RPW);
        return go(gemv_kernel<BF16, 8, 2, 2>, 2);
    } else if (dtype == VIDI_DT_F16) {
        if (M <= 1) return go(gemv_kernel<F16, 1, VIDI_GEMV_RPW, VIDI_GEMV_UNR>, VIDI_GEMV_RPW);
        if (M <= 2) return go(gemv_kernel<F16, 2, VIDI_GEMV_RPW, VIDI_GEMV_UNR>, VIDI_GEMV_RPW);
        if (M <= 4) return go(gemv_kernel<F16, 4, VIDI_GEMV_RPW, VIDI_GEMV_UNR>, VIDI_GEMV_RPW);
        return go(gemv_kernel<F16, 8, 2, 2>, 2);
    }
    return VIDI_ERR_DTYPE;
}
