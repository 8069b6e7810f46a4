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

// ---- decode: the Gemma2 norm pair FUSED INTO the projection that consumes it ------------------------------------------------------
//   vidi_resid_norm2 (s = A+B+C; Y1 = Res + gemma(s; W1); x = gemma(Y1; W2)) followed by a skinny projection of x was two launches: a
//   one-wave kernel of three dependent phases (6.8 us per call in the round-2 decode trace, 84 calls per token) and the GEMV.  Here
//   every block of the GEMV derives x itself — the row is 7 KB, all operands come from L2 — while its first two batches of weight
//   rows are already in flight: the operand loads of the norm are requested first, the weights right behind them, and the norm's
//   arithmetic runs under the weights' HBM latency.  Block 0 writes Y1 (it must not alias Res: the other blocks read Res at their own
//   pace).  x lives in LDS in the storage dtype and feeds the FMAs from there.
//   The element arithmetic and rounding points are resid_norm2's; the two sums of squares are block reductions (256 threads x 2 chunks)
//   instead of one wave's, so Y1 / x can differ from resid_norm2's in the last bit of the dtype.
//   The weight stream is software-pipelined over (row group, K step) pairs with two batches (RPW x UNR x 1 KB each) in flight per wave.
struct GemvNorm2Params {
    const u16* A; const u16* B; const u16* C; const u16* Res; const u16* W1; const u16* W2; u16* Y1;   // [M, K] rows, stride ld
    long long ld; float eps;
    const u16* W; u16* Y;                    // projection weight [N, K] (GLU: [2I, K] interleaved, N = I) and output [M, N]
    int M, N, K, ldw, ldy, silu;
};

template <typename T, int MMAX, int RPW, int UNR, bool GLU>
__global__ __launch_bounds__(256) void gemv_norm2_kernel(GemvNorm2Params p) {
    constexpr int NW = GLU ? 2 * RPW : RPW;            // weight rows a wave streams per step
    constexpr int MAXJ = 2;                            // row chunks per thread in the norm: K <= 4096
    extern __shared__ __attribute__((aligned(16))) u16 xs[];   // [MMAX][K] normalised input rows
    __shared__ float sred[2 * MMAX][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = p.K / 8;
    const int CS = (nchunk + 64 * UNR - 1) / (64 * UNR);        // K steps per row group
    const int stride = gridDim.x * 4 * RPW;
    const int first = (blockIdx.x * 4 + wave) * RPW;
    const int ngroups = first < p.N ? (p.N - first + stride - 1) / stride : 0;
    const int nsteps = ngroups * CS;

    // ---- operands of row 0's norm: requested first (L2 hits) ------------------------------------------------------------
    u32x4 ra[MAXJ], rb[MAXJ], rc[MAXJ], rr[MAXJ], rw1[MAXJ], rw2[MAXJ];
    const u16* Bp = p.B ? p.B : p.A;
    const u16* Cp = p.C ? p.C : p.A;
    auto load_row = [&](int m) {
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int ch = min(tid + 256 * j, nchunk - 1);
            const size_t o = (size_t)m * p.ld + ch * 8;
            ra[j] = *(const u32x4*)(p.A + o);
            rb[j] = *(const u32x4*)(Bp + o);               // absent operands re-read A (ignored below): no conditional load, so the
            rc[j] = *(const u32x4*)(Cp + o);               // waits below are counted exactly and never cover the weight batches
            rr[j] = *(const u32x4*)(p.Res + o);
        }
    };
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int ch = min(tid + 256 * j, nchunk - 1);
        rw1[j] = *(const u32x4*)(p.W1 + ch * 8);
        rw2[j] = *(const u32x4*)(p.W2 + ch * 8);
    }
    load_row(0);

    // ---- weight stream: step s = (row group s / CS, K step s % CS) ------------------------------------------------------------
    u32x4 wb[2][UNR][NW];
    auto issue = [&](int buf, int s) {
        const int nb = first + (s / CS) * stride, c0 = lane + (s % CS) * 64 * UNR;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int c = min(c0 + 64 * u, nchunk - 1);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int n = min(nb + r, p.N - 1);
                if constexpr (GLU) {
                    const u16* g = p.W + ((size_t)(n >> 5) * 64 + (n & 31)) * p.ldw + c * 8;
                    wb[buf][u][2 * r] = __builtin_nontemporal_load((const u32x4*)g);
                    wb[buf][u][2 * r + 1] = __builtin_nontemporal_load((const u32x4*)(g + (size_t)32 * p.ldw));
                } else {
                    wb[buf][u][r] = __builtin_nontemporal_load((const u32x4*)(p.W + (size_t)n * p.ldw + c * 8));
                }
            }
        }
    };
    issue(0, 0);                                       // unconditional (row indices are clamped): a wave without work re-reads the last rows
    issue(1, 1);
    __builtin_amdgcn_sched_barrier(0);

    // ---- the norm pair, row by row (decode: one row) -----------------------------------------------------------------------
    auto block_sum = [&](float v, int slot) {
        v = wave_sum(v);
        if (lane == 0) sred[slot][wave] = v;
        __syncthreads();
        return (sred[slot][0] + sred[slot][1]) + (sred[slot][2] + sred[slot][3]);
    };
    auto norm_row = [&](int m) {
        float x[MAXJ][8];
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            float z[8];
            unpack8<T>(ra[j], x[j]);
            if (p.B) {
                unpack8<T>(rb[j], z);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[j][e] = rnd<T>(x[j][e] + z[e]);
            }
            if (p.C) {
                unpack8<T>(rc[j], z);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[j][e] = rnd<T>(x[j][e] + z[e]);
            }
            if (tid + 256 * j < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s2 += x[j][e] * x[j][e];
            }
        }
        const float rs1 = rsqrtf(block_sum(s2, 2 * m) / p.K + p.eps);
        float t2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            float w[8], r[8];
            unpack8<T>(rw1[j], w);
            unpack8<T>(rr[j], r);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[j][e] = rnd<T>(r[e] + rnd<T>(x[j][e] * rs1 * (1.0f + w[e])));
            if (tid + 256 * j < nchunk) {
                if (blockIdx.x == 0) *(u32x4*)(p.Y1 + (size_t)m * p.ld + (tid + 256 * j) * 8) = pack8<T>(x[j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) t2 += x[j][e] * x[j][e];
            }
        }
        const float rs2 = rsqrtf(block_sum(t2, 2 * m + 1) / p.K + p.eps);
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            float w[8], y[8];
            unpack8<T>(rw2[j], w);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = x[j][e] * rs2 * (1.0f + w[e]);
            if (tid + 256 * j < nchunk) *(u32x4*)(xs + (size_t)m * p.K + (tid + 256 * j) * 8) = pack8<T>(y);
        }
    };
    norm_row(0);
    for (int m = 1; m < p.M; ++m) {
        load_row(m);
        norm_row(m);
    }
    __syncthreads();

    // ---- FMAs of one step; the row group's reduction and store after its last K step ---------------------------------------------
    float acc[NW][MMAX];
    auto step = [&](int buf, int s) {
        const int nb = first + (s / CS) * stride, ci = s % CS, c0 = lane + ci * 64 * UNR;
        if (ci == 0) {
#pragma unroll
            for (int r = 0; r < NW; ++r)
#pragma unroll
                for (int m = 0; m < MMAX; ++m) acc[r][m] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const bool live = c0 + 64 * u < nchunk;
            const int c = min(c0 + 64 * u, nchunk - 1);
            float xf[MMAX][8];
#pragma unroll
            for (int m = 0; m < MMAX; ++m) unpack8<T>(*(const u32x4*)(xs + (size_t)min(m, p.M - 1) * p.K + c * 8), xf[m]);
#pragma unroll
            for (int r = 0; r < NW; ++r) {
                float wf[8];
                unpack8<T>(live ? wb[buf][u][r] : u32x4{0, 0, 0, 0}, wf);
#pragma unroll
                for (int m = 0; m < MMAX; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[r][m] = fmaf(wf[e], xf[m][e], acc[r][m]);
            }
        }
        if (ci == CS - 1) {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int m = 0; m < MMAX; ++m) {
                    if constexpr (GLU) {
                        const float g = rnd<T>(wave_sum(acc[2 * r][m])), u = rnd<T>(wave_sum(acc[2 * r + 1][m]));
                        if (lane == 0 && nb + r < p.N && m < p.M)
                            p.Y[(size_t)m * p.ldy + nb + r] = T::from_f32(rnd<T>(p.silu ? silu_f(g) : gelu_tanh_f(g)) * u);
                    } else {
                        const float sum = wave_sum(acc[r][m]);
                        if (lane == 0 && nb + r < p.N && m < p.M) p.Y[(size_t)m * p.ldy + nb + r] = T::from_f32(sum);
                    }
                }
        }
    };
    int s = 0;
    for (; s + 3 < nsteps; s += 2) {                   // steady state: both refills exist, so every wait is an exact count
        step(0, s);
        issue(0, s + 2);
        step(1, s + 1);
        issue(1, s + 3);
    }
    if (s < nsteps) {
        step(0, s);
        if (s + 2 < nsteps) issue(0, s + 2);
    }
    if (s + 1 < nsteps) step(1, s + 1);
    if (s + 2 < nsteps) step(0, s + 2);
}

#ifndef VIDI_GEMV_FUSED_BLOCKS
#define VIDI_GEMV_FUSED_BLOCKS 512                     // 2 blocks per CU
#endif

int vidi_gemv_norm2_dispatch(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1,
                             long long ld, float eps, const void* W, void* Y, int M, int N, int K, int ldw, int ldy, int glu_act,
                             int dtype, hipStream_t st) {
    // glu_act < 0: plain projection (N rows of W); else gated pair with that activation (N = I features, W = [2I, K] interleaved)
    if (M <= 0 || M > 4 || N <= 0 || K <= 0 || K % 8 != 0 || K > 4096 || ldw % 8 != 0 || ld % 8 != 0) return VIDI_ERR_SHAPE;
    const bool glu = glu_act >= 0;
    if (glu && (N % 32 != 0 || (glu_act != ACT_GELU_TANH && glu_act != ACT_SILU))) return VIDI_ERR_ARG;
    if (Y1 == Res) return VIDI_ERR_ARG;                // every block reads Res while block 0 writes Y1
    for (const void* q : {A, Res, W1, W2, (const void*)Y1, W})
        if (((uintptr_t)q & 15)) return VIDI_ERR_ALIGN;
    if ((B && ((uintptr_t)B & 15)) || (C && ((uintptr_t)C & 15))) return VIDI_ERR_ALIGN;
    GemvNorm2Params p;
    p.A = (const u16*)A; p.B = (const u16*)B; p.C = (const u16*)C; p.Res = (const u16*)Res; p.W1 = (const u16*)W1; p.W2 = (const u16*)W2;
    p.Y1 = (u16*)Y1; p.ld = ld; p.eps = eps; p.W = (const u16*)W; p.Y = (u16*)Y; p.M = M; p.N = N; p.K = K; p.ldw = ldw; p.ldy = ldy;
    p.silu = glu_act == ACT_SILU;
    auto go = [&](auto kern, int mmax, int rpw) -> int {
        const int blocks = max(1, min((N + 4 * rpw - 1) / (4 * rpw), VIDI_GEMV_FUSED_BLOCKS));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), (size_t)mmax * K * 2, st, p);
        return (int)hipGetLastError();
    };
#define VIDI_GN2(TT)                                                                                            \
    do {                                                                                                        \
        if (glu) {                                                                                              \
            if (M <= 1) return go(gemv_norm2_kernel<TT, 1, VIDI_GLU_RPW, VIDI_GLU_UNR, true>, 1, VIDI_GLU_RPW); \
            if (M <= 2) return go(gemv_norm2_kernel<TT, 2, VIDI_GLU_RPW, VIDI_GLU_UNR, true>, 2, VIDI_GLU_RPW); \
            return go(gemv_norm2_kernel<TT, 4, 1, 2, true>, 4, 1);                                              \
        }                                                                                                       \
        if (M <= 1) return go(gemv_norm2_kernel<TT, 1, VIDI_GEMV_RPW, VIDI_GEMV_UNR, false>, 1, VIDI_GEMV_RPW); \
        if (M <= 2) return go(gemv_norm2_kernel<TT, 2, VIDI_GEMV_RPW, VIDI_GEMV_UNR, false>, 2, VIDI_GEMV_RPW); \
        return go(gemv_norm2_kernel<TT, 4, VIDI_GEMV_RPW, 2, false>, 4, VIDI_GEMV_RPW);                         \
    } while (0)
    if (dtype == VIDI_DT_BF16) VIDI_GN2(BF16);
    if (dtype == VIDI_DT_F16) VIDI_GN2(F16);
#undef VIDI_GN2
    return VIDI_ERR_DTYPE;
}
