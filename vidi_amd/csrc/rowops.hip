// Row-wise normalisations of the hot path — all HBM-bound: one wave per row, 16-byte loads, the
// whole row held in registers between the reduction and the write (one read + one write per
// element).  Algorithmic bytes = rows * H * 2 B * (reads + writes).
//
// Reference formulas:
//   GEMMA   : y = T( x32 * rsqrt(mean(x32^2)+eps) * (1 + w32) )                  TP gemma2:55-63
//   GEMMA_ADD: y = T( res + T(gemma(x)) )   — `residual + post_norm(x)`          gemma.py:201,237,121
//   MM      : y = T( w * T( x32 * rsqrt(mean(x32^2)+eps) ) )                     mm_layer/norm.py:9-25
//   MM_NOW  : y = T( x32 * rsqrt(mean(x32^2)+eps) )   (weight-free rms_norm)     mm_layer/norm.py:9-16
//   LLMNORM : mask = (sum|x| != 0) & sample_flag ; y = T(T(T(MM(x)) * mask) * normalizer)
//                                                     multimodal.py:201-206 + gemma.py:353-356
//   LAYERNORM: y = T( (x32-mean)*rsqrt(var+eps)*w + b )                           nn.LayerNorm
#include "kernels.h"



template <typename T, int MODE, int MAXC>
__global__ __launch_bounds__(256) void norm_kernel(NormParams p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nchunk = p.H / 8;
    float x[MAXC][8];
    float s1 = 0.f, s2 = 0.f, sa = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            if constexpr (MODE == NORM_MM_NOW) {
                if (p.XF32) {
                    const float* src = p.XF32 + row * p.ldx + ch * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[c][e] = rnd<T>(src[e]);
                } else {
                    unpack8<T>(*(const u32x4*)(p.X + row * p.ldx + ch * 8), x[c]);
                }
            } else {
                unpack8<T>(*(const u32x4*)(p.X + row * p.ldx + ch * 8), x[c]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += x[c][e];
                s2 += x[c][e] * x[c][e];
                sa += fabsf(x[c][e]);
            }
        }
    }
    s2 = wave_sum(s2);
    float mean = 0.f, rs;
    if constexpr (MODE == NORM_LAYER) {
        mean = wave_sum(s1) / p.H;
        // two-pass variance on the registers (matches torch's numerically stable form)
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (lane + c * 64 < nchunk)
#pragma unroll
                for (int e = 0; e < 8; ++e) v += (x[c][e] - mean) * (x[c][e] - mean);
        rs = rsqrtf(wave_sum(v) / p.H + p.eps);
    } else {
        rs = rsqrtf(s2 / p.H + p.eps);
    }
    float maskv = 1.f;
    if constexpr (MODE == NORM_LLM) {
        sa = wave_sum(sa);
        const int sflag = p.sample_flag ? *p.sample_flag : 1;
        maskv = (sa != 0.f && sflag) ? 1.f : 0.f;
        if (lane == 0 && p.Mask) p.Mask[row] = (unsigned char)(maskv != 0.f);
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float w[8], y[8];
            if constexpr (MODE != NORM_MM_NOW) unpack8<T>(*(const u32x4*)(p.Wt + ch * 8), w);
            if constexpr (MODE == NORM_GEMMA || MODE == NORM_GEMMA_ADD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = x[c][e] * rs * (1.0f + w[e]);
                if constexpr (MODE == NORM_GEMMA_ADD) {
                    float r[8];
                    unpack8<T>(*(const u32x4*)(p.Res + row * p.ldr + ch * 8), r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = r[e] + rnd<T>(y[e]);
                }
            } else if constexpr (MODE == NORM_MM) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = w[e] * rnd<T>(x[c][e] * rs);
            } else if constexpr (MODE == NORM_MM_NOW) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = x[c][e] * rs;
            } else if constexpr (MODE == NORM_LLM) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = rnd<T>(rnd<T>(w[e] * rnd<T>(x[c][e] * rs)) * maskv) * p.normalizer;
            } else {
                float bb[8];
                unpack8<T>(*(const u32x4*)(p.Bias + ch * 8), bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (x[c][e] - mean) * rs * w[e] + bb[e];
            }
            *(u32x4*)(p.Y + row * p.ldy + ch * 8) = pack8<T>(y);
        }
    }
}

// ---- text-stream fusion (Gemma2 wiring): add3 -> residual + post-norm -> next pre-norm in ONE pass over the row ------------
//   s  = T(T(a + b) + c)                      (b, c optional)                      gemma.py:236
//   y1 = T(res + T(gemma(s;  w1)))            -> Y1 (may alias res)                gemma.py:237 / :120-121
//   y2 = T(gemma(y1; w2))                     -> Y2                                gemma.py:118 / :162 of the next layer / :411
// Arithmetic, rounding points and summation order are those of add3_kernel + norm_kernel<GEMMA_ADD> + norm_kernel<GEMMA> run one
// after the other (bit-identical results); it replaces 3 (attention side) or 2 (FFN side) launches of the decode step.
// PRELOAD (decode, a handful of rows): every operand of the row is requested up front — one memory round trip for the whole kernel; at
// M = 1 a single wave runs this and the three dependent phases would otherwise each pay a full load latency.  !PRELOAD (the multimodal
// stream, 10^5 rows): operands are fetched phase by phase, so a row costs ~90 VGPRs instead of ~200 and 5 waves per SIMD stay resident
// (the preloading form measured 4.3 TB/s at 126 080 rows, two separate launches 5.1 TB/s).
template <typename T, int MAXC, bool PRELOAD>
__device__ __forceinline__ void resid_norm2_body(const u16* __restrict__ A, const u16* __restrict__ B, const u16* __restrict__ C,
                                                 const u16* Res, const u16* __restrict__ W1, const u16* __restrict__ W2,
                                                 u16* Y1, u16* __restrict__ Y2, int rows, int H, long long ld, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = H / 8;
    u32x4 ra[MAXC], rb[PRELOAD ? MAXC : 1], rc[PRELOAD ? MAXC : 1], rr[PRELOAD ? MAXC : 1], rw1[PRELOAD ? MAXC : 1], rw2[PRELOAD ? MAXC : 1];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            ra[c] = *(const u32x4*)(A + row * ld + ch * 8);
            if constexpr (PRELOAD) {
                if (B) rb[c] = *(const u32x4*)(B + row * ld + ch * 8);
                if (C) rc[c] = *(const u32x4*)(C + row * ld + ch * 8);
                rr[c] = *(const u32x4*)(Res + row * ld + ch * 8);
                rw1[c] = *(const u32x4*)(W1 + ch * 8);
                rw2[c] = *(const u32x4*)(W2 + ch * 8);
            }
        }
    }
    // the row between the phases is kept PACKED: every value is already rounded to T (rnd<T>), so packing is lossless and the row
    // costs 4 registers per chunk instead of 8
    u32x4 xp[MAXC];
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float x[8], z[8];
            unpack8<T>(ra[c], x);
            if (B) {
                unpack8<T>(PRELOAD ? rb[c] : *(const u32x4*)(B + row * ld + ch * 8), z);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = rnd<T>(x[e] + z[e]);
            }
            if (C) {
                unpack8<T>(PRELOAD ? rc[c] : *(const u32x4*)(C + row * ld + ch * 8), z);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = rnd<T>(x[e] + z[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s2 += x[e] * x[e];
            xp[c] = (B || C) ? pack8<T>(x) : ra[c];
        }
    }
    const float rs1 = rsqrtf(wave_sum(s2) / H + eps);
    if constexpr (!PRELOAD) __builtin_amdgcn_sched_barrier(0);
    float t2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float x[8], w[8], r[8];
            unpack8<T>(xp[c], x);
            unpack8<T>(PRELOAD ? rw1[c] : *(const u32x4*)(W1 + ch * 8), w);
            unpack8<T>(PRELOAD ? rr[c] : *(const u32x4*)(Res + row * ld + ch * 8), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = rnd<T>(r[e] + rnd<T>(x[e] * rs1 * (1.0f + w[e])));
            xp[c] = pack8<T>(x);
            *(u32x4*)(Y1 + row * ld + ch * 8) = xp[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) t2 += x[e] * x[e];
        }
    }
    const float rs2 = rsqrtf(wave_sum(t2) / H + eps);
    if constexpr (!PRELOAD) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float x[8], w[8], y[8];
            unpack8<T>(xp[c], x);
            unpack8<T>(PRELOAD ? rw2[c] : *(const u32x4*)(W2 + ch * 8), w);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = x[e] * rs2 * (1.0f + w[e]);
            *(u32x4*)(Y2 + row * ld + ch * 8) = pack8<T>(y);
        }
    }
}

// decode form (rows <= 64): all six operands of a row requested up front, latency over occupancy
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void resid_norm2_kernel(const u16* A, const u16* B, const u16* C, const u16* Res, const u16* W1,
                                                          const u16* W2, u16* Y1, u16* Y2, int rows, int H, long long ld, float eps) {
    resid_norm2_body<T, MAXC, true>(A, B, C, Res, W1, W2, Y1, Y2, rows, H, ld, eps);
}

// many-row form (the multimodal stream, 126 080 rows): operands requested phase by phase, registers capped for 4 waves per SIMD
template <typename T, int MAXC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void resid_norm2_rows_kernel(const u16* A, const u16* B, const u16* C, const u16* Res, const u16* W1, const u16* W2, u16* Y1, u16* Y2,
                             int rows, int H, long long ld, float eps) {
    resid_norm2_body<T, MAXC, false>(A, B, C, Res, W1, W2, Y1, Y2, rows, H, ld, eps);
}

// ---- LayerNorm statistics only: stats[row] = (mean, rsqrt(var + eps)), two-pass variance on the register-resident row ------
//   The consuming projection applies them in its epilogue (GemmParams::ln_stats): LayerNorm(x) is never written.
//   One read of the row: algorithmic bytes = rows * H * 2 B.
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void row_stats_kernel(const u16* __restrict__ X, float* __restrict__ stats, long long rows, int H,
                                                        long long ldx, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = H / 8;
    float x[MAXC][8];
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            unpack8<T>(*(const u32x4*)(X + row * ldx + ch * 8), x[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s1 += x[c][e];
        }
    }
    const float mean = wave_sum(s1) / H;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + c * 64 < nchunk)
#pragma unroll
            for (int e = 0; e < 8; ++e) v += (x[c][e] - mean) * (x[c][e] - mean);
    const float rs = rsqrtf(wave_sum(v) / H + eps);
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rs;
    }
}

// ---- LayerNorm statistics from the per-strip partial sums a producing GEMM left (GemmParams::stat_part) ------------------------------
//   part[row][entry] = (sum y, sum y^2) over a group of columns (a 128-column strip; for the 288-wide tile geometry 128 + 128 + 16 + 16
//   columns of a tile: vidi_stat_strips(N) entries per row);  stats[row] = (mean, rsqrt(E[y^2] - mean^2 + eps)).
//   One-pass variance in fp32: the relative error of var is ~1e-7 * E[y^2]/var, i.e. below the bf16 rounding of the consumer for any
//   row whose mean is within ~50 standard deviations of zero (the towers' residual streams are within a few).
//   A block finalizes 256 consecutive rows: their 256 x nstr entries are one contiguous piece of `part`, copied to LDS with consecutive
//   lanes on consecutive entries (512 B per wave instruction), one spare entry per row so that the per-row sums below read distinct
//   banks; then thread r adds row r's entries in ascending order (the order, and so every bit of the result, of a plain row loop).
//   (Round 4: one thread per row reading its own 8-byte entries nstr x 8 B apart ran at 0.75 TB/s — 89 us per call, 1 % of the prefill.)
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, long long rows, int nstr,
                                                          float invH, float eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x2_t* sp = (f32x2_t*)smem;                                                 // [256][nstr + 1]
    const long long row0 = (long long)blockIdx.x * 256;
    const int nrow = (int)(rows - row0 < 256 ? rows - row0 : 256);
    const int total = nrow * nstr;
    const f32x2_t* src = (const f32x2_t*)part + row0 * nstr;
    const int qstep = 256 / nstr, rstep = 256 % nstr;                             // entry e -> (row, strip), advanced by 256 entries per step
    int r = (int)threadIdx.x / nstr, c = (int)threadIdx.x % nstr;
    for (int e = threadIdx.x; e < total; e += 256) {
        sp[r * (nstr + 1) + c] = src[e];
        r += qstep; c += rstep;
        if (c >= nstr) { c -= nstr; ++r; }
    }
    __syncthreads();
    if ((int)threadIdx.x >= nrow) return;
    const f32x2_t* p = sp + threadIdx.x * (nstr + 1);
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < nstr; ++i) { const f32x2_t v = p[i]; s1 += v[0]; s2 += v[1]; }
    const float mean = s1 * invH;
    const float var = fmaxf(s2 * invH - mean * mean, 0.f);
    *((f32x2_t*)stats + row0 + threadIdx.x) = f32x2_t{mean, rsqrtf(var + eps)};
}

// rows wider than 31 strips (> 3 968 columns: none of the towers'): one thread per row, entries read in place
__global__ __launch_bounds__(256) void ln_finalize_wide_kernel(const float* __restrict__ part, float* __restrict__ stats, long long rows, int nstr,
                                                               float invH, float eps) {
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const f32x2_t* p = (const f32x2_t*)part + row * nstr;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < nstr; ++i) { const f32x2_t v = p[i]; s1 += v[0]; s2 += v[1]; }
    const float mean = s1 * invH;
    const float var = fmaxf(s2 * invH - mean * mean, 0.f);
    *((f32x2_t*)stats + row) = f32x2_t{mean, rsqrtf(var + eps)};
}

int vidi_ln_finalize_dispatch(const float* part, float* stats, long long rows, int nstr, int H, float eps, hipStream_t st) {
    if (rows <= 0 || nstr <= 0 || H <= 0) return VIDI_ERR_SHAPE;
    const size_t lds = (size_t)256 * (nstr + 1) * sizeof(f32x2_t);
    if (lds > 65536) {
        hipLaunchKernelGGL(ln_finalize_wide_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, part, stats, rows, nstr, 1.0f / H, eps);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), lds, st, part, stats, rows, nstr, 1.0f / H, eps);
    return (int)hipGetLastError();
}

// the same partial sums computed from a stored matrix (small problems whose GEMM runs on a tile kernel without the fused emission)
template <typename T>
__global__ __launch_bounds__(256) void row_partials_kernel(const u16* __restrict__ Y, float* __restrict__ part, long long rows, int N, long long ldy,
                                                           int nstr) {
    // nstr entries per row (>= ceil(N / 128): the layout the fused producers of this width write, vidi_stat_strips); entries past the last
    // real 128-column strip are written as (0, 0)
    const long long idx = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);          // one 16-lane row per (matrix row, strip)
    const int c = threadIdx.x & 15;
    const bool live = idx < rows * nstr;
    const long long row = live ? idx / nstr : 0;
    const int strip = live ? (int)(idx % nstr) : 0;
    const int n = strip * 128 + c * 8;
    float s1 = 0.f, s2 = 0.f;
    if (live && n < N) {
        float x[8];
        unpack8<T>(*(const u32x4*)(Y + row * ldy + n), x);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += x[e]; s2 = __builtin_fmaf(x[e], x[e], s2); }
    }
    row16_sum2(s1, s2);
    if (live && c == 0) *((f32x2_t*)part + idx) = f32x2_t{s1, s2};
}

int vidi_row_partials_dispatch(const void* Y, float* part, long long rows, int N, long long ldy, int nstr, int dtype, hipStream_t st) {
    if (rows <= 0 || N <= 0 || N % 8 || ldy % 8 || nstr < ((N + 127) >> 7)) return VIDI_ERR_SHAPE;
    const long long items = rows * nstr;
    const dim3 grid((unsigned)((items + 15) / 16));
    if (dtype == VIDI_DT_BF16) hipLaunchKernelGGL(row_partials_kernel<BF16>, grid, dim3(256), 0, st, (const u16*)Y, part, rows, N, ldy, nstr);
    else if (dtype == VIDI_DT_F16) hipLaunchKernelGGL(row_partials_kernel<F16>, grid, dim3(256), 0, st, (const u16*)Y, part, rows, N, ldy, nstr);
    else return VIDI_ERR_DTYPE;
    return (int)hipGetLastError();
}

int vidi_row_stats_dispatch(const void* X, float* stats, long long rows, int H, long long ldx, float eps, int dtype, hipStream_t st) {
    if (rows <= 0 || H <= 0 || H % 8 || ldx % 8) return VIDI_ERR_SHAPE;
    const int nchunk = H / 8;
    const int grid = (int)((rows + 3) / 4);
#define VIDI_RS(TT, MC) hipLaunchKernelGGL((row_stats_kernel<TT, MC>), dim3(grid), dim3(256), 0, st, (const u16*)X, stats, rows, H, ldx, eps)
    if (dtype == VIDI_DT_BF16) {
        if (nchunk <= 64) VIDI_RS(BF16, 1); else if (nchunk <= 192) VIDI_RS(BF16, 3); else if (nchunk <= 512) VIDI_RS(BF16, 8); else return VIDI_ERR_SHAPE;
    } else if (dtype == VIDI_DT_F16) {
        if (nchunk <= 64) VIDI_RS(F16, 1); else if (nchunk <= 192) VIDI_RS(F16, 3); else if (nchunk <= 512) VIDI_RS(F16, 8); else return VIDI_ERR_SHAPE;
    } else return VIDI_ERR_DTYPE;
#undef VIDI_RS
    return (int)hipGetLastError();
}

int vidi_resid_norm2_dispatch(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1,
                              void* Y2, int rows, int H, long long ld, float eps, int dtype, hipStream_t st) {
    if (rows <= 0 || H <= 0 || H % 8 || ld % 8) return VIDI_ERR_SHAPE;
    const int nchunk = H / 8, grid = (rows + 3) / 4;
#define VIDI_RN2_(TT, MC, KN)                                                                                                     \
    hipLaunchKernelGGL((KN<TT, MC>), dim3(grid), dim3(256), 0, st, (const u16*)A, (const u16*)B, (const u16*)C, \
                       (const u16*)Res, (const u16*)W1, (const u16*)W2, (u16*)Y1, (u16*)Y2, rows, H, ld, eps)
#define VIDI_RN2(TT, MC) do { if (rows <= 64) VIDI_RN2_(TT, MC, resid_norm2_kernel); else VIDI_RN2_(TT, MC, resid_norm2_rows_kernel); } while (0)
    if (dtype == VIDI_DT_BF16) {
        if (nchunk <= 64) VIDI_RN2(BF16, 1); else if (nchunk <= 192) VIDI_RN2(BF16, 3); else if (nchunk <= 512) VIDI_RN2(BF16, 8); else return VIDI_ERR_SHAPE;
    } else if (dtype == VIDI_DT_F16) {
        if (nchunk <= 64) VIDI_RN2(F16, 1); else if (nchunk <= 192) VIDI_RN2(F16, 3); else if (nchunk <= 512) VIDI_RN2(F16, 8); else return VIDI_ERR_SHAPE;
    } else return VIDI_ERR_DTYPE;
#undef VIDI_RN2_
#undef VIDI_RN2
    return (int)hipGetLastError();
}

template <typename T, int MODE>
static int norm_launch(const NormParams& p, hipStream_t st) {
    const int nchunk = p.H / 8;
    const int grid = (p.rows + 3) / 4;
    if (nchunk <= 64) hipLaunchKernelGGL((norm_kernel<T, MODE, 1>), dim3(grid), dim3(256), 0, st, p);
    else if (nchunk <= 192) hipLaunchKernelGGL((norm_kernel<T, MODE, 3>), dim3(grid), dim3(256), 0, st, p);
    else if (nchunk <= 512) hipLaunchKernelGGL((norm_kernel<T, MODE, 8>), dim3(grid), dim3(256), 0, st, p);
    else return VIDI_ERR_SHAPE;
    return (int)hipGetLastError();
}

template <typename T>
static int norm_mode(const NormParams& p, int mode, hipStream_t st) {
    switch (mode) {
        case NORM_GEMMA: return norm_launch<T, NORM_GEMMA>(p, st);
        case NORM_GEMMA_ADD: return norm_launch<T, NORM_GEMMA_ADD>(p, st);
        case NORM_MM: return norm_launch<T, NORM_MM>(p, st);
        case NORM_MM_NOW: return norm_launch<T, NORM_MM_NOW>(p, st);
        case NORM_LLM: return norm_launch<T, NORM_LLM>(p, st);
        case NORM_LAYER: return norm_launch<T, NORM_LAYER>(p, st);
        default: return VIDI_ERR_ARG;
    }
}

int vidi_norm_dispatch(const NormParams& p, int mode, int dtype, hipStream_t st) {
    if (p.rows <= 0 || p.H <= 0 || p.H % 8) return VIDI_ERR_SHAPE;
    if ((p.ldx % 8 && !p.XF32) || (p.ldy % 8)) return VIDI_ERR_ALIGN;
    if (mode == NORM_GEMMA_ADD && (!p.Res || p.ldr % 8)) return VIDI_ERR_ARG;
    if (mode != NORM_MM_NOW && !p.Wt) return VIDI_ERR_ARG;
    if (mode == NORM_LAYER && !p.Bias) return VIDI_ERR_ARG;
    if (dtype == VIDI_DT_BF16) return norm_mode<BF16>(p, mode, st);
    if (dtype == VIDI_DT_F16) return norm_mode<F16>(p, mode, st);
    return VIDI_ERR_DTYPE;
}
