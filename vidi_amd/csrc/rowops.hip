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
__global__ __launch_bounds__(256) void resid_norm2_kernel(const u16* __restrict__ A, const u16* __restrict__ B, const u16* __restrict__ C,
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
    float x[MAXC][8];
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            unpack8<T>(ra[c], x[c]);
            float z[8];
            if (B) {
                unpack8<T>(PRELOAD ? rb[c] : *(const u32x4*)(B + row * ld + ch * 8), z);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[c][e] = rnd<T>(x[c][e] + z[e]);
            }
            if (C) {
                unpack8<T>(PRELOAD ? rc[c] : *(const u32x4*)(C + row * ld + ch * 8), z);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[c][e] = rnd<T>(x[c][e] + z[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s2 += x[c][e] * x[c][e];
        }
    }
    const float rs1 = rsqrtf(wave_sum(s2) / H + eps);
    float t2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float w[8], r[8];
            unpack8<T>(PRELOAD ? rw1[c] : *(const u32x4*)(W1 + ch * 8), w);
            unpack8<T>(PRELOAD ? rr[c] : *(const u32x4*)(Res + row * ld + ch * 8), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c][e] = rnd<T>(r[e] + rnd<T>(x[c][e] * rs1 * (1.0f + w[e])));
            *(u32x4*)(Y1 + row * ld + ch * 8) = pack8<T>(x[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) t2 += x[c][e] * x[c][e];
        }
    }
    const float rs2 = rsqrtf(wave_sum(t2) / H + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float w[8], y[8];
            unpack8<T>(PRELOAD ? rw2[c] : *(const u32x4*)(W2 + ch * 8), w);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = x[c][e] * rs2 * (1.0f + w[e]);
            *(u32x4*)(Y2 + row * ld + ch * 8) = pack8<T>(y);
        }
    }
}

int vidi_resid_norm2_dispatch(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1,
                              void* Y2, int rows, int H, long long ld, float eps, int dtype, hipStream_t st) {
    if (rows <= 0 || H <= 0 || H % 8 || ld % 8) return VIDI_ERR_SHAPE;
    const int nchunk = H / 8, grid = (rows + 3) / 4;
#define VIDI_RN2_(TT, MC, PL)                                                                                                     \
    hipLaunchKernelGGL((resid_norm2_kernel<TT, MC, PL>), dim3(grid), dim3(256), 0, st, (const u16*)A, (const u16*)B, (const u16*)C, \
                       (const u16*)Res, (const u16*)W1, (const u16*)W2, (u16*)Y1, (u16*)Y2, rows, H, ld, eps)
#define VIDI_RN2(TT, MC) do { if (rows <= 64) VIDI_RN2_(TT, MC, true); else VIDI_RN2_(TT, MC, false); } while (0)
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
