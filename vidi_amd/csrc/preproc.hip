// GPU preprocessing for the Vidi hot path: what the reference does per video on the host CPU —
//   frames: PIL `Image.resize((S, S), BICUBIC)` + SigLIP rescale/normalise      (Vidi1.5_9B/vidi/dataset/img_utils.py:181-185)
//   audio : WhisperFeatureExtractor log-mel over 30-s windows                   (Vidi1.5_9B/vidi/dataset/vid_utils.py:53-64)
// as HBM-bound gfx950 kernels.  The frame path is BIT-EXACT with Pillow's 8-bit fixed-point resampler (integer
// accumulators, PRECISION_BITS = 22, the host supplies Pillow's coefficient tables) and with the processor's float
// arithmetic (256-entry per-channel table built on the host with that arithmetic).  The audio path computes the STFT as a
// windowed-DFT contraction (vidi_gemm_f32 over overlapping row views of the reflect-padded waveform); the kernels here
// are the data movement around it.
#include "common.h"
#include "../../include/vidi_hip.h"

#define PRECISION_BITS 22

namespace {

__device__ __forceinline__ unsigned clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return (unsigned)min(max(v, 0), 255);
}

// ---- pass 1, horizontal: in [rows = T*H0][W0][3] u8 -> tmp [rows][pitch] u8 (OW pixels x 3 bytes) --------------------------
// One block = RB consecutive source rows, staged in LDS with aligned dword loads (rows start at arbitrary byte offsets).
// A thread owns one output pixel column: it keeps that column's <= KMAX coefficients in registers (zero beyond the tap count,
// so no predication) and applies them to all RB rows.  The 3*KMAX source bytes of a pixel are fetched as aligned LDS dwords
// and re-aligned with v_alignbyte into a byte stream whose tap/channel positions are compile-time constants.  Results go to
// an LDS image of the output rows, stored as coalesced dwords.
template <int KMAX>
__global__ __launch_bounds__(512) void resize_h_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk,
                                                          long long rows, int W0, int OW, int pitch, int ksize, int RB,
                                                          int lds_row, long long in_bytes) {
    constexpr int ND = (3 * KMAX + 3) / 4;                              // dwords of the re-aligned byte stream
    extern __shared__ unsigned s_mem_h[];
    uint8_t* s_src = (uint8_t*)s_mem_h;                                // [RB][lds_row]   (lds_row % 4 == 0, >= 4*(ND+1) bytes of slack)
    uint8_t* s_dst = s_src + (size_t)RB * lds_row;                     // [RB][pitch]
    const int nthr = blockDim.x;
    const long long row0 = (long long)blockIdx.x * RB;
    const int nr = (int)min((long long)RB, rows - row0);
    const int nbytes = W0 * 3;
    for (int r = 0; r < nr; ++r) {
        const long long base = (row0 + r) * (long long)nbytes;
        const int mis = (int)((uintptr_t)(in + base) & 3);             // the row starts `mis` bytes after an aligned dword
        const uint8_t* abase = in + base - mis;
        const int ndw = (mis + nbytes + 3) >> 2;
        const long long last_full = (in_bytes - (base - mis)) >> 2;    // dwords lying wholly inside the buffer
        unsigned* d = (unsigned*)(s_src + (size_t)r * lds_row);
        for (int i = threadIdx.x; i < ndw; i += nthr) {
            unsigned v;
            if (i < last_full) v = ((const unsigned*)abase)[i];
            else {                                                      // last dword of the whole buffer: byte loads
                v = 0;
                for (int b = 0; b < 4; ++b) {
                    const long long off = base - mis + 4ll * i + b;
                    if (off < in_bytes) v |= (unsigned)in[off] << (8 * b);
                }
            }
            d[i] = v;
        }
    }
    for (int i = threadIdx.x; i < RB * pitch / 4; i += nthr) ((unsigned*)s_dst)[i] = 0;
    __syncthreads();
    for (int x = threadIdx.x; x < OW; x += nthr) {
        const int x0 = bounds[2 * x], n = bounds[2 * x + 1];
        int k[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) k[j] = j < n ? kk[(size_t)x * ksize + j] : 0;
        for (int r = 0; r < nr; ++r) {
            const long long base = (row0 + r) * (long long)nbytes;
            const int off = (int)((uintptr_t)(in + base) & 3) + x0 * 3;     // byte offset of the first tap inside the LDS row
            const unsigned* d = (const unsigned*)(s_src + (size_t)r * lds_row) + (off >> 2);
            const unsigned sh = off & 3;
            unsigned w[ND + 1];
#pragma unroll
            for (int i = 0; i <= ND; ++i) w[i] = d[i];
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const unsigned v = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);   // stream bytes 4i .. 4i+3
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int pos = 4 * i + b, j = pos / 3, c = pos - 3 * j;
                    if (j < KMAX) {
                        const int px = (int)((v >> (8 * b)) & 255u) * k[j];
                        if (c == 0) a0 += px; else if (c == 1) a1 += px; else a2 += px;
                    }
                }
            }
            uint8_t* o = s_dst + (size_t)r * pitch + 3 * x;
            o[0] = (uint8_t)clip8(a0); o[1] = (uint8_t)clip8(a1); o[2] = (uint8_t)clip8(a2);
        }
    }
    __syncthreads();
    unsigned* dst = (unsigned*)(tmp + row0 * (long long)pitch);
    for (int i = threadIdx.x; i < nr * pitch / 4; i += nthr) dst[i] = ((const unsigned*)s_dst)[i];
}

// ---- pass 2, vertical + normalise + HWC->CHW: tmp [T][H0][pitch] u8 -> out [T][3][OH][OW] (2- or 4-byte elements) -------
// One block = RB consecutive output rows of one frame.  An item = (row, dword of the interleaved row): it accumulates its 4
// bytes over KMAX taps (coefficients zero beyond the row's tap count, source row clamped — KMAX independent coalesced loads in
// flight; neighbouring output rows share source rows through L1/L2), maps the results through the per-channel value table
// and drops them into an LDS image of the three planes, which is stored coalesced.
template <typename E, int RB, int KMAX>
__global__ __launch_bounds__(256) void resize_v_u8_norm_kernel(const uint8_t* __restrict__ tmp, E* __restrict__ out,
                                                               const int* __restrict__ bounds, const int* __restrict__ kk,
                                                               const E* __restrict__ lut, int H0, int OW, int OH, int pitch, int ksize) {
    extern __shared__ unsigned s_mem[];
    int* s_b = (int*)s_mem;                     // [RB][2]
    int* s_k = s_b + 2 * RB;                    // [RB][KMAX]
    E* s_lut = (E*)(s_k + RB * KMAX);           // [3][256]
    E* s_out = s_lut + 768;                     // [RB][3][OW]
    const int bpf = (OH + RB - 1) / RB;         // blocks per frame
    const int t = blockIdx.x / bpf, y_base = (blockIdx.x - t * bpf) * RB;
    const int nr = min(RB, OH - y_base);
    for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
    for (int i = threadIdx.x; i < nr * 2; i += 256) s_b[i] = bounds[2 * y_base + i];
    for (int i = threadIdx.x; i < nr * KMAX; i += 256) {
        const int r = i / KMAX, j = i - r * KMAX;
        s_k[i] = j < bounds[2 * (y_base + r) + 1] ? kk[(size_t)(y_base + r) * ksize + j] : 0;
    }
    __syncthreads();
    const int row_dw = pitch / 4;
    const unsigned* frame = (const unsigned*)(tmp + (size_t)t * H0 * pitch);
    for (int it = threadIdx.x; it < nr * row_dw; it += 256) {
        const int r = it / row_dw, i = it - r * row_dw;
        const int y0 = s_b[2 * r];
        const int* k = s_k + r * KMAX;
        unsigned v[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) v[j] = frame[(size_t)min(y0 + j, H0 - 1) * row_dw + i];
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const int kj = k[j];
            a0 += (int)(v[j] & 255u) * kj; a1 += (int)((v[j] >> 8) & 255u) * kj;
            a2 += (int)((v[j] >> 16) & 255u) * kj; a3 += (int)(v[j] >> 24) * kj;
        }
        const int acc[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int idx = 4 * i + b, x = idx / 3, c = idx - 3 * x;
            if (x < OW) s_out[(r * 3 + c) * OW + x] = s_lut[c * 256 + clip8(acc[b])];
        }
    }
    __syncthreads();
    if (sizeof(E) == 2 && (OW & 1) == 0) {                              // two 16-bit elements per store
        const int ow2 = OW / 2;
        for (int i = threadIdx.x; i < nr * 3 * ow2; i += 256) {
            const int rc = i / ow2, x2 = i - rc * ow2, r = rc / 3, c = rc - 3 * r;
            ((unsigned*)(out + (((size_t)t * 3 + c) * OH + y_base + r) * OW))[x2] = ((const unsigned*)s_out)[i];
        }
    } else {
        for (int i = threadIdx.x; i < nr * 3 * OW; i += 256) {
            const int rc = i / OW, x = i - rc * OW, r = rc / 3, c = rc - 3 * r;
            out[(((size_t)t * 3 + c) * OH + y_base + r) * OW + x] = s_out[i];
        }
    }
}

// ---- audio: reflect padding of each zero-padded window (torch.stft center=True, pad_mode="reflect") -------------------
//   wave [C][n] f32 -> out [C][stride] f32: out[c][i] = wave[c][reflect(i - pad)] for i < n + 2 pad, 0 beyond
__global__ void reflect_pad_f32_kernel(const float* __restrict__ wave, float* __restrict__ out, int C, int n, int pad, int stride) {
    const size_t total = (size_t)C * stride;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % stride);
        const size_t c = idx / stride;
        float v = 0.f;
        if (i < n + 2 * pad) {
            int j = i - pad;
            if (j < 0) j = -j;
            else if (j >= n) j = 2 * (n - 1) - j;
            v = wave[c * n + j];
        }
        out[idx] = v;
    }
}

// ---- |X|^2: Y [M][ldy] = (re[0..nf) | im[nf..2nf)) -> P [M][ldp], columns >= nf zeroed (K padding of the mel GEMM) -------
__global__ void power_spectrum_kernel(const float* __restrict__ Y, float* __restrict__ P, long long M, int nf, int ldy, int ldp) {
    const size_t total = (size_t)M * ldp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % ldp);
        const size_t m = idx / ldp;
        float v = 0.f;
        if (k < nf) {
            const float re = Y[m * ldy + k], im = Y[m * ldy + nf + k];
            v = re * re + im * im;
        }
        P[idx] = v;
    }
}

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
    else atomicMin((unsigned*)addr, __float_as_uint(v));
}

// ---- log10(max(mel, 1e-10)) in place + per-window maximum over the valid frames ----------------------------------------
//   mel [C][R][nmel] f32 (R rows per window, first F valid); cmax [C] must be pre-filled with -inf
__global__ __launch_bounds__(256) void logmel_log_max_kernel(float* __restrict__ mel, float* __restrict__ cmax, int R, int F, int nmel) {
    const int c = blockIdx.y;
    const size_t per = (size_t)F * nmel;
    float* base = mel + (size_t)c * R * nmel;
    float m = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        const float v = log10f(fmaxf(base[i], 1e-10f));
        base[i] = v;
        m = fmaxf(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomic_max_f32(cmax + c, fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
}

// ---- floor at (max - 8), (x + 4) / 4, transpose [F][nmel] -> out [C][nmel][F] in the output element type -----------------
template <typename E, typename CVT>
__global__ __launch_bounds__(256) void logmel_finish_kernel(const float* __restrict__ mel, const float* __restrict__ cmax,
                                                            E* __restrict__ out, int R, int F, int nmel, CVT cvt) {
    __shared__ float tile[64][65];
    const int c = blockIdx.z, f0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
    const float floor_v = cmax[c] - 8.0f;
    const float* base = mel + (size_t)c * R * nmel;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int fr = i >> 6, ml = i & 63;
        float v = 0.f;
        if (f0 + fr < F && m0 + ml < nmel) v = (fmaxf(base[(size_t)(f0 + fr) * nmel + m0 + ml], floor_v) + 4.0f) / 4.0f;
        tile[fr][ml] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int ml = i >> 6, fr = i & 63;
        if (f0 + fr < F && m0 + ml < nmel) out[((size_t)c * nmel + m0 + ml) * F + f0 + fr] = cvt(tile[fr][ml]);
    }
}

struct CvtF32 { __device__ float operator()(float v) const { return v; } };
struct CvtBF16 { __device__ u16 operator()(float v) const { return f32_to_bf16(v); } };
struct CvtF16 { __device__ u16 operator()(float v) const { return f32_to_f16(v); } };

inline int grid_1d(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 65536 ? 65536 : (g ? g : 1));
}

}  // namespace

extern "C" {

int vidi_resize_h_u8(const void* in, void* tmp, const int* bounds, const int* kk, long long rows, int W0, int OW, int pitch,
                     int ksize, void* stream) {
    (void)hipGetLastError();
    if (!in || !tmp || !bounds || !kk) return VIDI_ERR_ARG;
    if (((uintptr_t)in & 3) || ((uintptr_t)tmp & 3)) return VIDI_ERR_ALIGN;
    if (rows <= 0 || W0 <= 0 || OW <= 0 || ksize <= 0 || ksize > 48 || pitch < OW * 3 || pitch % 4) return VIDI_ERR_SHAPE;
    const int kmax = ksize <= 8 ? 8 : ksize <= 12 ? 12 : ksize <= 16 ? 16 : ksize <= 24 ? 24 : ksize <= 32 ? 32 : 48;
    const int lds_row = (W0 * 3 + 3 + 3) / 4 * 4 + 4 * ((3 * kmax + 3) / 4 + 2);   // row + slack for the fixed-length tap fetch
    int RB = (30 * 1024) / (lds_row + pitch);                            // rows per block: ~30 KB of LDS (5 blocks per CU), <= 8
    if (RB > 8) RB = 8;
    if (RB < 1) RB = 1;
    const int lds = RB * (lds_row + pitch);
    if (lds > 64 * 1024) return VIDI_ERR_SHAPE;
    const long long nblk = (rows + RB - 1) / RB;
    if (nblk > 0x7fffffffll) return VIDI_ERR_SHAPE;
    const long long in_bytes = rows * (long long)W0 * 3;
    int nthr = (OW + 63) / 64 * 64;                                      // one output column per thread when it fits
    if (nthr > 512) nthr = 512;
    hipStream_t st = (hipStream_t)stream;
#define VIDI_LAUNCH_H(KM)                                                                                                     \
    hipLaunchKernelGGL(resize_h_u8_kernel<KM>, dim3((unsigned)nblk), dim3(nthr), lds, st, (const uint8_t*)in, (uint8_t*)tmp,   \
                       bounds, kk, rows, W0, OW, pitch, ksize, RB, lds_row, in_bytes)
    switch (kmax) {
        case 8: VIDI_LAUNCH_H(8); break;
        case 12: VIDI_LAUNCH_H(12); break;
        case 16: VIDI_LAUNCH_H(16); break;
        case 24: VIDI_LAUNCH_H(24); break;
        case 32: VIDI_LAUNCH_H(32); break;
        default: VIDI_LAUNCH_H(48); break;
    }
#undef VIDI_LAUNCH_H
    return (int)hipGetLastError();
}

int vidi_resize_v_u8_norm(const void* tmp, void* out, const int* bounds, const int* kk, const void* lut, int T, int H0, int OW, int OH,
                          int pitch, int ksize, int out_elem_bytes, void* stream) {
    (void)hipGetLastError();
    if (!tmp || !out || !bounds || !kk || !lut) return VIDI_ERR_ARG;
    if (T <= 0 || H0 <= 0 || OW <= 0 || OH <= 0 || ksize <= 0 || ksize > 48 || pitch < OW * 3 || pitch % 4) return VIDI_ERR_SHAPE;
    if (out_elem_bytes != 2 && out_elem_bytes != 4) return VIDI_ERR_DTYPE;
    constexpr int RB = 8;
    const int kmax = ksize <= 8 ? 8 : ksize <= 12 ? 12 : ksize <= 16 ? 16 : ksize <= 24 ? 24 : ksize <= 32 ? 32 : 48;
    const long long nblk = (long long)T * ((OH + RB - 1) / RB);
    if (nblk > 0x7fffffffll) return VIDI_ERR_SHAPE;
    const dim3 grid((unsigned)nblk);
    const size_t lds = (size_t)(2 * RB + RB * kmax) * 4 + (size_t)(768 + RB * 3 * OW) * out_elem_bytes;
    if (lds > 64 * 1024) return VIDI_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
#define VIDI_LAUNCH_V(KM)                                                                                                              \
    do {                                                                                                                               \
        if (out_elem_bytes == 2)                                                                                                       \
            hipLaunchKernelGGL((resize_v_u8_norm_kernel<u16, RB, KM>), grid, dim3(256), lds, st, (const uint8_t*)tmp, (u16*)out, bounds, \
                               kk, (const u16*)lut, H0, OW, OH, pitch, ksize);                                                         \
        else                                                                                                                           \
            hipLaunchKernelGGL((resize_v_u8_norm_kernel<float, RB, KM>), grid, dim3(256), lds, st, (const uint8_t*)tmp, (float*)out,    \
                               bounds, kk, (const float*)lut, H0, OW, OH, pitch, ksize);                                               \
    } while (0)
    switch (kmax) {
        case 8: VIDI_LAUNCH_V(8); break;
        case 12: VIDI_LAUNCH_V(12); break;
        case 16: VIDI_LAUNCH_V(16); break;
        case 24: VIDI_LAUNCH_V(24); break;
        case 32: VIDI_LAUNCH_V(32); break;
        default: VIDI_LAUNCH_V(48); break;
    }
#undef VIDI_LAUNCH_V
    return (int)hipGetLastError();
}

int vidi_reflect_pad_f32(const float* wave, float* out, int C, int n, int pad, int stride, void* stream) {
    (void)hipGetLastError();
    if (!wave || !out) return VIDI_ERR_ARG;
    if (C <= 0 || n <= pad || pad < 0 || stride < n + 2 * pad) return VIDI_ERR_SHAPE;
    hipLaunchKernelGGL(reflect_pad_f32_kernel, dim3(grid_1d((size_t)C * stride)), dim3(256), 0, (hipStream_t)stream, wave, out, C, n, pad, stride);
    return (int)hipGetLastError();
}

int vidi_power_spectrum_f32(const float* Y, float* P, long long M, int nf, int ldy, int ldp, void* stream) {
    (void)hipGetLastError();
    if (!Y || !P) return VIDI_ERR_ARG;
    if (M <= 0 || nf <= 0 || ldy < 2 * nf || ldp < nf) return VIDI_ERR_SHAPE;
    hipLaunchKernelGGL(power_spectrum_kernel, dim3(grid_1d((size_t)M * ldp)), dim3(256), 0, (hipStream_t)stream, Y, P, M, nf, ldy, ldp);
    return (int)hipGetLastError();
}

int vidi_logmel_finish(float* mel, float* cmax, void* out, int C, int R, int F, int nmel, int out_dtype, void* stream) {
    (void)hipGetLastError();
    if (!mel || !cmax || !out) return VIDI_ERR_ARG;
    if (C <= 0 || R < F || F <= 0 || nmel <= 0 || C > 65535) return VIDI_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const unsigned ninf = 0xff800000u;                                   // -inf: identity of the running maximum
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)cmax, (int)ninf, (size_t)C, st);
    if (e != hipSuccess) return (int)e;
    const int bx = (int)(((size_t)F * nmel + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(logmel_log_max_kernel, dim3(bx, C), dim3(256), 0, st, mel, cmax, R, F, nmel);
    const dim3 grid((F + 63) / 64, (nmel + 63) / 64, C);
    if (out_dtype == VIDI_DT_F32) hipLaunchKernelGGL((logmel_finish_kernel<float, CvtF32>), grid, dim3(256), 0, st, mel, cmax, (float*)out, R, F, nmel, CvtF32());
    else if (out_dtype == VIDI_DT_BF16) hipLaunchKernelGGL((logmel_finish_kernel<u16, CvtBF16>), grid, dim3(256), 0, st, mel, cmax, (u16*)out, R, F, nmel, CvtBF16());
    else if (out_dtype == VIDI_DT_F16) hipLaunchKernelGGL((logmel_finish_kernel<u16, CvtF16>), grid, dim3(256), 0, st, mel, cmax, (u16*)out, R, F, nmel, CvtF16());
    else return VIDI_ERR_DTYPE;
    return (int)hipGetLastError();
}

}  // extern "C"
