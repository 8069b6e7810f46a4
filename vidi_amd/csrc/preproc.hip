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

// ---- pass 1, horizontal: in [rows = T*H0][W0][3] u8 -> tmp [rows][OW][3] u8 ---------------------------------------------
// One block per input row.  The row (W0*3 bytes, arbitrary byte alignment) is staged in LDS with aligned dword loads;
// each thread then produces 4 consecutive output bytes and stores one dword (output rows are `pitch` bytes apart).
__global__ __launch_bounds__(256) void resize_h_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk,
                                                          long long rows, int W0, int OW, int pitch, int ksize, long long in_bytes) {
    extern __shared__ unsigned s_row_u32[];
    uint8_t* s_row = (uint8_t*)s_row_u32;
    const long long row = blockIdx.x;
    const long long base = row * (long long)W0 * 3;
    const int nbytes = W0 * 3;
    const int mis = (int)((uintptr_t)(in + base) & 3);                 // bytes between the aligned-down address and the row
    const uint8_t* abase = in + base - mis;
    const int ndw = (mis + nbytes + 3) >> 2;
    const long long last_full = (in_bytes - (base - mis)) >> 2;        // dwords that lie wholly inside the buffer
    for (int i = threadIdx.x; i < ndw; i += 256) {
        unsigned v;
        if (i < last_full) v = ((const unsigned*)abase)[i];
        else {                                                          // tail dword of the whole buffer: byte loads
            v = 0;
            for (int b = 0; b < 4; ++b) {
                const long long off = base - mis + 4ll * i + b;
                if (off < in_bytes) v |= (unsigned)in[off] << (8 * b);
            }
        }
        s_row_u32[i] = v;
    }
    __syncthreads();
    const uint8_t* src = s_row + mis;
    const int nout4 = pitch / 4;                                        // tmp rows are `pitch` = round_up(OW*3, 4) bytes apart
    unsigned* dst = (unsigned*)(tmp + row * (long long)pitch);
    for (int i = threadIdx.x; i < nout4; i += 256) {
        unsigned packed = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int idx = 4 * i + b, x = idx / 3, c = idx - 3 * x;
            if (x >= OW) break;                                         // pitch padding bytes stay zero
            const int x0 = bounds[2 * x], n = bounds[2 * x + 1];
            const int* k = kk + (size_t)x * ksize;
            int acc = 1 << (PRECISION_BITS - 1);
            for (int j = 0; j < n; ++j) acc += (int)src[(x0 + j) * 3 + c] * k[j];
            packed |= clip8(acc) << (8 * b);
        }
        dst[i] = packed;
    }
}

// ---- pass 2, vertical + normalise + HWC->CHW: tmp [T][H0][OW][3] u8 -> out [T][3][OH][OW] (2- or 4-byte elements) -------
// One block per output row (t, y).  A thread owns 4 consecutive bytes of the interleaved row across all taps (dword loads,
// rows are `pitch` bytes apart), looks the results up in the per-channel value table and the block writes the three
// planes coalesced through LDS.
template <typename E>
__global__ __launch_bounds__(256) void resize_v_u8_norm_kernel(const uint8_t* __restrict__ tmp, E* __restrict__ out,
                                                               const int* __restrict__ bounds, const int* __restrict__ kk,
                                                               const E* __restrict__ lut, int H0, int OW, int OH, int pitch, int ksize) {
    extern __shared__ unsigned s_mem[];
    E* s_lut = (E*)s_mem;                       // [3][256]
    E* s_out = s_lut + 768;                     // [3][OW]
    const int t = blockIdx.x / OH, y = blockIdx.x - t * OH;
    for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
    const int y0 = bounds[2 * y], n = bounds[2 * y + 1];
    const int* k = kk + (size_t)y * ksize;
    const int row_dw = pitch / 4;
    const unsigned* src = (const unsigned*)(tmp + ((size_t)t * H0 + y0) * pitch);
    __syncthreads();
    for (int i = threadIdx.x; i < row_dw; i += 256) {
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
        for (int j = 0; j < n; ++j) {
            const unsigned v = src[(size_t)j * row_dw + i];
            const int kj = k[j];
            a0 += (int)(v & 255u) * kj; a1 += (int)((v >> 8) & 255u) * kj;
            a2 += (int)((v >> 16) & 255u) * kj; a3 += (int)(v >> 24) * kj;
        }
        const int acc[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int idx = 4 * i + b, x = idx / 3, c = idx - 3 * x;
            if (x < OW) s_out[c * OW + x] = s_lut[c * 256 + clip8(acc[b])];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * OW; i += 256) {
        const int c = i / OW, x = i - c * OW;
        out[(((size_t)t * 3 + c) * OH + y) * OW + x] = s_out[i];
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
    if (rows <= 0 || W0 <= 0 || OW <= 0 || ksize <= 0 || pitch < OW * 3 || pitch % 4 || rows > 0x7fffffffll) return VIDI_ERR_SHAPE;
    const int lds = ((W0 * 3 + 3 + 3) / 4 + 1) * 4;
    if (lds > 64 * 1024) return VIDI_ERR_SHAPE;
    hipLaunchKernelGGL(resize_h_u8_kernel, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, (const uint8_t*)in, (uint8_t*)tmp,
                       bounds, kk, rows, W0, OW, pitch, ksize, rows * (long long)W0 * 3);
    return (int)hipGetLastError();
}

int vidi_resize_v_u8_norm(const void* tmp, void* out, const int* bounds, const int* kk, const void* lut, int T, int H0, int OW, int OH,
                          int pitch, int ksize, int out_elem_bytes, void* stream) {
    (void)hipGetLastError();
    if (!tmp || !out || !bounds || !kk || !lut) return VIDI_ERR_ARG;
    if (T <= 0 || H0 <= 0 || OW <= 0 || OH <= 0 || pitch < OW * 3 || pitch % 4 || (long long)T * OH > 0x7fffffffll) return VIDI_ERR_SHAPE;
    const dim3 grid((unsigned)((long long)T * OH));
    if (out_elem_bytes == 2) {
        hipLaunchKernelGGL(resize_v_u8_norm_kernel<u16>, grid, dim3(256), (768 + 3 * OW) * 2, (hipStream_t)stream, (const uint8_t*)tmp,
                           (u16*)out, bounds, kk, (const u16*)lut, H0, OW, OH, pitch, ksize);
    } else if (out_elem_bytes == 4) {
        hipLaunchKernelGGL(resize_v_u8_norm_kernel<float>, grid, dim3(256), (768 + 3 * OW) * 4, (hipStream_t)stream, (const uint8_t*)tmp,
                           (float*)out, bounds, kk, (const float*)lut, H0, OW, OH, pitch, ksize);
    } else return VIDI_ERR_DTYPE;
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
