// Box-speed reference kernels (diagnostic; not on the Vidi path).  bench.py runs them next to the timed steps and prints their rates in the
// JSON line so that two records taken on different boxes of the pool (whose sustained clocks differ by 2-3 %) can be divided by a workload
// that never changes: this file is FROZEN — editing it breaks the comparability of every record that carries its figures.
//   vidi_probe_mfma     : register-operand bf16 MFMA 16x16x32 loop on every SIMD (no LDS / memory traffic in the loop): the matrix pipe's
//                         sustained rate under the power limit with random operands
//   vidi_probe_hbm_read : one non-temporal 16-byte-per-lane sweep over a caller's buffer: the HBM read rate
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
typedef __attribute__((ext_vector_type(4))) float probe_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int probe_u32x4;

__global__ __launch_bounds__(512) void probe_mfma_kernel(const probe_bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = threadIdx.x;
    probe_bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = in[(i * 512 + t) & 4095];
    for (int i = 0; i < 4; ++i) b[i] = in[((i + 8) * 512 + t) & 4095];
    probe_f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * 512 + t] = s;
}

__global__ __launch_bounds__(256) void probe_hbm_read_kernel(const probe_u32x4* __restrict__ in, unsigned* __restrict__ out, long long n16) {
    probe_u32x4 x = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = (long long)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
        probe_u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = i + 256 * u;
            v[u] = __builtin_nontemporal_load(in + (j < n16 ? j : i));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) x ^= v[u];
    }
    out[blockIdx.x * 256 + threadIdx.x] = x[0] ^ x[1] ^ x[2] ^ x[3];
}

extern "C" {
// operands: 64 KB of bf16 values (4096 x 8); out: 2048 * 512 floats.  Work of one call: 2048 blocks x 8 waves x iters x 32 MFMAs x 16384 flop.
int vidi_probe_mfma(const void* operands, void* out, int iters, void* stream) {
    (void)hipGetLastError();
    if (!operands || !out || iters < 1) return -4;
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(2048), dim3(512), 0, (hipStream_t)stream, (const probe_bf16x8*)operands, (float*)out, iters);
    return (int)hipGetLastError();
}
// buf: `bytes` (a multiple of 16) to sweep once; out: 2048 * 256 uint32
int vidi_probe_hbm_read(const void* buf, void* out, long long bytes, void* stream) {
    (void)hipGetLastError();
    if (!buf || !out || bytes < 16 || (bytes & 15) || ((uintptr_t)buf & 15)) return -4;
    hipLaunchKernelGGL(probe_hbm_read_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const probe_u32x4*)buf, (unsigned*)out, bytes / 16);
    return (int)hipGetLastError();
}
}
