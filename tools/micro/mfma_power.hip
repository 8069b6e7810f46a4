// Micro-benchmark (diagnostic, not part of the library): sustained MFMA rate of the two bf16 shapes under the chip's
// power limit, register operands only (no LDS / global traffic in the loop), random vs zero data.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MI>
__global__ __launch_bounds__(512) void kern(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = threadIdx.x;
    float s = 0.f;
    if constexpr (MI == 32) {
        bf16x8 a[4], b[2];
        for (int i = 0; i < 4; ++i) a[i] = in[(i * 512 + t) & 4095];
        for (int i = 0; i < 2; ++i) b[i] = in[((i + 4) * 512 + t) & 4095];
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        bf16x8 a[8], b[4];
        for (int i = 0; i < 8; ++i) a[i] = in[(i * 512 + t) & 4095];
        for (int i = 0; i < 4; ++i) b[i] = in[((i + 8) * 512 + t) & 4095];
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    out[blockIdx.x * 512 + t] = s;
}

int main() {
    const int blocks = 256 * 8, iters = 20000;
    std::vector<unsigned short> h(4096 * 8);
    bf16x8* din; float* dout;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, blocks * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (auto& v : h) {               // bf16 bits: random sign/mantissa, exponent around 2^-6 ; or zeros
            const unsigned r = rand();
            v = data ? (unsigned short)(((r & 1) << 15) | ((120 + (r >> 1) % 4) << 7) | ((r >> 8) & 127)) : 0;
        }
        hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int mi = 0; mi < 2; ++mi) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mi == 0) hipLaunchKernelGGL(kern<32>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
                else hipLaunchKernelGGL(kern<16>, dim3(blocks), dim3(512), 0, 0, din, dout, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flop = (double)blocks * 8 * iters * (mi == 0 ? 8 * 32768.0 : 32 * 16384.0);
                printf("data=%s MI=%s rep%d: %.2f ms  %.0f TFLOP/s\n", data ? "random" : "zero", mi == 0 ? "32x32x16" : "16x16x32", rep, ms, flop / ms / 1e9);
            }
        }
    }
    return 0;
}
