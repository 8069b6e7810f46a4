// Bench-only (NOT part of libvidi_hip.so): tools/lab/gemm_skinny.h against the product's kernel for the text prompt's GEMM shapes
// (M = 39 rows; the decoder's q / kv / o / gate|up / down weights), random bf16 data; every result checked against a plain fp32
// dot-product kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vidi_amd/csrc tools/lab/skinny_lab.hip vidi_amd/csrc/gemm_skinny.hip -o tools/lab/skinny_lab
#include "gemm_tile.h"
#include "gemm_skinny.h"
#include "gemm_skinny_api.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(u16* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed ^ (unsigned)(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = f32_to_bf16(((h & 0xffffff) / 8388608.0f - 1.0f) * scale);
    }
}
// one wave per output element group: thread (m, n) plain loop — small M only
__global__ void naive_kernel(const u16* X, const u16* W, float* Y, int M, int N, int K) {
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = __builtin_fmaf(bf16_to_f32(X[(size_t)m * K + k]), bf16_to_f32(W[(size_t)n * K + k]), s);
    Y[(size_t)m * N + n] = s;
}
__global__ void diff_kernel(const u16* Y, const float* R, size_t n, float* out) {       // out[0] = max |y - r|, out[1] = max |r|
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float d = 0.f, a = 0.f;
    for (; i < n; i += stride) { d = fmaxf(d, fabsf(bf16_to_f32(Y[i]) - R[i])); a = fmaxf(a, fabsf(R[i])); }
    atomicMax((int*)out, __float_as_int(d));
    atomicMax((int*)out + 1, __float_as_int(a));
}

static int launch_tile(const GemmParams& p, hipStream_t st) {
    constexpr int BN = 128, BM = 128, STAGES = 2;
    constexpr int RING = STAGES * (BN + BM) * 64 * 2, CTILE = BM * (BN * 2 + 16), LDS = RING > CTILE ? RING : CTILE;
    auto kern = gemm_kernel<BF16, BN, BM, 2, 2, STAGES, MODE_PLAIN, false, 0, 32, LabNone>;
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); done = true; }
    const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
    hipLaunchKernelGGL(kern, dim3(tiles, 1), dim3(256), LDS, st, p);
    return (int)hipGetLastError();
}

struct Shape { const char* name; int N, K; };

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const std::vector<int> Ms = {39, 16, 64, 9, 117, 128};
    const std::vector<Shape> shapes = {{"q", 4096, 3584}, {"kv", 4096, 3584}, {"o", 3584, 4096}, {"gate_up", 28672, 3584}, {"down", 3584, 14336}, {"lm_head_slice", 32768, 3584}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* dmax; CK(hipMalloc(&dmax, 8));
    for (const Shape& sh : shapes) for (int M : Ms) {
        if (M != 39 && strcmp(sh.name, "o") && strcmp(sh.name, "down")) continue;
        u16 *X, *W, *Y; float *R, *P;
        const size_t nx = (size_t)128 * sh.K, nw = (size_t)sh.N * sh.K, ny = (size_t)M * sh.N;
        CK(hipMalloc(&X, nx * 2)); CK(hipMalloc(&W, nw * 2)); CK(hipMalloc(&Y, (size_t)128 * sh.N * 2)); CK(hipMalloc(&R, ny * 4)); CK(hipMalloc(&P, (size_t)16 * 128 * sh.N * 4));
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, X, nx, 0x1234u, 1.0f);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, nw, 0x9876u, 0.05f);
        hipLaunchKernelGGL(naive_kernel, dim3((sh.N + 255) / 256, M), dim3(256), 0, 0, X, W, R, M, sh.N, sh.K);
        CK(hipDeviceSynchronize());
        GemmParams p; memset(&p, 0, sizeof(p));
        p.X = X; p.W = W; p.Y = Y; p.M = M; p.N = sh.N; p.K = sh.K; p.ldx = sh.K; p.ldw = sh.K; p.ldy = sh.N; p.rmod = 0x7fffffff; p.group_m = 4;
        for (int which = 0; which < 2; ++which) {
            auto run = [&]() { return which == 0 ? launch_tile(p, 0) : vidi_gemm_skinny_dispatch(X, W, nullptr, Y, P, M, sh.N, sh.K, sh.K, sh.K, sh.N, VIDI_DT_BF16, 0); };
            CK(hipMemset(Y, 0xff, (size_t)128 * sh.N * 2));
            int rc = run();
            if (rc) { printf("{\"shape\": \"%s\", \"M\": %d, \"kernel\": \"%s\", \"rc\": %d}\n", sh.name, M, which ? "skinny" : "tile128", rc); continue; }
            CK(hipDeviceSynchronize());
            CK(hipMemset(dmax, 0, 8));
            hipLaunchKernelGGL(diff_kernel, dim3(256), dim3(256), 0, 0, Y, R, ny, dmax);
            float hm[2]; CK(hipMemcpy(hm, dmax, 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < 3; ++i) run();
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) run();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
            printf("{\"shape\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"kernel\": \"%s\", \"us\": %.1f, \"weight_TBps\": %.2f, \"max_err\": %.5f, \"max_ref\": %.3f, \"err_in_bf16_ulps_of_max\": %.2f, \"ksplit\": %d}\n",
                   sh.name, M, sh.N, sh.K, which ? "skinny" : "tile128", ms * 1e3, (double)nw * 2 / ms / 1e9, hm[0], hm[1], hm[0] / (hm[1] / 256.0f), which ? skinny_ksplit(sh.N, sh.K) : 0);
            fflush(stdout);
        }
        CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(Y)); CK(hipFree(R)); CK(hipFree(P));
    }
    return 0;
}
