// see gemm_skinny.h
#include "gemm_skinny.h"
#include "gemm_skinny_api.h"
#include "../../include/vidi_hip.h"

int vidi_gemm_skinny_ksplit(int N, int K) { return skinny_ksplit(N, K); }

template <typename T>
static int launch_skinny(const u16* X, const u16* W, const u16* bias, u16* Y, float* part, int M, int N, int K, int ldx, int ldw, int ldy, hipStream_t st) {
    const int ksplit = skinny_ksplit(N, K);
    if (ksplit == 0 || M < 1 || M > 128) return -100;
    const int ksteps = K / 64 / ksplit;
    const dim3 grid(N / 64, ksplit);
    const int mt = (M + 15) / 16;
#define VIDI_SK(MT_) hipLaunchKernelGGL((skinny_kernel<T, MT_>), grid, dim3(256), 4 * ((MT_ * 16 + 31) / 32) * 32 * 128, st, X, W, part, M, N, K, ldx, ldw, ksteps)
    switch (mt) {
        case 1: VIDI_SK(1); break; case 2: VIDI_SK(2); break; case 3: VIDI_SK(3); break; case 4: VIDI_SK(4); break;
        case 5: VIDI_SK(5); break; case 6: VIDI_SK(6); break; case 7: VIDI_SK(7); break; default: VIDI_SK(8); break;
    }
#undef VIDI_SK
    hipLaunchKernelGGL(skinny_reduce_kernel<T>, dim3((M * (N / 4) + 255) / 256), dim3(256), 0, st, part, bias, Y, M, N, ldy, ksplit);
    return (int)hipGetLastError();
}

int vidi_gemm_skinny_dispatch(const void* X, const void* W, const void* bias, void* Y, float* part, int M, int N, int K, int ldx, int ldw, int ldy,
                              int dtype, hipStream_t st) {
    if (dtype == VIDI_DT_BF16) return launch_skinny<BF16>((const u16*)X, (const u16*)W, (const u16*)bias, (u16*)Y, part, M, N, K, ldx, ldw, ldy, st);
    if (dtype == VIDI_DT_F16) return launch_skinny<F16>((const u16*)X, (const u16*)W, (const u16*)bias, (u16*)Y, part, M, N, K, ldx, ldw, ldy, st);
    return VIDI_ERR_DTYPE;
}
