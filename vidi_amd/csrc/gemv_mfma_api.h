// host entry points of gemv_mfma.hip (a header of their own, like gemm_skinny_api.h: kernels.h belongs to the digest that keys the GEMM
// family's PMC traffic record, which these kernels are not part of)
#pragma once
#include <hip/hip_runtime.h>
// 1 when (M, N, K) is taken: 1 <= M <= 32 (gated pair: 16), N % 16 == 0 (gated: N = I features, I % 32 == 0), K % 64 == 0
int vidi_gemvm_fits(int M, int N, int K, int glu);
// Y[M, N] = X[M, K] W[N, K]^T, or with glu_act >= 0 the gated pair act(x Wg^T) * (x Wu^T) on the interleaved gate/up layout ([2I, K], N = I)
int vidi_gemv_mfma_dispatch(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int glu_act, int dtype,
                            hipStream_t st);
