// host entry points of gemm_skinny.hip (a header of their own: kernels.h belongs to the digest that keys the GEMM family's PMC traffic
// record, which these kernels are not part of)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
// K slices the kernel would run for (N, K); 0: the shape is not taken (N % 64, K % 256)
int vidi_gemm_skinny_ksplit(int N, int K);
// Y[M, N] = X[M, K] W[N, K]^T (+ bias), 1 <= M <= 128; part: fp32 workspace of ksplit * M * N elements.  -100: shape not taken
int vidi_gemm_skinny_dispatch(const void* X, const void* W, const void* bias, void* Y, float* part, int M, int N, int K, int ldx, int ldw, int ldy,
                              int dtype, hipStream_t st);
