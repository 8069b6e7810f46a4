// persistent 4-wave GEMM, MODE_PLAIN instantiations, fp16 (see gemm_w4_launch.h)
#include "gemm_w4_launch.h"
int vidi_w4_plain_f16(const GemmParams& p, int batch, int repkv, hipStream_t st) { return w4_plain_dispatch<F16>(p, batch, repkv, st); }
