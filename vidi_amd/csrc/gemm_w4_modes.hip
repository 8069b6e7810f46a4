// persistent 4-wave GEMM, fused-epilogue modes (GeGLU / QKV + V-transpose / K,V cache tiles), both dtypes (see gemm_w4_launch.h)
#include "gemm_w4_launch.h"
template <typename T>
static int modes(const GemmParams& p, int batch, int mode, hipStream_t st) {
    switch (mode) {
        case MODE_GEGLU: return launch_w4<T, MODE_GEGLU, false, Epi<false, ACT_NONE, 0>>(p, batch, st);
        case MODE_QKV_VT: if (p.ln_stats) return vidi_w4_lnf(p, batch, MODE_QKV_VT, T::id, st);
                          return p.bias ? launch_w4<T, MODE_QKV_VT, false, Epi<true, ACT_NONE, 0>>(p, batch, st)
                                        : launch_w4<T, MODE_QKV_VT, false, Epi<false, ACT_NONE, 0>>(p, batch, st);
        case MODE_KV_CACHE: return launch_w4<T, MODE_KV_CACHE, false, Epi<false, ACT_NONE, 0>>(p, batch, st);
        default: return VIDI_W4_UNSUPPORTED;
    }
}
int vidi_w4_modes(const GemmParams& p, int batch, int mode, int dtype, hipStream_t st) {
    if (dtype == VIDI_DT_BF16) return modes<BF16>(p, batch, mode, st);
    if (dtype == VIDI_DT_F16) return modes<F16>(p, batch, mode, st);
    return VIDI_ERR_DTYPE;
}
