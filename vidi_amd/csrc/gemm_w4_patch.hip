// persistent 4-wave GEMM with the patch-embedding loader (GemmParams::pe_*): SiglipVisionEmbeddings' Conv2d(3, H, kernel = stride = P)
// + bias + position table (TP siglip/modeling_siglip.py:124-130, 178) as ONE launch that reads the NCHW pixels itself — no im2col
// buffer (see gemm_w4.h PATCH, gemm_w4_launch.h)
#include "gemm_w4_launch.h"
int vidi_w4_patch(const GemmParams& p, int dtype, hipStream_t st) {
    if (!p.bias || !p.R || p.pe_S <= 0 || p.pe_P <= 0 || p.pe_P > 16 || p.pe_side <= 0 || p.rmod != p.pe_side * p.pe_side) return VIDI_ERR_ARG;
    if (p.K % 64 || p.K < 192 || p.K < 3 * p.pe_P * 16 || p.N % 32 || (p.M % p.rmod)) return VIDI_ERR_SHAPE;
    if ((unsigned long long)(p.M / p.rmod) * 3ull * p.pe_S * p.pe_S * 2ull > 0xffffffffull) return VIDI_ERR_SHAPE;      // one buffer descriptor over the frames
    if (dtype == VIDI_DT_BF16) return launch_w4<BF16, MODE_PLAIN, false, Epi<true, ACT_NONE, 2>, 1>(p, 1, st);
    if (dtype == VIDI_DT_F16) return launch_w4<F16, MODE_PLAIN, false, Epi<true, ACT_NONE, 2>, 1>(p, 1, st);
    return VIDI_ERR_DTYPE;
}

// Vidi-7B's learned Conv2DPool (Vidi_7B/model/mm_vision/pool.py:19-26): Conv2d(C, N, kernel k, stride 1, no bias) over the tower's token-major
// feature map, the window gathered by the loader (was vidi_im2col_nhwc: 88 MB of im2col rows per frame at k = 14, C = 1152)
int vidi_w4_window(const GemmParams& p, int dtype, hipStream_t st) {
    if (p.bias || p.R || p.pe_S <= 0 || (p.pe_S % 64) || p.pe_P <= 0 || p.pe_side < p.pe_P) return VIDI_ERR_ARG;
    const int oc = p.pe_side - p.pe_P + 1;
    if (p.K != p.pe_P * p.pe_P * p.pe_S || p.K < 192 || p.N % 32 || (p.M % (oc * oc))) return VIDI_ERR_SHAPE;
    if ((unsigned long long)(p.M / (oc * oc)) * p.pe_side * p.pe_side * p.pe_S * 2ull > 0xffffffffull) return VIDI_ERR_SHAPE;   // one descriptor over the frames
    if (dtype == VIDI_DT_BF16) return launch_w4<BF16, MODE_PLAIN, false, Epi<false, ACT_NONE, 0>, 2>(p, 1, st);
    if (dtype == VIDI_DT_F16) return launch_w4<F16, MODE_PLAIN, false, Epi<false, ACT_NONE, 0>, 2>(p, 1, st);
    return VIDI_ERR_DTYPE;
}
