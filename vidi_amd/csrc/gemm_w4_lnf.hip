// persistent 4-wave GEMM, LayerNorm-folded epilogues of the encoder towers (see GemmParams::ln_stats and gemm_w4_launch.h):
//   SigLIP / Whisper  LayerNorm -> q/k/v projection (MODE_QKV_VT)  and  LayerNorm -> fc1 + GELU (MODE_PLAIN)
#include "gemm_w4_launch.h"
template <typename T>
static int lnf(const GemmParams& p, int batch, int mode, hipStream_t st) {
    if (p.stat_part) {          // producer side: bias + residual (row m) epilogue that also emits the rows' per-strip partial sums
        if (mode != MODE_PLAIN || batch != 1 || !p.bias || !p.R || p.rmod < p.M || p.act != ACT_NONE || p.ln_stats) return VIDI_ERR_ARG;
        return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_NONE, 1, false, true>>(p, batch, st);
    }
    if (batch != 1 || p.R || !p.ln_s || !p.ln_c) return VIDI_ERR_ARG;
    if (mode == MODE_QKV_VT) return launch_w4<T, MODE_QKV_VT, false, Epi<true, ACT_NONE, 0, true>>(p, batch, st);
    if (mode != MODE_PLAIN) return VIDI_ERR_ARG;
    if (p.hm_seq) {
        if (p.act != ACT_NONE) return VIDI_ERR_ARG;
        const int rc = vidi_w4n_ln_heads(p, T::id, st);              // widths of the 288 x 224 tile geometry (SigLIP q | k | v: N = 3 456)
        if (rc != VIDI_W4_UNSUPPORTED) return rc;
        return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_NONE, 0, true, false, true>>(p, batch, st);      // q/k/v projection, head-major
    }
    switch (p.act) {
        case ACT_NONE: return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_NONE, 0, true>>(p, batch, st);
        case ACT_GELU_TANH: return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_GELU_TANH, 0, true>>(p, batch, st);       // SigLIP fc1
        case ACT_GELU_ERF: return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_GELU_ERF, 0, true>>(p, batch, st);         // Whisper fc1
        default: return VIDI_ERR_ARG;
    }
}
int vidi_w4_lnf(const GemmParams& p, int batch, int mode, int dtype, hipStream_t st) {
    if (dtype == VIDI_DT_BF16) return lnf<BF16>(p, batch, mode, st);
    if (dtype == VIDI_DT_F16) return lnf<F16>(p, batch, mode, st);
    return VIDI_ERR_DTYPE;
}
