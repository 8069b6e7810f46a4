// persistent 4-wave GEMM, 288 x 224 tiles (gemm_w4n.h): the bias + residual (+ LayerNorm statistics) epilogues of the encoder towers'
// out_proj / fc2 at widths that are multiples of 288 but not of 256 (SigLIP: N = 1 152)
#include "gemm_w4n.h"
#include <stdlib.h>
#include "gemm_w4_launch.h"

// VIDI_W4N=0: the 256-wide kernel everywhere (the same-box A/B arm; the partial-sum layout follows the switch)
static bool w4n_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("VIDI_W4N"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

int vidi_w4n_bias_res(const GemmParams& p, int dtype, hipStream_t st) {
    if (!w4n_enabled() || !w4n_takes(p.N) || p.K % 64 || p.K < 192 || !p.bias || !p.R || p.rmod < p.M || p.act != ACT_NONE || p.ln_stats || p.hm_seq) return VIDI_W4_UNSUPPORTED;
    if ((p.ldy % 8) || (p.ldr % 8)) return VIDI_W4_UNSUPPORTED;
    if (p.stat_part) {
        if (dtype == VIDI_DT_BF16) return launch_w4n<BF16, Epi<true, ACT_NONE, 1, false, true>>(p, st);
        if (dtype == VIDI_DT_F16) return launch_w4n<F16, Epi<true, ACT_NONE, 1, false, true>>(p, st);
    } else {
        if (dtype == VIDI_DT_BF16) return launch_w4n<BF16, Epi<true, ACT_NONE, 1>>(p, st);
        if (dtype == VIDI_DT_F16) return launch_w4n<F16, Epi<true, ACT_NONE, 1>>(p, st);
    }
    return VIDI_ERR_DTYPE;
}

// LayerNorm folded into the q | k | v projection, head-major store (SigLIP: N = 3 456 = 12 x 288).  VIDI_W4N_QKV=0: the 256-wide kernel
int vidi_w4n_ln_heads(const GemmParams& p, int dtype, hipStream_t st) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("VIDI_W4N_QKV"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || !w4n_enabled() || !w4n_takes(p.N) || p.K % 64 || p.K < 192 || !p.ln_stats || !p.ln_s || !p.ln_c || !p.hm_seq || p.R || p.act != ACT_NONE || p.stat_part)
        return VIDI_W4_UNSUPPORTED;
    if (p.hm_hd % 8) return VIDI_W4_UNSUPPORTED;                             // a 16-byte chunk lies inside one head
    if (dtype == VIDI_DT_BF16) return launch_w4n<BF16, Epi<true, ACT_NONE, 0, 1, false, true>>(p, st);
    if (dtype == VIDI_DT_F16) return launch_w4n<F16, Epi<true, ACT_NONE, 0, 1, false, true>>(p, st);
    return VIDI_ERR_DTYPE;
}

int vidi_w4n_stat_strips(int N) { return (w4n_enabled() && w4n_takes(N)) ? w4n_stat_strips(N) : (N + 127) / 128; }
