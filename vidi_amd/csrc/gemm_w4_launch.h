// Host-side launch of the persistent 4-wave GEMM (gemm_w4.h).  The instantiations are spread over several translation units
// (gemm_w4_*.hip) so the library builds in parallel; every TU exports one plain dispatcher declared here.
#pragma once
#include "gemm_w4.h"

#define VIDI_W4_UNSUPPORTED (-100)      // epilogue combination not instantiated: the caller falls back to the 8-wave kernel

template <typename T, int MODE, bool REPKV, typename EPI, int PATCH = 0>
static int launch_w4(const GemmParams& p, int batch, hipStream_t st) {
    auto kern = gemm_w4_kernel<T, MODE, REPKV, true, EPI, LabNone, PATCH>;
    static bool attr_done = false;
    static int ncu = 0;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W4Geom::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        attr_done = true;
    }
    const long long tiles = (long long)((p.N + 255) / 256) * ((p.M + 255) / 256) * batch;
    const int grid = (int)(tiles < ncu ? tiles : ncu);                 // one block per CU walks the tiles
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), W4Geom::LDS_BYTES, st, p, batch);
    return (int)hipGetLastError();
}

// MODE_PLAIN: (bias, act, residual kind) combinations the Vidi engines use; others return VIDI_W4_UNSUPPORTED
template <typename T>
static int w4_plain_dispatch(const GemmParams& p, int batch, int repkv, hipStream_t st) {
    if (p.ln_stats || p.stat_part) return vidi_w4_lnf(p, batch, MODE_PLAIN, T::id, st);       // LayerNorm folded into the projection (gemm_w4_lnf.hip)
    const bool bias = p.bias != nullptr;
    const int res = p.R ? (p.rmod < p.M ? 2 : 1) : 0;
    const int act = p.act;
    if (repkv) {
        if (bias || act != ACT_NONE || res == 2) return VIDI_W4_UNSUPPORTED;
        return res ? launch_w4<T, MODE_PLAIN, true, Epi<false, ACT_NONE, 1>>(p, batch, st)
                   : launch_w4<T, MODE_PLAIN, true, Epi<false, ACT_NONE, 0>>(p, batch, st);
    }
    if (!bias) {
        if (act != ACT_NONE || res == 2) return VIDI_W4_UNSUPPORTED;
        return res ? launch_w4<T, MODE_PLAIN, false, Epi<false, ACT_NONE, 1>>(p, batch, st)       // x += proj(...)            (mistral.py:219-221, :135)
                   : launch_w4<T, MODE_PLAIN, false, Epi<false, ACT_NONE, 0>>(p, batch, st);      // bias-free projections
    }
    if (act == ACT_NONE) {
        if (res == 0) return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_NONE, 0>>(p, batch, st);   // projector second linear
        if (res == 1) {                                                                               // encoder out_proj / fc2 + residual
            if (batch == 1) {                                  // widths of the 288 x 224 tile geometry (N = 1 152): gemm_w4n.h
                const int rc = vidi_w4n_bias_res(p, T::id, st);
                if (rc != VIDI_W4_UNSUPPORTED) return rc;
            }
            return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_NONE, 1>>(p, batch, st);
        }
        return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_NONE, 2>>(p, batch, st);                  // patch embedding + position table
    }
    if (act == ACT_GELU_TANH) {
        if (res == 0) return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_GELU_TANH, 0>>(p, batch, st);   // SigLIP fc1
        return VIDI_W4_UNSUPPORTED;
    }
    if (act == ACT_GELU_ERF) {
        if (res == 0) return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_GELU_ERF, 0>>(p, batch, st);    // Whisper fc1 / conv1, projector first linear
        if (res == 1) return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_GELU_ERF, 1>>(p, batch, st);    // Whisper conv2 + embed_positions (one window per batch entry)
        return launch_w4<T, MODE_PLAIN, false, Epi<true, ACT_GELU_ERF, 2>>(p, batch, st);                  // the same with the table repeating inside M
    }
    return VIDI_W4_UNSUPPORTED;
}

int vidi_w4_lnf(const GemmParams& p, int batch, int mode, int dtype, hipStream_t st);          // LayerNorm-folded epilogues (PLAIN + act, QKV_VT)
int vidi_w4_plain_bf16(const GemmParams& p, int batch, int repkv, hipStream_t st);
int vidi_w4_plain_f16(const GemmParams& p, int batch, int repkv, hipStream_t st);
int vidi_w4_modes(const GemmParams& p, int batch, int mode, int dtype, hipStream_t st);       // GEGLU / QKV_VT / KV_CACHE
