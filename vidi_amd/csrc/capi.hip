// extern "C" entry points of libvidi_hip.so (declared in include/vidi_hip.h).
#include "kernels.h"
#include "attn_text_decode.h"
#include "gemm_skinny_api.h"
#include "gemv_mfma_api.h"
#include <stdlib.h>
#include "../../include/vidi_hip.h"

extern "C" {

int vidi_abi_version(void) { return VIDI_ABI_VERSION; }
const char* vidi_build_info(void) { return "vidi_hip gfx950 " __DATE__ " " __TIME__; }

static GemmParams base_params(const void* X, const void* W, const void* bias, void* Y, const void* R,
                              int M, int N, int K, int ldx, int ldw, int ldy, int ldr, int rmod) {
    GemmParams p;
    __builtin_memset(&p, 0, sizeof(p));
    p.X = (const u16*)X; p.W = (const u16*)W; p.bias = (const u16*)bias; p.Y = (u16*)Y; p.R = (const u16*)R;
    p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.ldr = ldr;
    p.rmod = rmod > 0 ? rmod : 0x7fffffff;
    {
        // tuning knobs (every setting computes the same results): m-tiles per scheduling group and the block -> tile order
        static int gm = -1, ord = -1;
        if (gm < 0) { const char* e = getenv("VIDI_GEMM_GROUP_M"); gm = e ? atoi(e) : 4; if (gm < 1) gm = 4; }     // 4 m-tiles per group: +0.8 % over 8 on the 60-min prefill (same binary, same box)
        if (ord < 0) { const char* e = getenv("VIDI_GEMM_ORDER"); ord = e ? atoi(e) : 1; if (ord != 0) ord = 1; }     // adjacent groups on the 8 XCDs: +0.5 % on the 60-min prefill, same binary
        p.group_m = gm;
        p.order = ord;
        // very wide outputs (the stream's gate/up projection: 112 n-tiles): 8 m-tiles per group measured +0.7…1.3 % over 4 on that shape,
        // same binary, bit-identical (profiles/r4_gemm_lab_clock.jsonl); narrower ones keep 4 (down_proj -2 %, fc1 -0.7 % at 8)
        if (!getenv("VIDI_GEMM_GROUP_M") && N >= 64 * 256) p.group_m = 8;
    }
    return p;
}

int vidi_gemm(const void* X, const void* W, const void* bias, void* Y, const void* R,
              int M, int N, int K, int ldx, int ldw, int ldy, int ldr, int rmod,
              long long bsX, long long bsY, long long bsR, int batch,
              int act, int repkv_hd, int repkv_g, int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!X || !W || !Y) return VIDI_ERR_ARG;
    if (R && (ldr % 4)) return VIDI_ERR_ALIGN;
    GemmParams p = base_params(X, W, bias, Y, R, M, N, K, ldx, ldw, ldy, ldr, rmod);
    p.bsX = bsX; p.bsY = bsY; p.bsR = bsR; p.act = act;
    const int repkv = (repkv_hd > 0 && repkv_g > 1) ? 1 : 0;
    if (repkv) {
        if (repkv_hd % 8) return VIDI_ERR_SHAPE;
        p.rep_hd = repkv_hd; p.rep_g = repkv_g;
    }
    return vidi_gemm_dispatch(p, batch, MODE_PLAIN, repkv, tile_cfg, dtype, (hipStream_t)stream);
}

int vidi_patch_embed(const void* px, const void* W, const void* bias, const void* pos, void* Y, int T, int S, int P, int N, int K,
                     int ldw, int ldy, int ldpos, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!px || !W || !bias || !pos || !Y) return VIDI_ERR_ARG;
    if (T <= 0 || S <= 0 || P <= 0 || P > 16 || S < P) return VIDI_ERR_SHAPE;
    // every 16-byte LDS-DMA piece of the gather starts at pixel (.. * S + px * P + dy * S): 4-byte aligned only when P and S are even
    // (tools/micro/lds_dma_align_probe.hip validated 4-byte, not 2-byte, alignment): odd sizes take the caller's im2col + vidi_gemm arm
    if ((P | S) & 1) return VIDI_ERR_ALIGN;
    if ((ldw % 8) || (ldy % 8) || (ldpos % 8)) return VIDI_ERR_ALIGN;
    if (((uintptr_t)px & 3) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 15) || ((uintptr_t)pos & 15)) return VIDI_ERR_ALIGN;
    const int side = S / P, n = side * side;
    GemmParams p = base_params(px, W, bias, Y, pos, T * n, N, K, 0, ldw, ldy, ldpos, n);
    p.pe_S = S; p.pe_P = P; p.pe_side = side;
    p.pe_nmagic = (unsigned)(0x100000000ull / (unsigned)n) + 1u;
    p.pe_smagic = (unsigned)(0x100000000ull / (unsigned)side) + 1u;
    if ((unsigned long long)T * n * (unsigned long long)n >= 0x100000000ull) return VIDI_ERR_SHAPE;      // exact m / n by multiply-high
    return vidi_w4_patch(p, dtype, (hipStream_t)stream);
}

int vidi_conv_window(const void* f, const void* W, void* Y, int T, int side, int C, int k, int N, int ldw, int ldy, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!f || !W || !Y) return VIDI_ERR_ARG;
    if (T <= 0 || side <= 0 || k <= 0 || k > side || C <= 0 || (C % 64)) return VIDI_ERR_SHAPE;
    if ((ldw % 8) || (ldy % 8)) return VIDI_ERR_ALIGN;
    if (((uintptr_t)f & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 15)) return VIDI_ERR_ALIGN;
    const int oc = side - k + 1, n = oc * oc;
    GemmParams p = base_params(f, W, nullptr, Y, nullptr, T * n, N, k * k * C, 0, ldw, ldy, 0, 0);
    p.pe_S = C; p.pe_P = k; p.pe_side = side;
    p.pe_nmagic = (unsigned)(0x100000000ull / (unsigned)n) + 1u;
    p.pe_smagic = (unsigned)(0x100000000ull / (unsigned)oc) + 1u;
    p.pe_cmagic = (unsigned)(0x100000000ull / (unsigned)(C / 64)) + 1u;
    p.pe_kmagic = (unsigned)(0x100000000ull / (unsigned)k) + 1u;
    if ((unsigned long long)T * n * (unsigned long long)n >= 0x100000000ull) return VIDI_ERR_SHAPE;      // exact m / n by multiply-high
    return vidi_w4_window(p, dtype, (hipStream_t)stream);
}

int vidi_gemm_geglu(const void* X, const void* Wgu, void* Y, int M, int I, int K, int ldx, int ldw, int ldy,
                    int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!X || !Wgu || !Y) return VIDI_ERR_ARG;
    if (I % 32) return VIDI_ERR_SHAPE;
    GemmParams p = base_params(X, Wgu, nullptr, Y, nullptr, M, 2 * I, K, ldx, ldw, ldy, 0, 0);
    if (tile_cfg == 3) return VIDI_ERR_ARG;
    return vidi_gemm_dispatch(p, 1, MODE_GEGLU, 0, tile_cfg, dtype, (hipStream_t)stream);
}

int vidi_gemm_glu(const void* X, const void* Wgu, void* Y, int M, int I, int K, int ldx, int ldw, int ldy,
                  int act, int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!X || !Wgu || !Y) return VIDI_ERR_ARG;
    if (I % 32) return VIDI_ERR_SHAPE;
    if (act != ACT_GELU_TANH && act != ACT_SILU) return VIDI_ERR_ARG;
    GemmParams p = base_params(X, Wgu, nullptr, Y, nullptr, M, 2 * I, K, ldx, ldw, ldy, 0, 0);
    p.act = act;
    if (tile_cfg == 3) return VIDI_ERR_ARG;
    return vidi_gemm_dispatch(p, 1, MODE_GEGLU, 0, tile_cfg, dtype, (hipStream_t)stream);
}

int vidi_gemm_qkv_vt(const void* X, const void* W, const void* bias, void* Yqk, void* Vt,
                     int M, int N, int K, int ldx, int ldw, int ldy,
                     int vstart, int hd, int seq, int seqpad, int nheads, int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!X || !W || !Yqk || !Vt) return VIDI_ERR_ARG;
    if (vstart % 4 || hd % 4 || seq <= 0 || seqpad % 16 || seqpad < ((seq + 15) / 16) * 16 || M % seq) return VIDI_ERR_SHAPE;
    if ((N - vstart) != nheads * hd) return VIDI_ERR_SHAPE;
    GemmParams p = base_params(X, W, bias, Yqk, nullptr, M, N, K, ldx, ldw, ldy, 0, 0);
    p.vstart = vstart; p.hd = hd; p.seq = seq; p.seqpad = seqpad; p.nheads = nheads; p.Vt = (u16*)Vt;
    if (tile_cfg == 3) return VIDI_ERR_ARG;
    return vidi_gemm_dispatch(p, 1, MODE_QKV_VT, 0, tile_cfg, dtype, (hipStream_t)stream);
}

int vidi_row_stats(const void* X, float* stats, long long rows, int H, long long ldx, float eps, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!X || !stats) return VIDI_ERR_ARG;
    return vidi_row_stats_dispatch(X, stats, rows, H, ldx, eps, dtype, (hipStream_t)stream);
}

int vidi_gemm_res_stats(const void* X, const void* W, const void* bias, void* Y, const void* R, float* part,
                        int M, int N, int K, int ldx, int ldw, int ldy, int ldr, int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!X || !W || !bias || !Y || !R || !part) return VIDI_ERR_ARG;
    if ((ldr % 8) || ((uintptr_t)part & 7)) return VIDI_ERR_ALIGN;
    GemmParams p = base_params(X, W, bias, Y, R, M, N, K, ldx, ldw, ldy, ldr, 0);
    const long long t256 = (long long)((N + 255) / 256) * ((M + 255) / 256);
    const int strips = vidi_w4n_stat_strips(N);               // entries per row of `part`: a function of N alone, whichever kernel runs
    if ((tile_cfg < 0 || tile_cfg == 5) && t256 >= 192 && K % 64 == 0 && K >= 192) {
        p.stat_part = part;                                   // fused: the persistent kernel's epilogue emits the partial sums
        if (strips == (N + 127) / 128) return vidi_gemm_dispatch(p, 1, MODE_PLAIN, 0, 5, dtype, (hipStream_t)stream);
        // widths of the 288 x 224 geometry (N = 1 152): its kernel, or — if it cannot take this call — the unfused form below
        const int rc = vidi_w4n_bias_res(p, dtype, (hipStream_t)stream);
        if (rc != -100) return rc;
        p.stat_part = nullptr;
    }
    // small problems run on a tile kernel without the fused emission: same partials from one pass over the stored rows
    const int rc = vidi_gemm_dispatch(p, 1, MODE_PLAIN, 0, tile_cfg == 5 ? -1 : tile_cfg, dtype, (hipStream_t)stream);
    if (rc != 0) return rc;
    return vidi_row_partials_dispatch(Y, part, M, N, ldy, strips, dtype, (hipStream_t)stream);
}

size_t vidi_gemm_skinny_workspace_bytes(int M, int N, int K) {
    if (M < 1 || M > 128 || N <= 0 || K <= 0) return 0;
    return (size_t)vidi_gemm_skinny_ksplit(N, K) * (size_t)M * (size_t)N * sizeof(float);
}

int vidi_gemm_skinny(const void* X, const void* W, const void* bias, void* Y, void* workspace, int M, int N, int K, int ldx, int ldw, int ldy,
                     int dtype, void* stream) {
    (void)hipGetLastError();
    if (!X || !W || !Y || !workspace) return VIDI_ERR_ARG;
    if (vidi_gemm_skinny_workspace_bytes(M, N, K) == 0) return VIDI_ERR_SHAPE;
    if ((ldx % 8) || (ldw % 8) || (ldy % 4) || ldx < K || ldw < K || ldy < N) return VIDI_ERR_ALIGN;
    if (((uintptr_t)X & 15) || ((uintptr_t)W & 15) || ((uintptr_t)Y & 7) || ((uintptr_t)workspace & 15) || (bias && ((uintptr_t)bias & 1))) return VIDI_ERR_ALIGN;
    const int rc = vidi_gemm_skinny_dispatch(X, W, bias, Y, (float*)workspace, M, N, K, ldx, ldw, ldy, dtype, (hipStream_t)stream);
    return rc == -100 ? VIDI_ERR_SHAPE : rc;
}

int vidi_stat_strips(int N) { return N > 0 ? vidi_w4n_stat_strips(N) : 0; }

int vidi_ln_finalize(const float* part, float* stats, long long rows, int N, float eps, void* stream) {
    (void)hipGetLastError();
    if (!part || !stats) return VIDI_ERR_ARG;
    if (((uintptr_t)part & 7) || ((uintptr_t)stats & 7)) return VIDI_ERR_ALIGN;
    return vidi_ln_finalize_dispatch(part, stats, rows, vidi_w4n_stat_strips(N), N, eps, (hipStream_t)stream);
}

int vidi_gemm_ln(const void* X, const void* Wf, const float* stats, const float* colsum, const float* shift, void* Y,
                 int M, int N, int K, int ldx, int ldw, int ldy, int act, int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!X || !Wf || !stats || !colsum || !shift || !Y) return VIDI_ERR_ARG;
    if (((uintptr_t)colsum & 15) || ((uintptr_t)shift & 15) || ((uintptr_t)stats & 7)) return VIDI_ERR_ALIGN;
    if (act != ACT_NONE && act != ACT_GELU_TANH && act != ACT_GELU_ERF) return VIDI_ERR_ARG;
    GemmParams p = base_params(X, Wf, nullptr, Y, nullptr, M, N, K, ldx, ldw, ldy, 0, 0);
    p.act = act; p.ln_stats = stats; p.ln_s = colsum; p.ln_c = shift;
    if (tile_cfg == 3) return VIDI_ERR_ARG;
    return vidi_gemm_dispatch(p, 1, MODE_PLAIN, 0, tile_cfg, dtype, (hipStream_t)stream);
}

int vidi_gemm_ln_heads(const void* X, const void* Wf, const float* stats, const float* colsum, const float* shift, void* Y,
                       int M, int N, int K, int ldx, int ldw, int seq, int hd, int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!X || !Wf || !stats || !colsum || !shift || !Y) return VIDI_ERR_ARG;
    if (((uintptr_t)colsum & 15) || ((uintptr_t)shift & 15) || ((uintptr_t)stats & 7)) return VIDI_ERR_ALIGN;
    if (seq <= 0 || hd <= 0 || hd % 8 || M % seq || N % (3 * hd)) return VIDI_ERR_SHAPE;
    if ((unsigned long long)M * (unsigned)seq >= (1ull << 32)) return VIDI_ERR_SHAPE;            // exactness range of the magic division
    GemmParams p = base_params(X, Wf, nullptr, Y, nullptr, M, N, K, ldx, ldw, N, 0, 0);
    p.ln_stats = stats; p.ln_s = colsum; p.ln_c = shift;
    p.hm_seq = seq; p.hm_hd = hd; p.hm_heads = N / (3 * hd); p.hm_magic = (unsigned)((1ull << 32) / (unsigned)seq) + 1u;
    if (tile_cfg == 3) return VIDI_ERR_ARG;
    return vidi_gemm_dispatch(p, 1, MODE_PLAIN, 0, tile_cfg, dtype, (hipStream_t)stream);
}

int vidi_gemm_qkv_vt_ln(const void* X, const void* Wf, const float* stats, const float* colsum, const float* shift, void* Yqk, void* Vt,
                        int M, int N, int K, int ldx, int ldw, int ldy,
                        int vstart, int hd, int seq, int seqpad, int nheads, int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!X || !Wf || !stats || !colsum || !shift || !Yqk || !Vt) return VIDI_ERR_ARG;
    if (((uintptr_t)colsum & 15) || ((uintptr_t)shift & 15) || ((uintptr_t)stats & 7)) return VIDI_ERR_ALIGN;
    if (vstart % 4 || hd % 4 || seq <= 0 || seqpad % 16 || seqpad < ((seq + 15) / 16) * 16 || M % seq) return VIDI_ERR_SHAPE;
    if ((N - vstart) != nheads * hd) return VIDI_ERR_SHAPE;
    GemmParams p = base_params(X, Wf, nullptr, Yqk, nullptr, M, N, K, ldx, ldw, ldy, 0, 0);
    p.vstart = vstart; p.hd = hd; p.seq = seq; p.seqpad = seqpad; p.nheads = nheads; p.Vt = (u16*)Vt;
    p.ln_stats = stats; p.ln_s = colsum; p.ln_c = shift;
    if (tile_cfg == 3) return VIDI_ERR_ARG;
    return vidi_gemm_dispatch(p, 1, MODE_QKV_VT, 0, tile_cfg, dtype, (hipStream_t)stream);
}

int vidi_gemm_kv_cache(const void* X, const void* W, void* Kc, void* Vtc, void* Vrow,
                       int M, int kvd, int K, int ldx, int ldw, int hd, int ntile64, int tok0,
                       int tile_cfg, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!X || !W || !Kc || !Vtc || !Vrow) return VIDI_ERR_ARG;
    if (kvd % hd || hd % 4 || tok0 < 0 || (tok0 + M + 63) / 64 > ntile64) return VIDI_ERR_SHAPE;
    GemmParams p = base_params(X, W, nullptr, Vrow, nullptr, M, 2 * kvd, K, ldx, ldw, kvd, 0, 0);
    p.Kc = (u16*)Kc; p.Vtc = (u16*)Vtc; p.Vrow = (u16*)Vrow; p.kvd = kvd; p.hd = hd; p.ntile64 = ntile64; p.tok0 = tok0;
    if (tile_cfg == 3) return VIDI_ERR_ARG;
    return vidi_gemm_dispatch(p, 1, MODE_KV_CACHE, 0, tile_cfg, dtype, (hipStream_t)stream);
}

int vidi_gemv(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy,
              int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!X || !W || !Y) return VIDI_ERR_ARG;
    return vidi_gemv_dispatch(X, W, Y, M, N, K, ldx, ldw, ldy, dtype, (hipStream_t)stream);
}

int vidi_gemv_mfma_fits(int M, int N, int K, int glu) { return vidi_gemvm_fits(M, N, K, glu); }

int vidi_gemv_mfma(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int glu_act, int dtype,
                   void* stream) {
    (void)hipGetLastError();
    if (!X || !W || !Y) return VIDI_ERR_ARG;
    return vidi_gemv_mfma_dispatch(X, W, Y, M, N, K, ldx, ldw, ldy, glu_act, dtype, (hipStream_t)stream);
}

int vidi_gemv_glu(const void* X, const void* Wgu, void* Y, int M, int I, int K, int ldx, int ldw, int ldy, int act, int dtype,
                  void* stream) {
    (void)hipGetLastError();
    if (!X || !Wgu || !Y) return VIDI_ERR_ARG;
    return vidi_gemv_glu_dispatch(X, Wgu, Y, M, I, K, ldx, ldw, ldy, act, dtype, (hipStream_t)stream);
}

int vidi_gemv_norm2(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1, long long ld,
                    float eps, const void* W, void* Y, int M, int N, int K, int ldw, int ldy, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!A || !Res || !W1 || !W2 || !Y1 || !W || !Y) return VIDI_ERR_ARG;
    return vidi_gemv_norm2_dispatch(A, B, C, Res, W1, W2, Y1, ld, eps, W, Y, M, N, K, ldw, ldy, -1, dtype, (hipStream_t)stream);
}

int vidi_gemv_glu_norm2(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1, long long ld,
                        float eps, const void* Wgu, void* Y, int M, int I, int K, int ldw, int ldy, int act, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!A || !Res || !W1 || !W2 || !Y1 || !Wgu || !Y || act < 0) return VIDI_ERR_ARG;
    return vidi_gemv_norm2_dispatch(A, B, C, Res, W1, W2, Y1, ld, eps, Wgu, Y, M, I, K, ldw, ldy, act, dtype, (hipStream_t)stream);
}

int vidi_gemm_f32(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K,
                  int ldx, int ldw, int ldy, int act, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!X || !W || !Y) return VIDI_ERR_ARG;
    return vidi_gemm_f32_dispatch(X, W, bias, Y, M, N, K, ldx, ldw, ldy, act, (hipStream_t)stream);
}

int vidi_attn_self(const void* QK, const void* Vt, void* O, int B, int N, int Npad, int H, int D,
                   int ldqk, int koff, int ldo, float scale, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!QK || !Vt || !O) return VIDI_ERR_ARG;
    AttnSelfParams p;
    p.QK = (const u16*)QK; p.Vt = (const u16*)Vt; p.O = (u16*)O;
    p.B = B; p.N = N; p.Npad = Npad; p.H = H; p.ldqk = ldqk; p.koff = koff; p.ldo = ldo; p.scale = scale;
    return vidi_attn_self_dispatch(p, D, dtype, (hipStream_t)stream);
}

int vidi_attn_self_rm(const void* QKV, void* O, int B, int N, int H, int D, int ld, long long koff, long long voff, long long bs, long long hs,
                      int ldo, float scale, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!QKV || !O) return VIDI_ERR_ARG;
    AttnSelfRmParams p;
    p.QKV = (const u16*)QKV; p.O = (u16*)O; p.B = B; p.N = N; p.H = H; p.ld = ld; p.koff = koff; p.voff = voff; p.ldo = ldo; p.scale = scale;
    p.bs = bs > 0 ? bs : (long long)N * ld;            // defaults: row-major [B*N, ld]
    p.hs = hs > 0 ? hs : D;
    return vidi_attn_self_rm_dispatch(p, D, dtype, (hipStream_t)stream);
}

size_t vidi_attn_cross_workspace_bytes(int zsplit, int nkv, int Rpad, int HD) {
    const size_t W = (size_t)zsplit;           // one partial per block (its 4 waves are merged in LDS)
    return W * nkv * Rpad * (size_t)(HD + 2) * sizeof(float);
}

int vidi_attn_cross_row_tiles_per_block(int Rpad, float softcap, int dtype) { return vidi_attn_cross_rtpb(Rpad, softcap, dtype); }

int vidi_attn_cross(const void* Q, const void* Kc, const void* Vtc, const void* mask, float* Opart, float* ML,
                    int R, int Rpad, int G, int nkv, int HD, int ldq, int ntile64, int key_start, int n_keys,
                    float scale, float softcap, int zsplit, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!Q || !Kc || !Vtc || !Opart || !ML) return VIDI_ERR_ARG;
    if ((key_start + n_keys + 63) / 64 > ntile64) return VIDI_ERR_SHAPE;
    AttnCrossParams p;
    p.Q = (const u16*)Q; p.Kc = (const u16*)Kc; p.Vtc = (const u16*)Vtc; p.mask = (const unsigned char*)mask;
    p.Opart = Opart; p.ML = ML; p.R = R; p.Rpad = Rpad; p.G = G; p.nkv = nkv; p.ldq = ldq;
    p.ntile64 = ntile64; p.key_start = key_start; p.n_keys = n_keys; p.scale = scale; p.softcap = softcap;
    return vidi_attn_cross_dispatch(p, HD, zsplit, dtype, (hipStream_t)stream);
}

int vidi_attn_cross2(const void* Q, const void* Kc, const void* Vtc,
                     const void* maskA, float* OpartA, float* MLA, int key_startA, int n_keysA, int zsplitA,
                     const void* maskB, float* OpartB, float* MLB, int key_startB, int n_keysB, int zsplitB,
                     int R, int Rpad, int G, int nkv, int HD, int ldq, int ntile64, float scale, float softcap, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!Q || !Kc || !Vtc || !OpartA || !MLA || !OpartB || !MLB) return VIDI_ERR_ARG;
    if ((key_startA + n_keysA + 63) / 64 > ntile64 || (key_startB + n_keysB + 63) / 64 > ntile64) return VIDI_ERR_SHAPE;
    AttnCrossParams a, b;
    a.Q = (const u16*)Q; a.Kc = (const u16*)Kc; a.Vtc = (const u16*)Vtc; a.mask = (const unsigned char*)maskA;
    a.Opart = OpartA; a.ML = MLA; a.R = R; a.Rpad = Rpad; a.G = G; a.nkv = nkv; a.ldq = ldq;
    a.ntile64 = ntile64; a.key_start = key_startA; a.n_keys = n_keysA; a.scale = scale; a.softcap = softcap;
    b = a;
    b.mask = (const unsigned char*)maskB; b.Opart = OpartB; b.ML = MLB; b.key_start = key_startB; b.n_keys = n_keysB;
    return vidi_attn_cross2_dispatch(a, b, HD, zsplitA, zsplitB, dtype, (hipStream_t)stream);
}

int vidi_attn_merge(const float* Opart, const float* ML, void* Out, float* OutF32, float* OutML,
                    int W, int nkv, int R, int Rpad, int G, int HD, int ldo, int zero_out, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!Opart || !ML || (!Out && !OutF32)) return VIDI_ERR_ARG;
    AttnMergeParams p;
    p.Opart = Opart; p.ML = ML; p.Out = (u16*)Out; p.OutF32 = OutF32; p.OutML = OutML;
    p.W = W; p.nkv = nkv; p.R = R; p.Rpad = Rpad; p.G = G; p.ldo = ldo; p.zero_out = zero_out;
    p.wsO = (long long)nkv * Rpad * HD; p.wsML = (long long)nkv * Rpad * 2; p.rpo = Rpad;
    return vidi_attn_merge_dispatch(p, HD, dtype, (hipStream_t)stream);
}

int vidi_attn_merge2(const float* OpartA, const float* MLA, void* OutA, int WA, int zeroA,
                     const float* OpartB, const float* MLB, void* OutB, int WB, int zeroB,
                     int nkv, int R, int Rpad, int G, int HD, int ldo, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!OpartA || !MLA || !OutA || !OpartB || !MLB || !OutB) return VIDI_ERR_ARG;
    AttnMergeParams a, b;
    a.Opart = OpartA; a.ML = MLA; a.Out = (u16*)OutA; a.OutF32 = nullptr; a.OutML = nullptr;
    a.W = WA; a.nkv = nkv; a.R = R; a.Rpad = Rpad; a.G = G; a.ldo = ldo; a.zero_out = zeroA;
    a.wsO = (long long)nkv * Rpad * HD; a.wsML = (long long)nkv * Rpad * 2; a.rpo = Rpad;
    b = a;
    b.Opart = OpartB; b.ML = MLB; b.Out = (u16*)OutB; b.W = WB; b.zero_out = zeroB;
    if (WA <= 0 || WB <= 0) return VIDI_ERR_SHAPE;
    return vidi_attn_merge2_dispatch(a, b, HD, dtype, (hipStream_t)stream);
}

int vidi_attn_merge2_sharded(const float* OpartA, const float* MLA, long long wsOA, long long wsMLA, void* OutA, float* OutF32A,
                             float* OutMLA, int WA, int zeroA,
                             const float* OpartB, const float* MLB, long long wsOB, long long wsMLB, void* OutB, float* OutF32B,
                             float* OutMLB, int WB, int zeroB,
                             int nkv, int R, int Rpad, int rpo, int G, int HD, int ldo, int dtype, void* stream) {
    (void)hipGetLastError();
    const bool hasA = OutA || OutF32A, hasB = OutB || OutF32B;
    if (!hasA && !hasB) return VIDI_ERR_ARG;
    if ((hasA && WA > 0 && (!OpartA || !MLA)) || (hasB && WB > 0 && (!OpartB || !MLB))) return VIDI_ERR_ARG;
    if ((OutF32A && !OutMLA) || (OutF32B && !OutMLB) || rpo < R) return VIDI_ERR_ARG;
    AttnMergeParams a, b;
    a.Opart = OpartA; a.ML = MLA; a.Out = (u16*)OutA; a.OutF32 = OutF32A; a.OutML = OutMLA;
    a.W = WA; a.nkv = nkv; a.R = R; a.Rpad = Rpad; a.G = G; a.ldo = ldo; a.zero_out = zeroA; a.wsO = wsOA; a.wsML = wsMLA; a.rpo = rpo;
    b = a;
    b.Opart = OpartB; b.ML = MLB; b.Out = (u16*)OutB; b.OutF32 = OutF32B; b.OutML = OutMLB;
    b.W = WB; b.zero_out = zeroB; b.wsO = wsOB; b.wsML = wsMLB;
    return vidi_attn_merge2_dispatch(a, b, HD, dtype, (hipStream_t)stream);
}

int vidi_attn_text(const void* Q, const void* Kc, const void* Vc, const void* kmask, void* O,
                   int B, int Lq, int Lmax, int nq, int nkv, int HD, int past_len, int window,
                   float scale, float softcap, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!Q || !Kc || !Vc || !O) return VIDI_ERR_ARG;
    AttnTextParams p;
    p.Q = (const u16*)Q; p.Kc = (const u16*)Kc; p.Vc = (const u16*)Vc; p.kmask = (const unsigned char*)kmask; p.O = (u16*)O;
    p.B = B; p.Lq = Lq; p.Lmax = Lmax; p.nq = nq; p.nkv = nkv; p.past_len = past_len; p.past_len_dev = nullptr; p.window = window;
    p.scale = scale; p.softcap = softcap;
    return vidi_attn_text_dispatch(p, HD, dtype, (hipStream_t)stream);
}

int vidi_attn_text_dyn(const void* Q, const void* Kc, const void* Vc, const void* kmask, void* O,
                       int B, int Lq, int Lmax, int nq, int nkv, int HD, const int* past_len_dev, int window,
                       float scale, float softcap, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!Q || !Kc || !Vc || !O || !past_len_dev) return VIDI_ERR_ARG;
    AttnTextParams p;
    p.Q = (const u16*)Q; p.Kc = (const u16*)Kc; p.Vc = (const u16*)Vc; p.kmask = (const unsigned char*)kmask; p.O = (u16*)O;
    p.B = B; p.Lq = Lq; p.Lmax = Lmax; p.nq = nq; p.nkv = nkv; p.past_len = 0; p.past_len_dev = past_len_dev; p.window = window;
    p.scale = scale; p.softcap = softcap;
    return vidi_attn_text_dispatch(p, HD, dtype, (hipStream_t)stream);
}

int vidi_attn_text_decode(const void* qkv, int ldqkv, void* Kc, void* Vc, const void* kmask, const void* cos_, const void* sin_, void* O,
                          int B, int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int window, float scale, float softcap,
                          int dtype, void* stream) {
    (void)hipGetLastError();
    if (!qkv || !Kc || !Vc || !cos_ || !sin_ || !O) return VIDI_ERR_ARG;
    return vidi_attn_text_decode_dispatch(qkv, ldqkv, Kc, Vc, kmask, cos_, sin_, O, B, Lmax, nq, nkv, HD, pos0, pos_dev, window, scale,
                                          softcap, dtype, (hipStream_t)stream);
}

int vidi_attn_text_decode_merge2(const void* qkv, int ldqkv, void* Kc, void* Vc, const void* kmask, const void* cos_, const void* sin_, void* O,
                                 int B, int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int window, float scale, float softcap,
                                 const float* OpartA, const float* MLA, void* OutA, int WA, int zeroA,
                                 const float* OpartB, const float* MLB, void* OutB, int WB, int zeroB,
                                 int R, int Rpad, int ldo, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!qkv || !Kc || !Vc || !cos_ || !sin_ || !O || !OpartA || !MLA || !OutA || !OpartB || !MLB || !OutB) return VIDI_ERR_ARG;
    AttnTextDecodeParams tp;
    size_t lds;
    const int rc = attn_text_decode_params(tp, lds, qkv, ldqkv, Kc, Vc, kmask, cos_, sin_, O, B, Lmax, nq, nkv, HD, pos0, pos_dev, window, scale,
                                           softcap);
    if (rc) return rc;
    AttnMergeParams a, b;
    a.Opart = OpartA; a.ML = MLA; a.Out = (u16*)OutA; a.OutF32 = nullptr; a.OutML = nullptr;
    a.W = WA; a.nkv = nkv; a.R = R; a.Rpad = Rpad; a.G = nq / nkv; a.ldo = ldo; a.zero_out = zeroA;
    a.wsO = (long long)nkv * Rpad * HD; a.wsML = (long long)nkv * Rpad * 2; a.rpo = Rpad;
    b = a;
    b.Opart = OpartB; b.ML = MLB; b.Out = (u16*)OutB; b.W = WB; b.zero_out = zeroB;
    return vidi_attn_text_decode_merge2_dispatch(tp, lds, a, b, HD, dtype, (hipStream_t)stream);
}

int vidi_rope(void* Q, void* K, const void* cos_, const void* sin_, int rows, int nq, int nkv, int HD,
              int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!Q || !K || !cos_ || !sin_) return VIDI_ERR_ARG;
    return vidi_rope_dispatch(Q, K, cos_, sin_, rows, nq, nkv, HD, dtype, (hipStream_t)stream);
}

int vidi_rope_cache(const void* qkv, int ldqkv, void* QR, void* Kc, void* Vc, const void* cos_, const void* sin_, int B, int Lq,
                    int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!qkv || !QR || !Kc || !Vc || !cos_ || !sin_) return VIDI_ERR_ARG;
    return vidi_rope_cache_dispatch(qkv, ldqkv, QR, Kc, Vc, cos_, sin_, B, Lq, Lmax, nq, nkv, HD, pos0, pos_dev, dtype, (hipStream_t)stream);
}

int vidi_resid_norm2(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1, void* Y2,
                     int rows, int H, long long ld, float eps, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!A || !Res || !W1 || !W2 || !Y1 || !Y2) return VIDI_ERR_ARG;
    return vidi_resid_norm2_dispatch(A, B, C, Res, W1, W2, Y1, Y2, rows, H, ld, eps, dtype, (hipStream_t)stream);
}

int vidi_norm(int mode, const void* X, const float* XF32, const void* W, const void* Bias, const void* Res,
              void* Y, void* Mask, int rows, int H, long long ldx, long long ldy, long long ldr,
              float eps, float normalizer, const int* sample_flag, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if ((!X && !XF32) || !Y) return VIDI_ERR_ARG;
    NormParams p;
    p.X = (const u16*)X; p.XF32 = XF32; p.Wt = (const u16*)W; p.Bias = (const u16*)Bias; p.Res = (const u16*)Res;
    p.Y = (u16*)Y; p.Mask = (unsigned char*)Mask; p.rows = rows; p.H = H; p.ldx = ldx; p.ldy = ldy; p.ldr = ldr;
    p.eps = eps; p.normalizer = normalizer; p.sample_flag = sample_flag;
    return vidi_norm_dispatch(p, mode, dtype, (hipStream_t)stream);
}

int vidi_im2col_patch(const void* px, void* A, int T, int S, int P, int Kpad, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    // S need not be a multiple of P: Conv2d(padding="valid") drops the remainder (384 = 27*14 + 6)
    if (!px || !A || T <= 0 || S < P || Kpad < 3 * P * P) return VIDI_ERR_ARG;
    void* a[2] = {(void*)px, A};
    const long long i[4] = {T, S, P, Kpad};
    return vidi_ew_dispatch(EW_IM2COL, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_pool_s2d(const void* f, void* out, int T, int side, int C, int h, int w, int m, int resize, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!f || !out || T <= 0 || m <= 0) return VIDI_ERR_ARG;
    if (!resize && (h != side + 1 || w != side + 1)) return VIDI_ERR_SHAPE;
    void* a[2] = {(void*)f, out};
    const long long i[7] = {T, side, C, h, w, m, resize};
    return vidi_ew_dispatch(EW_POOL, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_add_pos(void* f, const void* ph, const void* pw, const void* pt, int T, int oh, int ow, int H, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!f) return VIDI_ERR_ARG;
    void* a[4] = {f, (void*)ph, (void*)pw, (void*)pt};
    const long long i[4] = {T, oh, ow, H};
    return vidi_ew_dispatch(EW_ADDPOS, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_add3(const void* a_, const void* b, const void* c, void* y, long long n, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!a_ || !y) return VIDI_ERR_ARG;
    void* a[4] = {(void*)a_, (void*)b, (void*)c, y};
    const long long i[1] = {n};
    return vidi_ew_dispatch(EW_ADD3, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_embed(const long long* ids, const void* E, void* out, int n, int H, long long vocab, float normalizer, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!ids || !E || !out) return VIDI_ERR_ARG;
    void* a[3] = {(void*)ids, (void*)E, out};
    const long long i[3] = {n, H, vocab};
    const float f[1] = {normalizer};
    return vidi_ew_dispatch(EW_EMBED, a, i, f, dtype, (hipStream_t)stream);
}

int vidi_geglu_unpack(const void* Yp, void* out, int M, int I, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!Yp || !out) return VIDI_ERR_ARG;
    void* a[2] = {(void*)Yp, out};
    const long long i[3] = {M, I, ACT_GELU_TANH};
    return vidi_ew_dispatch(EW_GEGLU_UNPACK, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_glu_unpack(const void* Yp, void* out, int M, int I, int act, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!Yp || !out) return VIDI_ERR_ARG;
    if (act != ACT_GELU_TANH && act != ACT_SILU) return VIDI_ERR_ARG;
    void* a[2] = {(void*)Yp, out};
    const long long i[3] = {M, I, act};
    return vidi_ew_dispatch(EW_GEGLU_UNPACK, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_im2col_nhwc(const void* x, void* out, int T, int side, int C, int k, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!x || !out || T <= 0 || side <= 0 || k <= 0 || k > side || C % 8) return VIDI_ERR_ARG;
    void* a[2] = {(void*)x, out};
    const long long i[4] = {T, side, C, k};
    return vidi_ew_dispatch(EW_IM2COL_NHWC, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_resize_bilinear_ac(const void* x, void* out, int T, int s_in, int s_out, int C, int dtype, void* stream) {
    (void)hipGetLastError();
    if (!x || !out || T <= 0 || s_in <= 0 || s_out <= 0 || C % 8) return VIDI_ERR_ARG;
    void* a[2] = {(void*)x, out};
    const long long i[4] = {T, s_in, s_out, C};
    return vidi_ew_dispatch(EW_RESIZE_AC, a, i, nullptr, dtype, (hipStream_t)stream);
}

size_t vidi_softcap_argmax_workspace_bytes(int B) { return B > 0 ? (size_t)16 * (size_t)B : 0; }

int vidi_softcap_argmax(void* logits, long long* idx, int B, int V, long long ld, float cap, int dtype, void* workspace, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!logits || !idx || !workspace || B <= 0 || V <= 0) return VIDI_ERR_ARG;
    if ((uintptr_t)workspace & 7) return VIDI_ERR_ALIGN;
    void* a[3] = {logits, (void*)idx, workspace};
    const long long i[3] = {B, V, ld};
    const float f[1] = {cap};
    return vidi_ew_dispatch(EW_SOFTCAP_ARGMAX, a, i, f, dtype, (hipStream_t)stream);
}

int vidi_mel_transpose_pad(const void* mel, void* out, int C, int nmel, int L, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!mel || !out) return VIDI_ERR_ARG;
    void* a[2] = {(void*)mel, out};
    const long long i[3] = {C, nmel, L};
    return vidi_ew_dispatch(EW_MEL_T, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_scale(const void* x, void* y, long long n, float s, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!x || !y) return VIDI_ERR_ARG;
    void* a[2] = {(void*)x, y};
    const long long i[1] = {n};
    const float f[1] = {s};
    return vidi_ew_dispatch(EW_SCALE, a, i, f, dtype, (hipStream_t)stream);
}

int vidi_any_nonzero(const void* x, long long n, int* flag, int dtype, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!x || !flag) return VIDI_ERR_ARG;
    void* a[2] = {(void*)x, (void*)flag};
    const long long i[1] = {n};
    return vidi_ew_dispatch(EW_ANY_NONZERO, a, i, nullptr, dtype, (hipStream_t)stream);
}

int vidi_sinusoid(float* pe, const float* div_term, int rows, int i0, int l, int N, int d, void* stream) {
    (void)hipGetLastError();   // a launch status must not inherit an earlier, unrelated runtime error
    if (!pe || !div_term) return VIDI_ERR_ARG;
    return vidi_sinusoid_dispatch(pe, div_term, rows, i0, l, N, d, (hipStream_t)stream);
}

}  // extern "C"
