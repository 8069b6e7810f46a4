// Internal interface between the kernel translation units and the C ABI (capi.hip).
#pragma once
#include "common.h"

enum { MODE_PLAIN = 0, MODE_GEGLU = 1, MODE_QKV_VT = 2, MODE_KV_CACHE = 3 };
enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_SILU = 3 };
enum { NORM_GEMMA = 0, NORM_GEMMA_ADD = 1, NORM_MM = 2, NORM_MM_NOW = 3, NORM_LLM = 4, NORM_LAYER = 5 };
#define EW_IM2COL 0
#define EW_POOL 1
#define EW_ADDPOS 2
#define EW_ADD3 3
#define EW_EMBED 4
#define EW_GEGLU_UNPACK 5
#define EW_SOFTCAP_ARGMAX 6
#define EW_MEL_T 7
#define EW_SCALE 8
#define EW_ANY_NONZERO 9
#define EW_IM2COL_NHWC 10
#define EW_RESIZE_AC 11

struct GemmParams {
    const u16* X; const u16* W; const u16* bias; u16* Y; const u16* R;
    int M, N, K;
    int ldx, ldw, ldy, ldr;
    long long bsX, bsY, bsR;      // batch strides (elements); W/bias shared across the batch
    int act, rmod;
    int group_m;                  // m-tiles per scheduling group (L2 reuse shape)
    int order;                    // block -> tile order: 0 = per-XCD contiguous ranges, 1 = the 8 XCDs sweep adjacent groups (gemm_tile.h)
    unsigned long long* dbg;      // tools/lab builds only (phase time stamps); null in the product
    int rep_hd, rep_g;            // REPKV: logical k -> physical column (k/(g*hd))*hd + k%hd
    // MODE_QKV_VT: columns >= vstart go to Vt[b][h][d][seqpad] (key order permuted per 16-slab)
    int vstart, hd, seq, seqpad, nheads; u16* Vt;
    // MODE_KV_CACHE: N = 2*kvd; K half -> Kc tiled, V half -> Vrow row-major + Vtc tiled
    u16* Kc; u16* Vtc; u16* Vrow; int kvd, ntile64, tok0;
    // LayerNorm folded into the projection (encoder towers): X holds the UN-normalised rows, W the weight with the LayerNorm gain
    // folded in (W' = W * gamma), and the epilogue applies  y = rstd_m * (acc - mean_m * s_n) + c_n  with
    //   ln_stats[m] = (mean_m, rstd_m)  (vidi_row_stats),  ln_s[n] = sum_k W'[n][k],  ln_c[n] = sum_k W[n][k] * beta[k] + bias[n]   (fp32)
    // == Linear(LayerNorm(x)) without materialising LayerNorm(x).  Null ln_stats: plain bias epilogue.
    const float* ln_stats; const float* ln_s; const float* ln_c;
    float ln_eps;                 // Epi::lnf == 2 (statistics computed in the consumer's K loop): the LayerNorm's epsilon
    // producer side of the same fusion: a bias + residual epilogue that also leaves, per output row and 128-column strip, the
    // partial sums (sum y, sum y^2) of the values it stored — stat_part[m][strip][2], strips = ceil(N / 128) — so the NEXT
    // LayerNorm's statistics need no pass over Y (vidi_ln_finalize turns them into (mean, rstd)).  Null: off.
    float* stat_part;
    // head-major output (encoder q/k/v projection feeding vidi_attn_self_rm): column n = (which, head, d) of N = 3 * heads * hm_hd, row
    // m = (frame, token) of M = frames * hm_seq -> Y[which][frame][head][token][d]: every head's key rows contiguous, so the attention's
    // K / V tiles are whole 128-byte lines.  hm_seq == 0: row-major.  hm_magic = floor(2^32 / hm_seq) + 1 (exact m / hm_seq for M * seq < 2^32).
    int hm_seq, hm_hd, hm_heads; unsigned hm_magic;
    // patch-embedding loader (gemm_w4_kernel<..., PATCH = true>): X is the NCHW pixel tensor [T, 3, pe_S, pe_S]; output row m = (frame,
    // patch row, patch column) reads its pe_P x pe_P patch of every channel straight from the pixels.  The contraction index is
    // k = (c * pe_P + dy) * 16 + dx: one 32-byte run of 16 pixels per (channel, patch line) — the pe_P real ones and the first pixels of
    // the next patch, whose weight columns are zero — so a 64-wide K slice is four patch lines and every 16-byte LDS-DMA piece is 8
    // contiguous pixels (4-byte aligned: tools/micro/lds_dma_align_probe.hip).  pe_nmagic / pe_smagic: floor(2^32 / d) + 1 for d = the
    // patches per frame / per patch line.  pe_S == 0: X is a row-major matrix.
    int pe_S, pe_P, pe_side; unsigned pe_nmagic, pe_smagic;
    // window mode (PATCH = 2; Vidi-7B's learned Conv2DPool): X is the token-major feature map [frames, pe_side * pe_side, pe_S channels],
    // output row m = (frame, oy, ox) of the (pe_side - pe_P + 1)^2 valid positions of a pe_P x pe_P window, k = (dy, dx, channel): a 64-wide
    // K slice is 64 contiguous channels of one window position.  pe_cmagic / pe_kmagic: floor(2^32 / d) + 1 for d = pe_S / 64 and pe_P.
    unsigned pe_cmagic, pe_kmagic;
};

struct AttnSelfParams {
    const u16* QK; const u16* Vt; u16* O;
    int B, N, Npad, H;
    int ldqk, koff, ldo;
    float scale;
};

struct AttnSelfRmParams {         // Q | K | V of one projection in one buffer, V in natural (un-transposed) order
    const u16* QKV; u16* O;
    int B, N, H;
    int ld, ldo;                  // row strides (elements) of the q/k/v rows and of O
    long long koff, voff;         // element offsets of the K and V parts
    long long bs, hs;             // frame and head strides (elements): row-major rows N*ld, D; head-major H*N*D, N*D (ld = D)
    float scale;
};

struct AttnCrossParams {
    const u16* Q;            // [tokens_q, ldq]; head (kvh*G+g) at column (kvh*G+g)*HD
    const u16* Kc; const u16* Vtc; const unsigned char* mask;   // mask: [n_keys] (1 = valid) or null
    float* Opart; float* ML; // Opart[zsplit][nkv][Rpad][HD], ML[zsplit][nkv][Rpad][2] (one partial per block)
    int R;                   // query rows per kv head = tokens_q * G
    int Rpad;                // R rounded up to 32
    int G, nkv, ldq;
    int ntile64;             // tiles per kv head in the cache
    int key_start;           // first key (multiple of 64) of this modality's region in the cache
    int n_keys;              // number of keys in the region
    float scale, softcap;    // softcap <= 0: disabled
};

struct AttnMergeParams {
    const float* Opart; const float* ML; u16* Out; float* OutF32; float* OutML;
    int W, nkv, R, Rpad, G, ldo;
    int zero_out;            // 1: sample has no valid key at all -> output zeros (gemma.py:180-192)
    long long wsO, wsML;     // floats between successive partials w in Opart / ML (nkv*Rpad*HD and nkv*Rpad*2 when contiguous)
    int rpo;                 // row stride of the partial-form outputs OutF32 / OutML (Rpad, or a tighter packing for the all-gather)
};

struct AttnTextParams {
    const u16* Q;                 // [B, Lq, nq*HD]  (RoPE already applied)
    const u16* Kc; const u16* Vc; // [B, Lmax, nkv*HD] text KV cache (RoPE'd keys)
    const unsigned char* kmask;   // [B, Lmax] 1 = valid key (null: all valid)
    u16* O;                       // [B, Lq, nq*HD]
    int B, Lq, Lmax, nq, nkv;
    int past_len;                 // key index of query row 0
    const int* past_len_dev;      // if non-null, read past_len from device memory (graph-captured decode)
    int window;                   // <= 0: no sliding window
    float scale, softcap;
};

struct NormParams {
    const u16* X; const u16* Wt; const u16* Bias; const u16* Res; u16* Y; unsigned char* Mask;
    const float* XF32;       // NORM_MM_NOW only: fp32 input that is first rounded to T (pos-embed path)
    int rows, H; long long ldx, ldy, ldr;
    float eps, normalizer; const int* sample_flag;   // device flag (null = 1)
};

int vidi_resid_norm2_dispatch(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1,
                              void* Y2, int rows, int H, long long ld, float eps, int dtype, hipStream_t st);
int vidi_gemm_dispatch(const GemmParams& p, int batch, int mode, int repkv, int tile_cfg, int dtype, hipStream_t st);
int vidi_gemv_glu_dispatch(const void* X, const void* W, void* Y, int M, int I, int K, int ldx, int ldw, int ldy, int act, int dtype, hipStream_t st);
int vidi_gemv_norm2_dispatch(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1,
                             long long ld, float eps, const void* W, void* Y, int M, int N, int K, int ldw, int ldy, int glu_act,
                             int dtype, hipStream_t st);
int vidi_gemv_dispatch(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int dtype, hipStream_t st);
int vidi_gemm_f32_dispatch(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int ldx, int ldw, int ldy, int act, hipStream_t st);
int vidi_w4_patch(const GemmParams& p, int dtype, hipStream_t st);            // gemm_w4_patch.hip
int vidi_w4_window(const GemmParams& p, int dtype, hipStream_t st);           // gemm_w4_patch.hip
int vidi_attn_self_dispatch(const AttnSelfParams& p, int D, int dtype, hipStream_t st);
int vidi_attn_self_rm_dispatch(const AttnSelfRmParams& p, int D, int dtype, hipStream_t st);
#define VIDI_XROWS_MAX_CAP2 96.0f      // largest softcap x log2(e) the fixed-reference softmax of attn_cross_rows.hip is exact for (see there)
int vidi_attn_cross_rtpb(int Rpad, float softcap, int dtype);     // row tiles a block of the cross-attention launch covers (1 or 4)
int vidi_attn_cross_dispatch(const AttnCrossParams& p, int HD, int zsplit, int dtype, hipStream_t st);
int vidi_attn_cross2_dispatch(const AttnCrossParams& a, const AttnCrossParams& b, int HD, int za, int zb, int dtype, hipStream_t st);
int vidi_attn_merge_dispatch(const AttnMergeParams& p, int HD, int dtype, hipStream_t st);
int vidi_attn_merge2_dispatch(const AttnMergeParams& a, const AttnMergeParams& b, int HD, int dtype, hipStream_t st);
int vidi_attn_text_dispatch(const AttnTextParams& p, int HD, int dtype, hipStream_t st);
int vidi_attn_text_decode_dispatch(const void* qkv, int ldqkv, void* Kc, void* Vc, const void* kmask, const void* cs, const void* sn,
                                   void* O, int B, int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int window,
                                   float scale, float softcap, int dtype, hipStream_t st);
struct AttnTextDecodeParams;
int vidi_attn_text_decode_merge2_dispatch(const AttnTextDecodeParams& tp, size_t lds, const AttnMergeParams& a, const AttnMergeParams& b, int HD,
                                          int dtype, hipStream_t st);
int vidi_rope_dispatch(void* Q, void* K, const void* cs, const void* sn, int rows, int nq, int nkv, int HD, int dtype, hipStream_t st);
int vidi_rope_cache_dispatch(const void* qkv, int ldqkv, void* QR, void* Kc, void* Vc, const void* cs, const void* sn, int B, int Lq,
                             int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int dtype, hipStream_t st);
int vidi_norm_dispatch(const NormParams& p, int mode, int dtype, hipStream_t st);
int vidi_ln_finalize_dispatch(const float* part, float* stats, long long rows, int nstr, int H, float eps, hipStream_t st);
int vidi_row_partials_dispatch(const void* Y, float* part, long long rows, int N, long long ldy, int nstr, int dtype, hipStream_t st);
int vidi_w4n_stat_strips(int N);                                              // (sum, sum^2) entries per row of GemmParams::stat_part for an output width N
int vidi_w4n_ln_heads(const GemmParams& p, int dtype, hipStream_t st);       // ... its LN-fold + head-major epilogue (q | k | v at N = 12 x 288)
int vidi_w4n_bias_res(const GemmParams& p, int dtype, hipStream_t st);       // 288 x 224 tile geometry (gemm_w4n.h); VIDI_W4_UNSUPPORTED (-100) when it does not apply
int vidi_row_stats_dispatch(const void* X, float* stats, long long rows, int H, long long ldx, float eps, int dtype, hipStream_t st);
int vidi_ew_dispatch(int op, void** a, const long long* i, const float* f, int dtype, hipStream_t st);
int vidi_sinusoid_dispatch(float* pe, const float* div, int rows, int i0, int l, int N, int d, hipStream_t st);
