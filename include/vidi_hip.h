/* vidi_hip.h — C ABI of libvidi_hip.so: the MI355X (gfx950) kernels behind the Vidi inference hot path.
 *
 * The reference (bytedance/vidi) is pure Python; it has no FFI of its own.  The seam this ABI sits
 * under is the set of third-party kernel CALL SITES of the inference forward:
 * every entry point below names the reference call site(s) whose arithmetic it replaces.  Paths
 * are relative to Vidi1.5_9B/vidi/ ; "TP/" = transformers (pinned 4.50.0 by the reference).
 *
 * Conventions
 *   - plain C: raw device pointers, ints, floats, a hipStream_t passed as void*.  No torch types.
 *   - every call only ENQUEUES on `stream` (no allocation, no sync, graph-capture safe).
 *   - return 0 on success; negative = VIDI_ERR_* (bad shape/dtype/alignment/argument);
 *     positive = hipError_t from the launch.  Nothing throws across the ABI.
 *   - dtype: VIDI_DT_BF16 / VIDI_DT_F16 select the storage + MFMA input type (fp32 accumulate).
 *   - "T(x)" below means "rounded to the storage dtype": kernels round where eager PyTorch rounds.
 */
#ifndef VIDI_HIP_H
#define VIDI_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VIDI_ABI_VERSION 5 /* 2: vidi_softcap_argmax takes a caller-owned workspace (the library holds no device state); 3: + vidi_gemm_skinny; 4: (a lab-only revision: LayerNorm statistics in the consumer's K loop, measured 28 % slower, never shipped); 5: + vidi_gemv_mfma (a batch of decode rows on the matrix pipe), vidi_attn_cross_row_tiles_per_block, vidi_probe_* */
#define VIDI_DT_BF16 0
#define VIDI_DT_F16 1
#define VIDI_DT_F32 2 /* output type of the preprocessing kernels only */
#define VIDI_OK 0
#define VIDI_ERR_SHAPE (-1)
#define VIDI_ERR_DTYPE (-2)
#define VIDI_ERR_ALIGN (-3)
#define VIDI_ERR_ARG (-4)

#define VIDI_ACT_NONE 0
#define VIDI_ACT_GELU_TANH 1
#define VIDI_ACT_GELU_ERF 2
#define VIDI_ACT_SILU 3

#define VIDI_NORM_GEMMA 0      /* TP/models/gemma2/modeling_gemma2.py:55-63                        */
#define VIDI_NORM_GEMMA_ADD 1  /* residual + Gemma2RMSNorm(x): lmm/dattn/gemma.py:121,201,237       */
#define VIDI_NORM_MM 2         /* model/mm_layer/norm.py:19-25 (weight * rms_norm(x))               */
#define VIDI_NORM_MM_NOW 3     /* model/mm_layer/norm.py:9-16 (weight-free rms_norm)                */
#define VIDI_NORM_LLM 4        /* multimodal.py:201-206 (+ gemma.py:353-356 normalizer), emits mask */
#define VIDI_NORM_LAYER 5      /* nn.LayerNorm of the SigLIP / Whisper towers                       */

int vidi_abi_version(void);
const char* vidi_build_info(void);

/* ---- dense projections: nn.Linear / Conv-as-GEMM call sites (cuBLAS in the reference) ------------
 * Y[m][n] = epi( sum_k X[m][k] W[n][k] + bias[n] ),  X:[M,K] (ldx), W:[N,K] (ldw), Y:[M,N] (ldy).
 * epi: T(.) -> act -> optional residual add  Y = T(T(.) + R[m % rmod][n]).
 * batch>1 strides X/Y/R by bsX/bsY/bsR elements (W shared).  K % 64 == 0, N % 32 == 0.
 * repkv_hd/repkv_g != 0: X column for logical k is (k/(g*hd))*hd + k%hd, i.e. the GEMM consumes
 *   repeat_kv(V) without materialising it (gemma.py:77-78,96,196-197).
 * tile_cfg: -1 auto, 0 = 128x128, 1 = 128x256, 2 = 256x256, 3 = conservative register-staged.
 * Replaces: SigLIP q/k/v/out/fc1/fc2 + patch-embed conv (TP/models/siglip/modeling_siglip.py:124-130,
 * 250-357), Whisper convs/linears (TP/models/whisper/modeling_whisper.py:566-567,279-282,375-376),
 * projector MLP (model/mm_layer/mlp.py:9-28), Conv1d audio pool (multimodal.py:85-88,232),
 * o_proj on V (gemma.py:196-197), down_proj, text q/k/v/o (TP gemma2:224-236). */
int vidi_gemm(const void* X, const void* W, const void* bias, void* Y, const void* R,
              int M, int N, int K, int ldx, int ldw, int ldy, int ldr, int rmod,
              long long bsX, long long bsY, long long bsR, int batch,
              int act, int repkv_hd, int repkv_g, int tile_cfg, int dtype, void* stream);

/* Gemma2MLP gate/up + GeGLU fused (TP gemma2:79-82 via gemma.py:116-123):
 * Wgu:[2*I,K] holds gate/up rows interleaved in blocks of 32 (rows 64j..64j+31 = gate[32j..],
 * rows 64j+32..64j+63 = up[32j..]);  Y[m][i] = T( T(gelu_tanh(T(g))) * T(u) ),  Y:[M,I]. */
int vidi_gemm_geglu(const void* X, const void* Wgu, void* Y, int M, int I, int K, int ldx, int ldw, int ldy,
                    int tile_cfg, int dtype, void* stream);

/* Encoder QKV projection: columns < vstart (Q|K) go row-major to Yqk (ldy); columns >= vstart (V)
 * go to Vt[b][head][d][seqpad] (b = m / seq, key order perm16 inside each 16-slab) for vidi_attn_self.
 * Replaces SiglipAttention / WhisperAttention q/k/v_proj (TP siglip:277-279, whisper:309-333). */
int vidi_gemm_qkv_vt(const void* X, const void* W, const void* bias, void* Yqk, void* Vt,
                     int M, int N, int K, int ldx, int ldw, int ldy,
                     int vstart, int hd, int seq, int seqpad, int nheads, int tile_cfg, int dtype, void* stream);

/* LayerNorm folded into the projection that consumes it (the encoder towers: HF SiglipEncoderLayer layer_norm1 -> q/k/v_proj and
 * layer_norm2 -> mlp.fc1, TP siglip:305-357; WhisperEncoderLayer self_attn_layer_norm / final_layer_norm, TP whisper:386-411):
 *   vidi_row_stats   stats[m] = (mean_m, rsqrt(var_m + eps)) of row m of X (fp32, two-pass variance) — ONE read of X;
 *   vidi_gemm_ln / vidi_gemm_qkv_vt_ln   take the UN-normalised X, the weight with the LayerNorm gain folded in (Wf[n][k] =
 *   T(W[n][k] * gamma[k])) and two fp32 vectors colsum[n] = sum_k Wf[n][k], shift[n] = sum_k W[n][k] * beta[k] + bias[n], and
 *   compute  Y[m][n] = act( rstd_m * (sum_k X[m][k] Wf[n][k] - mean_m * colsum[n]) + shift[n] )  in the fp32 epilogue ==
 *   act(Linear(LayerNorm(x))) without ever writing LayerNorm(x) (one row pass instead of a read + a write per LayerNorm).
 *   act: VIDI_ACT_NONE / GELU_TANH / GELU_ERF.  Layout of the qkv variant as vidi_gemm_qkv_vt.
 *   vidi_gemm_res_stats  is the PRODUCER of such an X (SiglipEncoderLayer out_proj / fc2 + residual, TP siglip:345-356; Whisper
 *   likewise): Y = X W^T + bias + R, and for every stored row the partial sums part[m][entry] = (sum y, sum y^2) of the values it
 *   stored, over vidi_stat_strips(N) groups of columns per row (128-column strips, or the column groups of the 288-wide tile
 *   geometry that serves N = 1152: the caller sizes `part` as M * vidi_stat_strips(N) * 2 floats); vidi_ln_finalize turns them into
 *   stats[m] = (mean, rstd).  With it the next LayerNorm costs no pass over Y at all.  (Small problems: the partials come from one
 *   pass over Y inside the call.) */
int vidi_stat_strips(int N);
int vidi_row_stats(const void* X, float* stats, long long rows, int H, long long ldx, float eps, int dtype, void* stream);
int vidi_gemm_res_stats(const void* X, const void* W, const void* bias, void* Y, const void* R, float* part,
                        int M, int N, int K, int ldx, int ldw, int ldy, int ldr, int tile_cfg, int dtype, void* stream);
int vidi_ln_finalize(const float* part, float* stats, long long rows, int N, float eps, void* stream);
int vidi_gemm_ln(const void* X, const void* Wf, const float* stats, const float* colsum, const float* shift, void* Y,
                 int M, int N, int K, int ldx, int ldw, int ldy, int act, int tile_cfg, int dtype, void* stream);
/* vidi_gemm_ln with the HEAD-MAJOR output vidi_attn_self_rm reads: N = 3 * heads * hd columns (q | k | v), M = frames * seq rows ->
 * Y[which][frame][head][token][d] (contiguous, same size as [M, N]). */
int vidi_gemm_ln_heads(const void* X, const void* Wf, const float* stats, const float* colsum, const float* shift, void* Y,
                       int M, int N, int K, int ldx, int ldw, int seq, int hd, int tile_cfg, int dtype, void* stream);
int vidi_gemm_qkv_vt_ln(const void* X, const void* Wf, const float* stats, const float* colsum, const float* shift, void* Yqk, void* Vt,
                        int M, int N, int K, int ldx, int ldw, int ldy,
                        int vstart, int hd, int seq, int seqpad, int nheads, int tile_cfg, int dtype, void* stream);

/* LLM K/V projection of the multimodal stream straight into the cross-attention caches
 * (gemma.py:59-65: k_proj, v_proj, DynamicCache.update).  W:[2*kvd,K] = [Wk;Wv].
 * Kc[kvh][tile64][64][hd], Vtc[kvh][tile32][hd][32 (perm16)] (2*ntile64 sub-tiles), Vrow:[M,kvd] row-major copy of V for the
 * diagonal-stream o_proj.  Token index = tok0 + m. */
int vidi_gemm_kv_cache(const void* X, const void* W, void* Kc, void* Vtc, void* Vrow,
                       int M, int kvd, int K, int ldx, int ldw, int hd, int ntile64, int tok0,
                       int tile_cfg, int dtype, void* stream);

/* Skinny projection (M <= 8), HBM-bound weight streaming for decode: text q/k/v/o/gate/up/down and
 * lm_head (gemma.py:565) at Lq = 1. */
int vidi_gemv(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy,
              int dtype, void* stream);
/* Projection of a FEW rows (8 < M <= 128: the text prompt at prefill — HF Gemma2Attention q/k/v/o and Gemma2MLP down_proj through
 * gemma.py:165-175, 116-123 at Lq > 1; vidi_gemm serves any M, this entry point streams the weights at several times its rate for
 * these M): Y[M,N] = X W^T (+ bias), split-K with fp32 partial sums in the caller's workspace, rounded once.
 * vidi_gemm_skinny_workspace_bytes: bytes of `workspace` for (M, N, K), or 0 when the shape is not taken (M outside 1..128, N % 64,
 * K % 256) — then call vidi_gemm.  The library keeps no state: the workspace must stay untouched until the call's work on `stream`
 * is done. */
size_t vidi_gemm_skinny_workspace_bytes(int M, int N, int K);
int vidi_gemm_skinny(const void* X, const void* W, const void* bias, void* Y, void* workspace, int M, int N, int K, int ldx, int ldw,
                     int ldy, int dtype, void* stream);
/* The same projections for a BATCH of decode rows (several queries sharing one video, BASELINE configs[4]: 8 rows; their o_proj over the
 * three attention streams: 24 rows) on the matrix pipe — vidi_gemv's FMAs are VALU work that saturates near M = 8.  One pass over W, no
 * workspace, 1 <= M <= 32: Y[M,N] = X W^T;  with glu_act = VIDI_ACT_GELU_TANH / VIDI_ACT_SILU (M <= 16): vidi_gemv_glu's gated pair on the
 * interleaved gate/up weight (N = I features).  glu_act < 0: plain.  vidi_gemv_mfma_fits: 1 when (M, N, K) is taken (N % 16 — % 32 for the gated pair —, K % 64),
 * otherwise call vidi_gemv / vidi_gemm.  Values: vidi_gemv's up to the fp32 summation order. */
int vidi_gemv_mfma_fits(int M, int N, int K, int glu);
int vidi_gemv_mfma(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int glu_act, int dtype,
                   void* stream);
/* Skinny gated-MLP front half (M <= 8): Y[m][i] = T( T(act(T(gate_i . x_m))) * T(up_i . x_m) ) on the vidi_gemm_geglu weight
 * layout (gate/up rows interleaved in blocks of 32); act = VIDI_ACT_GELU_TANH (Gemma2MLP) or VIDI_ACT_SILU (MistralMLP).
 * Same values as vidi_gemv + vidi_geglu_unpack / vidi_glu_unpack, one launch. */
int vidi_gemv_glu(const void* X, const void* Wgu, void* Y, int M, int I, int K, int ldx, int ldw, int ldy, int act, int dtype,
                  void* stream);
/* Decode: vidi_resid_norm2 FUSED INTO the skinny projection that consumes its second output (gemma.py:236-237 + :118 -> Gemma2MLP
 * gate/up, and :120-121 + the next layer's :162 -> q/k/v):  s = T(T(A+B)+C) (B, C optional);  Y1 = T(Res + T(gemma(s; W1)));
 * x = T(gemma(Y1; W2));  Y = x W^T  (vidi_gemv_norm2)  or  Y = act(x Wg^T) * (x Wu^T) on the interleaved gate/up weight
 * (vidi_gemv_glu_norm2).  Every block derives x itself under the HBM latency of its first weight rows; block 0 writes Y1.
 * M <= 4 rows, K <= 4096, Y1 MUST NOT alias Res (VIDI_ERR_ARG).  Element arithmetic and rounding points are vidi_resid_norm2's; the
 * sums of squares are reduced in a different order, so Y1 / x may differ from it in the last bit of the dtype. */
int vidi_gemv_norm2(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1, long long ld,
                    float eps, const void* W, void* Y, int M, int N, int K, int ldw, int ldy, int dtype, void* stream);
int vidi_gemv_glu_norm2(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1, long long ld,
                        float eps, const void* Wgu, void* Y, int M, int I, int K, int ldw, int ldy, int act, int dtype, void* stream);

/* fp32 projection on exact-fp32 MFMA: LearnablePosEmbd's fp32 MLP (mm_vision/pos.py:36-39,55;
 * model/mm_layer/mlp.py:31-40). act: VIDI_ACT_NONE / VIDI_ACT_GELU_ERF. */
int vidi_gemm_f32(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K,
                  int ldx, int ldw, int ldy, int act, void* stream);

/* ---- attention --------------------------------------------------------------------------------
 * Encoder self-attention, non-causal (flash-attn call inside HF Siglip/Whisper attention,
 * multimodal.py:44-57).  QK:[B*N, ldqk] (Q at col h*D, K at col koff+h*D), Vt from vidi_gemm_qkv_vt,
 * O:[B*N, ldo]. D in {72, 64, 32, 16}. */
int vidi_attn_self(const void* QK, const void* Vt, void* O, int B, int N, int Npad, int H, int D,
                   int ldqk, int koff, int ldo, float scale, int dtype, void* stream);

/* The same attention reading Q, K AND V from ONE projection output with V in natural order: V is transposed on the fly by the LDS
 * transpose read (ds_read_b64_tr_b16), so the q/k/v projection needs no scattered V^T stores.  Row r of head h of frame b lies at
 * base + b*bs + h*hs + r*ld (elements), base = QKV (Q), QKV + koff (K), QKV + voff (V):
 *   row-major  [B*N, ld]            bs = 0 (-> N*ld), hs = 0 (-> D), koff / voff = column offsets            (vidi_gemm / vidi_gemm_ln output)
 *   head-major [3][B][H][N][D]      ld = D, bs = H*N*D, hs = N*D, koff = B*H*N*D, voff = 2*B*H*N*D           (vidi_gemm_ln_heads output)
 * Head-major makes every K / V tile whole 128-byte lines (+12 % on this kernel).  O:[B*N, ldo] row-major, head h at column h*D.
 * Same results as vidi_attn_self (bit-identical: same MFMA operands in the same slots).
 * scale > 0: softmax(scale * q.k).  scale <= 0: the caller has folded scale * log2(e) into Q (e.g. into the q projection's weights): the
 * kernel computes 2^(q.k) normalised — the same function, without a per-score multiply; with D = 72 the running maximum then rides
 * in the spare contraction chunk of the QK^T product (rounded to the storage dtype), so the exponent's argument comes straight out of
 * the matrix pipe. */
int vidi_attn_self_rm(const void* QKV, void* O, int B, int N, int H, int D, int ld, long long koff, long long voff, long long bs, long long hs,
                      int ldo, float scale, int dtype, void* stream);

/* Text->video / text->audio cross-attention, split-KV partial pass (flash_attn_func /
 * flash_attn_varlen_func: lmm/dattn/xattn.py:123,253 via gemma.py:81-91).  Rows r = token*G + g
 * for each kv head; keys [key_start, key_start+n_keys) of the tiled caches; mask: optional
 * uint8[n_keys] key-padding mask (image/audio_attention_mask).  Writes W = zsplit partials (one per block):
 * Opart:[W][nkv][Rpad][HD] fp32, ML:[W][nkv][Rpad][2] fp32 (base-2 running max, sum). */
size_t vidi_attn_cross_workspace_bytes(int zsplit, int nkv, int Rpad, int HD);
/* 32-row tiles ONE block of the launch covers for a launch of Rpad rows per kv head: 1 (a block = one row tile, its four waves split the
 * key slice) or 4 (two row tiles and more — a prompt, a batch of prompts — with a logit softcap, in bf16: the block's four waves own four
 * row tiles and share one K / V stream, so the keys are read once per four tiles; the softmax there runs against a fixed reference,
 * which needs the softcap's bound on the logits and bf16's exponent range).  The launch runs nkv x ceil(Rpad / 32 / this) x zsplit
 * blocks: size zsplit with it. */
int vidi_attn_cross_row_tiles_per_block(int Rpad, float softcap, int dtype);
int vidi_attn_cross(const void* Q, const void* Kc, const void* Vtc, const void* mask, float* Opart, float* ML,
                    int R, int Rpad, int G, int nkv, int HD, int ldq, int ntile64, int key_start, int n_keys,
                    float scale, float softcap, int zsplit, int dtype, void* stream);
/* T2V and T2A of one layer in ONE launch (gemma.py:81-91 runs them back to back): two vidi_attn_cross sweeps over disjoint key regions
 * [key_startA, +n_keysA) and [key_startB, +n_keysB) of the same caches, with zsplitA / zsplitB key slices and their own masks and
 * workspaces; identical partials to the two separate calls.  At decode a launch costs ~8 us on top of its bytes. */
int vidi_attn_cross2(const void* Q, const void* Kc, const void* Vtc,
                     const void* maskA, float* OpartA, float* MLA, int key_startA, int n_keysA, int zsplitA,
                     const void* maskB, float* OpartB, float* MLB, int key_startB, int n_keysB, int zsplitB,
                     int R, int Rpad, int G, int nkv, int HD, int ldq, int ntile64, float scale, float softcap, int dtype, void* stream);
/* Merge partials -> Out:[tokens, ldo] (head (kvh*G+g) at column (kvh*G+g)*HD).  Optional OutF32
 * [nkv][Rpad][HD] / OutML [nkv][Rpad][2] receive the merged result in PARTIAL form (numerator, m, l)
 * — one slice of the Opart/ML layout — so per-GPU results can be all-gathered and merged again.
 * zero_out=1 reproduces gemma.py:180-192 for a sample with no valid key. */
int vidi_attn_merge(const float* Opart, const float* ML, void* Out, float* OutF32, float* OutML,
                    int W, int nkv, int R, int Rpad, int G, int HD, int ldo, int zero_out, int dtype, void* stream);
/* Two merges in one launch: the T2V (A) and T2A (B) partials of one decoder layer (same R / nkv / G / ldo). */
int vidi_attn_merge2(const float* OpartA, const float* MLA, void* OutA, int WA, int zeroA,
                     const float* OpartB, const float* MLB, void* OutB, int WB, int zeroB,
                     int nkv, int R, int Rpad, int G, int HD, int ldo, int dtype, void* stream);
/* Frame-sharded (multi-GPU) form of the two merges, SURVEY 8(e): the keys of a video are sharded over ranks, so what the
 * reference gets from ONE flash_attn_func call over all keys (gemma.py:81-91) is assembled from per-rank partials.
 * Every set (A = T2V, B = T2A) reads W partials whose slices are wsO / wsML floats apart (a rank's own zsplit partials:
 * nkv*Rpad*HD / nkv*Rpad*2; the all-gathered packed buffers of all ranks: the packed per-rank length), rows Rpad apart
 * inside a slice, and writes Out (model dtype, may be null) and/or the partial form OutF32 [nkv][rpo][HD] + OutML
 * [nkv][rpo][2] (rpo >= R: row stride of the packed buffer).  W == 0 emits the neutral partial (m = -inf, l = 0) of a
 * rank that holds no key of the modality; a set with neither Out nor OutF32 is skipped. */
int vidi_attn_merge2_sharded(const float* OpartA, const float* MLA, long long wsOA, long long wsMLA, void* OutA, float* OutF32A,
                             float* OutMLA, int WA, int zeroA,
                             const float* OpartB, const float* MLB, long long wsOB, long long wsMLB, void* OutB, float* OutF32B,
                             float* OutMLB, int WB, int zeroB,
                             int nkv, int R, int Rpad, int rpo, int G, int HD, int ldo, int dtype, void* stream);

/* Text causal self-attention with softcap / sliding window / key mask (gemma.py:165-175 ->
 * TP gemma2:248-288 under FA2) over the text KV cache [B,Lmax,nkv*HD]. */
int vidi_attn_text(const void* Q, const void* Kc, const void* Vc, const void* kmask, void* O,
                   int B, int Lq, int Lmax, int nq, int nkv, int HD, int past_len, int window,
                   float scale, float softcap, int dtype, void* stream);
/* Same, with the number of cached keys read from device memory at run time (`past_len_dev`, int32): nothing
 * in the launch depends on the decode position, so a greedy decode step can be captured in a hipGraph and
 * replayed (own generate() loop without per-token host work; replaces the HF loop gemma.py:646-687). */
int vidi_attn_text_dyn(const void* Q, const void* Kc, const void* Vc, const void* kmask, void* O,
                       int B, int Lq, int Lmax, int nq, int nkv, int HD, const int* past_len_dev, int window,
                       float scale, float softcap, int dtype, void* stream);
/* The single-token decode step's T2T in one launch: rope(q), rope(k) of the new token (TP gemma2:146-168), the KV-cache append
 * (TP gemma2:262-275) and the attention of its nq heads over cache slots [max(0, pos - window), pos] (gemma.py:165-175) — what
 * vidi_rope_cache followed by vidi_attn_text / vidi_attn_text_dyn compute at Lq = 1 (same scores and probabilities; the fp32
 * summation order of the dot products differs).  qkv [B][ldqkv] = (q | k | v); cos/sin [B][HD]; O [B][nq*HD]; pos = *pos_dev when
 * pos_dev is non-null (hipGraph-capturable), else pos0.  VIDI_ERR_SHAPE when the scores of the visible keys do not fit 64 KB of LDS
 * (callers then use the two-launch form). */
int vidi_attn_text_decode(const void* qkv, int ldqkv, void* Kc, void* Vc, const void* kmask, const void* cos_, const void* sin_, void* O,
                          int B, int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int window, float scale, float softcap,
                          int dtype, void* stream);
/* vidi_attn_text_decode and vidi_attn_merge2 in ONE launch (the decode step runs the T2V + T2A partial pass first): neither fills the
 * chip and neither depends on the other; the o_proj that follows (gemma.py:94) needs both.  Arguments as in the two calls; results
 * bit-identical to them.  HD 128 or 256. */
int vidi_attn_text_decode_merge2(const void* qkv, int ldqkv, void* Kc, void* Vc, const void* kmask, const void* cos_, const void* sin_, void* O,
                                 int B, int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int window, float scale, float softcap,
                                 const float* OpartA, const float* MLA, void* OutA, int WA, int zeroA,
                                 const float* OpartB, const float* MLB, void* OutB, int WB, int zeroB,
                                 int R, int Rpad, int ldo, int dtype, void* stream);
/* apply_rotary_pos_emb in place (TP gemma2:146-168); cos/sin:[rows,HD] in the storage dtype. */
int vidi_rope(void* Q, void* K, const void* cos_, const void* sin_, int rows, int nq, int nkv, int HD,
              int dtype, void* stream);
/* RoPE + text KV-cache append fused (Gemma2Attention.forward: apply_rotary_pos_emb then past_key_value.update, TP gemma2:262-275):
 * qkv [B*Lq][ldqkv] = (q | k | v) -> QR [B*Lq][nq*HD] = rope(q); Kc[b][pos0+i] = rope(k); Vc[b][pos0+i] = v, caches
 * [B][Lmax][nkv*HD].  pos_dev (device int, may be null) overrides pos0: capturable in a hipGraph. */
int vidi_rope_cache(const void* qkv, int ldqkv, void* QR, void* Kc, void* Vc, const void* cos_, const void* sin_, int B, int Lq,
                    int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int dtype, void* stream);

/* ---- Vidi-7B (Mistral) variants of the same path (Vidi_7B/model/lmm/dattn/mistral.py) ---------------- */
/* Gated MLP with selectable activation: act = VIDI_ACT_GELU_TANH (Gemma2MLP, same as vidi_gemm_geglu) or
 * VIDI_ACT_SILU (MistralMLP, mistral.py:131-137 via transformers MistralMLP): Y = T(act(T(x Wg^T))) * T(x Wu^T). */
int vidi_gemm_glu(const void* X, const void* Wgu, void* Y, int M, int I, int K, int ldx, int ldw, int ldy,
                  int act, int tile_cfg, int dtype, void* stream);
/* Decode-path companion: applies the gate to the raw interleaved [M, 2I] GEMV output. */
int vidi_glu_unpack(const void* Yp, void* out, int M, int I, int act, int dtype, void* stream);
/* Learned Conv2DPool of Vidi-7B (Vidi_7B/model/mm_vision/pool.py:19-26) = im2col (this) + vidi_gemm with the
 * [d_out, k*k*C] repacked kernel + align_corners=True bilinear resize (below).  x:[T, side*side, C] tower
 * features, out:[T*(side-k+1)^2, k*k*C], column = (dy*k+dx)*C + c. */
int vidi_im2col_nhwc(const void* x, void* out, int T, int side, int C, int k, int dtype, void* stream);
/* F.interpolate(mode="bilinear", align_corners=True) on NHWC: x:[T,s_in,s_in,C] -> out:[T,s_out,s_out,C]. */
int vidi_resize_bilinear_ac(const void* x, void* out, int T, int s_in, int s_out, int C, int dtype, void* stream);

/* ---- row-wise normalisations (see VIDI_NORM_*) ------------------------------------------------ */
int vidi_norm(int mode, const void* X, const float* XF32, const void* W, const void* Bias, const void* Res,
              void* Y, void* Mask, int rows, int H, long long ldx, long long ldy, long long ldr,
              float eps, float normalizer, const int* sample_flag /* device int, null = 1 */, int dtype, void* stream);
/* Text-stream fusion of the Gemma2 wiring, one pass per row, results identical to vidi_add3 -> vidi_norm(GEMMA_ADD) ->
 * vidi_norm(GEMMA):  s = T(T(A+B)+C) (B, C optional);  Y1 = T(Res + T(gemma(s; W1)));  Y2 = T(gemma(Y1; W2)).
 * gemma.py:236-237 + :118 (attention side) and :120-121 + the next layer's :162 / the final :411 (FFN side).
 * All tensors [rows, H] with row stride ld; Y1 may alias Res. */
int vidi_resid_norm2(const void* A, const void* B, const void* C, const void* Res, const void* W1, const void* W2, void* Y1, void* Y2,
                     int rows, int H, long long ld, float eps, int dtype, void* stream);

/* SiglipVisionEmbeddings (TP siglip/modeling_siglip.py:124-130, 178): Conv2d(3, N, kernel = stride = P, valid) over px:[T,3,S,S] + bias
 * + position table pos:[(S/P)^2, N] -> Y:[T*(S/P)^2, N], ONE launch of the persistent GEMM whose activation loader reads the NCHW
 * pixels directly (16-byte LDS-DMA pieces of 8 contiguous pixels; no im2col buffer).  W:[N, K] is the conv weight re-laid as
 * k = (c*P + dy)*16 + dx (dx < P; columns dx >= P and k >= 3*P*16 are zero), K a multiple of 64 >= 3*P*16, P <= 16. */
int vidi_patch_embed(const void* px, const void* W, const void* bias, const void* pos, void* Y, int T, int S, int P, int N, int K,
                     int ldw, int ldy, int ldpos, int dtype, void* stream);

/* Learned Conv2DPool of Vidi-7B (Vidi_7B/model/mm_vision/pool.py:19-26): Conv2d(C, N, kernel k, stride 1, valid, no bias) over the tower's
 * token-major features f:[T, side*side, C] -> Y:[T*(side-k+1)^2, N], the k x k window gathered by the persistent GEMM's loader (64
 * contiguous channels of one window position per K slice; no im2col buffer).  W:[N, k*k*C] with k index (dy*k + dx)*C + c (the layout
 * vidi_im2col_nhwc + vidi_gemm used), C % 64 == 0. */
int vidi_conv_window(const void* f, const void* W, void* Y, int T, int side, int C, int k, int N, int ldw, int ldy, int dtype, void* stream);

/* ---- data movement / elementwise -------------------------------------------------------------- */
/* SiglipVisionEmbeddings conv as GEMM input (TP siglip:124-130,178): px:[T,3,S,S] -> A:[T*(S/P)^2,Kpad] */
int vidi_im2col_patch(const void* px, void* A, int T, int S, int P, int Kpad, int dtype, void* stream);
/* Conv2DPool.forward + space_to_depth (mm_vision/pool.py:23-32, utils.py:134-150):
 * f:[T,side*side,C] -> out:[T,h/m,w/m,C*m*m]; resize=0 means hw == padded grid (the "28" sentinel). */
int vidi_pool_s2d(const void* f, void* out, int T, int side, int C, int h, int w, int m, int resize, int dtype, void* stream);
/* f[t,y,x,:] = T(T(T(f+ph[y])+pw[x])+pt[t]) in place (multimodal.py:194-197,242); null tables skipped */
int vidi_add_pos(void* f, const void* ph, const void* pw, const void* pt, int T, int oh, int ow, int H, int dtype, void* stream);
/* y = T(T(a+b)+c), b/c optional (gemma.py:236) ; n elements, n % 8 == 0 */
int vidi_add3(const void* a, const void* b, const void* c, void* y, long long n, int dtype, void* stream);
/* embed_tokens gather * normalizer (multimodal.py:385, gemma.py:353-354); id < 0 -> zero row */
int vidi_embed(const long long* ids, const void* E, void* out, int n, int H, long long vocab, float normalizer, int dtype, void* stream);
/* GeGLU on the interleaved gate/up layout for the vidi_gemv path */
int vidi_geglu_unpack(const void* Yp, void* out, int M, int I, int dtype, void* stream);
/* final-logit softcap in place + greedy argmax (gemma.py:565-569, do_sample=False).  workspace: vidi_softcap_argmax_workspace_bytes(B)
 * bytes (8-byte aligned), zeroed ONCE by the caller; every call leaves it zeroed.  Calls on different streams need different workspaces
 * (the library keeps no device state: SURVEY 8b "stateless, re-entrant ... workspace passed in explicitly"). */
size_t vidi_softcap_argmax_workspace_bytes(int B);
int vidi_softcap_argmax(void* logits, long long* idx, int B, int V, long long ld, float cap, int dtype, void* workspace, void* stream);
/* mel:[C,nmel,L] -> [C,L+2,nmel] zero-padded rows for the conv1-as-GEMM view (TP whisper:566,618) */
int vidi_mel_transpose_pad(const void* mel, void* out, int C, int nmel, int L, int dtype, void* stream);
/* y = T(x*s): `embeds * normalizer` for externally supplied embeddings (gemma.py:353-356); n % 8 == 0 */
int vidi_scale(const void* x, void* y, long long n, float s, int dtype, void* stream);
/* *flag |= any(x != 0): the per-sample `sum(|x|) != 0` mask (multimodal.py:202,246); caller zeroes flag; x 16-byte aligned */
int vidi_any_nonzero(const void* x, long long n, int* flag, int dtype, void* stream);
/* FractionalSinusoidalEmbedding rows i0..i0+rows of l (mm_vision/pos.py:11-26,47-53), fp32 */
int vidi_sinusoid(float* pe, const float* div_term, int rows, int i0, int l, int N, int d, void* stream);

/* ---- host preprocessing moved to the GPU (dataset/img_utils.py:181-185, dataset/vid_utils.py:53-64) ---------------------
 * Frames: `image.resize((S, S), Image.BICUBIC)` + `image_processor.preprocess` — BIT-EXACT with Pillow's 8-bit resampler
 * (src/libImaging/Resample.c: two passes, int32 accumulators, PRECISION_BITS = 22) and with the processor's float
 * arithmetic.  The host supplies Pillow's per-output-position tables: bounds[out][2] = (first source index, tap count) and
 * kk[out][ksize] = fixed-point coefficients, and lut[3][256] = normalised value of every byte per channel in the output
 * element type (2 bytes: bf16/f16 bits, 4 bytes: f32).  `in`/`tmp` 4-byte aligned; tmp rows are `pitch` >= OW*3 bytes apart,
 * pitch % 4 == 0.
 *   pass 1: in [rows = T*H0][W0][3] u8 -> tmp [rows][pitch] u8 (OW pixels x 3)
 *   pass 2: tmp [T][H0][pitch] u8 -> out [T][3][OH][OW] */
int vidi_resize_h_u8(const void* in, void* tmp, const int* bounds, const int* kk, long long rows, int W0, int OW, int pitch,
                     int ksize, void* stream);
int vidi_resize_v_u8_norm(const void* tmp, void* out, const int* bounds, const int* kk, const void* lut, int T, int H0, int OW,
                          int OH, int pitch, int ksize, int out_elem_bytes, void* stream);
/* Audio: WhisperFeatureExtractor._torch_extract_fbank_features (TP/models/whisper/feature_extraction_whisper.py) as
 *   reflect pad -> STFT = vidi_gemm_f32 over overlapping row views (ldx = hop) against the Hann-windowed DFT matrix ->
 *   |.|^2 -> vidi_gemm_f32 against the mel filter bank -> log10 / per-window (max - 8) floor / (x + 4) / 4 / transpose.
 * reflect_pad: wave [C][n] -> out [C][stride], out[c][i] = wave[c][reflect(i - pad)] for i < n + 2*pad, 0 beyond.
 * power_spectrum: Y [M][ldy] = (re[0..nf) | im[nf..2nf)) -> P [M][ldp] = re^2 + im^2, columns >= nf zero.
 * logmel_finish: mel [C][R][nmel] f32 (first F rows of each window valid; overwritten with log10(max(.,1e-10))),
 *   cmax [C] scratch, out [C][nmel][F] in out_dtype (VIDI_DT_*). */
int vidi_reflect_pad_f32(const float* wave, float* out, int C, int n, int pad, int stride, void* stream);
int vidi_power_spectrum_f32(const float* Y, float* P, long long M, int nf, int ldy, int ldp, void* stream);
int vidi_logmel_finish(float* mel, float* cmax, void* out, int C, int R, int F, int nmel, int out_dtype, void* stream);

/* Box-speed reference (diagnostic, not on the Vidi path; csrc/probe.hip is frozen): bench.py prints these rates beside the metric so that
 * records from different boxes of a pool can be normalised.  vidi_probe_mfma: a register-operand bf16 MFMA 16x16x32 loop on every SIMD
 * (operands: 64 KB of bf16 values; out: 2048 * 512 floats; flop = 2048 * 8 * iters * 32 * 16384).  vidi_probe_hbm_read: one non-temporal
 * sweep over `bytes` (multiple of 16) of buf (out: 2048 * 256 uint32). */
int vidi_probe_mfma(const void* operands, void* out, int iters, void* stream);
int vidi_probe_hbm_read(const void* buf, void* out, long long bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
