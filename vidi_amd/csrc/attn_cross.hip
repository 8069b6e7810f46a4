// Text -> multimodal cross-attention (T2V / T2A of Decomposed Attention): skinny Q (Lq ~ 40 at
// prefill, 1 at decode) against ~10^5 cached video/audio keys.  Replaces the reference's
// flash_attn_func / flash_attn_varlen_func call sites (Vidi1.5_9B/vidi/model/lmm/dattn/xattn.py:123,253
// via gemma.py:81-91) — non-causal, logits softcap, key-padding mask, GQA.
//
// HBM-bound: every K/V byte is read once per (layer, modality): algorithmic bytes =
// n_keys * 2 * n_kv_heads * head_dim * 2 B.  Design:
//   * GQA-native: the G query heads sharing a KV head are stacked as rows (row = (token, g)), so
//     K/V are never repeat_kv'ed (the reference doubles the bytes, gemma.py:77-78).
//   * K cache is tile-contiguous  Kc[kvh][tile64][64 keys][HD];  V cache is stored transposed in 32-key
//     sub-tiles  Vtc[kvh][tile32][HD][32 positions]  with the perm16 key order, both written
//     by the KV-projection GEMM epilogue.  A wave's 32-key sub-tile is 16 KB + 16 KB of perfectly
//     linear 16-byte global_load_lds DMA.
//   * split-KV at WAVE granularity: every wave owns a contiguous key range and its private 32 KB
//     LDS ring and runs its own online softmax (no block barriers in the loop); the 4 waves of a block are merged
//     in LDS and the block emits one partial (O, m, l); `attn_merge_kernel` combines the blocks' partials (exact:
//     the tanh softcap is per logit).
//   * swapped MFMAs (S^T = K Q^T, O^T = Vt P^T): per-lane softmax statistics, P stays in registers.
#include "kernels.h"
#include "attn_text_decode.h"
#include <stdlib.h>

#ifndef VIDI_XATTN_INTERLEAVE
#define VIDI_XATTN_INTERLEAVE 1
#endif
#ifndef VIDI_XATTN_NT
#define VIDI_XATTN_NT 1
#endif


// z / zsplit: this block's key slice and the number of slices of ITS modality (the dual launch below runs the slices of two
// modalities in one grid)
template <typename T, int HD>
__device__ __forceinline__ void attn_cross_body(const AttnCrossParams& p, const int z, const int zsplit) {
    constexpr int QROW = HD * 2;                 // bytes per row
    constexpr int CPR = HD / 8;                  // 16-byte chunks per K/Q row (32 for HD=256)
    constexpr int KST = HD / 16;                 // k16 steps of QK^T
    constexpr int DT = HD / 32;                  // output d tiles
    constexpr int KBYTES = 32 * QROW;            // K sub-tile (32 keys)
    constexpr int VBYTES = HD * 64;              // Vt sub-tile (HD rows x 32 positions x 2 B)
    constexpr int KLD = KBYTES / 1024;           // DMA instructions per K sub-tile
    constexpr int VLD = VBYTES / 1024;
    static_assert(CPR >= 16 && (CPR & (CPR - 1)) == 0, "HD must be 128 or 256");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem;                             // [32 rows][QROW], chunk-swizzled
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    char* sK = smem + 32 * QROW + wave * (KBYTES + VBYTES);
    char* sV = sK + KBYTES;

    const int kvh = blockIdx.x, rt = blockIdx.y;
    const int r0 = rt * 32;

    // ---- stage the 32-row Q tile (rows beyond R are zero) -------------------------------------
    for (int i = tid; i < 32 * CPR; i += 256) {
        const int row = i / CPR, c = i % CPR, r = r0 + row;
        u32x4 v = {0, 0, 0, 0};
        if (r < p.R) {
            const int tq = r / p.G, g = r % p.G;
            v = *(const u32x4*)(p.Q + (size_t)tq * p.ldq + (kvh * p.G + g) * HD + c * 8);
        }
        *(u32x4*)(sQ + row * QROW + ((c ^ (row & 15)) << 4)) = v;
    }
    __syncthreads();

    // Q fragments stay in registers for the whole key sweep (B operand: column = row l31)
    u32x4 qf[KST];
#pragma unroll
    for (int ks = 0; ks < KST; ++ks)
        qf[ks] = *(const u32x4*)(sQ + l31 * QROW + (((2 * ks + hi) ^ (l31 & 15)) << 4));

    // ---- this wave's key range (32-key sub-tiles) ----------------------------------------------
    const int nsub = (p.n_keys + 31) / 32;
    const int wtot = zsplit * 4;
    const int wg = z * 4 + wave;
    // key slices INTERLEAVED over the waves (wave wg sweeps sub-tiles wg, wg + wtot, ...): at any moment the waves of a kv head read
    // one contiguous window of its K and of its V cache (wtot x 16 KB each) that slides through memory, instead of wtot separate
    // streams half a megabyte apart (VIDI_XATTN_INTERLEAVE=0: a contiguous range per wave, the round-1 form)
#if VIDI_XATTN_INTERLEAVE
    const int st_begin = min(wg, nsub), st_end = nsub, st_step = wtot;
#else
    const int per = (nsub + wtot - 1) / wtot;
    const int st_begin = min(wg * per, nsub), st_end = min(st_begin + per, nsub), st_step = 1;
#endif

    const u16* kc_head = p.Kc + (size_t)kvh * p.ntile64 * 64 * HD;
    const u16* vt_head = p.Vtc + (size_t)kvh * p.ntile64 * HD * 64;

    // streamed once: when a single row tile sweeps the keys (decode, short prompts) every K/V byte is read by exactly one CU, and the
    // non-temporal policy on the DMA shortens issued -> landed (MI355X_MICROARCH.md "nt-weights"); with several row tiles the later
    // tiles re-read the slice from L2 and the default policy is kept
    const bool nt = VIDI_XATTN_NT && gridDim.y == 1;
    auto issue_k = [&](int st) {
        const int kb = p.key_start + st * 32;
        const u16* src = kc_head + (size_t)kb * HD;                 // 32 consecutive keys are contiguous
        if (nt) {
#pragma unroll
            for (int j = 0; j < KLD; ++j) {
                const int pidx = j * 64 + lane, row = pidx / CPR, cl = pidx % CPR;
                glds16<2>(src + row * HD + (cl ^ (row & 15)) * 8, sK + j * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < KLD; ++j) {
                const int pidx = j * 64 + lane, row = pidx / CPR, cl = pidx % CPR;
                glds16(src + row * HD + (cl ^ (row & 15)) * 8, sK + j * 1024);
            }
        }
    };
    auto issue_v = [&](int st) {
        const int kb = p.key_start + st * 32;
        const u16* src = vt_head + (size_t)(kb >> 5) * HD * 32;          // [HD][32 positions]: 16 KB of linear 64-byte rows
        if (nt) {
#pragma unroll
            for (int j = 0; j < VLD; ++j) {
                const int pidx = j * 64 + lane, d = pidx >> 2, cl = pidx & 3;
                glds16<2>(src + d * 32 + (cl ^ ((d >> 2) & 3)) * 8, sV + j * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < VLD; ++j) {
                const int pidx = j * 64 + lane, d = pidx >> 2, cl = pidx & 3;
                glds16(src + d * 32 + (cl ^ ((d >> 2) & 3)) * 8, sV + j * 1024);
            }
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float L2E = 1.4426950408889634f;
    const bool use_cap = p.softcap > 0.f;
    const float pre = use_cap ? p.scale / p.softcap : p.scale * L2E;   // x/cap, or base-2 logits
    const float capl2 = p.softcap * L2E;

    if (st_begin < st_end) {
        issue_k(st_begin);
        issue_v(st_begin);
    }
    for (int st = st_begin; st < st_end; st += st_step) {
        const bool has_next = (st + st_step < st_end);
        // K(st) landed?  (V(st) may still be in flight)
        wait_vmcnt<VLD>();
        // ---- S^T = K Q^T : Q fragments are register-resident; K fragments stream from LDS in
        //      batches of 4 k16-steps, the next batch's reads issued ahead of this batch's MFMAs ----
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#ifdef VIDI_XATTN_DIAG_NOCOMPUTE                          // lab builds only (wrong results): the DMA / wait skeleton without MFMAs and softmax
        if (p.R < 0)
#endif
        {
            u32x4 kf[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                kf[0][e] = *(const u32x4*)(sK + l31 * QROW + ((((2 * e + hi)) ^ (l31 & 15)) << 4));
#pragma unroll
            for (int kb4 = 0; kb4 < KST / 4; ++kb4) {
                if (kb4 + 1 < KST / 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        kf[(kb4 + 1) & 1][e] = *(const u32x4*)(sK + l31 * QROW + (((2 * ((kb4 + 1) * 4 + e) + hi) ^ (l31 & 15)) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = T::mfma32(kf[kb4 & 1][e], qf[kb4 * 4 + e], s);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // all K-fragment reads are consumed by the MFMAs above -> the K ring slot is free
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (has_next) issue_k(st + st_step);

        // ---- logits: scale, softcap, mask; online softmax in base 2 --------------------------
        const int kb_local = st * 32;
        u32x4 pf0 = {0, 0, 0, 0}, pf1 = {0, 0, 0, 0};
#ifdef VIDI_XATTN_DIAG_NOCOMPUTE
        if (p.R < 0)
#endif
        {
        float mx = -INFINITY;
        if (use_cap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // cap*tanh(y) in base-2 units: tanh(y) = 1 - 2/(exp(2y)+1)
                const float e2 = __expf(2.0f * s[r] * pre);
                s[r] = capl2 * (1.0f - 2.0f / (e2 + 1.0f));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] *= pre;
        }
        if (kb_local + 32 > p.n_keys) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb_local + krow32(r, hi) >= p.n_keys) s[r] = -INFINITY;
        }
        if (p.mask) {
            // 32 mask bytes of this sub-tile (the cache region is padded to 64 keys => in bounds)
            const unsigned char* mp = p.mask + kb_local + 4 * hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned mv = *(const unsigned*)(mp + 8 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((mv >> (8 * e)) & 0xffu) == 0) s[4 * j + e] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : fast_exp2(m_run - m_use);
        m_run = m_new;
        float pv[16], psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pv[r] = fast_exp2(s[r] - m_use);
            psum += pv[r];
        }
        l_run = l_run * alpha + psum;
        pf0 = pack8<T>(pv);
        pf1 = pack8<T>(pv + 8);
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        }
        }
        // V(st) landed?  (K(st+1) may be in flight)
        if (has_next) wait_vmcnt<KLD>(); else wait_vmcnt<0>();
        // ---- O^T += Vt P^T : 2 d-tiles (4 fragments) per batch, reads one batch ahead ----------
#ifdef VIDI_XATTN_DIAG_NOCOMPUTE
        if (p.R < 0)
#endif
        {
            u32x4 vf[2][4];
            auto read_v = [&](int buf, int dt0) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int d = (dt0 + e) * 32 + l31;
                    const int swz = (d >> 2) & 3;
                    vf[buf][2 * e] = *(const u32x4*)(sV + d * 64 + (((0 + hi) ^ swz) << 4));
                    vf[buf][2 * e + 1] = *(const u32x4*)(sV + d * 64 + (((2 + hi) ^ swz) << 4));
                }
            };
            read_v(0, 0);
#pragma unroll
            for (int b2 = 0; b2 < DT / 2; ++b2) {
                if (b2 + 1 < DT / 2) read_v((b2 + 1) & 1, (b2 + 1) * 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    o[b2 * 2 + e] = T::mfma32(vf[b2 & 1][2 * e], pf0, o[b2 * 2 + e]);
                    o[b2 * 2 + e] = T::mfma32(vf[b2 & 1][2 * e + 1], pf1, o[b2 * 2 + e]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (has_next) issue_v(st + st_step);
    }

    // ---- merge the block's 4 waves in LDS (their private K/V rings are free now) and emit ONE partial ---------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float* so = (float*)sK;                                   // [32 rows][HD] fp32 = exactly this wave's ring bytes
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 ov = {o[dt][4 * j], o[dt][4 * j + 1], o[dt][4 * j + 2], o[dt][4 * j + 3]};
            *(f32x4*)(so + l31 * HD + dt * 32 + 8 * j + 4 * hi) = ov;
        }
    float* sml = (float*)sQ;                                  // Q tile is dead (fragments live in registers)
    if (hi == 0) {
        sml[(wave * 32 + l31) * 2] = m_run;                   // base-2 logit units
        sml[(wave * 32 + l31) * 2 + 1] = l_tot;
    }
    __syncthreads();
    const float* so_all = (const float*)(smem + 32 * QROW);
    constexpr int WSTRIDE = (KBYTES + VBYTES) / 4;            // floats between two waves' tiles
    for (int idx = tid; idx < 32 * (HD / 4); idx += 256) {
        const int row = idx / (HD / 4), c4 = idx % (HD / 4);
        const int r = r0 + row;
        if (r >= p.R) continue;
        float mw[4], m = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) { mw[w] = sml[(w * 32 + row) * 2]; m = fmaxf(m, mw[w]); }
        f32x4 accv = {0.f, 0.f, 0.f, 0.f};
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (mw[w] == -INFINITY) continue;
            const float scw = fast_exp2(mw[w] - m);
            const f32x4 v = *(const f32x4*)(so_all + w * WSTRIDE + row * HD + c4 * 4);
            accv += v * scw;
            l += scw * sml[(w * 32 + row) * 2 + 1];
        }
        const size_t base = ((size_t)z * p.nkv + kvh) * p.Rpad + r;
        *(f32x4*)(p.Opart + base * HD + c4 * 4) = accv;
        if (c4 == 0) {
            p.ML[base * 2] = m;
            p.ML[base * 2 + 1] = l;
        }
    }
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_cross_kernel(AttnCrossParams p) { attn_cross_body<T, HD>(p, blockIdx.z, gridDim.z); }

// T2V and T2A of one layer in ONE launch (decode: a launch of this kernel costs ~8 us on top of its bytes, measured from the two
// modalities' durations; the two key regions are disjoint slices of the same caches): the first `za` z-slices sweep set a's keys,
// the rest set b's.  Same per-slice arithmetic as two attn_cross_kernel launches with zsplit = za and gridDim.z - za.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_cross2_kernel(AttnCrossParams a, AttnCrossParams b, int za) {
    const int bz = blockIdx.z;
    if (bz < za) attn_cross_body<T, HD>(a, bz, za);
    else attn_cross_body<T, HD>(b, bz - za, (int)gridDim.z - za);
}

// Combine W partials per (row, kv head): O = sum_w 2^(m_w-m) O_w / sum_w 2^(m_w-m) l_w.
// Optionally emits the merged result in PARTIAL form (numerator, m, l) for a further cross-GPU merge.

// One (row r, kv head) merged by threads d = 0 .. HD-1 of the block; threads with d >= HD (a 256-thread caller at HD = 128) only take
// part in the barriers.
template <typename T, int HD>
__device__ __forceinline__ void attn_merge_row(const AttnMergeParams& p, const int r, const int kvh, const int d, float* s_m, float* s_l) {
    // the W (m, l) pairs of this (row, kv head) are fetched by W threads at once and shared through LDS; the weights
    // 2^(m_w - m) are then uniform values and the numerator loads of all partials are independent of each other
    const bool act = d < HD;
    float m = -INFINITY;
    for (int w0 = 0; w0 < p.W; w0 += 256) {
        __syncthreads();
        const int nw = min(256, p.W - w0);
        for (int w = d; act && w < nw; w += HD) {
            const size_t base = (size_t)(w0 + w) * p.wsML + ((size_t)kvh * p.Rpad + r) * 2;
            s_m[w] = p.ML[base];
            s_l[w] = p.ML[base + 1];
        }
        __syncthreads();
        for (int w = 0; w < nw; ++w) m = fmaxf(m, s_m[w]);
    }
    float num = 0.f, den = 0.f;
    if (m != -INFINITY) {
        for (int w0 = 0; w0 < p.W; w0 += 256) {
            const int nw = min(256, p.W - w0);
            if (p.W > 256) {                                            // re-stage this window (single window: still resident)
                __syncthreads();
                for (int w = d; act && w < nw; w += HD) {
                    const size_t base = (size_t)(w0 + w) * p.wsML + ((size_t)kvh * p.Rpad + r) * 2;
                    s_m[w] = p.ML[base];
                    s_l[w] = p.ML[base + 1];
                }
                __syncthreads();
            }
#pragma unroll 8
            for (int w = 0; w < nw; ++w) {
                const float mw = s_m[w];
                const float sc = (mw == -INFINITY) ? 0.f : fast_exp2(mw - m);
                den += sc * s_l[w];
                const float v = p.Opart[(size_t)(w0 + w) * p.wsO + ((size_t)kvh * p.Rpad + r) * HD + (act ? d : 0)];
                num += (mw == -INFINITY) ? 0.f : sc * v;                // a skipped partial may hold stale (even non-finite) data
            }
        }
    }
    if (!act) return;
    const int tq = r / p.G, g = r % p.G;
    const size_t oidx = (size_t)tq * p.ldo + (kvh * p.G + g) * HD + d;
    float out = (den > 0.f && !p.zero_out) ? num / den : 0.f;
    if (p.Out) p.Out[oidx] = T::from_f32(out);
    // partial form (same layout as one slice of Opart/ML): lets a second merge combine per-GPU results
    const size_t pbase = (size_t)kvh * p.rpo + r;
    if (p.OutF32) p.OutF32[pbase * HD + d] = num;
    if (p.OutML && d == 0) {
        p.OutML[pbase * 2] = m;
        p.OutML[pbase * 2 + 1] = den;
    }
}

template <typename T, int HD>
__device__ __forceinline__ void attn_merge_body(const AttnMergeParams& p) {
    __shared__ float s_m[256], s_l[256];
    attn_merge_row<T, HD>(p, blockIdx.x, blockIdx.y, threadIdx.x, s_m, s_l);
}

template <typename T, int HD>
__global__ __launch_bounds__(HD) void attn_merge_kernel(AttnMergeParams p) { attn_merge_body<T, HD>(p); }

// two independent merges (the T2V and the T2A partials of one layer) in one launch: blockIdx.z picks the set
template <typename T, int HD>
__global__ __launch_bounds__(HD) void attn_merge2_kernel(AttnMergeParams a, AttnMergeParams b) {
    const AttnMergeParams& p = blockIdx.z == 0 ? a : b;
    if (!p.Out && !p.OutF32) return;                        // absent set (a modality the sample does not have)
    attn_merge_body<T, HD>(p);
}

// The decode step's two small launches in ONE: the T2T of the new token (attn_text_decode_body, nkv * B blocks) and the merge of the
// T2V / T2A partials (attn_merge_row, 2 * nkv * R blocks).  Neither fills the chip, neither depends on the other (the cross-attention
// launch that produced the partials comes first; the o_proj that follows needs both): one launch of ≈9 us instead of 8.8 + 6.7 us and a gap.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_text_decode_merge2_kernel(AttnTextDecodeParams tp, AttnMergeParams a, AttnMergeParams b, int nt) {
    if ((int)blockIdx.x < nt) {
        attn_text_decode_body<T, HD>(tp, blockIdx.x % tp.nkv, blockIdx.x / tp.nkv);
        return;
    }
    // the merge role's (m, l) staging lives at the head of the launch's DYNAMIC allocation (sized for the T2T role, never below 2 KB):
    // a static array on top of it would push a T2T plan of exactly 64 KB over the limit
    extern __shared__ float s_mrg[];
    float* s_m = s_mrg;
    float* s_l = s_mrg + 256;
    const int mb = blockIdx.x - nt;
    const int r = mb % a.R, kvh = (mb / a.R) % a.nkv, set = mb / (a.R * a.nkv);
    const AttnMergeParams& p = set == 0 ? a : b;
    attn_merge_row<T, HD>(p, r, kvh, threadIdx.x, s_m, s_l);
}

int vidi_attn_text_decode_merge2_dispatch(const AttnTextDecodeParams& tp, size_t lds, const AttnMergeParams& a, const AttnMergeParams& b, int HD,
                                          int dtype, hipStream_t st) {
    if (a.R <= 0 || a.W <= 0 || b.W <= 0 || a.R != b.R || a.nkv != b.nkv || a.nkv != tp.nkv || !a.Out || !b.Out) return VIDI_ERR_SHAPE;
    if (HD != 256 && HD != 128) return VIDI_ERR_SHAPE;
    const int nt = tp.nkv * tp.B;
    const dim3 grid(nt + 2 * a.nkv * a.R);
    if (lds < 2 * 256 * sizeof(float)) lds = 2 * 256 * sizeof(float);       // the merge role's staging
    if (dtype == VIDI_DT_BF16) {
        if (HD == 256) hipLaunchKernelGGL((attn_text_decode_merge2_kernel<BF16, 256>), grid, dim3(256), lds, st, tp, a, b, nt);
        else hipLaunchKernelGGL((attn_text_decode_merge2_kernel<BF16, 128>), grid, dim3(256), lds, st, tp, a, b, nt);
    } else if (dtype == VIDI_DT_F16) {
        if (HD == 256) hipLaunchKernelGGL((attn_text_decode_merge2_kernel<F16, 256>), grid, dim3(256), lds, st, tp, a, b, nt);
        else hipLaunchKernelGGL((attn_text_decode_merge2_kernel<F16, 128>), grid, dim3(256), lds, st, tp, a, b, nt);
    } else return VIDI_ERR_DTYPE;
    return (int)hipGetLastError();
}

int vidi_attn_merge2_dispatch(const AttnMergeParams& a, const AttnMergeParams& b, int HD, int dtype, hipStream_t st) {
    // W == 0: a rank that holds no key of the modality emits the neutral partial (m = -inf, l = 0, numerator 0)
    if (a.R <= 0 || a.W < 0 || b.W < 0 || a.R != b.R || a.nkv != b.nkv) return VIDI_ERR_SHAPE;
    const dim3 grid(a.R, a.nkv, 2);
    if (dtype == VIDI_DT_BF16) {
        if (HD == 256) hipLaunchKernelGGL((attn_merge2_kernel<BF16, 256>), grid, dim3(256), 0, st, a, b);
        else if (HD == 128) hipLaunchKernelGGL((attn_merge2_kernel<BF16, 128>), grid, dim3(128), 0, st, a, b);
        else return VIDI_ERR_SHAPE;
    } else if (dtype == VIDI_DT_F16) {
        if (HD == 256) hipLaunchKernelGGL((attn_merge2_kernel<F16, 256>), grid, dim3(256), 0, st, a, b);
        else if (HD == 128) hipLaunchKernelGGL((attn_merge2_kernel<F16, 128>), grid, dim3(128), 0, st, a, b);
        else return VIDI_ERR_SHAPE;
    } else {
        return VIDI_ERR_DTYPE;
    }
    return (int)hipGetLastError();
}

// Row tiles (32 rows) a block of the cross-attention launch covers: 4 — the shared-stream kernel above — from two row tiles on
// (VIDI_XATTN_ROWS=0: never, the A/B arm), else 1.  Callers size the key split with it (blocks = nkv x ceil(row tiles / this) x slices).
int vidi_attn_cross_rtpb(int Rpad, float softcap, int dtype) {
    static const int on = [] { const char* e = getenv("VIDI_XATTN_ROWS"); return (e && atoi(e) == 0) ? 0 : 1; }();
    // every dtype / softcap has a form of the shared-stream kernel (attn_cross_rows.hip: the fixed-reference softmax for bf16 with a cap whose
    // VALUE admits it — softcap x log2(e) <= VIDI_XROWS_MAX_CAP2, not merely softcap > 0 —, a running reference otherwise)
    (void)softcap;
    return (on && Rpad > 32 && (dtype == VIDI_DT_BF16 || dtype == VIDI_DT_F16)) ? 4 : 1;
}

int vidi_attn_cross_rows_launch(const AttnCrossParams& a, const AttnCrossParams& b, int za, int zb, int HD, int dtype, hipStream_t st);      // attn_cross_rows.hip

int vidi_attn_cross_dispatch(const AttnCrossParams& p, int HD, int zsplit, int dtype, hipStream_t st) {
    if (p.R <= 0 || p.n_keys <= 0 || p.G <= 0 || p.nkv <= 0 || zsplit <= 0) return VIDI_ERR_SHAPE;
    if (p.key_start % 64 != 0 || p.Rpad % 32 != 0 || p.Rpad < p.R) return VIDI_ERR_SHAPE;
    if ((p.ldq % 8) || ((uintptr_t)p.Q & 15) || ((uintptr_t)p.Kc & 15) || ((uintptr_t)p.Vtc & 15)) return VIDI_ERR_ALIGN;
    if (HD != 256 && HD != 128) return VIDI_ERR_SHAPE;
    if (vidi_attn_cross_rtpb(p.Rpad, p.softcap, dtype) == 4) return vidi_attn_cross_rows_launch(p, p, zsplit, 0, HD, dtype, st);
    const dim3 grid(p.nkv, p.Rpad / 32, zsplit);
    const int lds = 32 * HD * 2 + 4 * (32 * HD * 2 + HD * 64);
#define LAUNCH(TT, HH)                                                                        \
    do {                                                                                      \
        auto kern = attn_cross_kernel<TT, HH>;                                                \
        static bool done = false;                                                             \
        if (!done) {                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e != hipSuccess) return (int)e;                                               \
            done = true;                                                                      \
        }                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);                                \
    } while (0)
    if (dtype == VIDI_DT_BF16) { if (HD == 256) LAUNCH(BF16, 256); else LAUNCH(BF16, 128); }
    else if (dtype == VIDI_DT_F16) { if (HD == 256) LAUNCH(F16, 256); else LAUNCH(F16, 128); }
    else return VIDI_ERR_DTYPE;
#undef LAUNCH
    return (int)hipGetLastError();
}

int vidi_attn_cross2_dispatch(const AttnCrossParams& a, const AttnCrossParams& b, int HD, int za, int zb, int dtype, hipStream_t st) {
    if (a.R <= 0 || a.n_keys <= 0 || b.n_keys <= 0 || a.G <= 0 || a.nkv <= 0 || za <= 0 || zb <= 0) return VIDI_ERR_SHAPE;
    if (a.key_start % 64 != 0 || b.key_start % 64 != 0 || a.Rpad % 32 != 0 || a.Rpad < a.R) return VIDI_ERR_SHAPE;
    if ((a.ldq % 8) || ((uintptr_t)a.Q & 15) || ((uintptr_t)a.Kc & 15) || ((uintptr_t)a.Vtc & 15)) return VIDI_ERR_ALIGN;
    if (HD != 256 && HD != 128) return VIDI_ERR_SHAPE;
    if (vidi_attn_cross_rtpb(a.Rpad, a.softcap, dtype) == 4) return vidi_attn_cross_rows_launch(a, b, za, zb, HD, dtype, st);
    const dim3 grid(a.nkv, a.Rpad / 32, za + zb);
    const int lds = 32 * HD * 2 + 4 * (32 * HD * 2 + HD * 64);
#define LAUNCH(TT, HH)                                                                        \
    do {                                                                                      \
        auto kern = attn_cross2_kernel<TT, HH>;                                               \
        static bool done = false;                                                             \
        if (!done) {                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e != hipSuccess) return (int)e;                                               \
            done = true;                                                                      \
        }                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, b, za);                         \
    } while (0)
    if (dtype == VIDI_DT_BF16) { if (HD == 256) LAUNCH(BF16, 256); else LAUNCH(BF16, 128); }
    else if (dtype == VIDI_DT_F16) { if (HD == 256) LAUNCH(F16, 256); else LAUNCH(F16, 128); }
    else return VIDI_ERR_DTYPE;
#undef LAUNCH
    return (int)hipGetLastError();
}

int vidi_attn_merge_dispatch(const AttnMergeParams& p, int HD, int dtype, hipStream_t st) {
    if (p.R <= 0 || p.W <= 0) return VIDI_ERR_SHAPE;
    const dim3 grid(p.R, p.nkv);
    if (dtype == VIDI_DT_BF16) {
        if (HD == 256) hipLaunchKernelGGL((attn_merge_kernel<BF16, 256>), grid, dim3(256), 0, st, p);
        else if (HD == 128) hipLaunchKernelGGL((attn_merge_kernel<BF16, 128>), grid, dim3(128), 0, st, p);
        else return VIDI_ERR_SHAPE;
    } else if (dtype == VIDI_DT_F16) {
        if (HD == 256) hipLaunchKernelGGL((attn_merge_kernel<F16, 256>), grid, dim3(256), 0, st, p);
        else if (HD == 128) hipLaunchKernelGGL((attn_merge_kernel<F16, 128>), grid, dim3(128), 0, st, p);
        else return VIDI_ERR_SHAPE;
    } else {
        return VIDI_ERR_DTYPE;
    }
    return (int)hipGetLastError();
}
