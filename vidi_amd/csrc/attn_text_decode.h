// Decode-step T2T body shared by attn_text_decode_kernel (attn_text.hip) and the combined T2T + merge launch (attn_cross.hip).
#pragma once
#include "kernels.h"

// ---- the decode step's T2T in ONE launch: RoPE of the new token's q and k, the KV-cache append and the attention over the cache ----
//   (was rope_cache_kernel + attn_text_kernel: 5 + 17 us per layer at ~70 keys, the second a chain of ~18 dependent L2 round trips in
//   one wave).  One 256-thread block per (kv head, batch row) serves the G query heads that share the head's K/V:
//     1. rope(q_g), rope(k) with rope_cache_kernel's roundings; k and v are written to cache slot `pos` (and read back below:
//        the block's own stores are visible after the barrier);
//     2. scores: 4 lanes per key, 128 keys per step with all of the step's 16-byte loads requested before the first FMA;
//     3. softmax per head by one wave each (P rounded to T, the sum over the unrounded P — attn_text_kernel's and flash-attn's);
//     4. PV: a thread owns 8 output columns and every (256 / (HD/8))-th key, 8 V rows in flight; the key groups are summed in LDS.
//   Scores and probabilities are the values attn_text_kernel computes; only the fp32 summation ORDER of the dot products differs.
struct AttnTextDecodeParams {
    const u16* qkv; int ldqkv;            // [B][q | k | v] raw projection output of the new token
    u16* Kc; u16* Vc;                     // [B, Lmax, nkv*HD]
    const unsigned char* kmask;           // [B, Lmax] or null
    const u16* cs; const u16* sn;         // [B, HD] rope rows of the new token
    u16* O;                               // [B, nq*HD]
    int B, Lmax, nq, nkv;
    int pos0; const int* pos_dev;         // cache slot of the new token (= number of cached tokens)
    int window; float scale, softcap;
    int lcap;                             // score slots per head in LDS (>= number of visible keys)
};

template <typename T, int HD>
__device__ __forceinline__ void attn_text_decode_body(const AttnTextDecodeParams& p, const int kvh, const int b) {
    constexpr int half = HD / 2;
    constexpr int NC = HD / 8;                        // 16-byte chunks per row
    constexpr int CPL = NC / 4;                       // chunks per lane in the score pass (4 lanes per key)
    constexpr int NKG = 256 / NC;                     // key groups in the PV pass
    static_assert(NC % 4 == 0 && 256 % NC == 0, "HD must be 64, 128 or 256");
    extern __shared__ float s_dec[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = p.nq / p.nkv;
    float* sq = s_dec;                                // [G][HD] roped query rows (values already rounded to T)
    float* sl = sq + G * HD;                          // [G] softmax denominators (padded to 8)
    float* sc = sl + 8;                               // [G][lcap] scores -> probabilities
    float* red = sc + (size_t)G * p.lcap;             // [NKG][G][HD] PV partials
    const int pos = p.pos_dev ? *p.pos_dev : p.pos0;
    const int lo = (p.window > 0) ? max(0, pos - p.window) : 0;
    const int n = pos - lo + 1;                       // visible keys lo .. pos
    const size_t kvstride = (size_t)p.nkv * HD;
    u16* kslot = p.Kc + ((size_t)b * p.Lmax + pos) * kvstride + kvh * HD;
    u16* vslot = p.Vc + ((size_t)b * p.Lmax + pos) * kvstride + kvh * HD;
    const u16* src = p.qkv + (size_t)b * p.ldqkv;

    // ---- 1. rope + cache append ----------------------------------------------------------------
    for (int w = tid; w < (G + 1) * half; w += 256) {
        const int hh = w / half, d = w % half;
        const u16* in = (hh < G) ? src + (kvh * G + hh) * HD : src + p.nq * HD + kvh * HD;
        const float x1 = T::to_f32(in[d]), x2 = T::to_f32(in[d + half]);
        const float c1 = T::to_f32(p.cs[b * HD + d]), c2 = T::to_f32(p.cs[b * HD + d + half]);
        const float s1 = T::to_f32(p.sn[b * HD + d]), s2 = T::to_f32(p.sn[b * HD + d + half]);
        const float y1 = rnd<T>(rnd<T>(x1 * c1) + rnd<T>(-x2 * s1));
        const float y2 = rnd<T>(rnd<T>(x2 * c2) + rnd<T>(x1 * s2));
        if (hh < G) {
            sq[hh * HD + d] = y1;
            sq[hh * HD + d + half] = y2;
        } else {
            kslot[d] = T::from_f32(y1);
            kslot[d + half] = T::from_f32(y2);
        }
    }
    for (int w = tid; w < half; w += 256)
        *(unsigned*)(vslot + 2 * w) = *(const unsigned*)(src + (p.nq + p.nkv) * HD + kvh * HD + 2 * w);
    __threadfence_block();
    __syncthreads();

    // ---- 2. scores --------------------------------------------------------------------------------
    const u16* kbase = p.Kc + (size_t)b * p.Lmax * kvstride + kvh * HD;
    const unsigned char* km = p.kmask ? p.kmask + (size_t)b * p.Lmax : nullptr;
    const int part = tid & 3;
    for (int j0 = lo; j0 <= pos; j0 += 128) {
        u32x4 kv[2][CPL];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int j = min(j0 + 64 * h2 + (tid >> 2), pos);
            const u16* kr = kbase + (size_t)j * kvstride;
#pragma unroll
            for (int i = 0; i < CPL; ++i) kv[h2][i] = *(const u32x4*)(kr + (i * 4 + part) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int j = j0 + 64 * h2 + (tid >> 2);
            for (int g = 0; g < G; ++g) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    float f[8];
                    unpack8<T>(kv[h2][i], f);
                    const float* qp = sq + g * HD + (i * 4 + part) * 8;
                    const f32x4 q0 = *(const f32x4*)qp, q1 = *(const f32x4*)(qp + 4);
                    a0 = fmaf(q0[0], f[0], a0); a1 = fmaf(q0[1], f[1], a1); a0 = fmaf(q0[2], f[2], a0); a1 = fmaf(q0[3], f[3], a1);
                    a0 = fmaf(q1[0], f[4], a0); a1 = fmaf(q1[1], f[5], a1); a0 = fmaf(q1[2], f[6], a0); a1 = fmaf(q1[3], f[7], a1);
                }
                float d = a0 + a1;
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                if (part == 0 && j <= pos) {
                    d *= p.scale;
                    if (p.softcap > 0.f) d = p.softcap * tanhf(d / p.softcap);
                    if (km && km[j] == 0) d = -INFINITY;
                    sc[(size_t)g * p.lcap + (j - lo)] = d;
                }
            }
        }
    }
    __syncthreads();

    // ---- 3. softmax: one wave per query head ----------------------------------------------------------
    for (int g = wave; g < G; g += 4) {
        float* sg = sc + (size_t)g * p.lcap;
        float mx = -INFINITY;
        for (int j = lane; j < n; j += 64) mx = fmaxf(mx, sg[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float m_use = (mx == -INFINITY) ? 0.f : mx;
        float lsum = 0.f;
        for (int j = lane; j < n; j += 64) {
            const float pj = __expf(sg[j] - m_use);
            lsum += pj;
            sg[j] = rnd<T>(pj);
        }
        lsum = wave_sum(lsum);
        if (lane == 0) sl[g] = lsum;
    }
    __syncthreads();

    // ---- 4. PV ---------------------------------------------------------------------------------------
    const int dc = tid % NC, kg = tid / NC;
    const u16* vb = p.Vc + (size_t)b * p.Lmax * kvstride + kvh * HD + dc * 8;
    for (int g0 = 0; g0 < G; g0 += 2) {               // two query heads per sweep over V (registers); G = 2: one sweep
        float o0[8], o1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
        const float* p0 = sc + (size_t)g0 * p.lcap;
        const float* p1 = sc + (size_t)min(g0 + 1, G - 1) * p.lcap;
        for (int jb = kg; jb < n; jb += NKG * 8) {
            u32x4 vv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) vv[u] = *(const u32x4*)(vb + (size_t)(lo + min(jb + u * NKG, n - 1)) * kvstride);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jb + u * NKG;
                if (j < n) {
                    float f[8];
                    unpack8<T>(vv[u], f);
                    const float pa = p0[j], pb = p1[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { o0[e] = fmaf(pa, f[e], o0[e]); o1[e] = fmaf(pb, f[e], o1[e]); }
                }
            }
        }
        float* r0 = red + ((size_t)kg * G + g0) * HD + dc * 8;
        *(f32x4*)r0 = f32x4{o0[0], o0[1], o0[2], o0[3]};
        *(f32x4*)(r0 + 4) = f32x4{o0[4], o0[5], o0[6], o0[7]};
        if (g0 + 1 < G) {
            *(f32x4*)(r0 + HD) = f32x4{o1[0], o1[1], o1[2], o1[3]};
            *(f32x4*)(r0 + HD + 4) = f32x4{o1[4], o1[5], o1[6], o1[7]};
        }
    }
    __syncthreads();
    for (int idx = tid; idx < G * HD; idx += 256) {
        const int g = idx / HD, d = idx % HD;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NKG; ++k) acc += red[((size_t)k * G + g) * HD + d];
        const float lsum = sl[g];
        const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
        p.O[(size_t)b * p.nq * HD + (kvh * G + g) * HD + d] = T::from_f32(acc * inv);
    }
}

// argument checks + parameter block + dynamic LDS bytes of the T2T role (shared by the two dispatchers)
static inline int attn_text_decode_params(AttnTextDecodeParams& p, size_t& lds, const void* qkv, int ldqkv, void* Kc, void* Vc,
                                          const void* kmask, const void* cs, const void* sn, void* O, int B, int Lmax, int nq, int nkv,
                                          int HD, int pos0, const int* pos_dev, int window, float scale, float softcap) {
    if (B <= 0 || nq <= 0 || nkv <= 0 || nq % nkv || nq / nkv > 8 || (ldqkv % 8)) return VIDI_ERR_SHAPE;
    if (HD != 64 && HD != 128 && HD != 256) return VIDI_ERR_SHAPE;
    if (!pos_dev && (pos0 < 0 || pos0 >= Lmax)) return VIDI_ERR_SHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)Kc & 15) || ((uintptr_t)Vc & 15)) return VIDI_ERR_ALIGN;
    p.qkv = (const u16*)qkv; p.ldqkv = ldqkv; p.Kc = (u16*)Kc; p.Vc = (u16*)Vc; p.kmask = (const unsigned char*)kmask;
    p.cs = (const u16*)cs; p.sn = (const u16*)sn; p.O = (u16*)O; p.B = B; p.Lmax = Lmax; p.nq = nq; p.nkv = nkv;
    p.pos0 = pos0; p.pos_dev = pos_dev; p.window = window; p.scale = scale; p.softcap = softcap;
    const int G = nq / nkv;
    int nvis = pos_dev ? Lmax : pos0 + 1;                           // device-side position: room for the whole cache
    if (window > 0) nvis = min(nvis, window + 1);
    p.lcap = (nvis + 3) & ~3;
    lds = ((size_t)G * HD + 8 + (size_t)G * p.lcap + (size_t)(256 / (HD / 8)) * G * HD) * 4;
    return lds > 64 * 1024 ? VIDI_ERR_SHAPE : 0;
}

