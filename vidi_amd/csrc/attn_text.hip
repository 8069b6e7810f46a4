// T2T causal self-attention over the short text stream (prompt ~40 tokens + <=1024 generated) and
// the RoPE that precedes it.  Reference: Gemma2Attention.forward under FA2
// (Vidi1.5_9B/vidi/model/lmm/dattn/gemma.py:165-175; TP gemma2/modeling_gemma2.py:146-168, 248-288):
// causal, logits softcap, sliding window on even layers (FA2 window_size=(W,W): i-W <= j <= i),
// right-padding key mask, GQA.  Tiny FLOPs; one wave per (batch, head, query row), one key per lane in the score pass.
#include "kernels.h"
#include "attn_text_decode.h"


template <typename T, int HD>
__global__ __launch_bounds__(64) void attn_text_kernel(AttnTextParams p) {
    constexpr int EPL = HD / 64;                      // elements per lane
    extern __shared__ float s_att[];
    float* sq = s_att;                                // [HD] query row in fp32 (read as LDS broadcasts)
    float* sc = s_att + HD;                           // [Lk] scores -> probabilities
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int kvh = h / (p.nq / p.nkv);
    const int qi = (p.past_len_dev ? *p.past_len_dev : p.past_len) + i;
    const int lo = (p.window > 0) ? max(0, qi - p.window) : 0;
    const int hiK = qi;                               // inclusive
    const u16* q = p.Q + ((size_t)b * p.Lq + i) * p.nq * HD + h * HD + lane * EPL;
#pragma unroll
    for (int e = 0; e < EPL; ++e) sq[lane * EPL + e] = T::to_f32(q[e]);
    const size_t kvstride = (size_t)p.nkv * HD;
    const u16* kbase = p.Kc + (size_t)b * p.Lmax * kvstride + kvh * HD;
    const u16* vb = p.Vc + (size_t)b * p.Lmax * kvstride + kvh * HD + lane * EPL;
    const unsigned char* km = p.kmask ? p.kmask + (size_t)b * p.Lmax : nullptr;
    __syncthreads();

    // scores: ONE KEY PER LANE (64 keys per pass, each lane walks its key row in 16-byte loads against the broadcast
    // query) — the keys of a pass are independent, so nothing serialises on a per-key wave reduction
    float mx = -INFINITY;
    for (int j0 = lo; j0 <= hiK; j0 += 64) {
        const int j = j0 + lane;
        float d = -INFINITY;
        if (j <= hiK) {
            const u16* kr = kbase + (size_t)j * kvstride;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll 8                                                   // 8 row chunks (128 B) in flight per lane and pass
            for (int c = 0; c < HD / 8; ++c) {
                float f[8];
                unpack8<T>(*(const u32x4*)(kr + c * 8), f);
                const f32x4 q0 = *(const f32x4*)(sq + c * 8), q1 = *(const f32x4*)(sq + c * 8 + 4);
                a0 = fmaf(q0[0], f[0], a0); a1 = fmaf(q0[1], f[1], a1); a0 = fmaf(q0[2], f[2], a0); a1 = fmaf(q0[3], f[3], a1);
                a0 = fmaf(q1[0], f[4], a0); a1 = fmaf(q1[1], f[5], a1); a0 = fmaf(q1[2], f[6], a0); a1 = fmaf(q1[3], f[7], a1);
            }
            d = (a0 + a1) * p.scale;
            if (p.softcap > 0.f) d = p.softcap * tanhf(d / p.softcap);
            if (km && km[j] == 0) d = -INFINITY;
            sc[j - lo] = d;
        }
        mx = fmaxf(mx, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __syncthreads();
    const int n = hiK - lo + 1;
    float lsum = 0.f;
    const float m_use = (mx == -INFINITY) ? 0.f : mx;
    for (int j = lane; j < n; j += 64) {
        const float pj = __expf(sc[j] - m_use);
        lsum += pj;
        sc[j] = rnd<T>(pj);                           // flash-attn feeds P in the model dtype to PV
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    float o[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) o[e] = 0.f;
    // PV: the V rows are independent loads -> 16 in flight per lane (the loop is one L2 round trip per batch of rows)
    int j = 0;
    for (; j + 16 <= n; j += 16) {
        u16 vv[16][EPL];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int e = 0; e < EPL; ++e) vv[u][e] = vb[(size_t)(lo + j + u) * kvstride + e];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float pj = sc[j + u];
#pragma unroll
            for (int e = 0; e < EPL; ++e) o[e] = fmaf(pj, T::to_f32(vv[u][e]), o[e]);
        }
    }
    for (; j < n; ++j) {
        const float pj = sc[j];
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = fmaf(pj, T::to_f32(vb[(size_t)(lo + j) * kvstride + e]), o[e]);
    }
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
    u16* op = p.O + ((size_t)b * p.Lq + i) * p.nq * HD + h * HD + lane * EPL;
#pragma unroll
    for (int e = 0; e < EPL; ++e) op[e] = T::from_f32(o[e] * inv);
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_text_decode_kernel(AttnTextDecodeParams p) { attn_text_decode_body<T, HD>(p, blockIdx.x, blockIdx.y); }

int vidi_attn_text_decode_dispatch(const void* qkv, int ldqkv, void* Kc, void* Vc, const void* kmask, const void* cs, const void* sn,
                                   void* O, int B, int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int window,
                                   float scale, float softcap, int dtype, hipStream_t st) {
    AttnTextDecodeParams p;
    size_t lds;
    const int rc = attn_text_decode_params(p, lds, qkv, ldqkv, Kc, Vc, kmask, cs, sn, O, B, Lmax, nq, nkv, HD, pos0, pos_dev, window, scale, softcap);
    if (rc) return rc;
    const dim3 grid(nkv, B);
#define VIDI_ATD(TT, D) hipLaunchKernelGGL((attn_text_decode_kernel<TT, D>), grid, dim3(256), lds, st, p)
    if (dtype == VIDI_DT_BF16) {
        if (HD == 256) VIDI_ATD(BF16, 256); else if (HD == 128) VIDI_ATD(BF16, 128); else if (HD == 64) VIDI_ATD(BF16, 64); else return VIDI_ERR_SHAPE;
    } else if (dtype == VIDI_DT_F16) {
        if (HD == 256) VIDI_ATD(F16, 256); else if (HD == 128) VIDI_ATD(F16, 128); else if (HD == 64) VIDI_ATD(F16, 64); else return VIDI_ERR_SHAPE;
    } else return VIDI_ERR_DTYPE;
#undef VIDI_ATD
    return (int)hipGetLastError();
}

// RoPE (rotate_half form) applied in place to Q [rows, nq*HD] and K [rows, nkv*HD];
// cos/sin [rows, HD] are in the model dtype like the reference's (cast after fp32 computation).
// Each product and the sum round to the model dtype as eager torch does.
template <typename T>
__global__ void rope_kernel(u16* Q, u16* K, const u16* cs, const u16* sn, int rows, int nq, int nkv, int HD) {
    const int half = HD / 2;
    const size_t total = (size_t)rows * (nq + nkv) * half;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int d = idx % half;
        const size_t t = idx / half;
        const int hh = t % (nq + nkv);
        const size_t row = t / (nq + nkv);
        u16* base = (hh < nq) ? Q + (row * nq + hh) * HD : K + (row * nkv + (hh - nq)) * HD;
        const float x1 = T::to_f32(base[d]), x2 = T::to_f32(base[d + half]);
        const float c1 = T::to_f32(cs[row * HD + d]), c2 = T::to_f32(cs[row * HD + d + half]);
        const float s1 = T::to_f32(sn[row * HD + d]), s2 = T::to_f32(sn[row * HD + d + half]);
        const float y1 = rnd<T>(x1 * c1) + rnd<T>(-x2 * s1);
        const float y2 = rnd<T>(x2 * c2) + rnd<T>(x1 * s2);
        base[d] = T::from_f32(y1);
        base[d + half] = T::from_f32(y2);
    }
}

// RoPE + KV-cache append in ONE pass over the fused projection output (was: 2 device copies, rope in place, 2 more
// copies per layer and step):  qkv [M][q | k | v] -> QR [M][nq*HD] = rope(q);  Kc[b][pos][nkv*HD] = rope(k);
// Vc[b][pos][nkv*HD] = v, with row m = b*Lq + i and pos = (pos_dev ? *pos_dev : pos0) + i.  Same roundings as rope_kernel.
template <typename T>
__global__ void rope_cache_kernel(const u16* __restrict__ qkv, int ldqkv, u16* __restrict__ QR, u16* __restrict__ Kc,
                                  u16* __restrict__ Vc, const u16* __restrict__ cs, const u16* __restrict__ sn, int B, int Lq,
                                  int Lmax, int nq, int nkv, int HD, int pos0, const int* __restrict__ pos_dev) {
    const int half = HD / 2;
    const int per_row = (nq + 2 * nkv) * half;                         // rope pairs of q and k, then pairs of v elements
    const size_t total = (size_t)B * Lq * per_row;
    const int p0 = pos_dev ? *pos_dev : pos0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(idx % per_row);
        const size_t row = idx / per_row;
        const int b = (int)(row / Lq), i = (int)(row % Lq);
        const int hh = w / half, d = w % half;
        const u16* src = qkv + row * (size_t)ldqkv;
        const size_t slot = ((size_t)b * Lmax + p0 + i) * (size_t)(nkv * HD);
        if (hh >= nq + nkv) {                                          // V: plain copy, two elements per item
            const int c = (hh - nq - nkv) * HD + 2 * d;
            *(unsigned*)(Vc + slot + c) = *(const unsigned*)(src + (nq + nkv) * HD + c);
            continue;
        }
        const u16* in = src + hh * HD;
        const float x1 = T::to_f32(in[d]), x2 = T::to_f32(in[d + half]);
        const float c1 = T::to_f32(cs[row * HD + d]), c2 = T::to_f32(cs[row * HD + d + half]);
        const float s1 = T::to_f32(sn[row * HD + d]), s2 = T::to_f32(sn[row * HD + d + half]);
        const float y1 = rnd<T>(x1 * c1) + rnd<T>(-x2 * s1);
        const float y2 = rnd<T>(x2 * c2) + rnd<T>(x1 * s2);
        u16* out = (hh < nq) ? QR + (row * nq + hh) * HD : Kc + slot + (hh - nq) * HD;
        out[d] = T::from_f32(y1);
        out[d + half] = T::from_f32(y2);
    }
}

int vidi_rope_cache_dispatch(const void* qkv, int ldqkv, void* QR, void* Kc, void* Vc, const void* cs, const void* sn, int B, int Lq,
                             int Lmax, int nq, int nkv, int HD, int pos0, const int* pos_dev, int dtype, hipStream_t st) {
    if (B <= 0 || Lq <= 0 || HD % 2 || (ldqkv % 2) || (!pos_dev && (pos0 < 0 || pos0 + Lq > Lmax))) return VIDI_ERR_SHAPE;
    const size_t total = (size_t)B * Lq * (nq + 2 * nkv) * (HD / 2);
    const int blocks = (int)min((total + 255) / 256, (size_t)4096);
    if (dtype == VIDI_DT_BF16)
        hipLaunchKernelGGL(rope_cache_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const u16*)qkv, ldqkv, (u16*)QR, (u16*)Kc, (u16*)Vc,
                           (const u16*)cs, (const u16*)sn, B, Lq, Lmax, nq, nkv, HD, pos0, pos_dev);
    else if (dtype == VIDI_DT_F16)
        hipLaunchKernelGGL(rope_cache_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const u16*)qkv, ldqkv, (u16*)QR, (u16*)Kc, (u16*)Vc,
                           (const u16*)cs, (const u16*)sn, B, Lq, Lmax, nq, nkv, HD, pos0, pos_dev);
    else return VIDI_ERR_DTYPE;
    return (int)hipGetLastError();
}

int vidi_attn_text_dispatch(const AttnTextParams& p, int HD, int dtype, hipStream_t st) {
    if (p.B <= 0 || p.Lq <= 0 || p.nq <= 0 || p.nkv <= 0 || p.nq % p.nkv) return VIDI_ERR_SHAPE;
    if (p.past_len + p.Lq > p.Lmax) return VIDI_ERR_SHAPE;
    const dim3 grid(p.Lq, p.nq, p.B);
    const int lds = (HD + (p.past_len_dev ? p.Lmax : p.past_len + p.Lq)) * 4;   // query row + scores (device-side length: whole cache)
    if (lds > 64 * 1024) return VIDI_ERR_SHAPE;
    if (dtype == VIDI_DT_BF16) {
        if (HD == 256) hipLaunchKernelGGL((attn_text_kernel<BF16, 256>), grid, dim3(64), lds, st, p);
        else if (HD == 128) hipLaunchKernelGGL((attn_text_kernel<BF16, 128>), grid, dim3(64), lds, st, p);
        else if (HD == 64) hipLaunchKernelGGL((attn_text_kernel<BF16, 64>), grid, dim3(64), lds, st, p);
        else return VIDI_ERR_SHAPE;
    } else if (dtype == VIDI_DT_F16) {
        if (HD == 256) hipLaunchKernelGGL((attn_text_kernel<F16, 256>), grid, dim3(64), lds, st, p);
        else if (HD == 128) hipLaunchKernelGGL((attn_text_kernel<F16, 128>), grid, dim3(64), lds, st, p);
        else if (HD == 64) hipLaunchKernelGGL((attn_text_kernel<F16, 64>), grid, dim3(64), lds, st, p);
        else return VIDI_ERR_SHAPE;
    } else {
        return VIDI_ERR_DTYPE;
    }
    return (int)hipGetLastError();
}

int vidi_rope_dispatch(void* Q, void* K, const void* cs, const void* sn, int rows, int nq, int nkv, int HD, int dtype, hipStream_t st) {
    if (rows <= 0 || HD % 2) return VIDI_ERR_SHAPE;
    const size_t total = (size_t)rows * (nq + nkv) * (HD / 2);
    const int blocks = (int)min((total + 255) / 256, (size_t)4096);
    if (dtype == VIDI_DT_BF16) hipLaunchKernelGGL(rope_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (u16*)Q, (u16*)K, (const u16*)cs, (const u16*)sn, rows, nq, nkv, HD);
    else if (dtype == VIDI_DT_F16) hipLaunchKernelGGL(rope_kernel<F16>, dim3(blocks), dim3(256), 0, st, (u16*)Q, (u16*)K, (const u16*)cs, (const u16*)sn, rows, nq, nkv, HD);
    else return VIDI_ERR_DTYPE;
    return (int)hipGetLastError();
}
