// Data-movement / elementwise kernels of the Vidi hot path (all HBM-bound, coalesced along the
// channel axis).  Each kernel cites the reference lines whose arithmetic it reproduces.
#include "kernels.h"

#define GRID_STRIDE(idx, total) \
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (total); idx += (size_t)gridDim.x * blockDim.x)

static inline int grid_for(size_t total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

// ---- SigLIP patch embedding as a GEMM: im2col of non-overlapping PxP patches --------------------
// pixel [T,3,S,S] -> A [T*side*side, Kpad], k = c*P*P + iy*P + ix (= Conv2d weight.flatten(1) order,
// TP siglip/modeling_siglip.py:124-130,178); columns >= 3*P*P are zero padding for the 64-wide K step.
template <typename T>
__global__ void im2col_patch_kernel(const u16* __restrict__ px, u16* __restrict__ A, int Tn, int S, int P, int Kpad) {
    const int side = S / P, PP = P * P;
    const size_t total = (size_t)Tn * side * side * Kpad;
    GRID_STRIDE(idx, total) {
        const int k = idx % Kpad;
        const size_t row = idx / Kpad;
        u16 v = 0;
        if (k < 3 * PP) {
            const int c = k / PP, iy = (k % PP) / P, ix = k % P;
            const int pxi = row % side, pyi = (row / side) % side;
            const size_t t = row / ((size_t)side * side);
            v = px[((t * 3 + c) * S + (size_t)pyi * P + iy) * S + (size_t)pxi * P + ix];
        }
        A[idx] = v;
    }
}

// ---- frame-token pooling: zero-pad (side -> side+1), optional bilinear resize to (h,w),
//      space_to_depth(m) — mm_vision/pool.py:23-32, vidi/utils.py:134-150.
// feats [T, side*side, C] (token-major tower output) -> out [T, h/m, w/m, C*m*m],
// out channel = c*m*m + dy*m + dx.  Bilinear: align_corners=False, fp32 opmath, one rounding.
template <typename T>
__global__ void pool_s2d_kernel(const u16* __restrict__ f, u16* __restrict__ out, int Tn, int side, int C, int h, int w,
                                int m, int resize) {
    const int oh = h / m, ow = w / m, P = side + 1;
    const size_t total = (size_t)Tn * oh * ow * C;
    const float sy_scale = (float)P / (float)h, sx_scale = (float)P / (float)w;
    GRID_STRIDE(idx, total) {
        const int c = idx % C;
        const int ox = (idx / C) % ow, oy = (idx / ((size_t)C * ow)) % oh;
        const size_t t = idx / ((size_t)C * ow * oh);
        const u16* ft = f + t * (size_t)side * side * C + c;
        auto pad_at = [&](int yy, int xx) -> float {
            return (yy < side && xx < side) ? T::to_f32(ft[((size_t)yy * side + xx) * C]) : 0.f;
        };
        u16* o = out + (((t * oh + oy) * ow + ox) * (size_t)C + c) * m * m;
        for (int dy = 0; dy < m; ++dy)
            for (int dx = 0; dx < m; ++dx) {
                const int Y = oy * m + dy, X = ox * m + dx;
                float v;
                if (!resize) {
                    v = pad_at(Y, X);
                } else {
                    float sy = sy_scale * ((float)Y + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
                    float sx = sx_scale * ((float)X + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
                    const int y0 = (int)sy, x0 = (int)sx;
                    const int y1 = y0 + (y0 < P - 1 ? 1 : 0), x1 = x0 + (x0 < P - 1 ? 1 : 0);
                    const float ly = sy - (float)y0, lx = sx - (float)x0;
                    const float hy = 1.f - ly, hx = 1.f - lx;
                    v = hy * (hx * pad_at(y0, x0) + lx * pad_at(y0, x1)) + ly * (hx * pad_at(y1, x0) + lx * pad_at(y1, x1));
                }
                o[dy * m + dx] = T::from_f32(v);
            }
    }
}

// ---- positional adds: f = T(T(T(f + ph[y]) + pw[x]) + pt[t])  — multimodal.py:194-197 ------------
// pt may be null (used for the two spatial adds only) ; ph/pw null => skip (audio: only pt).
template <typename T>
__global__ void add_pos_kernel(u16* __restrict__ f, const u16* __restrict__ ph, const u16* __restrict__ pw,
                               const u16* __restrict__ pt, int Tn, int oh, int ow, int H) {
    const size_t total = (size_t)Tn * oh * ow * (H / 8);
    const int hc = H / 8;
    GRID_STRIDE(idx, total) {
        const int ch = idx % hc;
        const int x = (idx / hc) % ow, y = (idx / ((size_t)hc * ow)) % oh;
        const size_t t = idx / ((size_t)hc * ow * oh);
        u16* p = f + (idx / hc) * (size_t)H + ch * 8;
        float v[8], a[8];
        unpack8<T>(*(const u32x4*)p, v);
        if (ph) {
            unpack8<T>(*(const u32x4*)(ph + (size_t)y * H + ch * 8), a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rnd<T>(v[e] + a[e]);
        }
        if (pw) {
            unpack8<T>(*(const u32x4*)(pw + (size_t)x * H + ch * 8), a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rnd<T>(v[e] + a[e]);
        }
        if (pt) {
            unpack8<T>(*(const u32x4*)(pt + t * H + ch * 8), a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rnd<T>(v[e] + a[e]);
        }
        *(u32x4*)p = pack8<T>(v);
    }
}

// ---- y = T(T(a + b) + c) (c optional) — gemma.py:236 (text + image + audio) ----------------------
template <typename T>
__global__ void add3_kernel(const u16* a, const u16* b, const u16* c, u16* y, size_t n8) {
    GRID_STRIDE(idx, n8) {
        float x[8], z[8];
        unpack8<T>(*(const u32x4*)(a + idx * 8), x);
        if (b) {
            unpack8<T>(*(const u32x4*)(b + idx * 8), z);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = rnd<T>(x[e] + z[e]);
        }
        if (c) {
            unpack8<T>(*(const u32x4*)(c + idx * 8), z);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = rnd<T>(x[e] + z[e]);
        }
        *(u32x4*)(y + idx * 8) = pack8<T>(x);
    }
}

// ---- Vidi-7B learned Conv2DPool (Vidi_7B/model/mm_vision/pool.py:19-26) as im2col + GEMM + resize -------------
// im2col of a stride-1, bias-free k x k convolution over NHWC tower features x[T][side*side][C]:
// out[(t, oy, ox)][(dy*k + dx)*C + c] = x[t][(oy+dy)*side + (ox+dx)][c]; 16-byte copies.
template <typename T>
__global__ void im2col_nhwc_kernel(const u16* __restrict__ x, u16* __restrict__ out, int Tn, int side, int C, int k) {
    const int oc = side - k + 1, c8 = C / 8;
    const size_t total = (size_t)Tn * oc * oc * k * k * c8;
    GRID_STRIDE(idx, total) {
        const int c = idx % c8;
        size_t r = idx / c8;
        const int dx = r % k; r /= k;
        const int dy = r % k; r /= k;
        const int ox = r % oc; r /= oc;
        const int oy = r % oc;
        const size_t t = r / oc;
        const u32x4 v = *(const u32x4*)(x + ((t * side + (oy + dy)) * side + (ox + dx)) * (size_t)C + c * 8);
        *(u32x4*)(out + idx * 8) = v;
    }
}

// F.interpolate(mode='bilinear', align_corners=True) on NHWC x[T][s_in][s_in][C] -> out[T][s_out][s_out][C];
// source coordinate = dst * (s_in-1)/(s_out-1) (0 when s_out == 1), fp32 opmath, one rounding.
template <typename T>
__global__ void resize_ac_kernel(const u16* __restrict__ x, u16* __restrict__ out, int Tn, int s_in, int s_out, int C) {
    const int c8 = C / 8;
    const size_t total = (size_t)Tn * s_out * s_out * c8;
    const float sc = (s_out > 1) ? (float)(s_in - 1) / (float)(s_out - 1) : 0.f;
    GRID_STRIDE(idx, total) {
        const int c = idx % c8;
        size_t r = idx / c8;
        const int ox = r % s_out; r /= s_out;
        const int oy = r % s_out;
        const size_t t = r / s_out;
        const float fy = sc * oy, fx = sc * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, s_in - 1), x1 = min(x0 + 1, s_in - 1);
        const float ly = fy - y0, lx = fx - x0;
        float a[8], b[8], cc[8], d[8], y[8];
        const u16* base = x + t * (size_t)s_in * s_in * C + c * 8;
        unpack8<T>(*(const u32x4*)(base + ((size_t)y0 * s_in + x0) * C), a);
        unpack8<T>(*(const u32x4*)(base + ((size_t)y0 * s_in + x1) * C), b);
        unpack8<T>(*(const u32x4*)(base + ((size_t)y1 * s_in + x0) * C), cc);
        unpack8<T>(*(const u32x4*)(base + ((size_t)y1 * s_in + x1) * C), d);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            y[e] = (1.f - ly) * ((1.f - lx) * a[e] + lx * b[e]) + ly * ((1.f - lx) * cc[e] + lx * d[e]);
        *(u32x4*)(out + idx * 8) = pack8<T>(y);
    }
}

// ---- embed_tokens gather * normalizer — multimodal.py:385 + gemma.py:353-354 --------------------
template <typename T>
__global__ void embed_kernel(const long long* __restrict__ ids, const u16* __restrict__ E, u16* __restrict__ out,
                             int n, int H, float normalizer, long long vocab) {
    const int hc = H / 8;
    const size_t total = (size_t)n * hc;
    GRID_STRIDE(idx, total) {
        const int ch = idx % hc;
        const size_t i = idx / hc;
        const long long id = ids[i];
        float v[8];
        if (id >= 0 && id < vocab) {
            unpack8<T>(*(const u32x4*)(E + (size_t)id * H + ch * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= normalizer;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;     // padding slot (multimodal.py:423-426 zero rows)
        }
        *(u32x4*)(out + i * H + ch * 8) = pack8<T>(v);
    }
}

// ---- GeGLU on the 32-row-interleaved [gate|up] layout produced by the skinny GEMM path ----------
//   out[m][i] = T(T(gelu_tanh(g)) * u), g = Y[m][(i/32)*64 + i%32], u = Y[m][(i/32)*64 + 32 + i%32]
template <typename T>
__global__ void geglu_unpack_kernel(const u16* __restrict__ Yp, u16* __restrict__ out, int M, int I, int silu) {
    const size_t total = (size_t)M * I;
    GRID_STRIDE(idx, total) {
        const int i = idx % I;
        const size_t m = idx / I;
        const size_t base = m * 2 * (size_t)I + (size_t)(i >> 5) * 64 + (i & 31);
        const float g = T::to_f32(Yp[base]), u = T::to_f32(Yp[base + 32]);
        out[idx] = T::from_f32(rnd<T>(silu ? silu_f(g) : gelu_tanh_f(g)) * u);
    }
}

// ---- final-logit softcap + greedy argmax — gemma.py:565-569 + do_sample=False ------------------
//   logits <- T(T(tanh(T(x/cap))) * cap) in place ; idx[b] = first argmax
//   One block per row walked the 256 000 logits in 123 us per decode step (round 2 trace); now a row is spread over up to 128 blocks:
//   each reduces its slice to (value, first index), folds it into a per-row 64-bit key with atomicMax (value in the high word as an
//   order-preserving integer, ~index in the low word: max = largest value, then smallest index) and the last block to arrive writes
//   idx[b] and clears the row's scratch for the next call.  The scratch is CALLER-OWNED (vidi_softcap_argmax_workspace_bytes: 16 bytes per
//   row = a 64-bit key + a 32-bit arrival counter; zeroed once by the caller, left zeroed by every call): the library holds no state, so
//   calls on different streams are independent as long as they use different workspaces (stateless, re-entrant).
struct AmSlot { unsigned long long best; unsigned int count; unsigned int pad; };

template <typename T>
__device__ __forceinline__ float softcap_value(float x, float cap) { return rnd<T>(rnd<T>(tanhf(rnd<T>(x / cap))) * cap); }

template <typename T>
__global__ __launch_bounds__(256) void softcap_argmax_kernel(u16* __restrict__ logits, long long* __restrict__ idx, int V,
                                                             long long ld, float cap, int vec, AmSlot* __restrict__ ws) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.y, tid = threadIdx.x;
    u16* row = logits + (size_t)b * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (vec) {                                           // V % 8 == 0 and 16-byte aligned rows
        for (int c = blockIdx.x * 256 + tid; c < V / 8; c += gridDim.x * 256) {
            float x[8];
            unpack8<T>(*(const u32x4*)(row + (size_t)c * 8), x);
            if (cap > 0.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = softcap_value<T>(x[e], cap);
                *(u32x4*)(row + (size_t)c * 8) = pack8<T>(x);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (x[e] > best || (x[e] == best && c * 8 + e < bi)) { best = x[e]; bi = c * 8 + e; }
        }
    } else {
        for (int i = blockIdx.x * 256 + tid; i < V; i += gridDim.x * 256) {
            float x = T::to_f32(row[i]);
            if (cap > 0.f) {
                x = softcap_value<T>(x, cap);
                row[i] = T::from_f32(x);
            }
            if (x > best || (x == best && i < bi)) { best = x; bi = i; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        if (bi != 0x7fffffff) {                          // a slice of NaNs only contributes nothing (as `x > best` never held for them)
            unsigned u = __float_as_uint(best);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            atomicMax(&ws[b].best, ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)bi));
        }
        __threadfence();
        if (atomicAdd(&ws[b].count, 1u) == gridDim.x - 1) {
            const unsigned long long key = atomicExch(&ws[b].best, 0ull);
            ws[b].count = 0;
            idx[b] = key ? (long long)(0xffffffffu - (unsigned)(key & 0xffffffffull)) : 0x7fffffffll;
        }
    }
}

// ---- Whisper stem input: mel [C, nmel, L] -> melT [C, L+2, nmel] with zero rows 0 and L+1 so that
//      the k=3/pad=1 Conv1d becomes a GEMM over overlapping row views (TP whisper:566-567,618) ----
template <typename T>
__global__ void mel_transpose_pad_kernel(const u16* __restrict__ mel, u16* __restrict__ out, int C, int nmel, int L) {
    const size_t total = (size_t)C * (L + 2) * nmel;
    GRID_STRIDE(idx, total) {
        const int k = idx % nmel;
        const int r = (idx / nmel) % (L + 2);
        const size_t c = idx / ((size_t)nmel * (L + 2));
        out[idx] = (r == 0 || r == L + 1) ? (u16)0 : mel[(c * nmel + k) * L + (r - 1)];
    }
}

// ---- y = T(x * s) — `embeds * normalizer` (gemma.py:353-356) for externally supplied embeddings ---
template <typename T>
__global__ void scale_kernel(const u16* x, u16* y, size_t n8, float s) {
    GRID_STRIDE(idx, n8) {
        float v[8];
        unpack8<T>(*(const u32x4*)(x + idx * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= s;
        *(u32x4*)(y + idx * 8) = pack8<T>(v);
    }
}

// ---- flag |= any(x != 0) — `torch.sum(torch.abs(x)) != 0` sample mask (multimodal.py:202,246) -----
__global__ void any_nonzero_kernel(const u16* x, size_t n8, size_t n, int* flag) {
    int found = 0;
    if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) found |= (x[n8 * 8 + threadIdx.x] & 0x7fffu) != 0;   // tail
    GRID_STRIDE(idx, n8) {
        const u32x4 v = *(const u32x4*)(x + idx * 8);
        // +0 and -0 both count as zero
        found |= ((v[0] | v[1] | v[2] | v[3]) & 0x7fff7fffu) != 0;
    }
    if (__any(found)) {
        if ((threadIdx.x & 63) == 0) atomicOr(flag, 1);
    }
}

// ---- FractionalSinusoidalEmbedding — mm_vision/pos.py:11-26,47-53 (fp32) -------------------------
//   p = i/(l-1)*(N-1) ; pe[i][2j] = sin(p*div[j]) ; pe[i][2j+1] = cos(p*div[j]); i = i0 + local row
__global__ void sinusoid_kernel(float* __restrict__ pe, const float* __restrict__ div, int rows, int i0, int l, int N, int d) {
    const size_t total = (size_t)rows * (d / 2);
    GRID_STRIDE(idx, total) {
        const int j = idx % (d / 2);
        const size_t r = idx / (d / 2);
        float p = (float)(i0 + (long long)r);
        p = p / (float)(l - 1) * (float)(N - 1);
        const float a = p * div[j];
        pe[r * d + 2 * j] = sinf(a);
        pe[r * d + 2 * j + 1] = cosf(a);
    }
}

// =============================================================================================

template <typename T>
static int ew_dispatch_T(int op, void** a, const long long* i, const float* f, hipStream_t st) {
    switch (op) {
        case EW_IM2COL: {
            const size_t total = (size_t)i[0] * (i[1] / i[2]) * (i[1] / i[2]) * i[3];
            hipLaunchKernelGGL(im2col_patch_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const u16*)a[0], (u16*)a[1],
                               (int)i[0], (int)i[1], (int)i[2], (int)i[3]);
            break;
        }
        case EW_POOL: {
            if (i[3] % i[5] || i[4] % i[5]) return VIDI_ERR_SHAPE;
            const size_t total = (size_t)i[0] * (i[3] / i[5]) * (i[4] / i[5]) * i[2];
            hipLaunchKernelGGL(pool_s2d_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const u16*)a[0], (u16*)a[1],
                               (int)i[0], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (int)i[5], (int)i[6]);
            break;
        }
        case EW_ADDPOS: {
            if (i[3] % 8) return VIDI_ERR_SHAPE;
            const size_t total = (size_t)i[0] * i[1] * i[2] * (i[3] / 8);
            hipLaunchKernelGGL(add_pos_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (u16*)a[0], (const u16*)a[1],
                               (const u16*)a[2], (const u16*)a[3], (int)i[0], (int)i[1], (int)i[2], (int)i[3]);
            break;
        }
        case EW_ADD3: {
            if (i[0] % 8) return VIDI_ERR_SHAPE;
            const size_t n8 = (size_t)i[0] / 8;
            hipLaunchKernelGGL(add3_kernel<T>, dim3(grid_for(n8)), dim3(256), 0, st, (const u16*)a[0], (const u16*)a[1],
                               (const u16*)a[2], (u16*)a[3], n8);
            break;
        }
        case EW_EMBED: {
            if (i[1] % 8) return VIDI_ERR_SHAPE;
            const size_t total = (size_t)i[0] * (i[1] / 8);
            hipLaunchKernelGGL(embed_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const long long*)a[0], (const u16*)a[1],
                               (u16*)a[2], (int)i[0], (int)i[1], f[0], i[2]);
            break;
        }
        case EW_GEGLU_UNPACK: {
            if (i[1] % 32) return VIDI_ERR_SHAPE;
            const size_t total = (size_t)i[0] * i[1];
            hipLaunchKernelGGL(geglu_unpack_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const u16*)a[0], (u16*)a[1],
                               (int)i[0], (int)i[1], (int)(i[2] == ACT_SILU));
            break;
        }
        case EW_SOFTCAP_ARGMAX: {
            const int B = (int)i[0], V = (int)i[1];
            const int vec = (V % 8 == 0) && (i[2] % 8 == 0) && (((uintptr_t)a[0] & 15) == 0);
            const int items = vec ? V / 8 : V;
            const int nblk = max(1, min(128, (items + 255) / 256));
            for (int b0 = 0; b0 < B; b0 += 32768)             // gridDim.y limit
                hipLaunchKernelGGL(softcap_argmax_kernel<T>, dim3(nblk, min(32768, B - b0)), dim3(256), 0, st,
                                   (u16*)a[0] + (size_t)b0 * i[2], (long long*)a[1] + b0, V, i[2], f[0], vec, (AmSlot*)a[2] + b0);
            break;
        }
        case EW_MEL_T: {
            const size_t total = (size_t)i[0] * (i[2] + 2) * i[1];
            hipLaunchKernelGGL(mel_transpose_pad_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const u16*)a[0], (u16*)a[1],
                               (int)i[0], (int)i[1], (int)i[2]);
            break;
        }
        case EW_SCALE: {
            if (i[0] % 8) return VIDI_ERR_SHAPE;
            const size_t n8 = (size_t)i[0] / 8;
            hipLaunchKernelGGL(scale_kernel<T>, dim3(grid_for(n8)), dim3(256), 0, st, (const u16*)a[0], (u16*)a[1], n8, f[0]);
            break;
        }
        case EW_IM2COL_NHWC: {
            const long long oc = i[1] - i[3] + 1;
            const size_t total = (size_t)i[0] * oc * oc * i[3] * i[3] * (i[2] / 8);
            hipLaunchKernelGGL(im2col_nhwc_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const u16*)a[0], (u16*)a[1],
                               (int)i[0], (int)i[1], (int)i[2], (int)i[3]);
            break;
        }
        case EW_RESIZE_AC: {
            const size_t total = (size_t)i[0] * i[2] * i[2] * (i[3] / 8);
            hipLaunchKernelGGL(resize_ac_kernel<T>, dim3(grid_for(total)), dim3(256), 0, st, (const u16*)a[0], (u16*)a[1],
                               (int)i[0], (int)i[1], (int)i[2], (int)i[3]);
            break;
        }
        case EW_ANY_NONZERO: {
            const size_t n8 = (size_t)i[0] / 8;
            hipLaunchKernelGGL(any_nonzero_kernel, dim3(grid_for(n8 ? n8 : 1)), dim3(256), 0, st, (const u16*)a[0], n8, (size_t)i[0], (int*)a[1]);
            break;
        }
        default: return VIDI_ERR_ARG;
    }
    return (int)hipGetLastError();
}

int vidi_ew_dispatch(int op, void** a, const long long* i, const float* f, int dtype, hipStream_t st) {
    if (dtype == VIDI_DT_BF16) return ew_dispatch_T<BF16>(op, a, i, f, st);
    if (dtype == VIDI_DT_F16) return ew_dispatch_T<F16>(op, a, i, f, st);
    return VIDI_ERR_DTYPE;
}

int vidi_sinusoid_dispatch(float* pe, const float* div, int rows, int i0, int l, int N, int d, hipStream_t st) {
    if (rows <= 0 || l < 2 || d % 2) return VIDI_ERR_SHAPE;
    const size_t total = (size_t)rows * (d / 2);
    hipLaunchKernelGGL(sinusoid_kernel, dim3(grid_for(total)), dim3(256), 0, st, pe, div, rows, i0, l, N, d);
    return (int)hipGetLastError();
}
