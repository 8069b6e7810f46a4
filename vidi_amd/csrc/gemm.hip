// MFMA GEMM family for the Vidi hot path (gfx950).
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )        X:[M,K] activations, W:[N,K] nn.Linear weight
//
// Both operands are K-contiguous, so both are staged with 16-byte `global_load_lds` DMA into an
// XOR-swizzled LDS image and read back as 8-element MFMA fragments with ds_read_b128.  The MFMA is
// issued "swapped": the weight tile is the A operand and the activation tile the B operand, so the
// 32x32 accumulator holds D[row = n][col = m].  Each lane then owns ONE token (m = lane & 31) and
// four consecutive output features per register quad, which makes every epilogue a plain 8-byte
// store and lets GeGLU / bias / residual / KV-cache layouts be applied per lane without shuffles.
//
// Roofline: MFMA-bound (2.5 PFLOP/s dense bf16/fp16).  Algorithmic FLOPs = 2*M*N*K.
#include "kernels.h"
#include <stdlib.h>
#include "gemm_w4_launch.h"

// ---- conservative variant: register-staged (global_load -> ds_write), same LDS image ----------
// Used by the self-test to cross-check the LDS-DMA path and as a fallback selectable from the ABI.
template <typename T, int MODE, bool REPKV>
__global__ __launch_bounds__(256) void gemm_kernel_regstage(GemmParams p) {
    constexpr int BN = 128, BM = 128, WM = 2, TN = 2, TM = 2, NT = 256;
    __shared__ __attribute__((aligned(16))) char smem[(BN + BM) * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave / WM, wm = wave % WM;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int n0 = tile_n * BN, m0 = tile_m * BM;
    const long long bz = blockIdx.y;
    const u16* Xb = p.X + bz * p.bsX;
    f32x16 acc[TN][TM];
    for (int a = 0; a < TN; ++a) for (int b = 0; b < TM; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
    char* sW = smem; char* sX = smem + BN * 128;
    for (int kt = 0; kt < p.K / 64; ++kt) {
        const int k0 = kt * 64;
        u32x4 wr[4], xr[4];
        for (int j = 0; j < 4; ++j) {
            const int pidx = j * NT + tid, row = pidx >> 3, c = pidx & 7;
            const int n = min(n0 + row, p.N - 1), m = min(m0 + row, p.M - 1);
            wr[j] = *(const u32x4*)(p.W + (size_t)n * p.ldw + k0 + c * 8);
            int k = k0 + c * 8;
            if constexpr (REPKV) k = (k / (p.rep_g * p.rep_hd)) * p.rep_hd + (k % p.rep_hd);
            xr[j] = *(const u32x4*)(Xb + (size_t)m * p.ldx + k);
        }
        __syncthreads();
        for (int j = 0; j < 4; ++j) {
            const int pidx = j * NT + tid, row = pidx >> 3, c = pidx & 7;
            const int off = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
            *(u32x4*)(sW + off) = wr[j];
            *(u32x4*)(sX + off) = xr[j];
        }
        __syncthreads();
        for (int s = 0; s < 4; ++s) {
            const int coff = ((2 * s + hi) ^ sw) << 4;
            u32x4 wf[TN], xf[TM];
            for (int a = 0; a < TN; ++a) wf[a] = *(const u32x4*)(sW + (wn * 64 + a * 32 + l31) * 128 + coff);
            for (int b = 0; b < TM; ++b) xf[b] = *(const u32x4*)(sX + (wm * 64 + b * 32 + l31) * 128 + coff);
            for (int a = 0; a < TN; ++a) for (int b = 0; b < TM; ++b) acc[a][b] = T::mfma32(wf[a], xf[b], acc[a][b]);
        }
    }
    u16* Yb = p.Y + bz * p.bsY;
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * 64 + b * 32 + l31;
        if (m >= p.M) continue;
        for (int a = 0; a < TN; ++a) for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + a * 32 + 8 * j + 4 * hi;
            if (n >= p.N) continue;
            float v[4];
            for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * j + e];
            if (p.bias) for (int e = 0; e < 4; ++e) v[e] += T::to_f32(p.bias[n + e]);
            const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
            *(u32x2*)(Yb + (size_t)m * p.ldy + n) = o;
        }
    }
}

// ---- fp32 GEMM on f32-input MFMA (exact fp32): the positional-embedding MLPs run in fp32 -------
//   Y[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]); all fp32, K % 16 == 0.
//   mfma_f32_32x32x2f32: A lane l = A[i=l&31][k=l>>5], B lane l = B[k=l>>5][j=l&31].
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ Y,
                                                       int M, int N, int K, int ldx, int ldw, int ldy, int act) {
    constexpr int BN = 128, BM = 128, BK = 16, LDT = BK + 1;   // +1 pad: conflict-free column reads
    __shared__ float sW[BN * LDT], sX[BM * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave >> 1, wm = wave & 1;
    const int tiles_n = (N + BN - 1) / BN;
    const int n0 = (blockIdx.x % tiles_n) * BN, m0 = (blockIdx.x / tiles_n) * BM;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    const int l31 = lane & 31, hi = lane >> 5;
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        for (int i = tid; i < BN * BK / 4; i += 256) {              // 4 floats per thread-step
            const int row = i / (BK / 4), c4 = (i % (BK / 4)) * 4;
            const int n = min(n0 + row, N - 1), m = min(m0 + row, M - 1);
            const f32x4 wv = *(const f32x4*)(W + (size_t)n * ldw + k0 + c4);
            const f32x4 xv = *(const f32x4*)(X + (size_t)m * ldx + k0 + c4);
            for (int e = 0; e < 4; ++e) { sW[row * LDT + c4 + e] = wv[e]; sX[row * LDT + c4 + e] = xv[e]; }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float wf[2], xf[2];
            for (int a = 0; a < 2; ++a) wf[a] = sW[(wn * 64 + a * 32 + l31) * LDT + 2 * s + hi];
            for (int b = 0; b < 2; ++b) xf[b] = sX[(wm * 64 + b * 32 + l31) * LDT + 2 * s + hi];
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
    }
    for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + l31;
        if (m >= M) continue;
        for (int a = 0; a < 2; ++a) for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + a * 32 + 8 * j + 4 * hi;
            for (int e = 0; e < 4; ++e) {
                if (n + e >= N) continue;
                float v = acc[a][b][4 * j + e] + (bias ? bias[n + e] : 0.f);
                if (act == ACT_GELU_ERF) v = gelu_erf_f(v);
                Y[(size_t)m * ldy + n + e] = v;
            }
        }
    }
}

// =============================================================================================
// host-side dispatch (C ABI in capi.hip calls these)
// =============================================================================================
template <typename K>
static hipError_t set_lds(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <typename T, int BN, int BM, int WN, int WM, int STAGES, int MODE, bool REPKV, int SCHED, int MI>
static int launch_cfg(const GemmParams& p, int batch, hipStream_t st) {
    constexpr int RING = STAGES * (BN + BM) * 64 * 2;
    constexpr int CTILE = BM * ((MODE == MODE_GEGLU ? BN / 2 : BN) * 2 + 16);
    constexpr int LDS = RING > CTILE ? RING : CTILE;
    auto kern = gemm_kernel<T, BN, BM, WN, WM, STAGES, MODE, REPKV, SCHED, MI, LabNone>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = set_lds(kern, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
    hipLaunchKernelGGL(kern, dim3(tiles, batch), dim3(WN * WM * 64), LDS, st, p);
    return (int)hipGetLastError();
}

// persistent 4-wave kernel (gemm_w4.h); VIDI_W4_UNSUPPORTED when this epilogue combination is not instantiated
template <typename T, int MODE, bool REPKV>
static int launch_w4_any(const GemmParams& p, int batch, hipStream_t st) {
    if (p.K % 64 || p.K < 192 || (REPKV && (p.rep_hd % 64))) return VIDI_W4_UNSUPPORTED;      // the K loop needs >= 3 slices
    if constexpr (MODE == MODE_PLAIN) {
        if constexpr (T::id == VIDI_DT_BF16) return vidi_w4_plain_bf16(p, batch, REPKV ? 1 : 0, st);
        else return vidi_w4_plain_f16(p, batch, REPKV ? 1 : 0, st);
    } else {
        return vidi_w4_modes(p, batch, MODE, T::id, st);
    }
}

template <typename T, int MODE, bool REPKV>
static int launch_mode(const GemmParams& p, int batch, int tile_cfg, hipStream_t st) {
    if (tile_cfg == 3) {
        if (p.ln_stats) return VIDI_ERR_ARG;
        if constexpr (MODE == MODE_PLAIN) {
            const int tiles = ((p.N + 127) / 128) * ((p.M + 127) / 128);
            hipLaunchKernelGGL((gemm_kernel_regstage<T, MODE, REPKV>), dim3(tiles, batch), dim3(256), 0, st, p);
            return (int)hipGetLastError();
        } else {
            return VIDI_ERR_ARG;
        }
    }
    if (tile_cfg < 0) {
        // measured on MI355X (tools/lab): the 256x256 tile wins from ~1 wave of blocks up, also when N is not a multiple of
        // 256 (edge tiles are masked); small problems take the 128x128 tile
        const long long t256 = (long long)((p.N + 255) / 256) * ((p.M + 255) / 256) * batch;
        static int big = -1;                      // VIDI_GEMM_CFG: 4 or 5 (A/B of the two large-tile kernels; same results)
        if (big < 0) { const char* e = getenv("VIDI_GEMM_CFG"); big = e ? atoi(e) : 5; if (big != 4 && big != 5) big = 5; }
        if (t256 >= 192 && (big == 5 || p.ln_stats)) {            // persistent 4-wave kernel; the 8-wave kernel covers epilogue combinations it lacks
            const int rc = launch_w4_any<T, MODE, REPKV>(p, batch, st);
            if (rc != VIDI_W4_UNSUPPORTED) return rc;
        }
        tile_cfg = (t256 >= 192 && !p.ln_stats) ? 4 : 0;
    }
    if ((p.ln_stats || p.hm_seq) && tile_cfg != 0 && tile_cfg != 5) return VIDI_ERR_ARG;       // folded LayerNorm: persistent kernel or the 128x128 tile
    switch (tile_cfg) {
        case 0: return launch_cfg<T, 128, 128, 2, 2, 2, MODE, REPKV, SCHED_RING, 32>(p, batch, st);
        case 1: return launch_cfg<T, 128, 256, 2, 4, 3, MODE, REPKV, SCHED_RING, 32>(p, batch, st);
        case 2: return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV, SCHED_RING, 32>(p, batch, st);
        case 4: return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV, SCHED_LATE, 16>(p, batch, st);    // 8 waves x (128x64), 16x16x32 MFMA
        case 5: {                                                                                      // persistent, 4 waves x (128x128)
            const int rc = launch_w4_any<T, MODE, REPKV>(p, batch, st);
            return rc == VIDI_W4_UNSUPPORTED ? VIDI_ERR_ARG : rc;
        }
        case 11:                                                                                        // 192-wide n tile (N = 1152: 6 exact tiles instead of 4.5)
            if constexpr (MODE == MODE_PLAIN) return launch_cfg<T, 192, 256, 2, 4, 2, MODE, REPKV, SCHED_LATE, 16>(p, batch, st);
            else return VIDI_ERR_ARG;
        default: return VIDI_ERR_ARG;
    }
}

template <typename T>
static int launch_dtype(const GemmParams& p, int batch, int mode, int repkv, int tile_cfg, hipStream_t st) {
    switch (mode) {
        case MODE_PLAIN: return repkv ? launch_mode<T, MODE_PLAIN, true>(p, batch, tile_cfg, st)
                                      : launch_mode<T, MODE_PLAIN, false>(p, batch, tile_cfg, st);
        case MODE_GEGLU: return launch_mode<T, MODE_GEGLU, false>(p, batch, tile_cfg, st);
        case MODE_QKV_VT: return launch_mode<T, MODE_QKV_VT, false>(p, batch, tile_cfg, st);
        case MODE_KV_CACHE: return launch_mode<T, MODE_KV_CACHE, false>(p, batch, tile_cfg, st);
        default: return VIDI_ERR_ARG;
    }
}

int vidi_gemm_dispatch(const GemmParams& p, int batch, int mode, int repkv, int tile_cfg, int dtype, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || batch <= 0) return VIDI_ERR_SHAPE;
    if (p.K % 64 != 0 || p.N % 32 != 0) return VIDI_ERR_SHAPE;
    if (mode == MODE_GEGLU && p.N % 64 != 0) return VIDI_ERR_SHAPE;
    if ((p.ldx % 8) || (p.ldw % 8) || (p.ldy % 8)) return VIDI_ERR_ALIGN;
    if (((uintptr_t)p.X & 15) || ((uintptr_t)p.W & 15) || ((uintptr_t)p.Y & 15)) return VIDI_ERR_ALIGN;
    if (p.R && ((p.ldr % 8) || ((uintptr_t)p.R & 15))) return VIDI_ERR_ALIGN;
    if (mode == MODE_KV_CACHE && (((uintptr_t)p.Kc & 15) || ((uintptr_t)p.Vrow & 15) || (p.hd % 8))) return VIDI_ERR_ALIGN;
    if (dtype == VIDI_DT_BF16) return launch_dtype<BF16>(p, batch, mode, repkv, tile_cfg, st);
    if (dtype == VIDI_DT_F16) return launch_dtype<F16>(p, batch, mode, repkv, tile_cfg, st);
    return VIDI_ERR_DTYPE;
}

int vidi_gemm_f32_dispatch(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K,
                           int ldx, int ldw, int ldy, int act, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 16 != 0 || (ldx % 4) || (ldw % 4)) return VIDI_ERR_SHAPE;
    const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, st, X, W, bias, Y, M, N, K, ldx, ldw, ldy, act);
    return (int)hipGetLastError();
}
