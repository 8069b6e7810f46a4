// Bench-only GEMM lab (NOT part of libvidi_hip.so): times schedules of vidi_amd/csrc/gemm_tile.h on the 60-min workload's
// GEMM shapes with random bf16 data, checks every variant's output checksum against the shipped schedule (all correct-result
// variants must be bit-identical), and hosts the timing diagnostics (LAB policies: no DMA / no epilogue / phase stamps) that
// the product library does not contain.  No torch, no Python: one hipcc binary, seconds per run on the GPU box.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vidi_amd/csrc tools/lab/gemm_lab.hip -o tools/lab/gemm_lab
//   tools/lab/gemm_lab [set]          set: all | quick | stamps | store
#include "gemm_w4.h"
#include "gemm_w4n.h"
#include "gemm_w4h.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct LabNoDma : LabNone { static constexpr bool no_dma = true; };
struct LabNoEpi : LabNone { static constexpr bool no_epilogue = true; };
struct LabNoStore : LabNone { static constexpr bool no_store = true; };
struct LabStamps : LabNone { static constexpr bool stamps = true; };
struct LabStampsNoEpi : LabStamps { static constexpr bool no_epilogue = true; };
struct LabStampsNoEpiNoDma : LabStampsNoEpi { static constexpr bool no_dma = true; };

__global__ void fill_kernel(u16* p, size_t n, unsigned seed, float scale, int zero) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed ^ (unsigned)(i >> 32) * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float u = ((h & 0xffffff) / 8388608.0f - 1.0f) * scale;      // uniform [-scale, scale)
        p[i] = zero ? (u16)0 : f32_to_bf16(u);
    }
}
__global__ void checksum_kernel(const u16* p, size_t n, unsigned long long* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long s = 0, x = 0;
    for (; i < n; i += stride) { const unsigned long long v = p[i]; s += v * (unsigned long long)((i % 1021) + 1); x ^= (v << (i % 47)); }
    atomicAdd(out, s);
    atomicXor(out + 1, x);
}

// LayerNorm-fold operands of the lab: per-row (mean, rstd) of X (one thread per row, fp32), per-column sums of W, zero shift
__global__ void rowstats_kernel(const u16* X, float* st, int M, int K, float eps) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < K; ++k) { const float v = bf16_to_f32(X[(size_t)m * K + k]); s1 += v; s2 = __builtin_fmaf(v, v, s2); }
    const float mean = s1 / K;
    st[2 * m] = mean; st[2 * m + 1] = rsqrtf(fmaxf(s2 / K - mean * mean, 0.f) + eps);
}
__global__ void colsum_kernel(const u16* W, float* cs, float* sh, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += bf16_to_f32(W[(size_t)n * K + k]);
    cs[n] = s; sh[n] = 0.01f * (float)(n % 17);
}
// out[0] = max |a - b| (as float bits), out[1] = number of elements that differ by more than one bf16 ulp of the larger magnitude
__global__ void ydiff_kernel(const u16* A, const u16* B, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float d = 0.f; unsigned bad = 0;
    for (; i < n; i += stride) {
        const float a = bf16_to_f32(A[i]), b = bf16_to_f32(B[i]);
        const float e = fabsf(a - b);
        d = fmaxf(d, e);
        if (e > fmaxf(fabsf(a), fabsf(b)) * (1.0f / 128.0f) + 1e-3f) ++bad;
    }
    atomicMax(out, __float_as_uint(d));
    atomicAdd(out + 1, bad);
}

// store-path micro-benchmark: every block writes `bytes` with 16-byte coalesced stores
__global__ void store_kernel(u32x4* out, int chunks_per_thread) {
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    u32x4* o = out + (size_t)blockIdx.x * blockDim.x * chunks_per_thread;
    for (int i = 0; i < chunks_per_thread; ++i) o[(size_t)i * blockDim.x + threadIdx.x] = v;
}

typedef int (*launch_fn)(const GemmParams&, hipStream_t);

template <typename T, int BN, int BM, int WN, int WM, int STAGES, int MODE, bool REPKV, int SCHED, int MI, typename LAB>
static int lab_launch(const GemmParams& p, hipStream_t st) {
    constexpr int RING = STAGES * (BN + BM) * 64 * 2;
    constexpr int CTILE = BM * ((MODE == MODE_GEGLU ? BN / 2 : BN) * 2 + 16);
    constexpr int LDS = RING > CTILE ? RING : CTILE;
    auto kern = gemm_kernel<T, BN, BM, WN, WM, STAGES, MODE, REPKV, SCHED, MI, LAB>;
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); done = true; }
    const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
    hipLaunchKernelGGL(kern, dim3(tiles, 1), dim3(WN * WM * 64), LDS, st, p);
    return (int)hipGetLastError();
}

template <typename T, int MODE, bool PERSIST, typename LAB, typename EPI = Epi<false, ACT_NONE, 0>>
static int lab_launch_w4(const GemmParams& p, hipStream_t st) {
    auto kern = gemm_w4_kernel<T, MODE, false, PERSIST, EPI, LAB>;
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W4Geom::LDS_BYTES)); done = true; }
    const int tiles = ((p.N + 255) / 256) * ((p.M + 255) / 256);
    const int grid = PERSIST ? (tiles < 256 ? tiles : 256) : tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), W4Geom::LDS_BYTES, st, p, 1);
    return (int)hipGetLastError();
}

template <typename EPI, typename LAB>
static int lab_launch_w4n(const GemmParams& p, hipStream_t st) { return launch_w4n<BF16, EPI, LAB>(p, st); }

template <typename EPI, typename LAB>
static int lab_launch_w4h(const GemmParams& p, hipStream_t st) { return launch_w4h<BF16, EPI, LAB>(p, st); }

struct Variant { const char* name; launch_fn fn; int mode; bool correct; int order; int group_m; int epi = 0; bool yonly = false; };   // yonly: checksum of Y alone, against the bias + residual reference   // epi: 0 none, 1 bias + residual, 2 bias + GELU(tanh), 3 bias + GELU(erf), 4 bias + residual + row statistics

#define LATE(MODE, LAB) lab_launch<BF16, 256, 256, 2, 4, 2, MODE, false, SCHED_LATE, 16, LAB>
#define W4(MODE, PERSIST, WAITMODE, LAB) lab_launch_w4<BF16, MODE, PERSIST, LAB>

static std::vector<Variant> variants() {
    return {
        {"late", LATE(MODE_PLAIN, LabNone), MODE_PLAIN, true, 0, 4},
        {"late_o1", LATE(MODE_PLAIN, LabNone), MODE_PLAIN, true, 1, 4},
        {"w4s", W4(MODE_PLAIN, false, 0, LabNone), MODE_PLAIN, true, 0, 4},          // strip epilogue, one block per tile
        {"w4p", W4(MODE_PLAIN, true, 0, LabNone), MODE_PLAIN, true, 0, 4},           // persistent, drain before the next tile
        {"w4pc", W4(MODE_PLAIN, true, 1, LabNone), MODE_PLAIN, true, 0, 4},          // persistent, counted wait
        {"w4p_o1", W4(MODE_PLAIN, true, 0, LabNone), MODE_PLAIN, true, 1, 4},
        {"w4pc_o1", W4(MODE_PLAIN, true, 1, LabNone), MODE_PLAIN, true, 1, 4},
        {"w4p_o1_g2", W4(MODE_PLAIN, true, 0, LabNone), MODE_PLAIN, true, 1, 2},
        {"w4p_g8", W4(MODE_PLAIN, true, 0, LabNone), MODE_PLAIN, true, 0, 8},
        {"w4p_o1_g8", W4(MODE_PLAIN, true, 0, LabNone), MODE_PLAIN, true, 1, 8},
        {"w4p_nodma", W4(MODE_PLAIN, true, 0, LabNoDma), MODE_PLAIN, false, 0, 4},
        {"w4p_noepi", W4(MODE_PLAIN, true, 0, LabNoEpi), MODE_PLAIN, false, 0, 4},
        {"w4p_nostore", W4(MODE_PLAIN, true, 0, LabNoStore), MODE_PLAIN, false, 0, 4},
        {"late_stamps", LATE(MODE_PLAIN, LabStamps), MODE_PLAIN, true, 0, 4},
        {"w4p_stamps", W4(MODE_PLAIN, true, 0, LabStamps), MODE_PLAIN, true, 0, 4},
        // round 4: tile order x effective clock (cycle stamps / wall time) — does fabric traffic cost clock?  group_m = 1: the 32 CUs of an
        // XCD share ONE X m-panel and walk n (33 operand slabs per 32 tiles instead of 12); group_m = 8 / 16: fewer, taller groups
        {"w4p_stamps_g1", W4(MODE_PLAIN, true, 0, LabStamps), MODE_PLAIN, true, 1, 1},
        {"w4p_stamps_g2", W4(MODE_PLAIN, true, 0, LabStamps), MODE_PLAIN, true, 1, 2},
        {"w4p_stamps_g4", W4(MODE_PLAIN, true, 0, LabStamps), MODE_PLAIN, true, 1, 4},
        {"w4p_stamps_g8", W4(MODE_PLAIN, true, 0, LabStamps), MODE_PLAIN, true, 1, 8},
        {"w4p_stamps_g16", W4(MODE_PLAIN, true, 0, LabStamps), MODE_PLAIN, true, 1, 16},
        {"w4p_stamps_o0g4", W4(MODE_PLAIN, true, 0, LabStamps), MODE_PLAIN, true, 0, 4},
        {"w4p_g1", W4(MODE_PLAIN, true, 0, LabNone), MODE_PLAIN, true, 1, 1},
        {"w4p_g2", W4(MODE_PLAIN, true, 0, LabNone), MODE_PLAIN, true, 1, 2},
        {"w4p_g16", W4(MODE_PLAIN, true, 0, LabNone), MODE_PLAIN, true, 1, 16},
        // round 4: 288 x 224 tiles (gemm_w4n.h) for N = 1 152: Y must equal the 256-wide kernel's bit for bit (the partial sums have another layout)
        {"w4n_br", lab_launch_w4n<Epi<true, ACT_NONE, 1>, LabNone>, MODE_PLAIN, true, 1, 4, 1, true},
        {"w4n_brs", lab_launch_w4n<Epi<true, ACT_NONE, 1, false, true>, LabNone>, MODE_PLAIN, true, 1, 4, 4, true},
        {"w4n_brs_g8", lab_launch_w4n<Epi<true, ACT_NONE, 1, false, true>, LabNone>, MODE_PLAIN, true, 1, 8, 4, true},
        {"w4n_brs_g2", lab_launch_w4n<Epi<true, ACT_NONE, 1, false, true>, LabNone>, MODE_PLAIN, true, 1, 2, 4, true},
        {"w4n_brs_stamps", lab_launch_w4n<Epi<true, ACT_NONE, 1, false, true>, LabStamps>, MODE_PLAIN, true, 1, 4, 4, true},
        {"w4n_br_noepi", lab_launch_w4n<Epi<true, ACT_NONE, 1>, LabNoEpi>, MODE_PLAIN, false, 1, 4, 1, true},
        // round 4 probe: 256 x 128 tiles (gemm_w4h.h), half the accumulator file: the K-loop rate a two-accumulator-set kernel would start from
        {"w4h_br", lab_launch_w4h<Epi<true, ACT_NONE, 1>, LabNone>, MODE_PLAIN, true, 1, 4, 1},
        {"w4h_brs", lab_launch_w4h<Epi<true, ACT_NONE, 1, false, true>, LabNone>, MODE_PLAIN, true, 1, 4, 4},
        {"w4h_br_g8", lab_launch_w4h<Epi<true, ACT_NONE, 1>, LabNone>, MODE_PLAIN, true, 1, 8, 1},
        {"w4h_br_noepi", lab_launch_w4h<Epi<true, ACT_NONE, 1>, LabNoEpi>, MODE_PLAIN, false, 1, 4, 1},
        {"w4h_br_nodma", lab_launch_w4h<Epi<true, ACT_NONE, 1>, LabNoDma>, MODE_PLAIN, false, 1, 4, 1},
        {"w4h_br_stamps", lab_launch_w4h<Epi<true, ACT_NONE, 1>, LabStamps>, MODE_PLAIN, true, 1, 4, 1},
        // the shader clock under the K loop alone (cycle stamps / wall time): is the MFMA-dense phase power-throttled below the whole kernel's clock?
        {"w4p_noepi_stamps", lab_launch_w4<BF16, MODE_PLAIN, true, LabStampsNoEpi, Epi<true, ACT_NONE, 1>>, MODE_PLAIN, false, 1, 4, 1},
        {"w4p_noepi_nodma_stamps", lab_launch_w4<BF16, MODE_PLAIN, true, LabStampsNoEpiNoDma, Epi<true, ACT_NONE, 1>>, MODE_PLAIN, false, 1, 4, 1},
        {"w4h_noepi_stamps", lab_launch_w4h<Epi<true, ACT_NONE, 1>, LabStampsNoEpi>, MODE_PLAIN, false, 1, 4, 1},
        {"w4h_noepi_nodma_stamps", lab_launch_w4h<Epi<true, ACT_NONE, 1>, LabStampsNoEpiNoDma>, MODE_PLAIN, false, 1, 4, 1},
        // next step (profiles/r4_notes.md): LayerNorm statistics in the consumer's K loop (Epi::lnf == 2) against the statistics input (lnf == 1)
        {"w4p_lnf1_bt", lab_launch_w4<BF16, MODE_PLAIN, true, LabNone, Epi<true, ACT_GELU_TANH, 0, 1>>, MODE_PLAIN, false, 1, 4, 5},
        {"w4p_lnf2_bt", lab_launch_w4<BF16, MODE_PLAIN, true, LabNone, Epi<true, ACT_GELU_TANH, 0, 2>>, MODE_PLAIN, false, 1, 4, 6},
        {"w4p_lnf1_noepi", lab_launch_w4<BF16, MODE_PLAIN, true, LabNoEpi, Epi<true, ACT_GELU_TANH, 0, 1>>, MODE_PLAIN, false, 1, 4, 5},
        {"w4p_lnf2_noepi", lab_launch_w4<BF16, MODE_PLAIN, true, LabNoEpi, Epi<true, ACT_GELU_TANH, 0, 2>>, MODE_PLAIN, false, 1, 4, 6},
        {"w4p_lnf1_stamps", lab_launch_w4<BF16, MODE_PLAIN, true, LabStamps, Epi<true, ACT_GELU_TANH, 0, 1>>, MODE_PLAIN, false, 1, 4, 5},
        {"w4p_lnf2_stamps", lab_launch_w4<BF16, MODE_PLAIN, true, LabStamps, Epi<true, ACT_GELU_TANH, 0, 2>>, MODE_PLAIN, false, 1, 4, 6},
        {"w4p_br_noepi_o1", lab_launch_w4<BF16, MODE_PLAIN, true, LabNoEpi, Epi<true, ACT_NONE, 1>>, MODE_PLAIN, false, 1, 4, 1},
        {"w4p_br_o1", lab_launch_w4<BF16, MODE_PLAIN, true, LabNone, Epi<true, ACT_NONE, 1>>, MODE_PLAIN, true, 1, 4, 1},
        {"w4p_brs_o1", lab_launch_w4<BF16, MODE_PLAIN, true, LabNone, Epi<true, ACT_NONE, 1, false, true>>, MODE_PLAIN, true, 1, 4, 4},
        {"w4p_brs_o1_stamps", lab_launch_w4<BF16, MODE_PLAIN, true, LabStamps, Epi<true, ACT_NONE, 1, false, true>>, MODE_PLAIN, true, 1, 4, 4},
        {"late_br", LATE(MODE_PLAIN, LabNone), MODE_PLAIN, true, 0, 4, 1},
        {"w4p_br", lab_launch_w4<BF16, MODE_PLAIN, true, LabNone, Epi<true, ACT_NONE, 1>>, MODE_PLAIN, true, 0, 4, 1},
        {"late_bt", LATE(MODE_PLAIN, LabNone), MODE_PLAIN, true, 0, 4, 2},
        {"w4p_bt", lab_launch_w4<BF16, MODE_PLAIN, true, LabNone, Epi<true, ACT_GELU_TANH, 0>>, MODE_PLAIN, true, 0, 4, 2},
        // N = 128 remainder column of the N = 1152 tower GEMMs (4.5 tiles of 256): candidates for a split launch
        {"t128x128", lab_launch<BF16, 128, 128, 2, 2, 2, MODE_PLAIN, false, SCHED_RING, 32, LabNone>, MODE_PLAIN, true, 0, 4},
        {"t128x256", lab_launch<BF16, 128, 256, 2, 4, 3, MODE_PLAIN, false, SCHED_RING, 32, LabNone>, MODE_PLAIN, true, 0, 4},
        {"late128x256", lab_launch<BF16, 128, 256, 2, 4, 2, MODE_PLAIN, false, SCHED_LATE, 16, LabNone>, MODE_PLAIN, true, 0, 4},
        {"t128x128_br", lab_launch<BF16, 128, 128, 2, 2, 2, MODE_PLAIN, false, SCHED_RING, 32, LabNone>, MODE_PLAIN, true, 0, 4, 1},
        {"late128x256_br", lab_launch<BF16, 128, 256, 2, 4, 2, MODE_PLAIN, false, SCHED_LATE, 16, LabNone>, MODE_PLAIN, true, 0, 4, 1},
        {"w4p_br_stamps", lab_launch_w4<BF16, MODE_PLAIN, true, LabStamps, Epi<true, ACT_NONE, 1>>, MODE_PLAIN, true, 0, 4, 1},
        {"w4p_bt_stamps", lab_launch_w4<BF16, MODE_PLAIN, true, LabStamps, Epi<true, ACT_GELU_TANH, 0>>, MODE_PLAIN, true, 0, 4, 2},
        {"w4p_brs_stamps", lab_launch_w4<BF16, MODE_PLAIN, true, LabStamps, Epi<true, ACT_NONE, 1, false, true>>, MODE_PLAIN, true, 0, 4, 4},
        {"w4p_brs", lab_launch_w4<BF16, MODE_PLAIN, true, LabNone, Epi<true, ACT_NONE, 1, false, true>>, MODE_PLAIN, true, 0, 4, 4},   // + per-strip row statistics
        {"w4p_be", lab_launch_w4<BF16, MODE_PLAIN, true, LabNone, Epi<true, ACT_GELU_ERF, 0>>, MODE_PLAIN, true, 0, 4, 3},
        {"late_geglu", LATE(MODE_GEGLU, LabNone), MODE_GEGLU, true, 0, 4},
        {"w4p_geglu", W4(MODE_GEGLU, true, 0, LabNone), MODE_GEGLU, true, 0, 4},
    };
}

struct Shape { const char* name; int M, N, K; };

int main(int argc, char** argv) {
    const std::string set = argc > 1 ? argv[1] : "quick";
    const int iters = argc > 2 ? atoi(argv[2]) : 5;
    const int zero = getenv("LAB_ZERO") ? 1 : 0;
    std::vector<Shape> shapes = {
        {"siglip_qkv", 262440, 3456, 1152}, {"siglip_o", 262440, 1152, 1152}, {"siglip_fc1", 262440, 4352, 1152},
        {"siglip_fc2", 262440, 1152, 4352}, {"mm_kv", 126080, 4096, 3584}, {"mm_o", 126080, 3584, 4096},
        {"mm_down", 126080, 3584, 14336}, {"whisper_fc1", 180000, 5120, 1280}, {"sq8k", 8192, 8192, 8192}, {"mm_gateup", 126080, 28672, 3584},
        {"o_n1024", 262440, 1024, 1152}, {"fc2_n1024", 262440, 1024, 4352}, {"o_rem128", 262440, 128, 1152}, {"fc2_rem128", 262440, 128, 4352},
    };
    std::vector<std::string> want;
    if (set == "quick") want = {"late", "w4s", "w4p", "w4pc"};
    else if (set == "order") want = {"late", "late_o1", "w4p", "w4p_o1", "w4pc_o1", "w4p_o1_g2", "w4p_g8", "w4p_o1_g8"};
    else if (set == "diag") want = {"late", "w4p", "w4p_nodma", "w4p_noepi", "w4p_nostore"};
    else if (set == "stamps") want = {"late_stamps", "w4p_stamps"};
    else if (set == "geglu") want = {"late_geglu", "w4p_geglu"};
    else if (set == "epi") want = {"late", "w4p", "w4pc", "late_br", "w4p_br", "late_bt", "w4p_bt", "late_geglu", "w4p_geglu"};
    else if (set == "all") want = {"late", "late_o1", "w4s", "w4p", "w4pc", "w4p_o1", "w4pc_o1", "w4p_o1_g2", "w4p_g8", "w4p_o1_g8", "w4p_nodma", "w4p_noepi", "w4p_nostore", "late_stamps", "w4p_stamps", "late_geglu", "w4p_geglu"};
    else if (set == "w4n") want = {"w4p_br_o1", "w4n_br", "w4p_brs_o1", "w4n_brs", "w4n_brs_g8", "w4n_brs_g2", "w4n_br_noepi", "w4p_brs_o1_stamps", "w4n_brs_stamps"};
    else if (set == "w4h") want = {"w4p_br_o1", "w4h_br", "w4p_brs_o1", "w4h_brs", "w4h_br_g8", "w4p_br_noepi_o1", "w4h_br_noepi", "w4h_br_nodma", "w4h_br_stamps"};
    else if (set == "lnstats") want = {"w4p_lnf1_bt", "w4p_lnf2_bt", "w4p_lnf1_noepi", "w4p_lnf2_noepi", "w4p_lnf1_stamps", "w4p_lnf2_stamps"};
    else if (set == "kclock") want = {"w4p_br_stamps", "w4p_noepi_stamps", "w4p_noepi_nodma_stamps", "w4h_br_stamps", "w4h_noepi_stamps", "w4h_noepi_nodma_stamps"};
    else if (set == "clock") want = {"w4p_stamps_g1", "w4p_stamps_g2", "w4p_stamps_g4", "w4p_stamps_g8", "w4p_stamps_g16", "w4p_stamps_o0g4"};
    else if (set == "orders") want = {"w4p_g1", "w4p_g2", "w4p_o1", "w4p_o1_g8", "w4p_g16", "w4p"};
    else if (set == "stamps_epi") want = {"w4p_stamps", "w4p_br_stamps", "w4p_brs_stamps", "w4p_bt_stamps"};
    else if (set == "split") want = {"w4p", "w4p_br", "t128x128", "t128x256", "late128x256", "t128x128_br", "late128x256_br"};
    else if (set == "store") want = {};
    else { want = {set}; }
    const char* only = getenv("LAB_SHAPE");

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* dsum; CK(hipMalloc(&dsum, 16));
    unsigned long long* dbg; CK(hipMalloc(&dbg, 1024 * 8 * 8));

    if (set == "store" || set == "all") {
        // 128 KB per block (one 256x256 bf16 tile), 1 block per CU worth of blocks x 8 rounds; 256 vs 512 threads; fewer active CUs
        u32x4* buf; const size_t bytes = (size_t)2048 * 128 * 1024; CK(hipMalloc(&buf, bytes));
        for (int threads : {256, 512}) for (int blocks : {32, 64, 128, 256, 2048}) {
            const int cpt = 128 * 1024 / 16 / threads;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(threads), 0, 0, buf, cpt);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("{\"bench\": \"store\", \"threads\": %d, \"blocks\": %d, \"us_per_launch\": %.2f, \"GBps\": %.1f, \"B_per_clk_per_block_at_2GHz\": %.2f}\n",
                                threads, blocks, ms * 250.0, 4.0 * blocks * 131072.0 / ms / 1e6, 131072.0 / (ms * 250.0 * 1e-6 * 2e9) * (blocks > 256 ? blocks / 256.0 : 1.0));
            }
        }
        CK(hipFree(buf));
        fflush(stdout);
    }

    auto vs = variants();
    for (const Shape& sh : shapes) {
        if (only && strcmp(only, sh.name)) continue;
        if (want.empty()) break;
        u16 *X, *W, *Y, *R, *Bv; float* SP;
        const size_t nx = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, ny = (size_t)sh.M * sh.N;
        const size_t spn = (size_t)sh.M * (size_t)((sh.N + 127) / 128 > w4n_stat_strips(sh.N) ? (sh.N + 127) / 128 : w4n_stat_strips(sh.N)) * 8;
        CK(hipMalloc(&X, nx * 2)); CK(hipMalloc(&W, nw * 2)); CK(hipMalloc(&Y, ny * 2)); CK(hipMalloc(&R, ny * 2)); CK(hipMalloc(&Bv, (size_t)sh.N * 2)); CK(hipMalloc(&SP, spn));
        float *LNst = nullptr, *LNs = nullptr, *LNc = nullptr; u16* Yref = nullptr;
        const bool want_ln = set == "lnstats";
        if (want_ln) {
            CK(hipMalloc(&LNst, (size_t)sh.M * 8)); CK(hipMalloc(&LNs, (size_t)sh.N * 4)); CK(hipMalloc(&LNc, (size_t)sh.N * 4)); CK(hipMalloc(&Yref, ny * 2));
        }
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, R, ny, 0x5555u, 1.0f, zero);
        hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, Bv, (size_t)sh.N, 0x7777u, 0.5f, zero);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, nx, 0x1234u, 1.7f, zero);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, W, nw, 0x9876u, 0.035f, zero);
        if (want_ln) {
            hipLaunchKernelGGL(rowstats_kernel, dim3((sh.M + 255) / 256), dim3(256), 0, 0, X, LNst, sh.M, sh.K, 1e-6f);
            hipLaunchKernelGGL(colsum_kernel, dim3((sh.N + 255) / 256), dim3(256), 0, 0, W, LNs, LNc, sh.N, sh.K);
        }
        CK(hipDeviceSynchronize());
        unsigned long long ref[16][2] = {};
        bool have_ref[16] = {};
        for (const std::string& wn : want) {
            const Variant* v = nullptr;
            for (auto& c : vs) if (wn == c.name) v = &c;
            if (!v) { printf("unknown variant %s\n", wn.c_str()); continue; }
            if (v->mode == MODE_GEGLU && (sh.N % 64)) continue;
            GemmParams p; memset(&p, 0, sizeof(p));
            p.X = X; p.W = W; p.Y = Y; p.M = sh.M; p.N = sh.N; p.K = sh.K; p.ldx = sh.K; p.ldw = sh.K;
            p.ldy = v->mode == MODE_GEGLU ? sh.N / 2 : sh.N; p.rmod = 0x7fffffff; p.group_m = v->group_m; p.order = v->order; p.act = ACT_GELU_TANH * (v->mode == MODE_GEGLU);
            p.dbg = dbg;
            if (v->epi == 1) { p.bias = Bv; p.R = R; p.ldr = sh.N; }
            if (v->epi == 2) { p.bias = Bv; p.act = ACT_GELU_TANH; }
            if (v->epi == 3) { p.bias = Bv; p.act = ACT_GELU_ERF; }
            if (v->epi == 4) { p.bias = Bv; p.R = R; p.ldr = sh.N; p.stat_part = SP; CK(hipMemset(SP, 0, spn)); }
            if (v->epi == 5 || v->epi == 6) {
                if (!want_ln) continue;
                p.bias = Bv; p.act = ACT_GELU_TANH; p.ln_s = LNs; p.ln_c = LNc; p.ln_eps = 1e-6f;
                p.ln_stats = v->epi == 5 ? LNst : nullptr;
            }
            if (v->yonly && !w4n_takes(sh.N)) continue;
            CK(hipMemset(Y, 0xff, ny * 2));
            CK(hipMemset(dbg, 0, 1024 * 64));
            int rc = v->fn(p, 0);
            if (rc) { printf("launch %s failed %d\n", v->name, rc); continue; }
            CK(hipDeviceSynchronize());
            CK(hipMemset(dsum, 0, 16));
            const size_t nyo = v->mode == MODE_GEGLU ? ny / 2 : ny;
            hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, 0, Y, nyo, dsum);
            if (v->epi == 4 && !v->yonly) hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, 0, (const u16*)SP, (size_t)sh.M * ((sh.N + 127) / 128) * 4, dsum);   // the statistics too
            unsigned long long cs[2]; CK(hipMemcpy(cs, dsum, 16, hipMemcpyDeviceToHost));
            if (wn == "w4p_lnf1_bt") CK(hipMemcpy(Yref, Y, ny * 2, hipMemcpyDeviceToDevice));
            if (wn == "w4p_lnf2_bt") {
                unsigned* dd; CK(hipMalloc(&dd, 8)); CK(hipMemset(dd, 0, 8));
                hipLaunchKernelGGL(ydiff_kernel, dim3(2048), dim3(256), 0, 0, Y, Yref, ny, dd);
                unsigned hd[2]; CK(hipMemcpy(hd, dd, 8, hipMemcpyDeviceToHost)); CK(hipFree(dd));
                float md; memcpy(&md, &hd[0], 4);
                printf("{\"shape\": \"%s\", \"check\": \"Y of the in-loop statistics against Y with the statistics input\", \"max_abs_diff\": %.6f, \"elements_beyond_one_ulp\": %u, \"elements\": %zu}\n",
                       sh.name, md, hd[1], ny);
            }
            v->fn(p, 0);                                             // warm
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) v->fn(p, 0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
            const char* verdict = "n/a";
            if (v->correct) {
                const int md = v->mode * 4 + (v->yonly ? 1 : v->epi);
                if (!have_ref[md]) { ref[md][0] = cs[0]; ref[md][1] = cs[1]; have_ref[md] = true; verdict = "ref"; }
                else verdict = (cs[0] == ref[md][0] && cs[1] == ref[md][1]) ? "bit-identical" : "MISMATCH";
            }
            printf("{\"shape\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"variant\": \"%s\", \"ms\": %.4f, \"tflops\": %.1f, \"checksum\": \"%016llx%016llx\", \"vs_ref\": \"%s\"}\n",
                   sh.name, sh.M, sh.N, sh.K, v->name, ms, 2.0 * sh.M * sh.N * sh.K / ms / 1e9, cs[0], cs[1], verdict);
            if (wn.find("stamps") != std::string::npos) {
                std::vector<unsigned long long> h(1024 * 8);
                CK(hipMemcpy(h.data(), dbg, 1024 * 64, hipMemcpyDeviceToHost));
                double a[4] = {0, 0, 0, 0}; int n = 0;
                for (int b = 0; b < 1024; ++b) {
                    const unsigned long long* d = &h[b * 8];
                    if (!d[4]) continue;
                    if (wn.find("w4") != std::string::npos) { a[0] += (double)d[0]; a[1] += (double)d[1]; a[2] += (double)d[2]; a[3] += (double)d[3]; }
                    else { a[0] += (double)(d[1] - d[0]); a[1] += (double)(d[2] - d[1]); a[2] += (double)(d[3] - d[2]); a[3] += (double)(d[4] - d[3]); }
                    ++n;
                }
                // late: per block (= per tile) setup / K loop / staging / copy-out; w4p: per block totals over its tiles of wait+frag reads / K loop / next-head issue / epilogue
                // effective shader clock of the launch = a block's stamped cycles (its whole life in the persistent kernel) / wall time
                if (n) printf("{\"shape\": \"%s\", \"variant\": \"%s\", \"blocks\": %d, \"cycles_0\": %.0f, \"cycles_1\": %.0f, \"cycles_2\": %.0f, \"cycles_3\": %.0f, \"k_iters\": %d, \"tiles\": %d, \"group_m\": %d, \"order\": %d, \"eff_clock_GHz\": %.3f}\n",
                              sh.name, v->name, n, a[0] / n, a[1] / n, a[2] / n, a[3] / n, sh.K / 64, ((sh.M + 255) / 256) * ((sh.N + 255) / 256), v->group_m, v->order,
                              wn.find("w4") != std::string::npos ? (a[0] + a[1] + a[2] + a[3]) / n / (ms * 1e6) : 0.0);
            }
            fflush(stdout);
        }
        CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(Y)); CK(hipFree(R)); CK(hipFree(Bv)); CK(hipFree(SP));
        if (want_ln) { CK(hipFree(LNst)); CK(hipFree(LNs)); CK(hipFree(LNc)); CK(hipFree(Yref)); }
    }
    return 0;
}
