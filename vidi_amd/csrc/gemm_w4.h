// Persistent 4-wave MFMA GEMM for the large projections of the Vidi hot path (gfx950).
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] )        X:[M,K] activations, W:[N,K] nn.Linear weight, 256x256x64 tiles
//
// One workgroup per CU, 4 waves, ONE wave per SIMD (up to 512 registers): every wave owns a 128(n) x 128(m) sub-tile,
// i.e. 256 accumulator registers = the whole AGPR file (pinned there with "+a" asm operands: left to itself hipcc splits them
// between the two files and copies them around every MFMA).  Compared with 8 waves x (128x64) this reads one third fewer
// fragment bytes from LDS per FLOP and frees the VGPR file for a full K slice of fragments.
//
// K loop (slice kt lives in LDS buffer kt & 1, 64 KB each, XOR-swizzled rows, filled by 16-byte LDS-DMA):
//   registers hold the fragments of a whole 64-wide slice, double-buffered (set 0 = k32 step 0, set 1 = step 1), so the X / W
//   halves of a buffer are dead half an iteration before the buffer's slice is finished and slice kt+2 is DMA'd over them:
//     phase 1 (64 MFMAs on set 0): read set 1 of slice kt; barrier 1 -> X pieces of slice kt+2; barrier 2 -> W pieces
//     phase 2 (64 MFMAs on set 1): remaining pieces; vmcnt(18) + barrier 3 -> read set 0 of slice kt+1 X; vmcnt(15) + barrier 4 -> W
//   The 16 DMA pieces and 32 fragment reads of an iteration are issued one at a time BETWEEN MFMAs (sched_barrier pins the
//   order), waits are counted (vmcnt never drains in the loop), the 4 barriers sit inside the MFMA stream.
//
// Persistent tile loop: block b walks tiles b, b + grid, ... (the same set of concurrently running tiles as a plain launch, so the
// L2 sharing pattern of tile_of_block is unchanged).  To the K loop the next tile's first two slices are simply slices nk and
// nk + 1: the last two iterations of a tile DMA them (under MFMA cover, over the buffers they free) and the last iteration
// already fetches the next tile's first fragments, so a tile starts its MFMAs right after the previous epilogue — no head
// burst, no wait, no barrier.  The epilogue needs no block barrier (each wave stages 16-row strips of its own sub-tile through
// a private 4 KB LDS scratch and stores whole 256-byte row segments), and a block never waits for its stores to be
// acknowledged before the next tile starts (a plain launch pays that at s_endpgm with one block per CU).  K >= 192.
//
// Roofline: MFMA-bound (2.5 PFLOP/s dense bf16/fp16).  Algorithmic FLOPs = 2*M*N*K.
#pragma once
#include "gemm_tile.h"

#ifndef VIDI_W4_RES_DEPTH
#define VIDI_W4_RES_DEPTH 3
#endif
// Epilogue form 2 (row-major outputs: MODE_PLAIN without the head-major layout, MODE_GEGLU): the tile's stores, residual loads and
// partial-sum stores go through per-tile buffer descriptors.  A lane's byte offset inside the tile is tile-invariant up to the row
// step, rows past M fall outside the descriptor's extent and columns past N get an out-of-range lane offset, so the hardware's range
// check replaces the per-store 64-bit address arithmetic (v_mad_i64 / v_lshl_add_u64), the row / column predicates and their exec-mask
// branches (about 650 of the 3 265 instructions of a bias + residual + statistics tile, 350 of a plain one); the statistics' eight row
// reductions of a strip run as one interleaved block (row16_sum8: no hazard s_nops).  The row offset is added on the VALU into the
// VGPR offset: the SGPR offset of a buffer instruction is NOT part of the range check.  Bit-identical results.
// Same-box A/B against form 1, alternating builds, identical checksums (profiles/r4_gemm_lab_epi2.jsonl): bias + residual + statistics
// 16.7 k -> 13.1 k epilogue cycles per tile (out_proj +2.8 %, fc2 +0.9 %), bias + GELU 16.4 k -> 13.5 k (SigLIP fc1 +2.3 %, Whisper fc1
// +1.2…1.5 %), bias + residual +0.5…1.3 %; the bias-free plain epilogue and GeGLU at K >= 3 584 measured -0.3…-0.6 % (their epilogue is 4 %
// of the tile; the descriptor set-up does not pay) and keep form 1.
#ifndef VIDI_W4_EPI2
#define VIDI_W4_EPI2 1
#endif

struct W4Geom {
    static constexpr int BN = 256, BM = 256, BK = 64, NT = 256, TN = 8, TM = 8, ROWB = 128;
    static constexpr int STAGE_BYTES = (BN + BM) * ROWB, RING = 2 * STAGE_BYTES;
    static constexpr int SCR_ROW = 128 * 2 + 16, SCR_BYTES = 16 * SCR_ROW;       // per-wave epilogue scratch: 16 rows x (128 cols + pad)
    // epilogue constants of a tile (bias, or the folded-LayerNorm column sums / shifts and row statistics): DMA'd into LDS during the
    // tile's first K iteration, two buffers by tile parity (a wave may start the next tile while another still reads in its epilogue)
    static constexpr int CST_OFF = RING + 4 * SCR_BYTES, CST_BYTES = 4096;
    static constexpr int LDS_BYTES = CST_OFF + 2 * CST_BYTES;
};

// compile-time epilogue shape of MODE_PLAIN (runtime flags made every strip a maze of scalar branches and put a vmcnt(0) —
// i.e. a wait for the in-flight DMA and for all earlier stores — on the path without a residual): bias add, activation
// (ACT_NONE / ACT_GELU_TANH / ACT_GELU_ERF), residual (0 none, 1 row m, 2 row m % rmod)
// LNF: LayerNorm folded into the projection (GemmParams::ln_stats / ln_s / ln_c): y = rstd_m * (acc - mean_m * s_n) + c_n replaces
// the bias add (c carries the bias).  1 (true): (mean, rstd) of the rows come from GemmParams::ln_stats.  2: they are computed HERE, in the
// K loop, from the X fragments the wave multiplies anyway (the contraction length is the LayerNorm width, so a tile sees whole rows): both
// n-waves of a row block read the same X fragments, each accumulates (sum, sum of squares) of its own 128 rows — one v_dot2c per MFMA slot,
// [LAB ONLY — measured in round 5 (profiles/r5_gemm_lab_lnstats.jsonl): the K loop runs at 941 instead of 1 584 TFLOP/s with the dot
// products in it, the whole fc1 kernel at 798 instead of 1 103 — no product path instantiates lnf == 2; tools/lab/gemm_lab lnstats does]
// on registers that are live anyway — and the results land in the lanes the epilogue reads them from: no exchange, no statistics input,
// no statistics epilogue in the producer.  GemmParams::ln_eps is the LayerNorm's epsilon.
// STATS: the stored rows' per-strip partial sums go to GemmParams::stat_part (bias + residual producers of a LayerNorm input)
// HEADS: head-major output Y[which][frame][head][token][d] (GemmParams::hm_*): the q/k/v projection of the encoder towers
template <bool BIAS, int ACT, int RES, int LNF = 0, bool STATS = false, bool HEADS = false>
struct Epi { static constexpr bool bias = BIAS; static constexpr int act = ACT, res = RES, lnf = LNF; static constexpr bool stats = STATS, heads = HEADS; };

// PATCH: the X operand is gathered by the loader instead of read from a row-major matrix (GemmParams::pe_*):
//   1  SigLIP patch embedding from NCHW pixels (no im2col buffer)
//   2  k x k convolution window (stride 1, valid) over a token-major [frames, side*side, C] feature map: Vidi-7B's learned Conv2DPool
template <typename T, int MODE, bool REPKV, bool PERSIST, typename EPI = Epi<false, ACT_NONE, 0>, typename LAB = LabNone, int PATCH = 0>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmParams p, int batch) {
    using G = W4Geom;
    constexpr int BN = G::BN, BM = G::BM, BK = G::BK, NT = G::NT, TN = G::TN, TM = G::TM, ROWB = G::ROWB, STAGE_BYTES = G::STAGE_BYTES;
    auto swz = [](int row) { return (row >> 1) & 7; };
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int l15 = lane & 15, hi = lane >> 4;
    const int sw = swz(l15);
    const int w_row_off = (wn * 128 + l15) * ROWB;
    const int x_row_off = BN * ROWB + (wm * 128 + l15) * ROWB;
    const int nk = p.K / BK;
    const int tiles_1 = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
    const int tiles = tiles_1 * batch;

    unsigned long long t_acc[4] = {0, 0, 0, 0}, t_mark = 0;
    auto stamp = [&](int slot) {
        if constexpr (LAB::stamps) { const unsigned long long t = __builtin_readcyclecounter(); t_acc[slot] += t - t_mark; t_mark = t; }
    };
    if constexpr (LAB::stamps) t_mark = __builtin_readcyclecounter();

    // ---- per-tile state ------------------------------------------------------------------------
    // DMA sources are buffer descriptors (SGPRs, rebuilt per tile by scalar code) + ONE tile-invariant per-lane offset per
    // operand: piece j of a tile covers rows j*32 + (tid >> 3), and the swizzled source chunk cg = (tid & 7) ^ swz(row) does not
    // depend on j, so  address = tile base + [(tid >> 3) * ld + cg * 8] (VGPR) + [k0 + j * 32 * ld] (SGPR).  Rows past the end
    // of the matrix are out of the descriptor's range and read as zeros (they are masked in the epilogue anyway).
    const int r0 = tid >> 3, cg0 = (tid & 7) ^ swz(r0);
    const unsigned offW = (unsigned)(r0 * p.ldw + cg0 * 8) * 2u, offX = (unsigned)(r0 * p.ldx + cg0 * 8) * 2u;
    __amdgpu_buffer_rsrc_t srdW, srdX, srdWn, srdXn;      // current tile / next tile
    int m0 = 0, n0 = 0, bz = 0, nm0 = 0, nn0 = 0, nbz = 0;
    int pb = 0;                                             // LDS buffer of the current tile's slice 0 (slice kt lives in buffer (pb + kt) & 1)
    auto make_srd = [&](const u16* base, int rows_left, int ld) {
        // valid extent = (rows_left - 1) * ld + K elements (rows may overlap: ld < K is the conv-as-GEMM view)
        const unsigned long long bytes = ((unsigned long long)(rows_left - 1) * (unsigned)ld + (unsigned)p.K) * 2ull;
        const unsigned nrec = bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes;
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nrec, 0x00020000);
    };
    // PATCH: per-lane byte offsets of the lane's 8 X rows (piece q covers row q*32 + r0) of the current / the next tile inside the pixel
    // tensor — (frame, patch row, patch column) -> first pixel of the patch in channel 0, + the lane's half of the 16-pixel run — and the
    // per-slice part: the lane's chunk cg0 of slice ks belongs to patch line rr = 4 ks + (cg0 >> 1) = (channel, dy)
    unsigned pxo[PATCH != 0 ? 8 : 1], pxn[PATCH != 0 ? 8 : 1];
    auto udiv = [](unsigned x, unsigned d, unsigned magic) { return d == 1u ? x : __umulhi(x, magic); };      // magic = floor(2^32 / d) + 1 does not exist for d = 1
    auto patch_rows = [&](int tm0, unsigned* dst) {
        if constexpr (PATCH == 1) {
            const unsigned n = (unsigned)(p.pe_side * p.pe_side);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned m = (unsigned)min(tm0 + q * 32 + r0, p.M - 1);        // rows past M repeat the last patch (masked in the epilogue)
                const unsigned f = udiv(m, n, p.pe_nmagic), rem = m - f * n;
                const unsigned py = udiv(rem, (unsigned)p.pe_side, p.pe_smagic), px = rem - py * (unsigned)p.pe_side;
                dst[q] = (((f * 3u * (unsigned)p.pe_S + py * (unsigned)p.pe_P) * (unsigned)p.pe_S + px * (unsigned)p.pe_P) << 1) + (unsigned)(cg0 & 1) * 16u;
            }
        } else if constexpr (PATCH == 2) {
            // window mode: pe_S = channels C, pe_P = window k, pe_side = input side; output row m = (frame, oy, ox) of an oc x oc grid,
            // oc = side - k + 1 (pe_nmagic / pe_smagic divide by oc*oc / oc): first input token of the window, + the lane's chunk
            const unsigned oc = (unsigned)(p.pe_side - p.pe_P + 1), n = oc * oc;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned m = (unsigned)min(tm0 + q * 32 + r0, p.M - 1);
                const unsigned f = udiv(m, n, p.pe_nmagic), rem = m - f * n;
                const unsigned oy = udiv(rem, oc, p.pe_smagic), ox = rem - oy * oc;
                dst[q] = ((((f * (unsigned)p.pe_side + oy) * (unsigned)p.pe_side + ox) * (unsigned)p.pe_S) << 1) + (unsigned)cg0 * 16u;
            }
        }
    };
    // the K-slice part of the gather address.  Patch mode: the lane's chunk cg0 of slice ks belongs to patch line rr = 4 ks + (cg0 >> 1) =
    // (channel, dy) -> a per-lane value.  Window mode: slice ks = 64 channels c0 .. of window position (dy, dx) = (ks * 64) / C -> the same
    // for every lane (pe_cmagic divides by C / 64 slices per window position; pe_kmagic by k)
    auto patch_koff = [&](int ks) -> unsigned {
        if constexpr (PATCH == 2) {
            const unsigned spc = (unsigned)p.pe_S >> 6;                                   // slices per window position
            const unsigned dd = udiv((unsigned)ks, spc, p.pe_cmagic), c0 = ((unsigned)ks - dd * spc) << 6;
            const unsigned dy = udiv(dd, (unsigned)p.pe_P, p.pe_kmagic), dx = dd - dy * (unsigned)p.pe_P;
            return (((dy * (unsigned)p.pe_side + dx) * (unsigned)p.pe_S) + c0) << 1;
        } else {
            const int rr = 4 * ks + (cg0 >> 1);
            const int c = (rr >= p.pe_P ? 1 : 0) + (rr >= 2 * p.pe_P ? 1 : 0);
            return (unsigned)(rr * p.pe_S + c * (p.pe_S - p.pe_P) * p.pe_S) << 1;        // channel c, line dy = rr - c P: (c S + dy) S pixels
        }
    };
    auto locate = [&](int vb, int& tm0, int& tn0, int& tbz) {
        const int b1 = vb % tiles_1;
        tbz = vb / tiles_1;
        int tile_m, tile_n;
        tile_of_block(p, BN, BM, b1, tiles_1, tile_m, tile_n);
        tn0 = tile_n * BN; tm0 = tile_m * BM;
    };
    // DMA piece q of K slice `ks` (of the current tile, or of the next one when NEXT) into `buf`: q < 8 -> X piece q, else W piece
    // q - 8 (1 KB per wave each)
    auto piece = [&](char* buf, int ks, int q, auto next_t) {
        if constexpr (LAB::no_dma) return;
        constexpr bool NEXT = decltype(next_t)::value;
        const int k0 = ks * BK;
        if constexpr (PATCH != 0) {
            if (q < 8) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdX, (__attribute__((address_space(3))) void*)(buf + BN * ROWB + (q * NT + wave * 64) * 16), 16,
                                                         (int)((NEXT ? pxn[q] : pxo[q]) + patch_koff(ks)), 0, 0, 0);
                return;
            }
        }
        if (q < 8) {
            int kx = k0;
            if constexpr (REPKV) kx = (k0 / (p.rep_g * p.rep_hd)) * p.rep_hd + (k0 % p.rep_hd);      // rep_hd % 64 == 0: the whole 64-wide slice maps together
            __builtin_amdgcn_raw_ptr_buffer_load_lds(NEXT ? srdXn : srdX, (__attribute__((address_space(3))) void*)(buf + BN * ROWB + (q * NT + wave * 64) * 16), 16,
                                                     offX, (unsigned)(kx + q * 32 * p.ldx) * 2u, 0, 0);
        } else {
            const int j = q - 8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(NEXT ? srdWn : srdW, (__attribute__((address_space(3))) void*)(buf + (j * NT + wave * 64) * 16), 16, offW,
                                                     (unsigned)(k0 + j * 32 * p.ldw) * 2u, 0, 0);
        }
    };
    auto bar = [&]() { __builtin_amdgcn_s_barrier(); };
    // accumulator element -> VGPR, exactly where it is written: the accumulators are asm operands pinned to the AGPR file, and
    // for plain uses hipcc copies ALL 256 of them to VGPRs right after the K loop (256 live registers, spills) — seen in the .s
    auto aread = [](float x) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(x)); return v; };

    f32x4 acc[TN][TM];                                   // written (not accumulated) by the first k32 step of every tile

    u32x4 fW[2][TN], fX[2][TM];
    auto rdW = [&](const char* buf, int a, int s) { return *(const u32x4*)(buf + w_row_off + a * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
    auto rdX = [&](const char* buf, int b, int s) { return *(const u32x4*)(buf + x_row_off + b * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
#define VIDI_PIN __builtin_amdgcn_sched_barrier(0)

    // ---- epilogue constants -> LDS (one 1 KB DMA per wave per tile, issued at the top of the tile's first iteration) ---------------
    // A global load at the start of the epilogue would retire only after the ~26 DMA pieces of the NEXT tile that are in flight by
    // then (vector memory operations retire in order): microseconds of exposed latency per tile.  Landing the constants in LDS
    // under the K loop removes every load from the bias / folded-LayerNorm epilogues.
    //   bias epilogues : wave 0 fetches bias[n0 .. n0+255] (bf16, 512 B; the upper half of the piece is never read)
    //   folded LN      : wave 0 colsum[n0 ..], wave 1 shift[n0 ..] (fp32, 1 KB each), waves 2, 3 (mean, rstd) of rows m0 .. m0+255
    constexpr bool cst_lnf = (MODE != MODE_GEGLU) && EPI::lnf != 0;
    constexpr bool ln_inloop = (MODE != MODE_GEGLU) && EPI::lnf == 2 && PATCH == 0 && !REPKV;      // row statistics from the X fragments (see Epi)
    static_assert(EPI::lnf != 2 || ln_inloop, "in-loop LayerNorm statistics need a plain row-major X");
    constexpr bool cst_bias = (MODE != MODE_GEGLU) && EPI::bias && EPI::lnf == 0;
    __amdgpu_buffer_rsrc_t srdC0, srdC1;
    int tpar = 0;                                           // parity of the current tile (constant buffer)
    auto rsrc_of = [](const void* base, unsigned long long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
    };
    if constexpr (cst_lnf) {
        srdC0 = rsrc_of(wave == 0 ? p.ln_s : (wave == 1 || ln_inloop ? p.ln_c : p.ln_stats), wave < 2 || ln_inloop ? (unsigned long long)p.N * 4 : (unsigned long long)p.M * 8);
    } else if constexpr (cst_bias) {
        srdC0 = rsrc_of(p.bias, (unsigned long long)p.N * 2);
    }
    (void)srdC1;
    auto issue_cst = [&]() {
        if constexpr (LAB::no_dma) return;
        char* dst = smem + G::CST_OFF + tpar * G::CST_BYTES;
        if constexpr (cst_lnf) {
            // out-of-range columns / rows read as zeros (buffer range check); they are masked in the epilogue
            const unsigned voff = wave < 2 ? (unsigned)(n0 + 4 * lane) * 4u : (unsigned)(m0 + (wave - 2) * 128 + 2 * lane) * 8u;
            if (!ln_inloop || wave < 2)                              // (in-loop statistics: no (mean, rstd) pieces; only waves 0, 1 count one more DMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdC0, (__attribute__((address_space(3))) void*)(dst + wave * 1024), 16, voff, 0, 0, 0);
        } else if constexpr (cst_bias) {
            if (wave == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdC0, (__attribute__((address_space(3))) void*)dst, 16, (unsigned)(n0 + 8 * lane) * 2u, 0, 0, 0);
        }
    };

    // in-loop LayerNorm statistics: (sum, sum of squares) of this lane's k-subset of rows wm*128 + b*16 + l15, b = 0..7
    float rs1[ln_inloop ? TM : 1], rs2[ln_inloop ? TM : 1];
    auto row_dot = [&](unsigned w, bool square, float& s1, float& s2) {
        // ONE v_dot2c per call (one dword of a fragment = 2 elements, picked at the call site; fp32 accumulate): against ones (sum) or
        // against itself (sum of squares).  64 calls per phase — one per MFMA — cover the phase's 8 fragments x 4 dwords x {sum, squares}.
        // asm volatile like the MFMAs: as ordinary instructions the last iteration's 256 were sunk below the K loop (seen in the ISA)
        if constexpr (T::id == VIDI_DT_BF16) {
            if (square) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(s2) : "v"(w));
            else asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(s1) : "v"(w));
        } else {
            if (square) asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(s2) : "v"(w));
            else asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(s1) : "v"(w));
        }
    };
    // one K iteration: 128 MFMAs on slice kt; slice kt+1's step-0 fragments are fetched and slice kt+2 is DMA'd over slice kt's
    // buffer.  FIRST: first slice of a tile, the step-0 MFMAs take C = 0 (no accumulator clearing).  NEXT: the last two iterations
    // of a tile — slices kt+1 / kt+2 are the NEXT tile's (slice kt+2-nk of its operands), issued only when `more` (a next tile
    // exists); without one the iteration still runs the same instruction stream (fragment reads of stale LDS, never used):
    // any control-flow merge over the 64 asm-pinned accumulators makes hipcc spill hundreds of registers
    auto body = [&](int kt, auto first_t, auto next_t, bool more) {
        constexpr bool FIRST = decltype(first_t)::value, NEXT = decltype(next_t)::value;
        constexpr bool HAS2 = true, HAS1 = true;
        char* bufc = smem + ((pb + kt) & 1) * STAGE_BYTES;
        const char* bufn = smem + ((pb + kt + 1) & 1) * STAGE_BYTES;
        const int ks2 = NEXT ? kt + 2 - nk : kt + 2;         // slice index (in its own tile) of the slice DMA'd this iteration
        auto dma = [&](int q) {
            if constexpr (NEXT) { if (more) piece(bufc, ks2, q, next_t); }
            else piece(bufc, ks2, q, next_t);
        };
        VIDI_PIN;
        if constexpr (FIRST) issue_cst();
        if constexpr (FIRST && ln_inloop) {
#pragma unroll
            for (int b = 0; b < TM; ++b) { rs1[b] = 0.f; rs2[b] = 0.f; }
        }
        VIDI_PIN;
        // ---------------- phase 1: step-0 MFMAs ----------------
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int a = i >> 3, b = i & 7;
            if constexpr (FIRST) T::mfma16_agpr_first(acc[a][b], fW[0][a], fX[0][b]);
            else T::mfma16_agpr(acc[a][b], fW[0][a], fX[0][b]);
            if (i < 16 && (i & 1) == 0) fX[1][i >> 1] = rdX(bufc, i >> 1, 1);                 // 8 X-fragment reads
            // in-loop statistics: the step-0 fragments (live through this phase), one v_dot2c in every MFMA's shadow
            if constexpr (ln_inloop) row_dot(fX[0][i >> 3][(i & 7) >> 1], (i & 1) != 0, rs1[i >> 3], rs2[i >> 3]);
            if constexpr (HAS2) {
                if (i == 19) wait_lgkm0();
                if (i == 20) bar();                                                            // barrier 1: X part of bufc is dead
                if (i >= 22 && i <= 38 && ((i - 22) & 3) == 0) dma((i - 22) >> 2);   // X pieces 0..4
            }
            if (i >= 24 && i <= 40 && ((i - 24) & 3) == 0) fW[1][(i - 24) >> 2] = rdW(bufc, (i - 24) >> 2, 1);   // W reads 0..4
            if (i == 42 || i == 44 || i == 46) fW[1][5 + ((i - 42) >> 1)] = rdW(bufc, 5 + ((i - 42) >> 1), 1); // W reads 5..7
            if constexpr (HAS2) {
                if (i == 51) wait_lgkm0();
                if (i == 52) bar();                                                            // barrier 2: W part of bufc is dead
                if (i == 53) dma(5);
                if (i == 56) dma(6);
                if (i == 58) dma(7);
                if (i == 61) dma(8);
            }
            VIDI_PIN;
        }
        // ---------------- phase 2: step-1 MFMAs ----------------
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const int a = i >> 3, b = i & 7;
            T::mfma16_agpr(acc[a][b], fW[1][a], fX[1][b]);
            if constexpr (ln_inloop) row_dot(fX[1][i >> 3][(i & 7) >> 1], (i & 1) != 0, rs1[i >> 3], rs2[i >> 3]);      // the step-1 fragments
            if constexpr (HAS2) {
                if (i == 1) dma(9);
                if (i == 21) dma(10);
                if (i == 23) dma(11);
                if (i == 25) dma(12);
                if (i == 32) dma(13);
                if (i == 36) dma(14);
                if (i == 60) dma(15);
            }
            if constexpr (HAS1) {
                // X part of slice kt+1: this iteration's 10 pieces + last iteration's 8 W pieces may stay in flight
                if (i == 3) { if constexpr (LAB::no_dma) {} else if (!NEXT || more) wait_vm<18>(); else wait_vm<8>(); }
                if (i == 4) bar();                                                             // barrier 3: everybody's X pieces landed
                if (i >= 5 && i <= 19 && ((i - 5) & 1) == 0) fX[0][(i - 5) >> 1] = rdX(bufn, (i - 5) >> 1, 0);
                if (i == 40) { if constexpr (LAB::no_dma) {} else if (!NEXT || more) wait_vm<15>(); else wait_vm<0>(); }
                if (i == 41) bar();                                                            // barrier 4: W pieces landed
                if (i >= 42 && i <= 56 && ((i - 42) & 1) == 0) fW[0][(i - 42) >> 1] = rdW(bufn, (i - 42) >> 1, 0);
            }
            VIDI_PIN;
        }
    };

    // ---- epilogue of the tile at (em0, en0, ebz): strips of 16 rows through the wave's private scratch ----------------
    char* scr = smem + G::RING + wave * G::SCR_BYTES;
    constexpr bool act_tanh = (EPI::act == ACT_GELU_TANH), act_erf = (EPI::act == ACT_GELU_ERF);
    const bool glu_silu = (p.act == ACT_SILU);
    constexpr bool GLU = (MODE == MODE_GEGLU);
    constexpr int SROW = GLU ? (64 * 2 + 16) : G::SCR_ROW;          // scratch row bytes
    constexpr int NRD = GLU ? 2 : 4;                                 // 16-byte reads per lane per strip
    constexpr int RPI = GLU ? 8 : 4;                                 // rows per read instruction
    constexpr int CPRW = GLU ? 8 : 16;                               // 16-byte chunks per strip row
    auto epilogue = [&](int em0, int en0, int ebz) {
        u16* Yb = p.Y + (long long)ebz * p.bsY;
        constexpr bool has_res = (MODE == MODE_PLAIN) && (EPI::res != 0);
        const u16* Rb = has_res ? p.R + (long long)ebz * p.bsR : nullptr;
        constexpr bool lnf = !GLU && EPI::lnf;
        constexpr bool has_bias = !GLU && EPI::bias && !lnf;          // (QKV_VT always carries a bias in the callers: Epi<true, ...>)
        constexpr bool wrap = (EPI::res == 2);
        const int Nout = GLU ? (p.N >> 1) : p.N;
        const int no0 = GLU ? (en0 >> 1) + wn * 64 : en0 + wn * 128;           // first output column of this wave's strips
        const char* cst = smem + G::CST_OFF + tpar * G::CST_BYTES;     // this tile's constants (landed during its K loop)
        u32x2 bq[TN];
        if constexpr (has_bias) {
#pragma unroll
            for (int a = 0; a < TN; ++a) bq[a] = *(const u32x2*)(cst + (wn * 128 + a * 16 + 4 * hi) * 2);
        }
        // folded LayerNorm: this lane's column sums / shifts (4 consecutive columns per tile) and the (mean, rstd) of its 8 rows
        f32x4 lnS[lnf ? TN : 1], lnC[lnf ? TN : 1];
        float lnMu[lnf ? TM : 1], lnRs[lnf ? TM : 1];
        if constexpr (lnf) {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                lnS[a] = *(const f32x4*)(cst + (wn * 128 + a * 16 + 4 * hi) * 4);
                lnC[a] = *(const f32x4*)(cst + 1024 + (wn * 128 + a * 16 + 4 * hi) * 4);
            }
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                if constexpr (ln_inloop) {
                    // the four lanes l15 + 16 hi hold the row's four k-subsets: two exchanges through the LDS crossbar (no memory), then the
                    // one-pass (mean, rstd) — the same arithmetic as vidi_ln_finalize, on the STORED values of the row
                    float s1 = rs1[b], s2 = rs2[b];
                    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                    const float invK = 1.0f / (float)p.K;
                    const float mean = s1 * invK;
                    lnMu[b] = mean;
                    lnRs[b] = rsqrtf(fmaxf(s2 * invK - mean * mean, 0.f) + p.ln_eps);
                } else {
                    const f32x2_t ms = *(const f32x2_t*)(cst + 2048 + (wm * 128 + b * 16 + l15) * 8);
                    lnMu[b] = ms[0];
                    lnRs[b] = ms[1];
                }
            }
        }
        const int rr = lane / CPRW, cc = lane % CPRW;                // this lane's (row in read group, chunk) of the read-back
        const int n = no0 + cc * 8;                                  // this lane's output column in the read-back
        // head-major output: this lane's column is (which, head, d) for the whole tile; rows split into (frame, token) per store
        size_t hm_col = 0;
        if constexpr (EPI::heads) {
            const int nc = min(n, p.N - 8), hdim = p.hm_heads * p.hm_hd;
            const int which = nc / hdim, nh = nc - which * hdim, hh = nh / p.hm_hd;
            hm_col = ((size_t)which * (p.M / p.hm_seq) * p.hm_heads + hh) * p.hm_seq * p.hm_hd + (nh - hh * p.hm_hd);
        }
        // epilogue form 2: per-tile descriptors + tile-invariant lane offsets (see VIDI_W4_EPI2 above)
        constexpr bool bufio = (VIDI_W4_EPI2 != 0) && !GLU && MODE == MODE_PLAIN && !EPI::heads &&
                               (VIDI_W4_EPI2 == 2 || EPI::bias || EPI::res != 0 || EPI::act != ACT_NONE || EPI::lnf);     // (2: every row-major epilogue, lab)
        constexpr bool stats_on = (MODE == MODE_PLAIN) && EPI::stats && (EPI::res != 0);
        __amdgpu_buffer_rsrc_t srdY, srdR, srdS;
        unsigned vY = 0, vR = 0, vS = 0;
        (void)srdY; (void)srdR; (void)srdS; (void)vY; (void)vR; (void)vS;
        if constexpr (bufio) {
            const unsigned rows_here = (unsigned)min(p.M - em0, 256);
            const bool col_ok = n < Nout;
            srdY = rsrc_of(Yb + (size_t)em0 * p.ldy, (unsigned long long)rows_here * (unsigned)p.ldy * 2ull);
            vY = (col_ok && !LAB::no_store) ? (unsigned)((wm * 128 + rr) * p.ldy + n) * 2u : 0x80000000u;
            if constexpr (has_res && !wrap) {
                // (a lane past the last column reads zeros: with a zero bias and zero accumulators there it adds nothing to the row sums)
                srdR = rsrc_of(Rb + (size_t)em0 * p.ldr, (unsigned long long)rows_here * (unsigned)p.ldr * 2ull);
                vR = col_ok ? (unsigned)((wm * 128 + rr) * p.ldr + n) * 2u : 0x80000000u;
            }
            if constexpr (stats_on) {
                const unsigned strips = (unsigned)((Nout + 127) >> 7);
                srdS = rsrc_of(p.stat_part + (size_t)em0 * strips * 2, (unsigned long long)rows_here * strips * 8ull);
                vS = (cc == 0 && no0 < Nout) ? ((unsigned)(wm * 128 + rr) * strips + (unsigned)(no0 >> 7)) * 8u : 0x80000000u;
            }
        }
        // ---- registers -> scratch (lane: row l15, 4 consecutive columns per 16-column tile) ----
        auto stage = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
            const int mrow0 = em0 + wm * 128 + b * 16;               // first global row of the strip
            if constexpr (GLU) {
                // the strip's 8 gate pairs (gate tiles a = 0, 1, 4, 5; up tiles a + 2 are consumed with them): rounded to T, activated in
                // lock-step on packed fp32 math, rounded again and multiplied by the rounded up values
                f32x2_t g[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int a = (i >> 1) * 4 + (i & 1);
#pragma unroll
                    for (int e = 0; e < 4; e += 2) g[i * 2 + (e >> 1)] = f32x2_t{rnd<T>(aread(acc[a][b][e])), rnd<T>(aread(acc[a][b][e + 1]))};
                }
                if (glu_silu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = silu_2(g[i]);
                } else {
                    gelu_tanh_pairs<8>(g);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int a = (i >> 1) * 4 + (i & 1);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2_t act = g[i * 2 + (e >> 1)];
                        v[e] = rnd<T>(act[0]) * rnd<T>(aread(acc[a + 2][b][e]));
                        v[e + 1] = rnd<T>(act[1]) * rnd<T>(aread(acc[a + 2][b][e + 1]));
                    }
                    const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                    const int col = (a >> 2) * 32 + (a & 1) * 16 + 4 * hi;
                    *(u32x2*)(scr + l15 * SROW + col * 2) = o;
                }
            } else {
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const int nn = en0 + wn * 128 + a * 16 + 4 * hi;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = aread(acc[a][b][e]);
                    if constexpr (has_bias) {
                        const u32x2 bv = bq[a];
                        v[0] += T::to_f32((u16)(bv[0] & 0xffff)); v[1] += T::to_f32((u16)(bv[0] >> 16));
                        v[2] += T::to_f32((u16)(bv[1] & 0xffff)); v[3] += T::to_f32((u16)(bv[1] >> 16));
                    }
                    if constexpr (lnf) {
                        // y = rstd * (acc - mean * s) + c  ==  rstd * acc + (c - rstd * mean * s): two packed FMAs per register pair
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        const float t = -lnRs[b] * lnMu[b];
                        const f32x2 tt = {t, t}, rr2 = {lnRs[b], lnRs[b]};
                        const f32x2 u0 = __builtin_elementwise_fma(tt, f32x2{lnS[a][0], lnS[a][1]}, f32x2{lnC[a][0], lnC[a][1]});
                        const f32x2 u1 = __builtin_elementwise_fma(tt, f32x2{lnS[a][2], lnS[a][3]}, f32x2{lnC[a][2], lnC[a][3]});
                        const f32x2 y0 = __builtin_elementwise_fma(rr2, f32x2{v[0], v[1]}, u0);
                        const f32x2 y1 = __builtin_elementwise_fma(rr2, f32x2{v[2], v[3]}, u1);
                        v[0] = y0[0]; v[1] = y0[1]; v[2] = y1[0]; v[3] = y1[1];
                    }
                    const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                    const int m = mrow0 + l15;
                    if constexpr (MODE == MODE_QKV_VT) {
                        if (nn >= p.vstart && nn < p.N && m < p.M) {    // V columns: transposed, perm16 key order (vidi_attn_self layout)
                            const int c = nn - p.vstart, h = c / p.hd, d = c % p.hd;
                            const int bi = m / p.seq, tok = m % p.seq;
                            const int pos = (tok & ~15) | perm16(tok & 15);
                            u16* dst = p.Vt + (((size_t)bi * p.nheads + h) * p.hd + d) * p.seqpad + pos;
                            dst[0] = (u16)(o[0] & 0xffff); dst[(size_t)p.seqpad] = (u16)(o[0] >> 16);
                            dst[2 * (size_t)p.seqpad] = (u16)(o[1] & 0xffff); dst[3 * (size_t)p.seqpad] = (u16)(o[1] >> 16);
                        }
                    } else if constexpr (MODE == MODE_KV_CACHE) {
                        if (nn >= p.kvd && nn < p.N && m < p.M) {       // V: also the transposed, perm16 32-key sub-tile image
                            const int tok = p.tok0 + m, tile = tok >> 5, tk = tok & 31;
                            const int c = nn - p.kvd, kvh = c / p.hd, d = c % p.hd;
                            const int pos = (tk & ~15) | perm16(tk & 15);
                            u16* dst = p.Vtc + (((size_t)kvh * p.ntile64 * 2 + tile) * p.hd + d) * 32 + pos;
                            dst[0] = (u16)(o[0] & 0xffff); dst[32] = (u16)(o[0] >> 16);
                            dst[64] = (u16)(o[1] & 0xffff); dst[96] = (u16)(o[1] >> 16);
                        }
                    }
                    *(u32x2*)(scr + l15 * SROW + (a * 16 + 4 * hi) * 2) = o;
                }
            }
        };
        // ---- scratch -> registers: whole 16-byte chunks; a row segment of the strip is contiguous.  LDS operations of one wave
        //      execute in order: these reads see the strip's writes, and the NEXT strip's writes (issued after them) come later ----
        // residual chunks are requested two strips ahead (3-deep register ring, strips 0 and 1 before the first store): vector
        // memory operations retire in issue order, so a load issued right before its use would first wait for the stores of the
        // strips before it to be acknowledged
        constexpr int RD = VIDI_W4_RES_DEPTH;                              // residual ring depth: strips b .. b + RD - 2 are in flight while strip b is stored
        u32x4 val[NRD], res[has_res ? RD : 1][NRD];
        auto load_res = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
            if constexpr (has_res && b < TM) {
                if constexpr (bufio && !wrap) {
#pragma unroll
                    for (int j = 0; j < NRD; ++j)
                        res[b % RD][j] = __builtin_amdgcn_raw_buffer_load_b128(srdR, vR + (unsigned)((b * 16 + j * RPI) * p.ldr) * 2u, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < NRD; ++j) {
                        const int mc = min(em0 + wm * 128 + b * 16 + j * RPI + rr, p.M - 1), mr = wrap ? mc % p.rmod : mc;
                        res[b % RD][j] = *(const u32x4*)(Rb + (size_t)mr * p.ldr + min(n, Nout - 8));
                    }
                }
            }
        };
        load_res(std::integral_constant<int, 0>{});
        load_res(std::integral_constant<int, 1>{});
        if constexpr (RD > 3) load_res(std::integral_constant<int, 2>{});
        if constexpr (RD > 4) load_res(std::integral_constant<int, 3>{});
        if constexpr (RD > 5) load_res(std::integral_constant<int, 4>{});
        auto fetch = [&](int b) {
#pragma unroll
            for (int j = 0; j < NRD; ++j) val[j] = *(const u32x4*)(scr + (j * RPI + rr) * SROW + cc * 16);
        };
        constexpr bool emit_stats = (MODE == MODE_PLAIN) && EPI::stats && (EPI::res != 0);
        auto store = [&](int b) {
            const int mrow0 = em0 + wm * 128 + b * 16;
            if constexpr (MODE == MODE_PLAIN && (act_tanh || act_erf)) {
                // the activation of the whole strip first (on the T-rounded staged values, as below), 8 register pairs in lock-step
#pragma unroll
                for (int j0 = 0; j0 < NRD; j0 += 2) {
                    f32x2_t xp[8];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        float x[8];
                        unpack8<T>(val[j0 + jj], x);
#pragma unroll
                        for (int e = 0; e < 4; ++e) xp[jj * 4 + e] = f32x2_t{x[2 * e], x[2 * e + 1]};
                    }
                    if constexpr (act_tanh) gelu_tanh_pairs<8>(xp);
                    else gelu_erf_pairs<8>(xp);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { x[2 * e] = xp[jj * 4 + e][0]; x[2 * e + 1] = xp[jj * 4 + e][1]; }
                        val[j0 + jj] = pack8<T>(x);
                    }
                }
            }
            if constexpr (bufio) {
                // form 2: no per-store predicate (rows past M / columns past N are out of the descriptors' range)
                if constexpr (emit_stats) {
                    float sv[2 * NRD];
                    u32x4 outv[NRD];
#pragma unroll
                    for (int j = 0; j < NRD; ++j) {
                        float x[8], r[8];
                        unpack8<T>(val[j], x);
                        unpack8<T>(res[b % RD][j], r);
                        f32x2_t sa = {0.f, 0.f}, sq = {0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2_t y = f32x2_t{x[e], x[e + 1]} + f32x2_t{r[e], r[e + 1]};
                            x[e] = y[0]; x[e + 1] = y[1];
                            sa += y;
                            sq = __builtin_elementwise_fma(y, y, sq);
                        }
                        sv[2 * j] = sa[0] + sa[1]; sv[2 * j + 1] = sq[0] + sq[1];
                        outv[j] = pack8<T>(x);
                    }
                    static_assert(NRD == 4, "row16_sum8 takes the strip's four rows");
                    row16_sum8(sv);
#pragma unroll
                    for (int j = 0; j < NRD; ++j) {
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                        const unsigned rowoff = (unsigned)(b * 16 + j * RPI);
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(sv[2 * j]), __float_as_uint(sv[2 * j + 1])}, srdS,
                                                              vS + rowoff * (unsigned)((Nout + 127) >> 7) * 8u, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(outv[j], srdY, vY + rowoff * (unsigned)p.ldy * 2u, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NRD; ++j) {
                        u32x4 v = val[j];
                        if constexpr (has_res) {
                            float x[8], r[8];
                            unpack8<T>(v, x);
                            unpack8<T>(res[b % RD][j], r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] += r[e];          // x is already T-rounded; the sum rounds on pack
                            v = pack8<T>(x);
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(v, srdY, vY + (unsigned)((b * 16 + j * RPI) * p.ldy) * 2u, 0, 0);
                    }
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < NRD; ++j) {
                const int m = mrow0 + j * RPI + rr;
                bool ok = (m < p.M) && (n < Nout);
                if constexpr (MODE == MODE_QKV_VT) ok = ok && (n < p.vstart);
                if constexpr (LAB::no_store) ok = ok && (p.M < 0);
                if constexpr (emit_stats) {
                    // every lane takes part (row reductions below run on whole 16-lane rows); out-of-range lanes contribute zeros.
                    // The residual chunk was loaded from a clamped address, the staged value is finite: nothing here can fault.
                    float x[8], r[8];
                    unpack8<T>(val[j], x);
                    unpack8<T>(res[b % RD][j], r);
                    // sums of the fp32 values BEFORE their rounding to T (packed adds / FMAs): the stored values differ by independent
                    // half-ulp roundings, which move a 1152-column mean by ~3e-5 of the row's spread — far below what the consumer's
                    // T-rounded output resolves; rounding first would cost two conversions per element here
                    f32x2_t sa = {0.f, 0.f}, sq = {0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2_t y = f32x2_t{x[e], x[e + 1]} + f32x2_t{r[e], r[e + 1]};
                        x[e] = y[0]; x[e + 1] = y[1];
                        sa += y;
                        sq = __builtin_elementwise_fma(y, y, sq);
                    }
                    float s1 = sa[0] + sa[1], s2 = sq[0] + sq[1];
                    if (!ok) { s1 = 0.f; s2 = 0.f; }
                    row16_sum2(s1, s2);                                    // the 16 lanes cc = 0..15 hold this row's 128 columns
                    const int strip = no0 >> 7;
                    if (cc == 0 && m < p.M && no0 < Nout)
                        *(f32x2_t*)(p.stat_part + ((size_t)m * ((Nout + 127) >> 7) + strip) * 2) = f32x2_t{s1, s2};
                    if (ok) *(u32x4*)(Yb + (size_t)m * p.ldy + n) = pack8<T>(x);
                    continue;
                }
                if (!ok) continue;
                if constexpr (MODE == MODE_KV_CACHE) {
                    if (n < p.kvd) {
                        const int tok = p.tok0 + m, kvh = n / p.hd, d = n % p.hd;
                        *(u32x4*)(p.Kc + ((size_t)kvh * p.ntile64 * 64 + tok) * p.hd + d) = val[j];
                    } else {
                        *(u32x4*)(p.Vrow + (size_t)m * p.kvd + (n - p.kvd)) = val[j];
                    }
                } else {
                    u32x4 v = val[j];
                    if constexpr (MODE == MODE_PLAIN) {
                        // the activation runs on the T-rounded staged values (same arithmetic as rounding first, then activating)
                        if constexpr (act_tanh) {
                            // (applied to the whole strip above)
                        }
                        if constexpr (has_res) {
                            float x[8], r[8];
                            unpack8<T>(v, x);
                            unpack8<T>(res[has_res ? b % RD : 0][j], r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] += r[e];          // x is already T-rounded; the sum rounds on pack
                            v = pack8<T>(x);
                        }
                    }
                    if constexpr (EPI::heads) {
                        const int fr = (int)__umulhi((unsigned)m, p.hm_magic), tok = m - fr * p.hm_seq;      // m / seq, m % seq
                        *(u32x4*)(Yb + hm_col + ((size_t)fr * p.hm_heads * p.hm_seq + tok) * p.hm_hd) = v;
                    } else {
                        *(u32x4*)(Yb + (size_t)m * p.ldy + n) = v;
                    }
                }
            }
        };
        // software pipeline over the 8 strips: the next strip's registers -> scratch pass is issued between a strip's reads and
        // the stores that consume them (sched_barrier pins the groups; LDS order makes the single scratch buffer safe)
        // software pipeline over the 8 strips: the next strip's registers -> scratch pass is issued between a strip's reads and
        // the stores that consume them (sched_barrier pins the groups; LDS order makes the single scratch buffer safe)
        auto step = [&](auto bt) {
            constexpr int b = decltype(bt)::value;
            VIDI_PIN;
            fetch(b);
            load_res(std::integral_constant<int, b + RD - 1>{});
            VIDI_PIN;
            if constexpr (b + 1 < TM) stage(std::integral_constant<int, b + 1>{});
            VIDI_PIN;
            store(b);
            VIDI_PIN;
        };
        stage(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    };

    // =========================================== tile loop ===========================================
    using TT = std::true_type; using FF = std::false_type;
    int vb = blockIdx.x;
    locate(vb, m0, n0, bz);
    srdW = make_srd(p.W + (size_t)n0 * p.ldw, p.N - n0, p.ldw);
    if constexpr (PATCH != 0) {
        // ONE descriptor over the whole pixel tensor (the row -> patch map is in the per-lane offsets): T * 3 * S * S pixels < 4 GB
        const int oc2 = (p.pe_side - p.pe_P + 1) * (p.pe_side - p.pe_P + 1);
        const unsigned long long bytes = PATCH == 1 ? (unsigned long long)(p.M / (p.pe_side * p.pe_side)) * 3ull * p.pe_S * p.pe_S * 2ull
                                                    : (unsigned long long)(p.M / oc2) * p.pe_side * p.pe_side * p.pe_S * 2ull;
        srdX = srdXn = rsrc_of(p.X, bytes);
        patch_rows(m0, pxo);
    } else {
        srdX = make_srd(p.X + (long long)bz * p.bsX + (size_t)m0 * p.ldx, p.M - m0, p.ldx);
    }
    // head of the block's first tile: slices 0 and 1 in flight, slice 0 landed, its step-0 fragments in registers
#pragma unroll
    for (int q = 0; q < 16; ++q) piece(smem, 0, q, FF{});
#pragma unroll
    for (int q = 0; q < 16; ++q) piece(smem + STAGE_BYTES, 1, q, FF{});
    if constexpr (!LAB::no_dma) wait_vm<16>();
    bar();
#pragma unroll
    for (int b = 0; b < TM; ++b) fX[0][b] = rdX(smem, b, 0);
#pragma unroll
    for (int a = 0; a < TN; ++a) fW[0][a] = rdW(smem, a, 0);
    stamp(0);
    while (true) {
        const int nvb = vb + gridDim.x;
        const bool has_next = PERSIST && nvb < tiles;
        if (has_next) {
            locate(nvb, nm0, nn0, nbz);
            srdWn = make_srd(p.W + (size_t)nn0 * p.ldw, p.N - nn0, p.ldw);
            if constexpr (PATCH != 0) patch_rows(nm0, pxn);
            else srdXn = make_srd(p.X + (long long)nbz * p.bsX + (size_t)nm0 * p.ldx, p.M - nm0, p.ldx);
        }
        body(0, TT{}, FF{}, true);
        int kt = 1;
        for (; kt + 2 < nk; ++kt) body(kt, FF{}, FF{}, true);
        body(kt, FF{}, TT{}, has_next);             // the next tile's slices 0, 1 are this K loop's slices nk, nk + 1
        body(kt + 1, FF{}, TT{}, has_next);
        // the MFMAs are asm statements: hipcc does not pad the MFMA-result -> reader hazard for them (12 wait states)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        stamp(1);
        if constexpr (LAB::no_epilogue) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) asm volatile("" ::"a"(acc[a][b]));
            if constexpr (ln_inloop) {                              // (lab: keep the K loop's statistics alive without an epilogue)
#pragma unroll
                for (int b = 0; b < TM; ++b) asm volatile("" ::"v"(rs1[b]), "v"(rs2[b]));
            }
        } else {
            epilogue(m0, n0, bz);
        }
        stamp(3);
        if (!has_next) break;
        vb = nvb; m0 = nm0; n0 = nn0; bz = nbz; srdW = srdWn; srdX = srdXn;
        if constexpr (PATCH != 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) pxo[q] = pxn[q];
        }
        pb = (pb + nk) & 1;
        tpar ^= 1;
    }
#undef VIDI_PIN
    if constexpr (LAB::stamps) {
        if (p.dbg && tid == 0 && blockIdx.x < 1024) {
            unsigned long long* d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = t_acc[0]; d[1] = t_acc[1]; d[2] = t_acc[2]; d[3] = t_acc[3]; d[4] = 1;
        }
    }
}
