// Non-causal multi-head self-attention for the encoder towers, ROW-MAJOR V variant (round 2).
//
// Same algorithm, tiling and numerics as attn_self.hip (read its header first); the difference is where V comes from:
//   QKV: Q, K and V of ONE projection GEMM, V in natural order.  Row r of head h of frame b at  base + b*bs + h*hs + r*ld  with base =
//   QKV (Q), QKV + koff (K), QKV + voff (V): row-major [B*N, ld] (bs = N*ld, hs = D) or HEAD-MAJOR [3][B][H][N][D] (bs = H*N*D, hs = N*D,
//   ld = D) — there a head's key rows are contiguous and the K / V tiles are whole 128-byte lines (+12 % on this kernel, same box:
//   a 144-byte head row inside a 6 912-byte token row costs two line fetches).
// attn_self.hip wants V transposed and key-permuted (Vt[b][h][d][Npad]), which made the QKV GEMM's epilogue scatter 2-byte stores for a
// third of its columns (-14 % on that GEMM).  Here the V tile is DMA'd into LDS exactly like the K tile ([64 keys][D], 16-byte pieces)
// and the PV MFMA's A fragments (lane = d row, 8 keys per lane) are read with gfx950's LDS transpose read, `ds_read_b64_tr_b16`: every
// 16-lane group reads a [4 keys][16 d] block — lane i supplies the address of key (i >> 2), d-columns 4 (i & 3) .. +3 — and each lane
// receives its d-column for the 4 keys (semantics pinned by tools/micro/tr_read_probe.hip).  The key order the swapped QK^T leaves in the
// P registers (keys 4 hi + {0..3} and 8 + 4 hi + {0..3} per 16-slab, krow32) is produced by the ADDRESSES of the two reads of a
// fragment, so memory holds V in natural order.  The softmax denominator's ones row (d = D when D % 32 != 0) is patched into the
// fragment registers of the lanes that own output row D.
//
// Round 3: (1) the transpose reads are inline asm (VIDI_ATTN_RM_TRASM: the builtin drew a wait for the next tile's LDS-DMA in front of them);
// (2) the d = 72 instantiation runs two 32-query sets per wave in a hand-pipelined tile body (VIDI_ATTN_RM_QS72), its tile DMA is five
// branch-free instructions per wave, keys past N are masked by a bias in a spare contraction slot; (3) template flag PS: the caller folded
// scale * log2(e) into Q (scale <= 0 at the C ABI) — the running maximum then rides in the contraction too, the exponent needs no FMA, and
// the pipeline is carried across the tile boundary with the rendezvous at the tile's midpoint.  profiles/r3_notes.md has the measurements.
//
// Algorithmic FLOPs = 4*N*N*D per (batch, head).
#include "kernels.h"
#include <stdlib.h>

#ifndef VIDI_ATTN_PRIO
#define VIDI_ATTN_PRIO 1               // see attn_self.hip
#endif
#define VIDI_ATTN_RM_PRIO_HI(bit) do { if (VIDI_ATTN_PRIO & (bit)) asm volatile("s_setprio 1" ::: "memory"); } while (0)
#define VIDI_ATTN_RM_PRIO_LO(bit) do { if (VIDI_ATTN_PRIO & (bit)) asm volatile("s_setprio 0" ::: "memory"); } while (0)


typedef short v4s16 __attribute__((ext_vector_type(4)));

#ifndef VIDI_ATTN_RM_QS72
#define VIDI_ATTN_RM_QS72 2            // query sets of 32 per wave in the d = 72 instantiation (SigLIP): 2 = a 256-query block whose waves run
#endif                                 // TWO softmax / PV chains against every K / V tile, software-pipelined by hand (see the tile body): the
                                       // K / V fragments are read from LDS once for both, the tile DMA and the rendezvous are per 256 queries, and
                                       // the matrix work of one chain is issued between the softmax instructions of the other.  238 VGPRs -> two
                                       // waves per SIMD.  Left to the compiler the two chains ran back to back (399 vs 636 useful TFLOP/s in
                                       // round 2); pipelined: 714 -> 747 on the SigLIP shape, bit-identical (profiles/r3_ab_attn_pipe.jsonl).
                                       // Every other head dim keeps one set per wave (three waves per SIMD).
#ifndef VIDI_ATTN_RM_QS64
#define VIDI_ATTN_RM_QS64 2            // the same two-set body for d = 64 (Whisper): no spare contraction chunk and no ones row there, so keys past N
#endif                                 // are masked through the first QK^T instruction's C input (zeros except in the last tile) and the row sums stay
                                       // on the VALU; ~190 VGPRs, two waves per SIMD
template <int D> constexpr int attn_rm_qs() { return D == 72 ? VIDI_ATTN_RM_QS72 : (D == 64 ? VIDI_ATTN_RM_QS64 : 1); }

#ifndef VIDI_ATTN_RM_TRASM
#define VIDI_ATTN_RM_TRASM 1           // 1: the V transpose reads are inline asm.  Through the builtin the compiler cannot tell that the LDS-DMA
#endif                                 // of tile t + 1 (issued at the top of tile t) writes the OTHER ring slot and puts `s_waitcnt vmcnt(0)` in
                                       // front of the first transpose read of every tile: the prefetch was waited for one instruction after its
                                       // issue.  As asm the reads carry no memory operand; their completion is waited for explicitly (tr_wait).
// one 64-bit LDS transpose read; OFF is an instruction immediate
// INVARIANT (not visible to the compiler): the read completes ASYNCHRONOUSLY, but to the compiler `dst` is written at the asm statement.
// Between a VIDI_TR_READ and the tr_wait that follows it nothing may read, copy or spill `dst` — a v_mov / scratch store of it there
// would move stale data, silently.  At 252-253 VGPRs the register allocator has no reason to, but nothing in the source forbids it:
// tests/test_build_resources.py::test_transpose_read_destinations_untouched_until_the_wait disassembles the built object and fails if
// any instruction between a transpose read and the next `s_waitcnt lgkmcnt(0)` names one of the pending destination registers (knob
// variants and compiler upgrades included); VIDI_ATTN_RM_TRASM=0 (the builtin, compiler-tracked) stays available as the reference arm.
#define VIDI_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
// every LDS read issued so far has returned; the 12 raw halves pass through so that their consumers are ordered behind the wait
template <int N>
__device__ __forceinline__ void tr_wait(u32x2 (&r)[12]) {
    static_assert(N == 4 || N == 8 || N == 12, "halves in use");
    if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
    else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]));
}
template <int V> struct IntC { static constexpr int value = V; };
#ifndef VIDI_ATTN_RM_FLATDMA
#define VIDI_ATTN_RM_FLATDMA 1         // d = 72, two query sets: the tile DMA as five branch-free instructions per wave (see issue_dma_flat)
#endif
#ifndef VIDI_ATTN_RM_XTILE
#define VIDI_ATTN_RM_XTILE 1           // d = 72: the two-set pipeline carried across the tile boundary (0: per tile)
#endif
#ifndef VIDI_ATTN_RM_PREX
#define VIDI_ATTN_RM_PREX 3            // matrix instructions in front of the running-max branch in the cross-tile loop (2 measured +4 % on one box and
#endif                                 // -1.5 % on another, and leaves the fp16 build 2 registers short: 252 / 253 VGPRs at 3)
#ifndef VIDI_ATTN_RM_VF0
#define VIDI_ATTN_RM_VF0 2             // softmax chunk of step 1 after which the V fragments of sub-tile 0 are requested
#endif
#ifndef VIDI_ATTN_RM_PRE1
#define VIDI_ATTN_RM_PRE1 2            // matrix instructions placed before the running-max branch of a pipelined step (steps 1 and 4 / 2 and 3)
#endif
#ifndef VIDI_ATTN_RM_PRE2
#define VIDI_ATTN_RM_PRE2 3
#endif
template <typename T, int D, int QS, bool PS>
__global__ __launch_bounds__(256, QS) void attn_self_rm_kernel(AttnSelfRmParams p) {      // (QS = 2: two waves per SIMD, 256 registers)
    constexpr int QB = 128 * QS;
    static_assert(!PS || (QS == 2 && D == 72), "the max-in-the-contraction form needs the spare contraction chunk of d = 72");
    constexpr int KS = (D + 15) / 16;          // k16 steps of the QK^T contraction
    constexpr int NCH = D / 8;                 // 16-byte chunks per head row
    constexpr int DT = (D + 31) / 32;          // 32-wide output d tiles
    constexpr int ROWB = NCH * 16;             // bytes per key row of a tile
    // Both operand tiles are [64 keys][NCH chunks], written by 16-byte LDS-DMA (lane-linear, rows unpadded); bank conflicts are avoided
    // by permuting which global chunk each lane fetches: chunk c of row r sits at slot c ^ swz(r).
    //   K tile: read as ds_read_b128 fragments (lane = key row)              -> kswz, as in attn_self.hip
    //   V tile: see MAINC / TAILC below
    // V tile image for the transpose reads (a 16-lane group reads a [4 keys][16 d] block, the two groups of a 32-lane half read
    // neighbouring column blocks of the same 4 keys and must not share banks — a [64][72] image with 144-byte rows lost 37 % of its LDS
    // cycles to conflicts):
    //   main block  [64 keys][MAINC d], MAINC = 64 / 32 / 16: whole 128-byte rows when D >= 64, fetched as whole rows (8 keys per 1 KB
    //               piece = 8 full lines, like the K tile); with 128-byte rows keys r and r + 2 would share banks, so chunk c of key r
    //               sits at slot c ^ 2 (r & 3) (bank-conflict-free for every (key, column-block) pair of a half: checked with
    //               SQ_LDS_BANK_CONFLICT)
    //   tail block  [64 keys][8 d] (D = 72: d 64..71), 16-byte rows, ONE piece; its missing columns (72..79) are supplied from the
    //               existing ones — they only feed output rows >= D, which are either the ones row (patched below) or never stored
    constexpr int MAINC = D >= 64 ? 64 : (D >= 32 ? 32 : 16), TAILC = D - MAINC, MCH = MAINC / 8, MROWB = MAINC * 2;
    static_assert(TAILC == 0 || TAILC == 8, "head dims: 16, 32, 64, 72");
    constexpr int KBYTES = 64 * NCH * 16, VMAIN = 64 * MROWB, VBYTES = VMAIN + (TAILC ? 1024 : 0), BUF = KBYTES + VBYTES;
    constexpr int KRND = (64 * NCH + 255) / 256, VRND = (MCH + 3) / 4;
    constexpr int ORW = (NCH % 2 == 0) ? (NCH + 1) * 16 : (NCH + 2) * 16;      // output staging row: odd number of 16-B chunks
    // When D is not a multiple of 32 the last d-tile has spare output rows; row D accumulates the softmax denominator: its lanes must
    // receive ONES from the transpose reads.  The lanes that supply the addresses of columns D .. D+3 point into a constant region whose
    // rows read {1, 0, 0, 0}, the suppliers of columns beyond into its zero half (same row stride as the real data of that d-tile, so
    // the per-read immediate offsets apply to them too): the sum of the T-rounded P comes out of the matrix pipe, no VALU adds, no patch.
    constexpr bool kOnesRow = (DT * 32 > D);
    constexpr int CROWB = TAILC ? 16 : MROWB, CBYTES = kOnesRow ? 64 * CROWB : 0;
    constexpr int RING = 2 * BUF > 128 * ORW ? 2 * BUF : 128 * ORW;
    static_assert(RING + CBYTES <= 65536, "static LDS");
    __shared__ __attribute__((aligned(16))) char smem[RING + CBYTES];   // 2-deep K/V ring (+ the ones / zeros constant rows); a 3-deep ring
                                                                        // (tile t + 2 in flight) measured 4 % slower: the DMA latency is covered
    auto kswz = [](int r) { return NCH == 8 ? ((r >> 1) & 7) : (NCH == 4 ? ((r >> 2) & 3) : 0); };

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nqt = (p.N + QB - 1) / QB, per_b = nqt * p.H;
    int qt, h, b;
    {
        const int L = blockIdx.x;
        const int nfull = (p.B / 8) * 8;                     // frames that can be dealt round-robin over the 8 XCDs
        if ((D * 2) % 128 == 0 && L < nfull * per_b) {
            const int xcd = L & 7, j = L >> 3;
            b = 8 * (j / per_b) + xcd;
            const int w = j % per_b;
            h = w / nqt; qt = w % nqt;
        } else if ((D * 2) % 128 == 0) {                      // remainder frames: plain order
            const int w = L - nfull * per_b;
            b = nfull + w / per_b;
            h = (w % per_b) / nqt; qt = (w % per_b) % nqt;
        } else if ((p.B * p.H) % 8 == 0) {
            // D=72: the q-tiles of one (frame, head) run on ONE XCD (they share its K/V through that L2), while neighbouring heads -
            // whose 144-byte output segments share 128-byte lines - go to different XCDs (see attn_self.hip)
            const int xcd = L & 7, j = L >> 3;
            const int unit = (j / nqt) * 8 + xcd;
            qt = j % nqt; b = unit / p.H; h = unit % p.H;
        } else {                                              // q-tile fastest, then head, then frame
            qt = L % nqt; h = (L / nqt) % p.H; b = L / per_b;
        }
    }
    // Q fragments (B operand: column = query, contraction chunk = 2s + hi); chunks past D are zero.  Query set qs of this wave: rows
    // qt*QB + qs*128 + wave*32 + l31 (each set is a 128-row band of the block, staged and stored as such in the epilogue)
    u32x4 qf[QS][KS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        const int qc = min(qt * QB + qs * 128 + wave * 32 + l31, p.N - 1);
        const u16* qrow = p.QKV + (size_t)b * p.bs + (size_t)h * p.hs + (size_t)qc * p.ld;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 2 * s + hi;
            qf[qs][s] = (c < NCH) ? *(const u32x4*)(qrow + c * 8) : u32x4{0, 0, 0, 0};
        }
        if constexpr (QS == 2 && D == 72) {      // the tail bias rides in contraction chunk 9 (see load_kf); PS: so does -max (slot 1, 0 at first)
            if (hi) qf[qs][KS - 1][0] = (unsigned)T::from_f32(1.0f);
        }
    }

    const u16* kbase_ptr = p.QKV + p.koff + (size_t)b * p.bs + (size_t)h * p.hs;
    const u16* vbase_ptr = p.QKV + p.voff + (size_t)b * p.bs + (size_t)h * p.hs;

    // per-thread DMA pieces (tile independent).  K: piece j = (key row, source column) as in attn_self.hip.  V main block: 1 KB piece pc
    // holds keys (64 / MCH) pc .. of the main block — lane L fetches the chunk that belongs at slot L % MCH of key row L / MCH.
    int krow[KRND], kcol[KRND];
#pragma unroll
    for (int j = 0; j < KRND; ++j) {
        const int i = j * 256 + tid;
        const int row = i / NCH, cs = i % NCH;
        krow[j] = row; kcol[j] = (cs ^ kswz(row)) * 8;
    }
    auto vswz = [](int r) { return MAINC == 64 ? 2 * (r & 3) : 0; };
    int vrow[VRND], vcol[VRND];
#pragma unroll
    for (int j = 0; j < VRND; ++j) {
        const int pc = j * 4 + wave;
        const int row = pc * (64 / MCH) + lane / MCH;
        vrow[j] = row; vcol[j] = ((lane % MCH) ^ vswz(row)) * 8;
    }
#ifndef VIDI_ATTN_RM_SRD
#define VIDI_ATTN_RM_SRD 1             // 1 (default): tile DMA through buffer descriptors — the per-lane byte offsets are tile-invariant and the
#endif                                 // tile advance is a SCALAR offset, so a piece costs no vector address arithmetic (the pointer form, 0,
                                       // spends v_add + v_min + v_mad_i64 + v_lshl_add per piece: 26 VALU instructions per tile); key rows past N
                                       // are outside the descriptor's range and read as zeros instead of repeating row N - 1 (both finite; their
                                       // scores are masked to -inf, their probabilities are exactly 0).  Same-box A/B, N = 729, d = 72, 360
                                       // frames: 654.9 -> 686.7 useful TFLOP/s on head-major input, 598.1 -> 629.6 on row-major, bit-identical
                                       // (profiles/r3_ab_attn_srd.jsonl)
    __amdgpu_buffer_rsrc_t srdK, srdV;
    if constexpr (VIDI_ATTN_RM_SRD != 0) {
        const unsigned bytes = (unsigned)(((size_t)(p.N - 1) * p.ld + D) * 2);
        // (inside a lambda: a target builtin called directly in the kernel body makes the HOST pass drop the kernel's launch stub; so does
        // an array element as the DMA's vector offset — the per-lane offsets below are loop-invariant expressions the compiler hoists)
        auto mk = [](const u16* base, unsigned nbytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nbytes, 0x00020000); };
        srdK = mk(kbase_ptr, bytes);
        srdV = mk(vbase_ptr, bytes);
    }
    auto issue_dma_srd = [&](int kb, auto bufi) {          // (generic on purpose: the LDS-DMA builtin in a non-dependent context makes the host pass drop the kernel's stub)
        char* sK = smem + bufi * BUF;
        char* sV = sK + KBYTES;
        const unsigned so = (unsigned)kb * (unsigned)p.ld * 2u;                 // wave-uniform: the tile's first key row
#pragma unroll
        for (int j = 0; j < KRND; ++j) {
            const int ib = j * 256 + wave * 64;
            if (ib < 64 * NCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srdK, (__attribute__((address_space(3))) void*)(sK + ib * 16), 16, (unsigned)(krow[j] * p.ld + kcol[j]) * 2u, so, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < VRND; ++j) {
            const int pc = j * 4 + wave;
            if (pc < MCH) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV, (__attribute__((address_space(3))) void*)(sV + pc * 1024), 16, (unsigned)(vrow[j] * p.ld + vcol[j]) * 2u, so, 0, 0);
        }
        if constexpr (TAILC != 0) {
            if (wave == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV, (__attribute__((address_space(3))) void*)(sV + VMAIN), 16, (unsigned)(lane * p.ld + MAINC) * 2u, so, 0, 0);
        }
    };
    // d = 72, two query sets: the same 18 pieces (9 K + 8 V + the V tail) as exactly FIVE instructions per wave and no branches — the
    // wave-dependent choices (which piece, its LDS address, which descriptor) are made once, here.  Slot 2 of waves 0-1 is K piece 8, of
    // waves 2-3 the V tail: both are fetched twice (same bytes to the same LDS address), which costs 2 KB of L2 reads per tile and saves
    // ~25 scalar instructions and 10 branches per tile and wave (the loop is instruction-issue-bound).
    constexpr bool kFlatDma = (QS == 2 && (D == 72 || D == 64) && VIDI_ATTN_RM_SRD != 0 && VIDI_ATTN_RM_FLATDMA != 0);   // (d = 64: 8 K + 8 V pieces, four per wave)
    unsigned fvo2 = 0;
    int fdst2 = 0;
    __amdgpu_buffer_rsrc_t srdM = srdK;
    if constexpr (kFlatDma && D == 72) {
        static_assert(!(kFlatDma && D == 72) || (KRND == 3 && VRND == 2 && MCH == 8 && TAILC == 8), "piece map of d = 72");
        const int i2 = 512 + lane;                                   // K piece 8: chunks 512 .. 575 (no swizzle at 9 chunks per row)
        fvo2 = wave < 2 ? (unsigned)((i2 / NCH) * p.ld + (i2 % NCH) * 8) * 2u : (unsigned)(lane * p.ld + MAINC) * 2u;
        fdst2 = wave < 2 ? 8 * 1024 : KBYTES + VMAIN;
        srdM = wave < 2 ? srdK : srdV;
    }
    auto issue_dma_flat = [&](int kb, auto bufi) {
        static_assert(!kFlatDma || D == 72 || (KRND == 2 && VRND == 2 && MCH == 8 && TAILC == 0), "piece map of d = 64");
        char* sK = smem + bufi * BUF;
        const unsigned so = (unsigned)kb * (unsigned)p.ld * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdK, (__attribute__((address_space(3))) void*)(sK + wave * 1024), 16, (unsigned)(krow[0] * p.ld + kcol[0]) * 2u, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdK, (__attribute__((address_space(3))) void*)(sK + 4096 + wave * 1024), 16, (unsigned)(krow[1] * p.ld + kcol[1]) * 2u, so, 0, 0);
        if constexpr (D == 72) __builtin_amdgcn_raw_ptr_buffer_load_lds(srdM, (__attribute__((address_space(3))) void*)(sK + fdst2), 16, (int)fvo2, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV, (__attribute__((address_space(3))) void*)(sK + KBYTES + wave * 1024), 16, (unsigned)(vrow[0] * p.ld + vcol[0]) * 2u, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srdV, (__attribute__((address_space(3))) void*)(sK + KBYTES + 4096 + wave * 1024), 16, (unsigned)(vrow[1] * p.ld + vcol[1]) * 2u, so, 0, 0);
    };
    auto issue_dma = [&](int kb, int bufi) {
        if constexpr (kFlatDma) { issue_dma_flat(kb, bufi); return; }
        if constexpr (VIDI_ATTN_RM_SRD != 0) { issue_dma_srd(kb, bufi); return; }
        char* sK = smem + bufi * BUF;
        char* sV = sK + KBYTES;
#pragma unroll
        for (int j = 0; j < KRND; ++j) {
            const int ib = j * 256 + wave * 64;              // wave-uniform: whole 64-chunk pieces only
            if (ib < 64 * NCH)      // rows past N re-read the last key (finite; masked / weighted 0)
                glds16(kbase_ptr + (size_t)min(kb + krow[j], p.N - 1) * p.ld + kcol[j], sK + ib * 16);
        }
#pragma unroll
        for (int j = 0; j < VRND; ++j) {
            const int pc = j * 4 + wave;
            if (pc < MCH) glds16(vbase_ptr + (size_t)min(kb + vrow[j], p.N - 1) * p.ld + vcol[j], sV + pc * 1024);
        }
        if constexpr (TAILC != 0) {
            if (wave == 3) glds16(vbase_ptr + (size_t)min(kb + lane, p.N - 1) * p.ld + MAINC, sV + VMAIN);      // key = lane, d MAINC .. +7
        }
    };

    // transpose-read addresses of this lane inside a V tile: 16-lane group g reads the [4 keys][16 d] block of d-columns
    // dt*32 + 16 (g & 1) .. +15; lane i of the group supplies key row 4 hi + (i >> 2) (+ 32 u + {0, 8, 16, 24} at the call), d-columns
    // 4 (i & 3) .. +3 and receives column (i & 15) of the block for those 4 keys
    int vaddr[DT], vstep[DT];                            // byte offset from the start of smem in ring buffer 0; + vstep in buffer 1
    {
        const int i = lane & 15, g1 = (lane >> 4) & 1, r = 4 * hi + (i >> 2);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 32 + 16 * g1 + 4 * (i & 3);                    // first of the 4 d-columns whose address this lane supplies
            int off, in_ring = 1;
            if (d0 < MAINC) off = KBYTES + r * MROWB + (((d0 >> 3) ^ vswz(r)) << 4) + (i & 1) * 8;
            else if (d0 < D) off = KBYTES + VMAIN + r * 16 + (i & 1) * 8;     // tail block: d MAINC .. D-1, 16-byte rows
            else { off = RING + r * CROWB + (d0 == D ? 0 : 8); in_ring = 0; }  // columns >= D: the ones row (d0 == D) / zeros
            vaddr[dt] = off;
            vstep[dt] = in_ring * BUF;
        }
    }

    f32x16 o[QS][DT];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs)
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qs][t][i] = 0.f;
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    // running maximum in base-2 logit units (score * scale * log2 e), kept as the two values the loop uses every sub-tile — its negative
    // (the addend of the exponent FMAs) and the rescale threshold m + TAU — so that the common no-rescale case costs one multiply and
    // one compare after the max tree (the loop is issue-bound: every instruction counts)
    float nm_run[QS], thr_run[QS], l_run[QS];
#pragma unroll
    for (int qs = 0; qs < QS; ++qs) { nm_run[qs] = PS ? 0.f : INFINITY; thr_run[qs] = -INFINITY; l_run[qs] = 0.f; }
    const float sc = p.scale > 0.f ? p.scale * 1.4426950408889634f : 1.0f;      // fold log2(e): softmax in base 2 (scale <= 0: Q carries it)

    const int ntiles = (p.N + 63) / 64;
    const bool half_last = (((p.N - 1) & 63) < 32);      // the last tile's second 32 keys are all past N
    const int ksw = kswz(l31);
    if constexpr (kOnesRow) {                            // constant rows: {1, 0, 0, 0 | 0 ...}
        const unsigned one1 = (unsigned)T::from_f32(1.0f);
        for (int i = tid; i < CBYTES / 4; i += 256) *(unsigned*)(smem + RING + i * 4) = (i % (CROWB / 4) == 0) ? one1 : 0u;
    }
    constexpr int ROWD = D - (DT - 1) * 32;
    // ---- pieces of the two-set pipelined tile body (d = 72; see the loops below) ----
    const __attribute__((address_space(3))) char* lds0 = (const __attribute__((address_space(3))) char*)smem;
    // d = 64 has no spare contraction chunk for the tail bias: there the first QK^T instruction of a sub-tile starts from cin[u] — zeros,
    // except in the last tile, where the rows of keys past N hold -inf (set once at the top of that tile)
    constexpr bool kBiasSlot = (D == 72), kCinMask = (QS == 2 && D == 64);
    f32x16 cin[kCinMask ? 2 : 1];
    if constexpr (kCinMask) { cin[0] = zero16; cin[1] = zero16; }
    auto QKm = [&](int s, const u32x4 (&kf)[KS], int qs, f32x16& S, int u = 0) __attribute__((always_inline)) {
        if constexpr (kCinMask) S = T::mfma32(kf[s], qf[qs][s], s == 0 ? cin[u] : S);
        else S = T::mfma32(kf[s], qf[qs][s], s == 0 ? zero16 : S);
    };
    auto PVm = [&](int j, const u32x4 (&vf)[DT][2], int qs, const u32x4& pf0, const u32x4& pf1) __attribute__((always_inline)) {
        const int dt = j % DT, m = j / DT;
        o[qs][dt] = T::mfma32(vf[dt][m], m ? pf1 : pf0, o[qs][dt]);
    };
    auto load_kf = [&](int u, u32x4 (&kf)[KS], const char* sK, int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KS; ++s) kf[s] = *(const u32x4*)(sK + (u * 32 + l31) * ROWB + (((2 * s + hi) ^ ksw) << 4));
        // keys past N, without a branch and without touching the scores: d = 72 leaves contraction chunk 9 (the hi lanes of the last
        // k-step) empty; Q carries 1.0 in its first slot, K a bias: 0 for a real key, -inf for one past N — the matrix pipe adds it
        // (x + 0 = x exactly; x - inf = -inf: what masking the score would have given)
        if constexpr (!kBiasSlot) return;
        constexpr unsigned NEG_INF = T::id == VIDI_DT_BF16 ? 0xff80u : 0xfc00u;      // -inf in T
        constexpr unsigned ONE_HI = PS ? ((unsigned)(T::id == VIDI_DT_BF16 ? 0x3f80u : 0x3c00u) << 16) : 0u;      // PS: K's slot 1 = 1.0 (times Q's -max)
        const unsigned bias = ((kb + u * 32 + l31 >= p.N) ? NEG_INF : 0u) | ONE_HI;
        kf[KS - 1][0] = hi ? bias : kf[KS - 1][0];
    };
    u32x2 vraw[12];
    auto load_vf = [&](int u, u32x4 (&vf)[DT][2], const int (&va)[DT]) __attribute__((always_inline)) {
        static_assert(DT * 4 <= 12, "raw halves");
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int rowb = dt * 32 < MAINC ? MROWB : 16;
                if constexpr (VIDI_ATTN_RM_TRASM != 0) {
                    const unsigned a = (unsigned)(uintptr_t)(lds0 + va[dt]);
                    VIDI_TR_READ(vraw[(dt * 2 + m) * 2], a, (u * 32 + 16 * m) * rowb);
                    VIDI_TR_READ(vraw[(dt * 2 + m) * 2 + 1], a, (u * 32 + 16 * m + 8) * rowb);
                } else {
                    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(lds0 + va[dt] + (u * 32 + 16 * m) * rowb));
                    const v4s16 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(lds0 + va[dt] + (u * 32 + 16 * m + 8) * rowb));
                    const u32x2 a = __builtin_bit_cast(u32x2, lo), c = __builtin_bit_cast(u32x2, up);
                    vf[dt][m] = u32x4{a[0], a[1], c[0], c[1]};
                }
            }
    };
    auto vf_ready = [&](u32x4 (&vf)[DT][2]) __attribute__((always_inline)) {        // before the first PV of a sub-tile
        if constexpr (VIDI_ATTN_RM_TRASM != 0) {
            tr_wait<DT * 4>(vraw);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const u32x2 a = vraw[(dt * 2 + m) * 2], c = vraw[(dt * 2 + m) * 2 + 1];
                    vf[dt][m] = u32x4{a[0], a[1], c[0], c[1]};
                }
        }
    };
    constexpr float TAU = 8.0f;
    // running max of an item (+ the rare rescale): ends in a branch.  NPRE matrix instructions go in front of it.
    auto sm_a = [&](int qs, f32x16& S, auto&& M, auto npre) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < decltype(npre)::value; ++c) M(c);
        float mx = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
        mx = xhalf_max(mx);
        if constexpr (PS) {
            // S already is (score - max~) in base-2 units: Q carries the scale (host contract: scale <= 0) and -max~ in contraction slot 73
            // (K has 1.0 there), max~ = the running maximum ROUNDED TO T (any reference works as long as every term uses the same one).
            // nm_run = -max~ (the slot's value, 0 before the first sub-tile), thr_run = -inf before the first sub-tile, then TAU.
            if (__builtin_expect(__any(mx > thr_run[qs]), 0)) {
                asm volatile("" : "+v"(mx));
                // The branch is taken wave-wide when ANY query triggers; a lane that did not (mx far below 0) must keep its maximum:
                // mx is clamped at 0 (= the old max~ in these units) except on the first sub-tile (floor -inf: nothing seen yet), so
                // max~ never moves down, delta <= 0 and alpha <= 1 for every lane (an early outlier followed by a much lower sub-tile
                // would otherwise give alpha = 2^(+large) -> inf)
                const float floor0 = fminf(thr_run[qs], 0.f);                          // -inf on the first sub-tile, else 0
                const float m_new = T::to_f32(T::from_f32(fmaxf(mx, floor0) - nm_run[qs]));   // max~ of everything seen so far
                const float delta = -nm_run[qs] - m_new;                               // exact: both are T values
                const float alpha = fast_exp2(delta + floor0);                         // 0 on the first sub-tile (o = 0 there; -max~ may be huge)
                nm_run[qs] = -m_new; thr_run[qs] = TAU;
#pragma unroll
                for (int r = 0; r < 16; ++r) S[r] += delta;                            // this item's scores were formed with the old slot value
                if (hi) qf[qs][KS - 1][0] = (unsigned)T::from_f32(1.0f) | ((unsigned)T::from_f32(-m_new) << 16);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[qs][dt][i] *= alpha;
            }
            return;
        }
        float mxs = mx * sc;
        if (__builtin_expect(__any(mxs > thr_run[qs]), 0)) {
            asm volatile("" : "+v"(mxs));                    // (keeps the rare path's arithmetic inside the branch: LLVM speculates it above)
            const float m_cand = fmaxf(-nm_run[qs], mxs);
            const float alpha = fast_exp2(-nm_run[qs] - m_cand);
            nm_run[qs] = -m_cand; thr_run[qs] = m_cand + TAU;
            if constexpr (!kOnesRow) l_run[qs] *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[qs][dt][i] *= alpha;
        }
    };
    // exponentials + packing of an item in 8 chunks of 2 probabilities; matrix instruction M(c) leads chunk c
    auto sm_b = [&](int qs, const f32x16& S, u32x4& pf0, u32x4& pf1, auto&& M) __attribute__((always_inline)) {
        unsigned w[8];
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            M(c);
            const float e0 = PS ? fast_exp2(S[2 * c]) : fast_exp2(__builtin_fmaf(S[2 * c], sc, nm_run[qs]));
            const float e1 = PS ? fast_exp2(S[2 * c + 1]) : fast_exp2(__builtin_fmaf(S[2 * c + 1], sc, nm_run[qs]));
            w[c] = pack2<T>(e0, e1);
            if constexpr (!kOnesRow) { ps0 += e0; ps1 += e1; }       // (no ones row at d = 64: the row sums stay on the VALU, as in the one-set body)
        }
        if constexpr (!kOnesRow) l_run[qs] += ps0 + ps1;
#pragma unroll
        for (int c = 8; c < 12; ++c) M(c);           // (matrix instructions beyond the 8 chunks, if the step has more)
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(w[c]));       // keep the exponentials in THIS block (they are pure: LLVM sinks them to their users)
        pf0 = u32x4{w[0], w[1], w[2], w[3]}; pf1 = u32x4{w[4], w[5], w[6], w[7]};
    };
    issue_dma(0, 0);

    // ---- d = 72, two query sets, pipeline carried ACROSS the tile boundary --------------------------------------------------------------
    // In the per-tile form below QK^T of (A,0) and PV of (B,1) have no softmax of their own wave beside them (11 of a tile's 44 matrix
    // instructions).  Here QK^T of the next tile's (A,0) runs beside the softmax of (B,1) and PV of (B,1) beside the next tile's (A,0):
    // every step is one softmax item beside 11 matrix instructions.  The rendezvous moves to the middle of the tile: by then every LDS read
    // of tile t has returned (K(1), V(0), V(1) are in registers after step 3), so slot t & 1 can take tile t + 2 at once, and tile t + 1 —
    // requested at the previous tile's midpoint, a whole tile ago — is waited for there.  Still a 2-slot ring, still one barrier per tile.
    // (only the prescaled-Q form: with the scale > 0 form's two extra state registers per set this loop spills)
    constexpr bool kXTile = (PS && QS == 2 && D == 72 && kFlatDma && VIDI_ATTN_RM_XTILE != 0);
    if constexpr (kXTile) {
        constexpr int P = VIDI_ATTN_RM_PREX;
        u32x4 kf[KS], vf[DT][2], pa0, pa1, pb0 = {0, 0, 0, 0}, pb1 = {0, 0, 0, 0};
        f32x16 SA, SB;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { vf[dt][0] = u32x4{0, 0, 0, 0}; vf[dt][1] = u32x4{0, 0, 0, 0}; }     // the first "previous PV" adds 0 * 0
        if (ntiles > 1) {
            issue_dma(64, 1);
            asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");       // tile 0 has landed (five pieces per wave and tile), tile 1 may be on its way
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        load_kf(0, kf, smem, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) QKm(s, kf, 0, SA);
        // (the two halves of a tile as lambdas: the LAST tile is peeled off the loop so that its second half can be skipped when its 32 keys are
        // all past N — N = 729: keys 736 .. 767 — without a second exit from the loop, which cost hundreds of spilled registers)
        auto first_half = [&](int t) __attribute__((always_inline)) {
            const int kb = t * 64, slot = t & 1;
            const char* sK = smem + slot * BUF;
            int va[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) va[dt] = vaddr[dt] + slot * vstep[dt];
            // step 1: softmax (A,0)  ||  PV (B,1) of tile t - 1, QK (B,0)
            { auto M = [&](int i) __attribute__((always_inline)) { if (i < 2 * DT) PVm(i, vf, 1, pb0, pb1); else if (i < 2 * DT + KS) QKm(i - 2 * DT, kf, 1, SB); };
              sm_a(0, SA, M, IntC<P>{});
              sm_b(0, SA, pa0, pa1, [&](int c) __attribute__((always_inline)) { M(c + P); if (c + P == 2 * DT - 1) load_vf(0, vf, va); if (c + P == 2 * DT + KS - 1) load_kf(1, kf, sK, kb); }); }
            // step 2: softmax (B,0)  ||  PV (A,0), QK (A,1)
            vf_ready(vf);
            { auto M = [&](int i) __attribute__((always_inline)) { if (i < 2 * DT) PVm(i, vf, 0, pa0, pa1); else if (i < 2 * DT + KS) QKm(i - 2 * DT, kf, 0, SA); };
              sm_a(1, SB, M, IntC<P>{});
              sm_b(1, SB, pb0, pb1, [&](int c) __attribute__((always_inline)) { M(c + P); }); }
        };
        auto second_half = [&](int t) __attribute__((always_inline)) {
            const int kb = t * 64, slot = t & 1;
            int va[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) va[dt] = vaddr[dt] + slot * vstep[dt];
            // step 3: softmax (A,1)  ||  PV (B,0), QK (B,1); V fragments of sub-tile 1 replace sub-tile 0's once its last PV has been issued
            { auto M = [&](int i) __attribute__((always_inline)) { if (i < 2 * DT) PVm(i, vf, 1, pb0, pb1); else if (i < 2 * DT + KS) QKm(i - 2 * DT, kf, 1, SB); };
              sm_a(0, SA, M, IntC<P>{});
              sm_b(0, SA, pa0, pa1, [&](int c) __attribute__((always_inline)) { M(c + P); if (c + P == 2 * DT - 1) load_vf(1, vf, va); }); }
            // midpoint: every LDS read of tile t has returned; tile t + 1 (requested a tile ago) has landed, for everyone
            vf_ready(vf);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (t + 2 < ntiles) issue_dma(kb + 128, slot);
            load_kf(0, kf, smem + (slot ^ 1) * BUF, kb + 64);            // (past the last tile: stale bytes, the scores are never used)
            // step 4: softmax (B,1)  ||  PV (A,1), QK (A,0) of tile t + 1
            { auto M = [&](int i) __attribute__((always_inline)) { if (i < 2 * DT) PVm(i, vf, 0, pa0, pa1); else if (i < 2 * DT + KS) QKm(i - 2 * DT, kf, 0, SA); };
              sm_a(1, SB, M, IntC<P>{});
              sm_b(1, SB, pb0, pb1, [&](int c) __attribute__((always_inline)) { M(c + P); }); }
        };
        for (int t = 0; t + 1 < ntiles; ++t) { first_half(t); second_half(t); }
        first_half(ntiles - 1);
        if (!half_last) second_half(ntiles - 1);         // else: vf / pb still hold V(0) / P(B,0): the PV below consumes exactly those
#pragma unroll
        for (int j = 0; j < 2 * DT; ++j) PVm(j, vf, 1, pb0, pb1);           // PV (B,1) of the last tile
    }

    if constexpr (!kXTile)
    for (int t = 0; t < ntiles; ++t) {
        const int kb = t * 64;
        wait_vmcnt<0>();                                  // my pieces of tile t have landed ...
        __syncthreads();                                  // ... everyone's have, and tile t-1's buffer is free
        const int slot = t & 1;
        if (t + 1 < ntiles) issue_dma(kb + 64, (t + 1) & 1);
        const char* sK = smem + slot * BUF;
        int va[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) va[dt] = vaddr[dt] + slot * vstep[dt];
        const bool tail = (kb + 64 > p.N);

      if constexpr (QS == 2 && (D == 72 || D == 64)) {
        // ---- two query sets, software-pipelined by hand: the matrix work of one set is issued between the softmax VALU of the other ----
        // items (set, 32-key sub-tile) in the order (A,0) (B,0) (A,1) (B,1); step i runs the softmax of item i on the VALU while the
        // matrix pipe takes QK^T of item i+1 and PV of item i-1.  K fragments of a sub-tile are read ONCE for both sets, so are V's.
        // Source order = the intended issue order (a matrix instruction leads every chunk of two probabilities); the compiler's own placement inside
        // a basic block measured better than sched_group_barrier patterns (-2 %), so only two things are pinned: the exponentials stay in their
        // step's block (LLVM would sink them to their users in the next one), and the rare rescale is out of line.
        constexpr int P1 = VIDI_ATTN_RM_PRE1, P2 = VIDI_ATTN_RM_PRE2;
        u32x4 kf[KS], vf[DT][2], pa0, pa1, pb0, pb1;
        f32x16 SA, SB;
        if constexpr (kCinMask) {
            if (tail) {                                  // the last tile: -inf under the keys past N (the matrix pipe adds the scores to it)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cin[u][r] = (kb + u * 32 + krow32(r, hi) >= p.N) ? -INFINITY : 0.f;
            }
        }
        load_kf(0, kf, sK, kb);
#pragma unroll
        for (int s = 0; s < KS; ++s) QKm(s, kf, 0, SA, 0);
        // step 1: softmax (A,0)  ||  QK (B,0)  [KS]
        { auto M = [&](int i) __attribute__((always_inline)) { if (i < KS) QKm(i, kf, 1, SB, 0); };
          sm_a(0, SA, M, IntC<P1>{});
          sm_b(0, SA, pa0, pa1, [&](int c) __attribute__((always_inline)) { M(c + P1); if (c + P1 == KS - 1) load_kf(1, kf, sK, kb); if (c == VIDI_ATTN_RM_VF0) load_vf(0, vf, va); }); }
        // step 2: softmax (B,0)  ||  PV (A,0) [2 DT], QK (A,1) [KS]
        vf_ready(vf);
        { auto M = [&](int i) __attribute__((always_inline)) { if (i < 2 * DT) PVm(i, vf, 0, pa0, pa1); else if (i < 2 * DT + KS) QKm(i - 2 * DT, kf, 0, SA, 1); };
          sm_a(1, SB, M, IntC<P2>{});
          sm_b(1, SB, pb0, pb1, [&](int c) __attribute__((always_inline)) { M(c + P2); }); }
        // step 3: softmax (A,1)  ||  PV (B,0), QK (B,1); V fragments of sub-tile 1 replace sub-tile 0's once its last PV has been issued
        { auto M = [&](int i) __attribute__((always_inline)) { if (i < 2 * DT) PVm(i, vf, 1, pb0, pb1); else if (i < 2 * DT + KS) QKm(i - 2 * DT, kf, 1, SB, 1); };
          sm_a(0, SA, M, IntC<P2>{});
          sm_b(0, SA, pa0, pa1, [&](int c) __attribute__((always_inline)) { M(c + P2); if (c + P2 == 2 * DT - 1) load_vf(1, vf, va); }); }
        // step 4: softmax (B,1)  ||  PV (A,1)
        vf_ready(vf);
        { auto M = [&](int i) __attribute__((always_inline)) { if (i < 2 * DT) PVm(i, vf, 0, pa0, pa1); };
          sm_a(1, SB, M, IntC<P1>{});
          sm_b(1, SB, pb0, pb1, [&](int c) __attribute__((always_inline)) { M(c + P1); }); }
        // step 5: PV (B,1)
#pragma unroll
        for (int j = 0; j < 2 * DT; ++j) PVm(j, vf, 1, pb0, pb1);
      } else
#pragma unroll
      for (int qs = 0; qs < QS; ++qs) {
        f32x16 s2[2];
        VIDI_ATTN_RM_PRIO_HI(1);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x4 kf = *(const u32x4*)(sK + (u * 32 + l31) * ROWB + (((2 * s + hi) ^ ksw) << 4));
                s2[u] = T::mfma32(kf, qf[qs][s], s == 0 ? zero16 : s2[u]);
            }
        VIDI_ATTN_RM_PRIO_LO(1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // (the asm reads are invisible to the compiler's lgkmcnt accounting: a K fragment consumed after them would wait for them too)
            if constexpr (VIDI_ATTN_RM_TRASM != 0) { if (u == 0) __builtin_amdgcn_sched_barrier(0); }
            // A fragments of the PV MFMAs: contraction slots 8 hi + {0..3 | 4..7} of MFMA 0 hold keys 4 hi + {0..3} and 8 + 4 hi + {0..3}
            // of the sub-tile (where the swapped QK^T left them in pv[0..7]); MFMA 1 the same 16 keys further
            u32x4 vf[DT][2];
            u32x2 vraw[12];
            static_assert(DT * 4 <= 12, "raw halves");
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int rowb = dt * 32 < MAINC ? MROWB : 16;
                    if constexpr (VIDI_ATTN_RM_TRASM != 0) {
                        const unsigned a = (unsigned)(uintptr_t)(lds0 + va[dt]);
                        VIDI_TR_READ(vraw[(dt * 2 + m) * 2], a, (u * 32 + 16 * m) * rowb);
                        VIDI_TR_READ(vraw[(dt * 2 + m) * 2 + 1], a, (u * 32 + 16 * m + 8) * rowb);
                    } else {
                        const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(lds0 + va[dt] + (u * 32 + 16 * m) * rowb));
                        const v4s16 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(lds0 + va[dt] + (u * 32 + 16 * m + 8) * rowb));
                        const u32x2 a = __builtin_bit_cast(u32x2, lo), c = __builtin_bit_cast(u32x2, up);
                        vf[dt][m] = u32x4{a[0], a[1], c[0], c[1]};
                    }
                }
            if (tail) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb + u * 32 + krow32(r, hi) >= p.N) s2[u][r] = -INFINITY;
            }
            // online softmax on raw scores (scale folded into the exponent), lazy rescaling — see attn_self.hip
            float mx = s2[u][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s2[u][r]);
            mx = xhalf_max(mx);
            constexpr float TAU = 8.0f;
            float mxs = mx * sc;
            if (__builtin_expect(__any(mxs > thr_run[qs]), 0)) {      // also the very first sub-tile (m = -inf)
                asm volatile("" : "+v"(mxs));                      // (keeps the rare path's arithmetic inside the branch: LLVM speculates it above)
                const float m_cand = fmaxf(-nm_run[qs], mxs);
                const float alpha = fast_exp2(-nm_run[qs] - m_cand); // 0 on the first sub-tile (o = l = 0 there)
                nm_run[qs] = -m_cand; thr_run[qs] = m_cand + TAU;
                l_run[qs] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[qs][dt][i] *= alpha;
            }
            float ps0 = 0.f, ps1 = 0.f;
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                pv[r] = fast_exp2(__builtin_fmaf(s2[u][r], sc, nm_run[qs]));      // (as packed pairs, v_pk_fma_f32: measured 2 % slower)
                pv[r + 1] = fast_exp2(__builtin_fmaf(s2[u][r + 1], sc, nm_run[qs]));
                if constexpr (!kOnesRow) { ps0 += pv[r]; ps1 += pv[r + 1]; }
            }
            const u32x4 pf0 = pack8<T>(pv), pf1 = pack8<T>(pv + 8);
            if constexpr (!kOnesRow) l_run[qs] += ps0 + ps1;
            if constexpr (VIDI_ATTN_RM_TRASM != 0) {                 // the V fragments have had the whole softmax to arrive
                tr_wait<DT * 4>(vraw);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const u32x2 a = vraw[(dt * 2 + m) * 2], c = vraw[(dt * 2 + m) * 2 + 1];
                        vf[dt][m] = u32x4{a[0], a[1], c[0], c[1]};
                    }
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[qs][dt] = T::mfma32(vf[dt][0], pf0, o[qs][dt]);
                o[qs][dt] = T::mfma32(vf[dt][1], pf1, o[qs][dt]);
            }
        }
      }
    }

#pragma unroll
    for (int qs = 0; qs < QS; ++qs) {
        float l_tot;
        if constexpr (kOnesRow) {
            // output row D of the last d-tile: register (D%32 -> j = row/8, e = row%4) of the lanes with hi == (row/4)%2
            constexpr int REG = 4 * (ROWD / 8) + (ROWD % 4), HI = (ROWD / 4) % 2;
            const float mine = (hi == HI) ? o[qs][DT - 1][REG] : 0.f;
            l_tot = xhalf_sum(mine);
        } else {
            l_tot = xhalf_sum(l_run[qs]);
        }
        const float inv = 1.0f / l_tot;
        // ---- epilogue: O^T registers -> LDS [128 q][D] -> row-contiguous 16-byte global stores (as attn_self.hip)
        __syncthreads();
        {
            char* so = smem + (wave * 32 + l31) * ORW;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = dt * 32 + 8 * j + 4 * hi;
                    if (d < D) {
                        const u32x2 ov = {pack2<T>(o[qs][dt][4 * j] * inv, o[qs][dt][4 * j + 1] * inv),
                                          pack2<T>(o[qs][dt][4 * j + 2] * inv, o[qs][dt][4 * j + 3] * inv)};
                        *(u32x2*)(so + d * 2) = ov;
                    }
                }
        }
        __syncthreads();
        for (int i = tid; i < 128 * NCH; i += 256) {
            const int row = i / NCH, c = i % NCH;
            const int qq = qt * QB + qs * 128 + row;
            if (qq < p.N)
                *(u32x4*)(p.O + ((size_t)b * p.N + qq) * p.ldo + h * D + c * 8) = *(const u32x4*)(smem + row * ORW + c * 16);
        }
    }
}

int vidi_attn_self_rm_dispatch(const AttnSelfRmParams& p, int D, int dtype, hipStream_t st) {
    if (p.B <= 0 || p.N <= 0 || p.H <= 0) return VIDI_ERR_SHAPE;
    if ((p.ld % 8) || (p.koff % 8) || (p.voff % 8) || (p.ldo % 8) || (p.bs % 8) || (p.hs % 8)) return VIDI_ERR_ALIGN;
    if (((uintptr_t)p.QKV & 15) || ((uintptr_t)p.O & 15)) return VIDI_ERR_ALIGN;
#define LAUNCH(TT, DD) do { constexpr int QB = 128 * attn_rm_qs<DD>();                                                                \
        constexpr bool CAN_PS = (DD == 72 && attn_rm_qs<DD>() == 2);                                                                      \
        const dim3 grid(((p.N + QB - 1) / QB) * p.H * p.B);                                                                               \
        if (CAN_PS && p.scale <= 0.f) hipLaunchKernelGGL((attn_self_rm_kernel<TT, DD, attn_rm_qs<DD>(), CAN_PS>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((attn_self_rm_kernel<TT, DD, attn_rm_qs<DD>(), false>), grid, dim3(256), 0, st, p); } while (0)
    if (dtype == VIDI_DT_BF16) {
        if (D == 72) LAUNCH(BF16, 72); else if (D == 64) LAUNCH(BF16, 64); else if (D == 16) LAUNCH(BF16, 16);
        else if (D == 32) LAUNCH(BF16, 32); else return VIDI_ERR_SHAPE;
    } else if (dtype == VIDI_DT_F16) {
        if (D == 72) LAUNCH(F16, 72); else if (D == 64) LAUNCH(F16, 64); else if (D == 16) LAUNCH(F16, 16);
        else if (D == 32) LAUNCH(F16, 32); else return VIDI_ERR_SHAPE;
    } else {
        return VIDI_ERR_DTYPE;
    }
#undef LAUNCH
    return (int)hipGetLastError();
}
