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



template <int MI> struct AccOf { typedef f32x16 type; };
template <> struct AccOf<16> { typedef f32x4 type; };

// MI selects the matrix instruction: 32 -> v_mfma_f32_32x32x16 (k16 steps), 16 -> v_mfma_f32_16x16x32 (k32 steps).
// Both run at the same peak rate, but the 16x16x32 form moves half the accumulator registers per FLOP and was
// measured to sustain ~12 % more under the chip's power limit on random data (tools/micro/mfma_power.hip).
template <typename T, int BN, int BM, int WN, int WM, int STAGES, int MODE, bool REPKV, int BK = 64, int STAG = 0, int MI = 32>
__global__ __launch_bounds__(WN* WM * 64) void gemm_kernel(GemmParams p) {
    constexpr int NT = WN * WM * 64;
    constexpr int TN = BN / WN / MI, TM = BM / WM / MI;
    static_assert(MI == 32 || (MI == 16 && TM <= TN && (((STAG == 0 || STAG == 6 || STAG == 7 || STAG == 10 || (STAG >= 16 && STAG <= 19)) && BK == 64) || (STAG == 9 && BK == 32))), "MI");
    // STAG 16..19 = schedule 6 with a cache policy on the DMA loads (experiments): nt / sc0 / sc1 / sc0+nt
    constexpr bool kS6 = (STAG == 6) || (STAG >= 16 && STAG <= 19);
    constexpr int AUX = (STAG == 16) ? 2 : (STAG == 17) ? 1 : (STAG == 18) ? 16 : (STAG == 19) ? 3 : 0;
    // MI == 16 schedules: STAG 0 = DMA issued right after the barrier; 6 = after the first fragment reads;
    // 7 = one DMA piece after each of the first W_LOADS + X_LOADS row groups of MFMAs
    constexpr bool kLate16 = (MI == 16) && (kS6 || STAG == 7);
    constexpr int CPR = BK / 8;                       // 16-byte chunks per LDS row (8 for BK=64, 4 for BK=32)
    constexpr int CSH = (CPR == 8) ? 3 : 2;
    constexpr int ROWB = BK * 2;                      // LDS row bytes
    constexpr int SWSH = (CPR == 8) ? 1 : 2;          // swizzle = (row >> SWSH) & (CPR-1): conflict-free ds_read_b128
    // 16-byte chunk c of LDS row r lives at chunk c ^ swz(r); chosen so that every ds_read_b128 lane group of the
    // fragment reads hits 64 distinct banks (32-row x 2-chunk lanes for MI=32, 16-row x 4-chunk lanes for MI=16)
    auto swz = [](int row) {
        if constexpr (CPR == 8) return (row >> 1) & 7;
        else if constexpr (MI == 32) return (row >> 2) & 3;
        else return (4 - ((row >> 2) & 3)) & 3;
    };
    constexpr int W_LOADS = BN * CPR / NT, X_LOADS = BM * CPR / NT;
    constexpr int LPT = W_LOADS + X_LOADS;
    constexpr int STAGE_BYTES = (BN + BM) * ROWB;
    static_assert(BK == 64 || BK == 32, "BK");
    static_assert(BN * CPR % NT == 0 && BM * CPR % NT == 0, "tile/threads mismatch");
    static_assert(MODE != MODE_GEGLU || ((TN * MI) % 64 == 0), "GeGLU needs whole 32-gate/32-up row blocks per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WM, wm = wave % WM;

    // ---- block -> tile: XCD-contiguous remap (bijective), then grouped ordering --------------
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    // Persistent tile loop: the grid may be smaller than the tile count (one block per CU); a block walks tiles
    // vb = blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x % 8 == 0 keeps vb on the block's XCD), so the output stores of
    // one tile drain while the next tile's first K slices are already being fetched, and no block relaunch sits between tiles.
    const int total_tiles = tiles_n * tiles_m;
  for (int vb = blockIdx.x; vb < total_tiles; vb += gridDim.x) {
    int tile_n, tile_m;
    {
        const int nb = total_tiles, b = vb;
        const int q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
        const int rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int GROUP_M = p.group_m;
        const int per_group = GROUP_M * tiles_n;
        const int g = rb / per_group, first_m = g * GROUP_M;
        const int gm = min(tiles_m - first_m, GROUP_M);
        const int in_g = rb - g * per_group;
        tile_m = first_m + in_g % gm;
        tile_n = in_g / gm;
    }
    const int n0 = tile_n * BN, m0 = tile_m * BM;
    const long long bz = blockIdx.y;
    const u16* Xb = p.X + bz * p.bsX;

    // ---- per-thread DMA source pointers (k0 added per stage) ----------------------------------
    const u16* wsrc[W_LOADS];
    const u16* xsrc[X_LOADS];
    int xcol[X_LOADS];
#pragma unroll
    for (int j = 0; j < W_LOADS; ++j) {
        const int pidx = j * NT + tid, row = pidx >> CSH, cl = pidx & (CPR - 1), cg = cl ^ swz(row);
        const int n = min(n0 + row, p.N - 1);
        wsrc[j] = p.W + (size_t)n * p.ldw + cg * 8;
    }
#pragma unroll
    for (int j = 0; j < X_LOADS; ++j) {
        const int pidx = j * NT + tid, row = pidx >> CSH, cl = pidx & (CPR - 1), cg = cl ^ swz(row);
        const int m = min(m0 + row, p.M - 1);
        xsrc[j] = Xb + (size_t)m * p.ldx;
        xcol[j] = cg * 8;
        if constexpr (!REPKV) xsrc[j] += cg * 8;
    }

    // loads [jw0,jw1) of the W tile and [jx0,jx1) of the X tile of K-slice kt into ring slot `stage`
    auto load_part = [&](int stage, int kt, int jw0, int jw1, int jx0, int jx1) {
        char* sW = smem + stage * STAGE_BYTES;
        char* sX = sW + BN * ROWB;
        const int k0 = (STAG == 8) ? 0 : kt * BK;          // STAG 8 (diagnostic): every DMA re-reads K slice 0 (cache hits)
#pragma unroll
        for (int j = 0; j < W_LOADS; ++j) if (j >= jw0 && j < jw1) glds16<AUX>(wsrc[j] + k0, sW + (j * NT + wave * 64) * 16);
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) {
            if (j < jx0 || j >= jx1) continue;
            if constexpr (REPKV) {
                const int k = k0 + xcol[j];
                const int phys = (k / (p.rep_g * p.rep_hd)) * p.rep_hd + (k % p.rep_hd);
                glds16<AUX>(xsrc[j] + phys, sX + (j * NT + wave * 64) * 16);
            } else {
                glds16<AUX>(xsrc[j] + k0, sX + (j * NT + wave * 64) * 16);
            }
        }
    };
    // STAG 12 (diagnostic): after the prologue only the W tile is re-loaded (half the DMA traffic; wrong results)
    auto load_stage = [&](int stage, int kt) { load_part(stage, kt, 0, W_LOADS, 0, (STAG == 12 && kt > 0) ? 0 : X_LOADS); };
    // STAG: the two waves that share a SIMD (w and w + NW/2) issue their DMA at different k16-steps, so one
    // wave's load-issue time overlaps the other's MFMAs instead of both stalling the matrix pipe in lockstep.
    constexpr int NS = BK / 16;
    // STAG == 7: DMA pieces are issued one at a time BETWEEN MFMAs (2 per k16-step), away from the ds_read burst
    constexpr int NPIECE = W_LOADS + X_LOADS, NMF = TN * TM;
    constexpr bool kInter = (STAG == 7) && (NS == 4) && (NPIECE % NS == 0) && ((NMF * NS) % NPIECE == 0);
    constexpr int GAP = kInter ? (NMF * NS) / NPIECE : 1;            // MFMAs between two DMA pieces
    constexpr bool kStag = (STAG != 0) && (STAG != 7) && (NS == 4) && (W_LOADS % 2 == 0) && (X_LOADS % 2 == 0) && (WN * WM == 8);
    const int grp = (wave >= (WN * WM) / 2) ? 1 : 0;

    typename AccOf<MI>::type acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int i = 0; i < (MI == 32 ? 16 : 4); ++i) acc[a][b][i] = 0.f;

    // lane -> (row inside an MI-row tile, 8-element k chunk): 32x32x16 = 32 rows x 2 chunks, 16x16x32 = 16 rows x 4 chunks
    const int l31 = (MI == 32) ? (lane & 31) : (lane & 15), hi = (MI == 32) ? (lane >> 5) : (lane >> 4);
    const int sw = swz(l31);                               // row swizzle (tile bases are multiples of 16)
    const int w_row_off = (wn * TN * MI + l31) * ROWB;
    const int x_row_off = (wm * TM * MI + l31) * ROWB;

    const int nk = p.K / BK;
  if constexpr (STAG == 9) {
    // ================= ping-pong schedule (8 waves, BK = 32 micro-tiles, R-deep ring) =================
    // The two waves that share a SIMD (w and w+4) alternate roles every barrier interval ("slot"):
    //   slot 2j   : group A (waves 0-3) runs L(j)  = ds_read the fragments of micro-tile j + issue its own DMA pieces
    //               group B (waves 4-7) runs M(j-1) = 16 MFMAs from registers
    //   slot 2j+1 : A runs M(j), B runs L(j)
    // so every SIMD's matrix pipe is fed back-to-back by one wave while its partner does all LDS/DMA work.
    // Micro-tile j is read by A in slot 2j and by B in slot 2j+1; its ring slot is refilled (micro-tile j+R)
    // by A in slot 2j+2 and by B in slot 2j+3, i.e. 2R-2 / 2R-3 slots before the first reader: loads are never
    // drained (counted vmcnt) and have ~3 K-steps of lead time.
    static_assert(BK == 32 && WN * WM == 8 && STAGES >= 3 && STAGES <= 5, "ping-pong geometry");
    constexpr int R = STAGES;
    auto wait_rem = [&](int rem) {            // allow `rem` younger batches (LPT DMA instructions each) in flight
        if (rem >= 3) wait_vmcnt<3 * LPT>(); else if (rem == 2) wait_vmcnt<2 * LPT>(); else if (rem == 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
    };
    constexpr int KST = (MI == 32) ? 2 : 1;                // MFMA k-steps per 32-wide micro-tile
    constexpr int CPS = (MI == 32) ? 2 : 4;                // 16-byte chunks per k-step
    u32x4 wf[KST][TN], xf[KST][TM];
    auto Lseg = [&](int j) {
        const char* sW = smem + (j % R) * STAGE_BYTES;
        const char* sX = sW + BN * ROWB;
#pragma unroll
        for (int st = 0; st < KST; ++st) {
            const int coff = ((CPS * st + hi) ^ sw) << 4;
#pragma unroll
            for (int a2 = 0; a2 < TN; ++a2) wf[st][a2] = *(const u32x4*)(sW + w_row_off + a2 * MI * ROWB + coff);
#pragma unroll
            for (int b2 = 0; b2 < TM; ++b2) xf[st][b2] = *(const u32x4*)(sX + x_row_off + b2 * MI * ROWB + coff);
        }
        if (j + R - 1 < nk) load_stage((j + R - 1) % R, j + R - 1);      // refill the slot freed by micro-tile j-1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // my LDS reads are done before I release the slot
    };
    auto Mseg = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int st = 0; st < KST; ++st)
#pragma unroll
            for (int a2 = 0; a2 < TN; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < TM; ++b2) {
                    if constexpr (MI == 32) acc[a2][b2] = T::mfma32(wf[st][a2], xf[st][b2], acc[a2][b2]);
                    else acc[a2][b2] = T::mfma16(wf[st][a2], xf[st][b2], acc[a2][b2]);
                }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // before the barrier that precedes A's L(j+1): this wave's pieces of micro-tile j+1 must have landed
    auto wait_next = [&](int j) { if (j + 1 < nk) wait_rem(min(R - 2, nk - 2 - j)); };
#pragma unroll
    for (int st = 0; st < R - 1; ++st)
        if (st < nk) load_stage(st, st);
    wait_rem(min(R - 2, nk - 1));                                          // micro-tile 0 landed
    if (grp == 0) {
        for (int j = 0; j < nk; ++j) { bar(); Lseg(j); bar(); Mseg(); wait_next(j); }
        bar();
    } else {
        bar();
        for (int j = 0; j < nk; ++j) { bar(); Lseg(j); wait_next(j); bar(); Mseg(); }
    }
  } else if constexpr (STAG == 15) {
    // ---- experiment: register-staged double buffer (global_load -> VGPR early, ds_write late) instead of LDS-DMA ----
    static_assert(!REPKV && STAGES == 2, "regstage experiment");
    u32x4 wreg[W_LOADS], xreg[X_LOADS];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < W_LOADS; ++j) wreg[j] = *(const u32x4*)(wsrc[j] + k0);
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) xreg[j] = *(const u32x4*)(xsrc[j] + k0);
    };
    auto lwrite = [&](int stage) {
        char* sW = smem + stage * STAGE_BYTES;
        char* sX = sW + BN * ROWB;
#pragma unroll
        for (int j = 0; j < W_LOADS; ++j) *(u32x4*)(sW + (j * NT + tid) * 16) = wreg[j];
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) *(u32x4*)(sX + (j * NT + tid) * 16) = xreg[j];
    };
    gload(0);
    lwrite(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const char* sW = smem + (kt & 1) * STAGE_BYTES;
        const char* sX = sW + BN * ROWB;
        u32x4 wf[2][TN], xf[2][TM];
        auto read_frags = [&](int buf, int st) {
            const int coff = ((2 * st + hi) ^ sw) << 4;
#pragma unroll
            for (int a2 = 0; a2 < TN; ++a2) wf[buf][a2] = *(const u32x4*)(sW + w_row_off + a2 * 32 * ROWB + coff);
#pragma unroll
            for (int b2 = 0; b2 < TM; ++b2) xf[buf][b2] = *(const u32x4*)(sX + x_row_off + b2 * 32 * ROWB + coff);
        };
        read_frags(0, 0);
#pragma unroll
        for (int st = 0; st < BK / 16; ++st) {
            if (st < BK / 16 - 1) read_frags((st + 1) & 1, st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a2 = 0; a2 < TN; ++a2)
#pragma unroll
                for (int b2 = 0; b2 < TM; ++b2) acc[a2][b2] = T::mfma32(wf[st & 1][a2], xf[st & 1][b2], acc[a2][b2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 < nk) lwrite((kt + 1) & 1);          // slot (kt+1)&1 was last read in iteration kt-1
        __syncthreads();
    }
  } else if constexpr (MI == 16 && STAG == 10) {
    // ============ mid-iteration barrier: the fragment pipeline never drains (2-deep ring, BK = 64) ============
    // Iteration kt = two k32 steps on stage kt.  All LDS reads of stage kt are issued during step 0 (its own
    // step-1 fragments, refilled in place); the wait + barrier sit BETWEEN the steps, where every wave already
    // holds the operands of its next 32 MFMAs.  After the barrier buffer kt%2 is free (DMA of stage kt+2 goes
    // there) and stage kt+1 has landed, so step 1 refills its registers with the step-0 fragments of stage kt+1:
    // no MFMA ever waits for a top-of-iteration LDS round trip.
    static_assert(BK == 64 && STAGES == 2, "mid-barrier schedule geometry");
    load_stage(0, 0);
    if (nk > 1) { load_stage(1, 1); wait_vmcnt<LPT>(); } else { wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    u32x4 wf[TN], xf[2][TM];
    auto rdW = [&](const char* sW, int a, int s) { return *(const u32x4*)(sW + w_row_off + a * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
    auto rdX = [&](const char* sX, int b, int s) { return *(const u32x4*)(sX + x_row_off + b * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
    {
        const char* sW = smem;
        const char* sX = sW + BN * ROWB;
#pragma unroll
        for (int b = 0; b < TM; ++b) xf[0][b] = rdX(sX, b, 0);
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = rdW(sW, a, 0);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const char* sW = smem + (kt & 1) * STAGE_BYTES;
        const char* sX = sW + BN * ROWB;
        // ---- step 0: MFMAs on the resident fragments; refill with this stage's step-1 fragments
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = T::mfma16(wf[a], xf[0][b], acc[a][b]);
            wf[a] = rdW(sW, a, 1);
            if (a < TM) xf[1][a] = rdX(sX, a, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- mid point: my reads of stage kt are complete, stage kt+1 has landed (all waves: barrier)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (p.diag != 4 && kt + 2 < nk) load_stage(kt & 1, kt + 2);
        // ---- step 1: MFMAs; refill with the next stage's step-0 fragments
        const bool more = (kt + 1 < nk);
        const char* nW = smem + ((kt + 1) & 1) * STAGE_BYTES;
        const char* nX = nW + BN * ROWB;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = T::mfma16(wf[a], xf[1][b], acc[a][b]);
            if (more) {
                wf[a] = rdW(nW, a, 0);
                if (a < TM) xf[0][a] = rdX(nX, a, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  #pragma unroll
      for (int s = 0; s < STAGES - 1; ++s)
          if (s < nk) load_stage(s, s);

      for (int kt = 0; kt < nk; ++kt) {
          const int rem = min(STAGES - 2, nk - 1 - kt);
          // tile kt must have landed; up to `rem` younger tiles may stay in flight (loads return in order)
          if constexpr (STAGES == 2) {
              wait_vmcnt<0>();
          } else if constexpr (STAGES == 3) {
              if (rem >= 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
          } else if constexpr (STAGES == 4) {
              if (rem >= 2) wait_vmcnt<2 * LPT>(); else if (rem == 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
          } else {
              if (rem >= 3) wait_vmcnt<3 * LPT>(); else if (rem == 2) wait_vmcnt<2 * LPT>(); else if (rem == 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();
          }
          if constexpr (STAG != 4) __builtin_amdgcn_s_barrier();
          const bool do_load = (STAG != 3 && STAG != 4) && (p.diag != 4) && (kt + STAGES - 1 < nk);   // diag 4: timing without DMA (wrong results)
          const int lstage = (kt + STAGES - 1) % STAGES, lkt = kt + STAGES - 1;
          if constexpr (!kStag && !kInter && !kLate16) { if (do_load) load_stage(lstage, lkt); }

          const char* sW = smem + (kt % STAGES) * STAGE_BYTES;
          const char* sX = sW + BN * ROWB;
          if constexpr (MI == 16) {
              // k32 steps; W fragments are refilled in place right after their last MFMA of the step, X fragments
              // are double-buffered (register budget: 2 waves/SIMD = 256 VGPRs, 128 of them accumulators)
              constexpr int NS16 = BK / 32;
              u32x4 wf[TN], xf[2][TM];
              auto rdW = [&](int a, int s) { return *(const u32x4*)(sW + w_row_off + a * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
              auto rdX = [&](int b, int s) { return *(const u32x4*)(sX + x_row_off + b * 16 * ROWB + (((4 * s + hi) ^ sw) << 4)); };
  #pragma unroll
              for (int b = 0; b < TM; ++b) xf[0][b] = rdX(b, 0);
  #pragma unroll
              for (int a = 0; a < TN; ++a) wf[a] = rdW(a, 0);
              if constexpr (kS6) {
                  __builtin_amdgcn_sched_barrier(0);
                  if (do_load) load_stage(lstage, lkt);
              }
  #pragma unroll
              for (int s = 0; s < NS16; ++s) {
  #pragma unroll
                  for (int a = 0; a < TN; ++a) {
                      __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
                      for (int b = 0; b < TM; ++b) acc[a][b] = T::mfma16(wf[a], xf[s & 1][b], acc[a][b]);
                      if (s + 1 < NS16) {
                          wf[a] = rdW(a, s + 1);
                          if (a < TM) xf[(s + 1) & 1][a] = rdX(a, s + 1);
                      }
                      if constexpr (STAG == 7) {
                          constexpr int PPG = (W_LOADS + X_LOADS + TN - 1) / TN;      // pieces per row group
                          if (s == 0 && do_load) {
  #pragma unroll
                              for (int q = a * PPG; q < (a + 1) * PPG && q < W_LOADS + X_LOADS; ++q) {
                                  if (q < W_LOADS) load_part(lstage, lkt, q, q + 1, 0, 0);
                                  else load_part(lstage, lkt, 0, 0, q - W_LOADS, q - W_LOADS + 1);
                              }
                          }
                      }
                  }
              }
              __builtin_amdgcn_sched_barrier(0);
          } else {
          // fragments of k16-step s+1 are read before the MFMAs of step s (software pipeline in regs)
          u32x4 wf[2][TN], xf[2][TM];
          auto read_frags = [&](int buf, int s) {
              const int coff = ((2 * s + hi) ^ sw) << 4;
  #pragma unroll
              for (int a = 0; a < TN; ++a) wf[buf][a] = *(const u32x4*)(sW + w_row_off + a * 32 * ROWB + coff);
  #pragma unroll
              for (int b = 0; b < TM; ++b) xf[buf][b] = *(const u32x4*)(sX + x_row_off + b * 32 * ROWB + coff);
          };
          read_frags(0, 0);
  #pragma unroll
          for (int s = 0; s < BK / 16; ++s) {
              if (s < BK / 16 - 1) read_frags((s + 1) & 1, s + 1);
              if constexpr (kStag) {
                  // group 0 issues its halves at steps 0,1 ; group 1 at steps 2,3
                  if (do_load && (s >> 1) == grp) {
                      const int h = s & 1;
                      load_part(lstage, lkt, h * (W_LOADS / 2), (h + 1) * (W_LOADS / 2), h * (X_LOADS / 2), (h + 1) * (X_LOADS / 2));
                  }
              }
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (STAG == 2) __builtin_amdgcn_s_setprio(1);
              if constexpr (kInter) {
  #pragma unroll
                  for (int i = 0; i < NMF; ++i) {
                      const int a = i / TM, b = i % TM;
                      acc[a][b] = T::mfma32(wf[s & 1][a], xf[s & 1][b], acc[a][b]);
                      if (i % GAP == (GAP > 1 ? 1 : 0)) {
                          __builtin_amdgcn_sched_barrier(0);
                          const int q = s * (NPIECE / NS) + i / GAP;              // piece index: W loads first, then X loads
                          if (do_load) {
                              if (q < W_LOADS) load_part(lstage, lkt, q, q + 1, 0, 0);
                              else load_part(lstage, lkt, 0, 0, q - W_LOADS, q - W_LOADS + 1);
                          }
                          __builtin_amdgcn_sched_barrier(0);
                      }
                  }
              } else {
  #pragma unroll
                  for (int a = 0; a < TN; ++a)
  #pragma unroll
                      for (int b = 0; b < TM; ++b) acc[a][b] = T::mfma32(wf[s & 1][a], xf[s & 1][b], acc[a][b]);
              }
              if constexpr (STAG == 2) __builtin_amdgcn_s_setprio(0);
              __builtin_amdgcn_sched_barrier(0);
          }
          }   // MI == 32
      }

  }

    // ---- epilogue -----------------------------------------------------------------------------
    // Row-major outputs are staged through LDS (the stage ring is free now) and leave as whole-row
    // 16-byte stores: per-lane 8-byte stores straight from the MFMA layout touch 32 rows per
    // instruction and were measured to cost ~20 K-iterations per 256x256 tile.  Transposed /
    // scattered destinations (the Vt images) still go straight from registers.
    u16* Yb = p.Y + bz * p.bsY;
    const u16* Rb = p.R ? p.R + bz * p.bsR : nullptr;
    const bool act_tanh = (p.act == ACT_GELU_TANH), act_erf = (p.act == ACT_GELU_ERF);
    const bool glu_silu = (p.act == ACT_SILU);                     // MODE_GEGLU: SiLU-GLU (Mistral) instead of GELU(tanh)-GLU (Gemma2)
    constexpr int BNO = (MODE == MODE_GEGLU) ? BN / 2 : BN;       // output columns of this tile
    constexpr int CROW = BNO * 2 + 16;                            // padded LDS row (bytes)
    __syncthreads();                                              // every wave is done with the last K slice
    if (p.diag == 2) continue;                                     // timing diagnostic: no epilogue at all
    char* sC = smem;
    // copy-out geometry (chunk i = it*NT + tid -> row i / CH, 16-byte chunk i % CH of the staged tile)
    constexpr int CH = BNO / 8, ITERS = BM * CH / NT, UNR = 4;
    static_assert((BM * CH) % NT == 0 && ITERS % UNR == 0, "copy-out geometry");
    const int no0 = (MODE == MODE_GEGLU) ? (n0 >> 1) : n0;
    const int Nout = (MODE == MODE_GEGLU) ? (p.N >> 1) : p.N;
    // the residual tile is fetched NOW, so its HBM latency hides under the staging pass below instead of
    // stalling the copy-out (the accumulators are still live: 128 + 4*ITERS registers)
    u32x4 res[ITERS];
    if constexpr (MODE != MODE_KV_CACHE) {
        if (Rb) {
            const bool wrap = p.rmod < p.M;                       // residual rows repeat every rmod rows (position tables)
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int i = it * NT + tid;
                const int m = min(m0 + i / CH, p.M - 1), n = min(no0 + (i % CH) * 8, Nout - 8);
                const int mr = wrap ? m % p.rmod : m;
                res[it] = *(const u32x4*)(Rb + (size_t)mr * p.ldr + n);
            }
        }
    }
    constexpr int NQ = (MI == 32) ? 4 : 1;                        // 4-row quads per lane per tile (32x32: rows 8j+4hi+e; 16x16: 4hi+e)
    // this lane's bias quads (one 8-byte load per n-quad, shared by all its m-tiles)
    const bool has_bias = (MODE != MODE_GEGLU) && (p.bias != nullptr);
    u32x2 bq[TN][NQ];
    if constexpr (MODE != MODE_GEGLU) {
        if (has_bias) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int n = n0 + wn * TN * MI + a * MI + 8 * j + 4 * hi;
                    bq[a][j] = *(const u32x2*)(p.bias + min(n, p.N - 4));
                }
        }
    }
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int ml = wm * TM * MI + b * MI + l31;               // row inside the tile
        const int m = m0 + ml;
        if constexpr (MODE == MODE_GEGLU) {
            // W rows are interleaved in blocks of 32 gate / 32 up rows: pair every gate quad with its up quad
            constexpr int TPB = 32 / MI;                          // MI-row tiles per 32-row block
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                if ((a / TPB) % 2) continue;                      // up tiles are consumed with their gate tile
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int nol = ((wn * TN * MI) >> 1) + (a / (2 * TPB)) * 32 + (a % TPB) * MI + 8 * j + 4 * hi;   // output column inside the tile
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float g = rnd<T>(acc[a][b][4 * j + e]);
                        const float u = rnd<T>(acc[a + TPB][b][4 * j + e]);
                        v[e] = rnd<T>(glu_silu ? silu_f(g) : gelu_tanh_f(g)) * u;
                    }
                    const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                    *(u32x2*)(sC + ml * CROW + nol * 2) = o;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int nl = wn * TN * MI + a * MI + 8 * j + 4 * hi;
                    const int n = n0 + nl;
                    if (n >= p.N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * j + e];
                    if (has_bias) {
                        const u32x2 bv = bq[a][j];
                        v[0] += T::to_f32((u16)(bv[0] & 0xffff)); v[1] += T::to_f32((u16)(bv[0] >> 16));
                        v[2] += T::to_f32((u16)(bv[1] & 0xffff)); v[3] += T::to_f32((u16)(bv[1] >> 16));
                    }
                    // (the activation runs in the copy-out pass on the T-rounded staged values — same arithmetic,
                    //  but one small loop body instead of TN*TM*NQ unrolled copies that overflowed the I-cache)
                    const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                    bool staged = true;
                    if constexpr (MODE == MODE_QKV_VT) {
                        if (n >= p.vstart) {
                            staged = false;
                            if (m < p.M) {
                                const int c = n - p.vstart, h = c / p.hd, d = c % p.hd;
                                const int bi = m / p.seq, tok = m % p.seq;
                                const int pos = (tok & ~15) | perm16(tok & 15);
                                u16* dst = p.Vt + (((size_t)bi * p.nheads + h) * p.hd + d) * p.seqpad + pos;
                                dst[0] = (u16)(o[0] & 0xffff); dst[(size_t)p.seqpad] = (u16)(o[0] >> 16);
                                dst[2 * (size_t)p.seqpad] = (u16)(o[1] & 0xffff); dst[3 * (size_t)p.seqpad] = (u16)(o[1] >> 16);
                            }
                        }
                    } else if constexpr (MODE == MODE_KV_CACHE) {
                        if (n >= p.kvd && m < p.M) {               // V: also the transposed, perm16 tile image
                            // Vtc[kvh][tile32][hd][32 positions (perm16)]: a 32-key sub-tile is 64-byte rows back to back, so the
                            // cross-attention fetches whole 128-byte lines (a [hd][64] tile made every sub-tile fetch half-lines
                            // and the other halves were evicted before the next sub-tile came: 1.5x the HBM bytes, measured)
                            const int tok = p.tok0 + m, tile = tok >> 5, tk = tok & 31;
                            const int c = n - p.kvd, kvh = c / p.hd, d = c % p.hd;
                            const int pos = (tk & ~15) | perm16(tk & 15);
                            u16* dst = p.Vtc + (((size_t)kvh * p.ntile64 * 2 + tile) * p.hd + d) * 32 + pos;
                            dst[0] = (u16)(o[0] & 0xffff); dst[32] = (u16)(o[0] >> 16);
                            dst[64] = (u16)(o[1] & 0xffff); dst[96] = (u16)(o[1] >> 16);
                        }
                    }
                    if (staged) *(u32x2*)(sC + ml * CROW + nl * 2) = o;
                }
            }
        }
    }
    __syncthreads();
    // ---- copy-out: 16-byte chunks, consecutive lanes = consecutive chunks of one row ------------
    // UNR chunks per pass: their LDS reads are issued together, then processed
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += UNR) {
        u32x4 val[UNR];
        int mm[UNR], nn[UNR];
        bool ok[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int i = (it0 + k) * NT + tid;
            const int ml = i / CH, c = i % CH;
            mm[k] = m0 + ml; nn[k] = no0 + c * 8;
            ok[k] = (mm[k] < p.M) && (nn[k] < Nout);
            if constexpr (MODE == MODE_QKV_VT) ok[k] = ok[k] && (nn[k] < p.vstart);
            val[k] = *(const u32x4*)(sC + ml * CROW + c * 16);
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            if (!ok[k]) continue;
            const int m = mm[k], n = nn[k];
            if constexpr (MODE == MODE_KV_CACHE) {
                if (n < p.kvd) {
                    const int tok = p.tok0 + m, kvh = n / p.hd, d = n % p.hd;
                    *(u32x4*)(p.Kc + ((size_t)kvh * p.ntile64 * 64 + tok) * p.hd + d) = val[k];
                } else {
                    *(u32x4*)(p.Vrow + (size_t)m * p.kvd + (n - p.kvd)) = val[k];
                }
            } else {
                u32x4 v = val[k];
                if (act_tanh) {
                    float x[8];
                    unpack8<T>(v, x);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = gelu_tanh_f(x[e]);
                    v = pack8<T>(x);
                } else if (act_erf) {
                    float x[8];
                    unpack8<T>(v, x);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = gelu_erf_f(x[e]);
                    v = pack8<T>(x);
                }
                if (Rb) {
                    float x[8], r[8];
                    unpack8<T>(v, x);
                    unpack8<T>(res[it0 + k], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += r[e];          // x is already T-rounded; the sum rounds on pack
                    v = pack8<T>(x);
                }
                *(u32x4*)(Yb + (size_t)m * p.ldy + n) = v;
            }
        }
    }
    __syncthreads();                                              // the staging tile / ring is reused by the next tile
  }
}

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

// ---- skinny GEMM (M <= 8): HBM-bound weight streaming for decode -------------------------------
//   Y[m][n] = sum_k X[m][k] W[n][k].  Every wave owns RPW weight rows per step and streams them
//   with 16-byte non-temporal loads straight into VGPRs (no LDS round trip: each weight byte is
//   used once); the few activation rows are re-read from L1/L2.  Wave-reduce at the end.
//   Algorithmic bytes = N*K*2 (weights); roofline = HBM.
template <typename T, int MMAX, int RPW>
__global__ __launch_bounds__(256) void gemv_kernel(const u16* __restrict__ X, const u16* __restrict__ W, u16* __restrict__ Y,
                                                   int M, int N, int K, int ldx, int ldw, int ldy) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = K / 8;                          // 16-byte chunks per row
    const int waves_total = gridDim.x * 4;
    for (int nb = (blockIdx.x * 4 + wave) * RPW; nb < N; nb += waves_total * RPW) {
        float acc[RPW][MMAX];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MMAX; ++m) acc[r][m] = 0.f;
        for (int c = lane; c < nchunk; c += 64) {
            u32x4 wv[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int n = min(nb + r, N - 1);
                wv[r] = __builtin_nontemporal_load((const u32x4*)(W + (size_t)n * ldw + c * 8));
            }
            float xf[MMAX][8];
#pragma unroll
            for (int m = 0; m < MMAX; ++m) {
                const int mm = min(m, M - 1);
                const u32x4 xv = *(const u32x4*)(X + (size_t)mm * ldx + c * 8);
                unpack8<T>(xv, xf[m]);
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float wf[8];
                unpack8<T>(wv[r], wf);
#pragma unroll
                for (int m = 0; m < MMAX; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[r][m] = fmaf(wf[e], xf[m][e], acc[r][m]);
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MMAX; ++m) {
                const float s = wave_sum(acc[r][m]);
                if (lane == 0 && nb + r < N && m < M) Y[(size_t)m * ldy + nb + r] = T::from_f32(s);
            }
    }
}

// ---- skinny gated-MLP front half: gate/up GEMV + act(gate) * up in one launch ------------------------------------------
//   W: [2I, K] with gate/up rows interleaved in blocks of 32 (the MODE_GEGLU weight layout); a wave owns RPW features i and
//   streams their gate row (i/32)*64 + i%32 and up row (+32).  Y[m][i] = T( T(act(T(g))) * T(u) ) — the values of
//   gemv_kernel followed by geglu_unpack_kernel, bit for bit (same per-lane accumulation order), one launch and no [M, 2I] round trip.
template <typename T, int MMAX, int RPW>
__global__ __launch_bounds__(256) void gemv_glu_kernel(const u16* __restrict__ X, const u16* __restrict__ W, u16* __restrict__ Y,
                                                       int M, int I, int K, int ldx, int ldw, int ldy, int silu) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = K / 8;
    const int waves_total = gridDim.x * 4;
    for (int ib = (blockIdx.x * 4 + wave) * RPW; ib < I; ib += waves_total * RPW) {
        float ag[RPW][MMAX], au[RPW][MMAX];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MMAX; ++m) { ag[r][m] = 0.f; au[r][m] = 0.f; }
        for (int c = lane; c < nchunk; c += 64) {
            u32x4 wg[RPW], wu[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int i = min(ib + r, I - 1);
                const size_t grow = (size_t)(i >> 5) * 64 + (i & 31);
                wg[r] = __builtin_nontemporal_load((const u32x4*)(W + grow * ldw + c * 8));
                wu[r] = __builtin_nontemporal_load((const u32x4*)(W + (grow + 32) * ldw + c * 8));
            }
            float xf[MMAX][8];
#pragma unroll
            for (int m = 0; m < MMAX; ++m) {
                const int mm = min(m, M - 1);
                unpack8<T>(*(const u32x4*)(X + (size_t)mm * ldx + c * 8), xf[m]);
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float gf[8], uf[8];
                unpack8<T>(wg[r], gf);
                unpack8<T>(wu[r], uf);
#pragma unroll
                for (int m = 0; m < MMAX; ++m) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ag[r][m] = fmaf(gf[e], xf[m][e], ag[r][m]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) au[r][m] = fmaf(uf[e], xf[m][e], au[r][m]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int m = 0; m < MMAX; ++m) {
                const float g = rnd<T>(wave_sum(ag[r][m])), u = rnd<T>(wave_sum(au[r][m]));
                if (lane == 0 && ib + r < I && m < M) Y[(size_t)m * ldy + ib + r] = T::from_f32(rnd<T>(silu ? silu_f(g) : gelu_tanh_f(g)) * u);
            }
    }
}

int vidi_gemv_glu_dispatch(const void* X, const void* W, void* Y, int M, int I, int K, int ldx, int ldw, int ldy, int act, int dtype,
                           hipStream_t st) {
    if (M <= 0 || M > 8 || I <= 0 || (I % 32) || K <= 0 || K % 8 != 0 || ldw % 8 != 0 || ldx % 8 != 0) return VIDI_ERR_SHAPE;
    if (((uintptr_t)W & 15) || ((uintptr_t)X & 15)) return VIDI_ERR_ALIGN;
    if (act != ACT_GELU_TANH && act != ACT_SILU) return VIDI_ERR_ARG;
    const int silu = act == ACT_SILU;
    auto go = [&](auto kern, int rpw) -> int {
        const int blocks = max(1, min((I + 4 * rpw - 1) / (4 * rpw), 256 * 8));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, (const u16*)X, (const u16*)W, (u16*)Y, M, I, K, ldx, ldw, ldy, silu);
        return (int)hipGetLastError();
    };
    if (dtype == VIDI_DT_BF16) {
        if (M <= 1) return go(gemv_glu_kernel<BF16, 1, 2>, 2);
        if (M <= 2) return go(gemv_glu_kernel<BF16, 2, 2>, 2);
        if (M <= 4) return go(gemv_glu_kernel<BF16, 4, 1>, 1);
        return go(gemv_glu_kernel<BF16, 8, 1>, 1);
    } else if (dtype == VIDI_DT_F16) {
        if (M <= 1) return go(gemv_glu_kernel<F16, 1, 2>, 2);
        if (M <= 2) return go(gemv_glu_kernel<F16, 2, 2>, 2);
        if (M <= 4) return go(gemv_glu_kernel<F16, 4, 1>, 1);
        return go(gemv_glu_kernel<F16, 8, 1>, 1);
    }
    return VIDI_ERR_DTYPE;
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

template <typename T, int BN, int BM, int WN, int WM, int STAGES, int MODE, bool REPKV, int BK = 64, int STAG = 0, int MI = 32>
static int launch_cfg(const GemmParams& p, int batch, hipStream_t st) {
    constexpr int RING = STAGES * (BN + BM) * BK * 2;
    constexpr int CTILE = BM * ((MODE == MODE_GEGLU ? BN / 2 : BN) * 2 + 16);
    constexpr int LDS = RING > CTILE ? RING : CTILE;
    auto kern = gemm_kernel<T, BN, BM, WN, WM, STAGES, MODE, REPKV, BK, STAG, MI>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = set_lds(kern, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
    // One block per CU, each walking several tiles (measured +1...5 % per shape over one block per tile: the stores of a tile
    // drain under the next tile's first loads and no block relaunch sits between tiles).  VIDI_GEMM_PERSIST=n overrides the
    // block count (multiple of 8 so a block's tiles stay on its XCD), 0 = one block per tile.
    static int persist = -1;
    if (persist < 0) {
        const char* e = getenv("VIDI_GEMM_PERSIST");
        if (e) persist = atoi(e);
        else {
            int dev = 0, ncu = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
            persist = ncu;
        }
        if (persist < 0) persist = 0;
        persist = persist / 8 * 8;
    }
    const int gx = (persist > 0 && tiles > persist) ? persist : tiles;
    hipLaunchKernelGGL(kern, dim3(gx, batch), dim3(WN * WM * 64), LDS, st, p);
    return (int)hipGetLastError();
}

template <typename T, int MODE, bool REPKV>
static int launch_mode(const GemmParams& p, int batch, int tile_cfg, hipStream_t st) {
    if (tile_cfg == 3) {
        if constexpr (MODE == MODE_PLAIN) {
            const int tiles = ((p.N + 127) / 128) * ((p.M + 127) / 128);
            hipLaunchKernelGGL((gemm_kernel_regstage<T, MODE, REPKV>), dim3(tiles, batch), dim3(256), 0, st, p);
            return (int)hipGetLastError();
        } else {
            return VIDI_ERR_ARG;
        }
    }
    if (tile_cfg < 0) {
        // measured on MI355X (tools/bench_gemm.py): the 256x256 tile wins from ~1 wave of blocks up, also when
        // N is not a multiple of 256 (edge tiles are masked); small problems take the 128x128 tile.  The
        // 16x16x32-MFMA body (cfg 4) beats the 32x32x16 one (cfg 2) by 12-20 % on every large shape (power-limited chip).
        const long long t256 = (long long)((p.N + 255) / 256) * ((p.M + 255) / 256) * batch;
        tile_cfg = (t256 >= 192) ? 4 : 0;
    }
    switch (tile_cfg) {
        case 0: return launch_cfg<T, 128, 128, 2, 2, 2, MODE, REPKV>(p, batch, st);
        case 1: return launch_cfg<T, 128, 256, 2, 4, 3, MODE, REPKV>(p, batch, st);
        case 2: return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV>(p, batch, st);
        case 4: return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV, 64, 6, 16>(p, batch, st);       // 16x16x32 MFMA, DMA issued after the first fragment reads
        case 11:                                                                                      // 192-wide n tile (N = 1152: 6 exact tiles instead of 4.5)
            if constexpr (MODE == MODE_PLAIN) return launch_cfg<T, 192, 256, 2, 4, 2, MODE, REPKV, 64, 6, 16>(p, batch, st);
            else return VIDI_ERR_ARG;
        case 12: return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV, 64, 10, 16>(p, batch, st);      // mid-iteration barrier schedule
        case 14: case 15:                                                                             // 4 waves x (128x128): 1 wave/SIMD, 1/3 fewer LDS fragment bytes per FLOP
            if constexpr (MODE == MODE_PLAIN && !REPKV) {
                if (!getenv("VIDI_GEMM_EXPERIMENTAL")) return VIDI_ERR_ARG;
                if (tile_cfg == 14) return launch_cfg<T, 256, 256, 2, 2, 2, MODE, REPKV, 64, 10, 16>(p, batch, st);
                return launch_cfg<T, 256, 256, 2, 2, 2, MODE, REPKV, 64, 6, 16>(p, batch, st);
            } else {
                return VIDI_ERR_ARG;
            }
        case 5: case 6: case 7: case 10:                                                              // other schedules (same results)
            if constexpr (MODE == MODE_PLAIN && !REPKV) {
                if (!getenv("VIDI_GEMM_EXPERIMENTAL")) return VIDI_ERR_ARG;
                if (tile_cfg == 7) return launch_cfg<T, 256, 256, 2, 4, 4, MODE, REPKV, 32, 9, 16>(p, batch, st);   // ping-pong, 16x16x32, 4-deep ring
                if (tile_cfg == 10) return launch_cfg<T, 256, 256, 2, 4, 5, MODE, REPKV, 32, 9, 16>(p, batch, st);  // ping-pong, 5-deep ring
                if (tile_cfg == 5) return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV, 64, 7, 16>(p, batch, st);
                return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV, 64, 0, 16>(p, batch, st);
            } else {
                return VIDI_ERR_ARG;
            }
        // ---- experimental / diagnostic schedules (only with VIDI_GEMM_EXPERIMENTAL=1; see profiles/r1_gemm_pmc.md) ----
        case 9: if (!getenv("VIDI_GEMM_EXPERIMENTAL")) return VIDI_ERR_ARG;
                return launch_cfg<T, 256, 256, 2, 4, 4, MODE, REPKV, 32, 9>(p, batch, st);      // ping-pong schedule (correct results)
        case 8: case 13:                                                                       // diagnostics: WRONG results
            if constexpr (MODE == MODE_PLAIN && !REPKV) {
                if (!getenv("VIDI_GEMM_EXPERIMENTAL")) return VIDI_ERR_ARG;
                if (tile_cfg == 8) return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV, 64, 8>(p, batch, st);   // DMA always re-reads K slice 0
                return launch_cfg<T, 256, 256, 2, 4, 2, MODE, REPKV, 64, 3>(p, batch, st);                        // no DMA in the loop
            } else {
                return VIDI_ERR_ARG;
            }
        // (deeper BK=32 rings, staggered DMA issue and 2-blocks/CU 128x256 tiles were measured slower on MI355X —
        //  DESIGN.md 'GEMM experiments' — the template parameters BK / STAG remain for future schedules)
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

int vidi_gemv_dispatch(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy,
                       int dtype, hipStream_t st) {
    if (M <= 0 || M > 8 || N <= 0 || K <= 0 || K % 8 != 0 || ldw % 8 != 0 || ldx % 8 != 0) return VIDI_ERR_SHAPE;
    if (((uintptr_t)W & 15) || ((uintptr_t)X & 15)) return VIDI_ERR_ALIGN;
    auto go = [&](auto kern, int rpw) -> int {
        const int blocks = max(1, min((N + 4 * rpw - 1) / (4 * rpw), 256 * 8));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, (const u16*)X, (const u16*)W, (u16*)Y, M, N, K, ldx, ldw, ldy);
        return (int)hipGetLastError();
    };
    if (dtype == VIDI_DT_BF16) {
        if (M <= 1) return go(gemv_kernel<BF16, 1, 4>, 4);
        if (M <= 2) return go(gemv_kernel<BF16, 2, 4>, 4);
        if (M <= 4) return go(gemv_kernel<BF16, 4, 2>, 2);
        return go(gemv_kernel<BF16, 8, 2>, 2);
    } else if (dtype == VIDI_DT_F16) {
        if (M <= 1) return go(gemv_kernel<F16, 1, 4>, 4);
        if (M <= 2) return go(gemv_kernel<F16, 2, 4>, 4);
        if (M <= 4) return go(gemv_kernel<F16, 4, 2>, 2);
        return go(gemv_kernel<F16, 8, 2>, 2);
    }
    return VIDI_ERR_DTYPE;
}

int vidi_gemm_f32_dispatch(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K,
                           int ldx, int ldw, int ldy, int act, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 16 != 0 || (ldx % 4) || (ldw % 4)) return VIDI_ERR_SHAPE;
    const int tiles = ((N + 127) / 128) * ((M + 127) / 128);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(tiles), dim3(256), 0, st, X, W, bias, Y, M, N, K, ldx, ldw, ldy, act);
    return (int)hipGetLastError();
}
