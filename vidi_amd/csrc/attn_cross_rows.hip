// Text -> multimodal cross-attention for MANY query rows (a prompt, a batch of prompts): see attn_cross.hip for the path, the cache layout
// and the one-row-tile kernel (decode).  Replaces the same call sites (flash_attn_func / flash_attn_varlen_func, lmm/dattn/xattn.py:123,253
// via gemma.py:81-91) when a launch has two row tiles and more, a logit softcap and bf16 operands (vidi_attn_cross_row_tiles_per_block).
// A translation unit of its own, compiled WITHOUT -amdgpu-mfma-vgpr-form (attn_cross.hip has it): the matrix instructions here are asm
// statements with pinned register files (accumulators in AGPRs, everything else in VGPRs) — see the comment at qk_mfma below for why and
// for the wait states that are kept by hand; tools/isa_mfma_hazards.py and tests/test_build_resources.py read them back from the ISA.
#include "kernels.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) unsigned int u32x8;

#ifndef VIDI_XROWS_AHEAD
#define VIDI_XROWS_AHEAD 3            // sub-tiles in flight ahead of the scores being formed (3: 128 KB of LDS at HD = 256; 4: 160 KB, all of it)
#endif

// ---- many query rows (a prompt: 2 x Lq rows per kv head; a batch of prompts: hundreds) ------------------------------------------------
// attn_cross_body gives every 32-row tile its own blocks, each of which streams its key slice through private per-wave rings: R / 32 row
// tiles read the whole K / V R / 32 times.  At the 8-prompt prefill of BASELINE configs[4] (608 rows = 19 row tiles) that is 14 GB of
// L2 -> LDS traffic per layer and modality for 0.74 GB of keys — the launch ran at 0.3 PFLOP/s and 0.5 TB/s of unique bytes, bound by the
// L2's bandwidth (round-4 verdict item 4; profiles/r5_notes.md section 3 has every step with its measurement).  Here:
//   * the four waves of a block own FOUR DIFFERENT row tiles and share ONE K / V stream: a 32-key sub-tile (K 32 x HD, Vt HD x 32: 32 KB at
//     HD = 256) is DMA'd into block-wide rings once — every wave issues a quarter of its 1 KB pieces — and consumed by all four waves, each
//     with its own register-resident Q fragments, row sums and accumulators; one barrier per sub-tile; K / V traffic per launch drops 4x
//     (and the row blocks of one kv head run on one XCD and share its L2);
//   * the softmax has NO running maximum: the tanh softcap bounds every logit, so one fixed reference serves all rows (see m_ref below) — no
//     cross-lane maximum, no rescale of the accumulators, and the accumulators are never touched by the VALU inside the loop;
//   * a three-stage software pipeline inside the wave: QK^T of sub-tile i + 1 and PV of sub-tile i - 1 are on the matrix pipe while the VALU
//     turns the scores of sub-tile i into probabilities, one matrix instruction per ~32 cycles of lock-stepped VALU work.
// A wave writes the partial (numerator, reference, sum) of its row tile itself — the partial layout and the merge kernels of attn_cross.hip.
// Values: the one-row-tile kernel's up to the softmax's reference (a power-of-two-free shift of the exponent: the probabilities differ in
// their last bits) and the tanh's division (v_rcp here, IEEE there).
//
// Two softmax forms (MODE):
//   XR_FIXED    bf16 + a softcap with softcap log2 e <= 96 (Gemma2 in bf16, the BASELINE dtype): no running maximum at all (see m_ref below);
//   XR_RUN_CAP  fp16 + softcap (the reference's inference dtype), or a cap too large for the fixed form;
//   XR_RUN      no softcap (Vidi-7B), both dtypes.
// The running forms keep a per-row reference that only ever moves UP, and rarely: a step forms its probabilities against the reference of
// the steps before it; only when one of them leaves the range T can hold (fp16: 2^14; bf16: 2^100) does the wave take a cold path that
// re-references the row — the accumulators (AGPRs) are scaled through VGPRs there, the row sum with them, and the step's probabilities
// are formed again.  The first step of a slice always takes it (the reference starts at "nothing seen"), afterwards a row takes it once per
// ~8+ binades its maximum climbs: a handful of times per slice at worst.  The hot path is the fixed form's plus a 16-way maximum.
enum { XR_FIXED = 0, XR_RUN_CAP = 1, XR_RUN = 2 };

template <typename T, int HD, int MODE>
__device__ __forceinline__ void attn_cross_rows_body(const AttnCrossParams& p, const int z, const int zsplit) {
    constexpr int QROW = HD * 2;
    constexpr int CPR = HD / 8;
    constexpr int KST = HD / 16;
    constexpr int DT = HD / 32;
    constexpr int KBYTES = 32 * QROW;
    constexpr int VBYTES = HD * 64;
    constexpr int KLD = KBYTES / 1024, VLD = VBYTES / 1024;      // 1 KB DMA pieces per sub-tile
    constexpr int KPW = KLD / 4, VPW = VLD / 4;                  // ... per wave
    constexpr int AHEAD = VIDI_XROWS_AHEAD;                      // sub-tiles requested ahead of the one whose scores are being formed
    constexpr int KSLOTS = AHEAD, VSLOTS = AHEAD + 2;            // rings: see the pipeline below
    static_assert(KLD % 4 == 0 && VLD % 4 == 0 && (KPW + VPW == 8 || KPW + VPW == 4), "HD must be 128 or 256");      // (wait_vmcnt<> knows 0, 4, 8, 16)
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [KSLOTS][K sub-tile] [VSLOTS][Vt sub-tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int kvh = blockIdx.x;
    const int r0 = (blockIdx.y * 4 + wave) * 32;                 // this wave's row tile
    const bool active = r0 < p.R;                                // (a block's last waves may have no rows: they run on zero Q fragments and store nothing)

    // Q fragments straight from global memory (B operand: column = row l31, contraction chunk 2 ks + hi); rows beyond R are zero
    u32x4 qf[KST];
    {
        const int r = r0 + l31;
        const bool live = r < p.R;
        const int rc = live ? r : 0;
        const u16* qrow = p.Q + (size_t)(rc / p.G) * p.ldq + (kvh * p.G + rc % p.G) * HD;
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) qf[ks] = live ? *(const u32x4*)(qrow + (2 * ks + hi) * 8) : u32x4{0, 0, 0, 0};
        // the fragments are consumed HERE as far as the compiler's wait-count bookkeeping goes: their loads are then complete before the first
        // DMA is issued, and the loop carries no pending ordinary load (with one pending at the loop header the compiler put vmcnt(0) — a
        // full drain of the K / V ring — in front of the first MFMA of every sub-tile: seen in the ISA)
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    const int nsub = (p.n_keys + 31) / 32;
    const int n_mine = z < nsub ? (nsub - z + zsplit - 1) / zsplit : 0;     // sub-tiles z, z + zsplit, ... (interleaved over the slices)
    const u16* kc_head = p.Kc + (size_t)kvh * p.ntile64 * 64 * HD;
    const u16* vt_head = p.Vtc + (size_t)kvh * p.ntile64 * HD * 64;
    char* const vring = smem + KSLOTS * KBYTES;
    auto issue = [&](int i) {                                    // sub-tile i of this block -> K slot i % 3, V slot i % 5; this wave's quarter of the pieces
        const int st = z + i * zsplit;
        const int kb = p.key_start + st * 32;
        char* sK = smem + (i % KSLOTS) * KBYTES;
        char* sV = vring + (i % VSLOTS) * VBYTES;
        const u16* ksrc = kc_head + (size_t)kb * HD;
        const u16* vsrc = vt_head + (size_t)(kb >> 5) * HD * 32;
#pragma unroll
        for (int jj = 0; jj < KPW; ++jj) {
            const int j = wave * KPW + jj;
            const int pidx = j * 64 + lane, row = pidx / CPR, cl = pidx % CPR;
            glds16(ksrc + row * HD + (cl ^ (row & 15)) * 8, sK + j * 1024);      // (default cache policy: the other row blocks re-read the slice from L2)
        }
#pragma unroll
        for (int jj = 0; jj < VPW; ++jj) {
            const int j = wave * VPW + jj;
            const int pidx = j * 64 + lane, d = pidx >> 2, cl = pidx & 3;
            glds16(vsrc + d * 32 + (cl ^ ((d >> 2) & 3)) * 8, sV + j * 1024);
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    // Softmax WITHOUT a running maximum.  With the tanh softcap every logit (base-2 units) lies in [-cap2, cap2], cap2 = softcap log2 e (72.1 for
    // Gemma2's 50), so ONE fixed reference serves every row of every launch: m_ref = 0, p = 2^logit in [2^-cap2, 2^cap2] — no overflow, no
    // underflow, no rescaling of the accumulators, no cross-lane maximum.  The margins are symmetric and the dispatcher admits the kernel only
    // while they hold (vidi_attn_cross_rtpb: cap2 <= VIDI_XROWS_MAX_CAP2 = 96): a row sum stays below 2^17 keys x 2^96 = 2^113 and a numerator
    // below that times |v|; the smallest probability, 2^-96, is a normal bf16 / fp32 number (bf16 has fp32's exponent range, fp16 does not:
    // see the dispatch).  The partial is (numerator, m_ref, l): the merge kernels take any reference.  Launches without a softcap (Vidi-7B),
    // with a larger one, or in fp16 go to the per-tile kernel, which keeps the running maximum.
    static_assert(MODE != XR_FIXED || T::id == VIDI_DT_BF16, "the fixed-reference softmax needs T's exponent range to be fp32's");
    constexpr bool CAP = MODE != XR_RUN;
    float l_run = 0.f;
    const float L2E = 1.4426950408889634f;
    const float pre2 = CAP ? 2.0f * (p.scale / p.softcap) * L2E : p.scale * L2E;   // CAP: exp(2 y) = 2^(score * pre2), y = score * scale / cap; else: logit = score * pre2
    const float capl2 = CAP ? p.softcap * L2E : 0.f;
    // running forms: the row's reference (the same in both half-waves of a row); "nothing seen yet" is a large finite negative number, so
    // that differences of references never form inf - inf
    constexpr float NOREF = -1.0e30f;
    constexpr float P_LIMIT = T::id == VIDI_DT_F16 ? 16384.0f : 1.2676506e30f;    // largest probability a step may hand to T (2^14; 2^100)
    constexpr float HEADROOM = 6.0f;                                               // a re-referenced row's maximum sits at 2^6
    float m_run = MODE == XR_FIXED ? 0.f : NOREF;                                  // the reference this wave's partial reports

    // Software pipeline over the sub-tiles, three stages deep (the wave is alone on its SIMD: whatever overlaps must overlap inside it, and a
    // wave issues in order).  Step i runs, score by score:
    //     matrix pipe:  one MFMA of QK^T(i + 1)   +   one MFMA of PV(i - 1)          (64 cycles of the pipe)
    //     VALU:         softcap + exponential of one score of sub-tile i            (~60 cycles: two v_exp, one v_rcp)
    // so the scores of sub-tile i + 1 and the products of sub-tile i - 1 are on the matrix pipe while the VALU turns sub-tile i's scores
    // into probabilities (PMC before this form: the wave spent 22 % of its cycles waiting on back-to-back PV instructions).  Data in LDS at
    // step i: K(i + 1) and V(i - 1) are being read, sub-tile i + 2 is landing, i + 3 is requested -> a K ring of 3 slots (K(i) was last
    // read in step i - 1) and a V ring of 5 (V(i - 2) was last read in step i - 1).  The barrier at the top of step i says "everybody's
    // pieces of sub-tile i + 1 have landed, everybody is done with K(i) and V(i - 2)", after which the DMA of i + 3 may overwrite exactly those.
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
    // Matrix instructions as asm with pinned register files (common.h: mfma32_*): the output accumulators in AGPRs ("+a": 128 registers at
    // HD = 256 that never compete with the fragments and are never copied), everything else — the scores the VALU works on, the Q / K / V / P
    // fragments — in VGPRs.  Through the builtins the compiler kept the accumulators in VGPRs across the loop and copied all 128 into AGPRs
    // and back around every sub-tile's PV (or, with -amdgpu-mfma-vgpr-form, parked the Q fragments in AGPRs and copied them out per use):
    // 300+ of the loop's VALU instructions either way (ISA + PMC, profiles/r5_notes.md).  To the compiler these are opaque statements: the
    // wait states gfx940+ needs around matrix instructions are kept BY HAND — a score tile is read in the step after the one whose MFMAs
    // wrote it; the P fragments a PV reads were written a whole step earlier; the accumulators are read only after the loop, behind s_nops;
    // no operand is named in the AGPR file unless it lives there (an "a" operand held in VGPRs is copied in right in front of its MFMA:
    // NaNs); every path through the loop issues the same matrix instructions (a path around them makes the compiler carry the accumulators
    // in VGPRs).  tools/isa_mfma_hazards.py reads the distances back from the ISA.
    auto qk_mfma = [&](f32x16& c, const u32x4& a, const u32x4& b, bool first) __attribute__((always_inline)) {
        if (first) T::mfma32_bV_first(c, a, b); else T::mfma32_bV(c, a, b);
    };
    auto k_frag = [&](const char* sK, int ks) __attribute__((always_inline)) {
        return *(const u32x4*)(sK + l31 * QROW + (((2 * ks + hi) ^ (l31 & 15)) << 4));
    };
    auto v_frag = [&](const char* sV, int j) __attribute__((always_inline)) {      // j = half * DT + dt: keys 16 half .. + 15 of d tile dt
        const int d = (j % DT) * 32 + l31;
        return *(const u32x4*)(sV + d * 64 + (((2 * (j / DT) + hi) ^ ((d >> 2) & 3)) << 4));
    };
    if (n_mine > 0) issue(0);
    if (n_mine > 1) issue(1);
    if (n_mine > 2) issue(2);
    if constexpr (AHEAD > 3) { if (n_mine > 3) issue(3); }
    f32x16 s_cur = zero16;
    u32x4 pp0 = {0, 0, 0, 0}, pp1 = {0, 0, 0, 0};                  // P fragments of the previous sub-tile (none yet: the first PV adds 0 x V(0))
    // A sub-tile's key-padding mask is 32 bytes that are the SAME for every lane: ONE scalar load into SGPRs per sub-tile, issued a whole
    // step ahead (at the end of step i - 1, behind that step's last LDS wait, so the next lgkmcnt wait — after the barrier and the first
    // fragment reads of step i — finds it long returned) and read after step i's matrix work.  (As ordinary global loads behind the MFMAs —
    // the form up to round 5 — the compiler put `s_waitcnt vmcnt(0)` in front of their use: a full drain of the hand-counted K / V DMA
    // ring once per sub-tile, on every launch of the product, which always passes a mask.  Scalar loads count in lgkmcnt, not vmcnt.)
    u32x8 mk = {0, 0, 0, 0, 0, 0, 0, 0};
    auto mask_prefetch = [&](int i) __attribute__((always_inline)) {
        const unsigned char* mp = p.mask + (size_t)(z + i * zsplit) * 32;
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(mk) : "s"(mp));
    };
    if (p.mask && n_mine > 0) mask_prefetch(0);
    if (n_mine > 0) {
        // (sub-tile 0 has landed; up to AHEAD - 1 younger ones may be in flight)
        if (AHEAD > 3 && n_mine > 3) wait_vmcnt<(AHEAD > 3 ? 3 : 2) * (KPW + VPW)>();
        else if (n_mine > 2) wait_vmcnt<2 * (KPW + VPW)>(); else if (n_mine > 1) wait_vmcnt<KPW + VPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        u32x4 kf[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) kf[ks] = k_frag(smem, ks);
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) qk_mfma(s_cur, kf[ks], qf[ks], ks == 0);
    }
    for (int i = 0; i < n_mine; ++i) {
        const bool more = i + 1 < n_mine;
        if (more) {
            // sub-tile i + 1 has landed; i + 2 (.. i + AHEAD - 1) may be in flight
            if (AHEAD > 3 && i + 3 < n_mine) wait_vmcnt<(AHEAD > 3 ? 2 : 1) * (KPW + VPW)>();
            else if (i + 2 < n_mine) wait_vmcnt<KPW + VPW>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (i + AHEAD < n_mine) issue(i + AHEAD);
        }
        // (a block's waves without rows run the same instructions on zero Q fragments, and the last step multiplies whatever the next K slot
        // holds — its scores are never used — so that every path through the loop issues the same matrix instructions)
        const char* sKn = smem + ((i + 1) % KSLOTS) * KBYTES;
        const char* sVp = vring + ((i > 0 ? i - 1 : 0) % VSLOTS) * VBYTES;      // step 0: V(0), landed, times P = 0
        const int st = z + i * zsplit;
        const int kb_local = st * 32;
        f32x16 s = s_cur;
        f32x16 s_next = zero16;
        float pv[16];
        {
            // K / V fragments four at a time, one batch ahead of the MFMAs that use them
            u32x4 kf[2][4], vf[2][4];
            auto load_k = [&](int batch) __attribute__((always_inline)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) kf[batch & 1][e] = k_frag(sKn, batch * 4 + e);
            };
            auto load_v = [&](int batch) __attribute__((always_inline)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[batch & 1][e] = v_frag(sVp, batch * 4 + e);
            };
            load_k(0);
            load_v(0);
            // Four scores at a time in lock-step, and ONE matrix instruction in front of every ~32 cycles of VALU work: a wave issues in order,
            // a 32x32x16 MFMA holds the matrix pipe for 32 cycles, and a second one issued right behind it stalls the wave for that long (PMC of
            // the forms with two MFMAs back to back: 21-23 % of the wave's cycles waiting on instruction issue).  A group's VALU work — four
            // multiplies, four v_exp (16 cycles each), four adds, four v_rcp, four fma, four v_exp — is cut into eight slots of about 32
            // cycles; its eight matrix instructions (QK^T(i + 1) and PV(i - 1) alternating: two links of the dependent QK chain are 64+ cycles
            // apart) lead the slots.  Fences keep the slots apart; inside a slot the order is free.
            auto mq = [&](int r) __attribute__((always_inline)) { if (r < KST) qk_mfma(s_next, kf[(r / 4) & 1][r % 4], qf[r], r == 0); };
            auto mp = [&](int r) __attribute__((always_inline)) { if (r < 2 * DT) T::mfma32_cA(o[r % DT], vf[(r / 4) & 1][r % 4], r < DT ? pp0 : pp1); };
            auto fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); };
            const float c2 = -2.0f * capl2;
            const float sh = capl2 - m_run;                        // (fixed form: capl2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r0 = 4 * g;
                if (r0 + 4 < KST) load_k(g + 1);
                if (r0 + 4 < 2 * DT) load_v(g + 1);
                float x[4];
                if constexpr (CAP) {
                    mq(r0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = s[r0 + e] * pre2;
                    x[0] = fast_exp2(x[0]);
                    fence();
                    mp(r0);
                    x[1] = fast_exp2(x[1]); x[2] = fast_exp2(x[2]);
                    fence();
                    mq(r0 + 1);
                    x[3] = fast_exp2(x[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = x[e] + 1.0f;
                    fence();
                    mp(r0 + 1);
                    x[0] = __builtin_amdgcn_rcpf(x[0]); x[1] = __builtin_amdgcn_rcpf(x[1]);
                    fence();
                    mq(r0 + 2);
                    x[2] = __builtin_amdgcn_rcpf(x[2]); x[3] = __builtin_amdgcn_rcpf(x[3]);
                    fence();
                    mp(r0 + 2);
                    // logit in base-2 units  cap2 tanh(y) = cap2 - 2 cap2 / (exp(2 y) + 1)  minus the reference (fixed form: 0)
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = __builtin_fmaf(c2, x[e], sh);
                    pv[r0] = fast_exp2(x[0]);
                    fence();
                    mq(r0 + 3);
                    pv[r0 + 1] = fast_exp2(x[1]); pv[r0 + 2] = fast_exp2(x[2]);
                    fence();
                    mp(r0 + 3);
                    pv[r0 + 3] = fast_exp2(x[3]);
                } else {
                    // no softcap: logit = score * scale * log2 e — one FMA and one exponential per score; the eight matrix instructions keep
                    // their alternating order, each in front of a quarter of the group's VALU work
                    mq(r0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = __builtin_fmaf(s[r0 + e], pre2, -m_run);
                    fence();
                    mp(r0);
                    pv[r0] = fast_exp2(x[0]);
                    fence();
                    mq(r0 + 1);
                    fence();
                    mp(r0 + 1);
                    pv[r0 + 1] = fast_exp2(x[1]);
                    fence();
                    mq(r0 + 2);
                    fence();
                    mp(r0 + 2);
                    pv[r0 + 2] = fast_exp2(x[2]);
                    fence();
                    mq(r0 + 3);
                    fence();
                    mp(r0 + 3);
                    pv[r0 + 3] = fast_exp2(x[3]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(pv[r0 + e]));       // (pure arithmetic: LLVM would sink it to its first user)
                fence();
            }
        }
        if (kb_local + 32 > p.n_keys) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb_local + krow32(r, hi) >= p.n_keys) pv[r] = 0.f;
        }
        bool mask_live = false;                                      // this sub-tile holds an invalid key (uniform)
        if (p.mask) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(mk));        // the scalar load issued a step ago (every ds_read of this step has been consumed by now)
            // a sub-tile whose 32 keys are all valid — every one but the few around padding / blank frames — costs eight scalar compares
            mask_live = (mk[0] & mk[1] & mk[2] & mk[3] & mk[4] & mk[5] & mk[6] & mk[7]) != 0x01010101u ||
                        (mk[0] | mk[1] | mk[2] | mk[3] | mk[4] | mk[5] | mk[6] | mk[7]) != 0x01010101u;
            if (mask_live) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned mv = hi ? mk[2 * j + 1] : mk[2 * j];  // keys 8 j + 4 hi .. + 3 of the sub-tile: register 4 j + e holds key krow32(4 j + e, hi)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (((mv >> (8 * e)) & 0xffu) == 0) pv[4 * j + e] = 0.f;
                }
            }
        }
        if constexpr (MODE != XR_FIXED) {
            // did a probability of this step leave T's range against the reference of the steps before?  (first step: every one did)
            float pmax = pv[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) pmax = fmaxf(pmax, pv[r]);
            const bool need_half = !(pmax <= P_LIMIT);                          // (also true for NaN / inf)
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(need_half) != 0, 0)) {
                // ---- cold path: re-reference the rows that need it ----
                // the row's largest VALID score of this sub-tile (masked / beyond-the-end keys do not count), over both half-waves
                float smax = -3.0e38f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    bool ok = kb_local + krow32(r, hi) < p.n_keys;
                    if (mask_live) ok = ok && (((hi ? mk[2 * (r >> 2) + 1] : mk[2 * (r >> 2)]) >> (8 * (r & 3))) & 0xffu) != 0;
                    smax = ok ? fmaxf(smax, s[r]) : smax;
                }
                smax = fmaxf(smax, __shfl_xor(smax, 32, 64));
                const bool need = __shfl_xor((int)need_half, 32, 64) != 0 || need_half;
                float tmax = smax * pre2;                                       // its logit in base-2 units
                if constexpr (CAP) tmax = capl2 - 2.0f * capl2 / (exp2f(tmax) + 1.0f);
                // (a row whose keys are all masked in this sub-tile has nothing to re-reference: its probabilities are zeros)
                const float new_ref = (need && smax > -1.0e38f) ? tmax - HEADROOM : m_run;
                const float f = exp2f(m_run - new_ref);                         // <= 1 up to rounding; 0 from "nothing seen"
                m_run = new_ref;
                l_run *= f;
                // the accumulators: PV(i - 1) was issued in this step's slots; its results must have landed before anything reads them
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) asm volatile("s_nop 7\n\ts_nop 7" : "+a"(o[dt]));
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] *= f;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) asm volatile("s_nop 1" : "+a"(o[dt]));
                // this step's probabilities against the new reference
                const float sh2 = capl2 - m_run;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float t = s[r] * pre2;
                    if constexpr (CAP) t = __builtin_fmaf(-2.0f * capl2, __builtin_amdgcn_rcpf(fast_exp2(t) + 1.0f), sh2);
                    else t = t - m_run;
                    bool ok = kb_local + krow32(r, hi) < p.n_keys;
                    if (mask_live) ok = ok && (((hi ? mk[2 * (r >> 2) + 1] : mk[2 * (r >> 2)]) >> (8 * (r & 3))) & 0xffu) != 0;
                    pv[r] = ok ? fast_exp2(t) : 0.f;
                }
            }
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) psum += pv[r];
        l_run += psum;
        pp0 = pack8<T>(pv);
        pp1 = pack8<T>(pv + 8);
        s_cur = s_next;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every LDS read of this step has returned before the wave can reach the next barrier
        if (p.mask && more) mask_prefetch(i + 1);
    }
    // PV of the last sub-tile
    if (n_mine > 0) {
        const char* sVp = vring + ((n_mine - 1) % VSLOTS) * VBYTES;
        u32x4 vf[2 * DT];
#pragma unroll
        for (int j = 0; j < 2 * DT; ++j) vf[j] = v_frag(sVp, j);
#pragma unroll
        for (int j = 0; j < 2 * DT; ++j) T::mfma32_cA(o[j % DT], vf[j], j < DT ? pp0 : pp1);
    }

    // ---- this wave's partial: numerator rows, (m, l) — the layout of one attn_cross_body partial ----
    if (!active) return;
    // the last PV results (8-pass MFMAs, invisible to the hazard recogniser) must have landed before anything reads them; naming the
    // accumulators as AGPR operands here also keeps the compiler's copies of them (for the stores below) behind this point
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) asm volatile("s_nop 7" : "+a"(o[dt]));
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int r = r0 + l31;
    if (r < p.R) {
        const size_t base = ((size_t)z * p.nkv + kvh) * p.Rpad + r;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 ov = {o[dt][4 * j], o[dt][4 * j + 1], o[dt][4 * j + 2], o[dt][4 * j + 3]};
                *(f32x4*)(p.Opart + base * HD + dt * 32 + 8 * j + 4 * hi) = ov;
            }
        if (hi == 0) {
            p.ML[base * 2] = m_run;
            p.ML[base * 2 + 1] = l_tot;
        }
    }
}

// one or two modalities' key regions in one launch (b.n_keys <= 0: only a); the first `za` z-slices sweep set a's keys
template <typename T, int HD, int MODE>
__global__ __launch_bounds__(256) void attn_cross_rows_kernel(AttnCrossParams a, AttnCrossParams b, int za) {
    const int bz = blockIdx.z;
    if (bz < za) attn_cross_rows_body<T, HD, MODE>(a, bz, za);
    else attn_cross_rows_body<T, HD, MODE>(b, bz - za, (int)gridDim.z - za);
}

int vidi_attn_cross_rows_launch(const AttnCrossParams& a, const AttnCrossParams& b, int za, int zb, int HD, int dtype, hipStream_t st) {
    const dim3 grid(a.nkv, (a.Rpad / 32 + 3) / 4, za + zb);
    const int lds = VIDI_XROWS_AHEAD * (32 * HD * 2) + (VIDI_XROWS_AHEAD + 2) * (HD * 64);          // K ring of AHEAD sub-tiles, V ring of AHEAD + 2
    if (((uintptr_t)a.mask & 3) || (zb > 0 && ((uintptr_t)b.mask & 3))) return VIDI_ERR_ALIGN;       // the mask is read by scalar dword loads
    const bool cap = a.softcap > 0.f;
    if (zb > 0 && (b.softcap > 0.f) != cap) return VIDI_ERR_ARG;
    const bool fixed = cap && a.softcap * 1.4426950408889634f <= VIDI_XROWS_MAX_CAP2 && dtype == VIDI_DT_BF16 && (zb <= 0 || b.softcap == a.softcap);
    const int mode = fixed ? XR_FIXED : (cap ? XR_RUN_CAP : XR_RUN);
#define LAUNCH(TT, HH, MM)                                                                    \
    do {                                                                                      \
        auto kern = attn_cross_rows_kernel<TT, HH, MM>;                                       \
        static bool done = false;                                                             \
        if (!done) {                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e != hipSuccess) return (int)e;                                               \
            done = true;                                                                      \
        }                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, b, za);                         \
    } while (0)
#define LAUNCH_HD(TT, MM) do { if (HD == 256) LAUNCH(TT, 256, MM); else LAUNCH(TT, 128, MM); } while (0)
    if (dtype == VIDI_DT_BF16) {
        if (mode == XR_FIXED) LAUNCH_HD(BF16, XR_FIXED);
        else if (mode == XR_RUN_CAP) LAUNCH_HD(BF16, XR_RUN_CAP);
        else LAUNCH_HD(BF16, XR_RUN);
    } else if (dtype == VIDI_DT_F16) {
        if (mode == XR_RUN_CAP) LAUNCH_HD(F16, XR_RUN_CAP);
        else LAUNCH_HD(F16, XR_RUN);
    } else return VIDI_ERR_DTYPE;
#undef LAUNCH_HD
#undef LAUNCH
    return (int)hipGetLastError();
}
