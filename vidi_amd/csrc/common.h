// Shared device helpers for the vidi_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// ---- C-ABI constants (mirrored in include/vidi_hip.h) -------------------------------------
#define VIDI_DT_BF16 0
#define VIDI_DT_F16 1
#define VIDI_DT_F32 2

#define VIDI_OK 0
#define VIDI_ERR_SHAPE (-1)
#define VIDI_ERR_DTYPE (-2)
#define VIDI_ERR_ALIGN (-3)
#define VIDI_ERR_ARG (-4)

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15); every lane of the row gets the total.  All 64 lanes must be active.
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});        // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});        // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});       // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});       // row_mirror
    return v;
}

// The same reduction for TWO values at once, as v_add_f32 with the DPP permutation ON the add (one instruction per value and step).  Written
// with the builtin the compiler packs the two chains into v_pk_add_f32, which cannot carry a DPP modifier, and emits per step two zeroing
// moves, two v_mov_b32_dpp and the packed add (20 instructions instead of 8 in the GEMM's statistics epilogue).  A VALU result needs two
// wait states before a DPP read: the two chains alternate and one s_nop separates the steps.  Same additions in the same order.
#ifndef VIDI_ROW16_ASM
#define VIDI_ROW16_ASM 1
#endif
__device__ __forceinline__ void row16_sum2(float& a, float& b) {
#if VIDI_ROW16_ASM
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0"
                 : "+v"(a), "+v"(b));
#else
    a = row16_sum(a); b = row16_sum(b);
#endif
}

// eight 16-lane row sums at once (the statistics epilogue's four rows x (sum, sum of squares)): the same four butterfly steps, each over
// the eight independent chains back to back — the chains themselves cover the two wait states a DPP read needs after a VALU write, so the
// per-step s_nop of row16_sum2 (224 of a tile's 3 265 epilogue instructions) disappears.  Same additions in the same order per value.
__device__ __forceinline__ void row16_sum8(float (&v)[8]) {
#define VIDI_R8_STEP(CTRL)                                                              \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t"
                 VIDI_R8_STEP("quad_perm:[1,0,3,2]")
                 VIDI_R8_STEP("quad_perm:[2,3,0,1]")
                 VIDI_R8_STEP("row_half_mirror")
                 VIDI_R8_STEP("row_mirror")
                 "s_nop 0"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#undef VIDI_R8_STEP
}

// ---- scalar conversions -------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(u16 v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ u16 f32_to_bf16(float f) {   // round-to-nearest-even: hardware v_cvt_pk_bf16_f32
    return __builtin_bit_cast(u16, (__bf16)f);
}
__device__ __forceinline__ float f16_to_f32(u16 v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return (float)h;
}
__device__ __forceinline__ u16 f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    u16 v;
    __builtin_memcpy(&v, &h, 2);
    return v;
}

struct BF16 {
    static constexpr int id = VIDI_DT_BF16;
    static __device__ __forceinline__ float to_f32(u16 v) { return bf16_to_f32(v); }
    static __device__ __forceinline__ u16 from_f32(float f) { return f32_to_bf16(f); }
    static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(short8_t, a), __builtin_bit_cast(short8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(short8_t, a), __builtin_bit_cast(short8_t, b), c, 0, 0, 0);
    }
    // accumulator pinned to the AGPR file ("+a"): a 128x128 wave tile is 256 accumulator registers = the whole AGPR file, and
    // left to itself hipcc splits them between the files and copies them around every MFMA (measured in the .s)
    static __device__ __forceinline__ void mfma16_agpr(f32x4& c, const u32x4& a, const u32x4& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma16_agpr_first(f32x4& c, const u32x4& a, const u32x4& b) {      // c = a*b (C operand = 0)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    }
    // 32x32x16 forms with operands pinned to a register file (gfx90a+ takes A / B from either file): the B operand read straight from
    // AGPRs (no v_accvgpr_read per use of a resident fragment), the accumulator in AGPRs or VGPRs.
    // (the `_first` forms' result is early-clobber: a matrix instruction's destination must not overlap its A / B operands)
    // Each carries `s_nop 1` in front: a VALU-written operand needs two wait states before a matrix instruction reads it, and the compiler
    // puts plain moves right in front of these statements.
    // CAUTION — to the compiler these are opaque asm statements, not MFMAs: it inserts NONE of the wait states gfx940+ needs in software
    // around matrix instructions (XDL result -> VALU / memory reader: passes + 2..3; a register copied into the other file right in front of
    // the MFMA that reads it).  Callers keep readers of a result far behind it (or put s_nop in between) and pick the form whose operands
    // are RESIDENT in the named file — attn_cross.hip's many-row kernel documents its distances.
    static __device__ __forceinline__ void mfma32_bA(f32x16& c, const u32x4& a, const u32x4& b_agpr) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b_agpr));
    }
    static __device__ __forceinline__ void mfma32_bA_first(f32x16& c, const u32x4& a, const u32x4& b_agpr) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b_agpr));
    }
    static __device__ __forceinline__ void mfma32_cA(f32x16& c_agpr, const u32x4& a, const u32x4& b) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c_agpr) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma32_bV(f32x16& c, const u32x4& a, const u32x4& b) {              // everything in VGPRs
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma32_bV_first(f32x16& c, const u32x4& a, const u32x4& b) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    }
};
struct F16 {
    static constexpr int id = VIDI_DT_F16;
    static __device__ __forceinline__ float to_f32(u16 v) { return f16_to_f32(v); }
    static __device__ __forceinline__ u16 from_f32(float f) { return f32_to_f16(f); }
    static __device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mfma16_agpr(f32x4& c, const u32x4& a, const u32x4& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma16_agpr_first(f32x4& c, const u32x4& a, const u32x4& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma32_bA(f32x16& c, const u32x4& a, const u32x4& b_agpr) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b_agpr));
    }
    static __device__ __forceinline__ void mfma32_bA_first(f32x16& c, const u32x4& a, const u32x4& b_agpr) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b_agpr));
    }
    static __device__ __forceinline__ void mfma32_cA(f32x16& c_agpr, const u32x4& a, const u32x4& b) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c_agpr) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma32_bV(f32x16& c, const u32x4& a, const u32x4& b) {              // everything in VGPRs
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma32_bV_first(f32x16& c, const u32x4& a, const u32x4& b) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    }
};

// round an fp32 value to the storage dtype and back (the reference rounds every module output)
template <typename T>
__device__ __forceinline__ float rnd(float f) { return T::to_f32(T::from_f32(f)); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    if constexpr (T::id == VIDI_DT_BF16) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // one v_cvt_pk_bf16_f32
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
// raw v_exp_f32 (no denormal fix-up sequence): softmax probabilities below 2^-126 flush to zero
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
template <typename T>
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = T::to_f32((u16)(v[i] & 0xffffu));
        f[2 * i + 1] = T::to_f32((u16)(v[i] >> 16));
    }
}
template <typename T>
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2<T>(f[2 * i], f[2 * i + 1]);
    return v;
}

// ---- math ---------------------------------------------------------------------------------
// 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715x^3) — torch gelu(approximate='tanh') — evaluated as x / (1 + exp(-2u)): one v_exp + one
// v_rcp, no branches (tanhf() is ~50 instructions with divergent ranges; the epilogues evaluate this once per output element with the
// matrix cores idle).  -2u*log2(e) = x * (K0 + K1 x^2) with the constants folded; the pair form runs the polynomial, the +1 and the final
// product on packed fp32 instructions (v_pk_mul/fma/add_f32: two elements per issue slot), the scalar form is the same arithmetic
// element by element (bit-identical results), so every kernel of the library rounds a GELU the same way.
#define VIDI_GELU_K0 (-2.0f * 0.7978845608028654f * 1.4426950408889634f)
#define VIDI_GELU_K1 (-2.0f * 0.7978845608028654f * 1.4426950408889634f * 0.044715f)
__device__ __forceinline__ f32x2_t gelu_tanh_2(f32x2_t x) {
    const f32x2_t k0 = {VIDI_GELU_K0, VIDI_GELU_K0}, k1 = {VIDI_GELU_K1, VIDI_GELU_K1}, one = {1.0f, 1.0f};
    const f32x2_t w = x * __builtin_elementwise_fma(k1, x * x, k0);
    const f32x2_t d = one + f32x2_t{__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};      // exp(-2u); +inf for very negative x -> result -0
    return x * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
// NP pairs in lock-step, one arithmetic step at a time: every packed / transcendental instruction is followed by NP - 1 independent ones
// before its result is used, so no hazard wait states (s_nop) separate dependent pairs — left to itself the compiler runs the chains of
// an epilogue strip one after the other (535 s_nop per fc1 tile in the ISA).  Same arithmetic as gelu_tanh_2 element by element.
#ifndef VIDI_GELU_LOCKSTEP
#define VIDI_GELU_LOCKSTEP 1
#endif
template <int NP>
__device__ __forceinline__ void gelu_tanh_pairs(f32x2_t* x) {
#if VIDI_GELU_LOCKSTEP
    const f32x2_t k0 = {VIDI_GELU_K0, VIDI_GELU_K0}, k1 = {VIDI_GELU_K1, VIDI_GELU_K1}, one = {1.0f, 1.0f};
    f32x2_t w[NP];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) w[i] = x[i] * x[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) w[i] = __builtin_elementwise_fma(k1, w[i], k0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) w[i] = x[i] * w[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) w[i] = f32x2_t{__builtin_amdgcn_exp2f(w[i][0]), __builtin_amdgcn_exp2f(w[i][1])};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) w[i] = one + w[i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) w[i] = f32x2_t{__builtin_amdgcn_rcpf(w[i][0]), __builtin_amdgcn_rcpf(w[i][1])};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = x[i] * w[i];
    __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = gelu_tanh_2(x[i]);
#endif
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float w = x * __builtin_fmaf(VIDI_GELU_K1, x * x, VIDI_GELU_K0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}
__device__ __forceinline__ float silu_f(float x) {                   // x * sigmoid(x) (MistralMLP act), branch-free
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ f32x2_t silu_2(f32x2_t x) {               // the same arithmetic on a pair (packed multiplies / add)
    const f32x2_t k = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
    const f32x2_t w = k * x;
    const f32x2_t d = one + f32x2_t{__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    return x * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ float gelu_erf_f(float x) {
    // 0.5*x*(1+erf(x/sqrt2)) with erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7), branch-free
    const float z = fabsf(x) * 0.7071067811865476f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float q = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // 1 - erf(z), z >= 0
    // x >= 0: 0.5x(2 - q);  x < 0: 0.5x q
    return 0.5f * x * (x >= 0.f ? 2.0f - q : q);
}

// erf-GELU of NP pairs in lock-step on packed fp32 math: the arithmetic of gelu_erf_f element by element (same operations, same
// association, so the same bits), two elements per polynomial / product instruction and no hazard wait states between dependent steps
template <int NP>
__device__ __forceinline__ void gelu_erf_pairs(f32x2_t* x) {
#if VIDI_GELU_LOCKSTEP
    auto c2 = [](float c) { return f32x2_t{c, c}; };
    f32x2_t z[NP], t[NP], p[NP];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) z[i] = f32x2_t{fabsf(x[i][0]), fabsf(x[i][1])} * c2(0.7071067811865476f);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) t[i] = __builtin_elementwise_fma(c2(0.3275911f), z[i], c2(1.0f));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) t[i] = f32x2_t{__builtin_amdgcn_rcpf(t[i][0]), __builtin_amdgcn_rcpf(t[i][1])};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(c2(1.061405429f), t[i], c2(-1.453152027f));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], c2(1.421413741f));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], c2(-0.284496736f));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], c2(0.254829592f));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) { p[i] = p[i] * t[i]; z[i] = (c2(-1.4426950408889634f) * z[i]) * z[i]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) z[i] = f32x2_t{__builtin_amdgcn_exp2f(z[i][0]), __builtin_amdgcn_exp2f(z[i][1])};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = p[i] * z[i];                                         // q = 1 - erf(z), z >= 0
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const f32x2_t two_q = c2(2.0f) - p[i];
        t[i] = f32x2_t{x[i][0] >= 0.f ? two_q[0] : p[i][0], x[i][1] >= 0.f ? two_q[1] : p[i][1]};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = (c2(0.5f) * x[i]) * t[i];
    __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = f32x2_t{gelu_erf_f(x[i][0]), gelu_erf_f(x[i][1])};
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// MFMA 32x32 C/D layout: lane l holds column (l & 31); register r holds row krow32(r, l >> 5).
__device__ __forceinline__ int krow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
// Key permutation inside a 16-slab so that MFMA operand slot (g = lane>>5, e) of a 16-deep
// contraction is register r = 8*slab + e of a swapped-QK^T score tile: position p = 8g+e holds
// key (e&3) + 8*(e>>2) + 4g.  perm16(key) gives the storage position of `key` (involution-free map).
__host__ __device__ __forceinline__ int perm16(int x) { return 8 * ((x >> 2) & 1) + (x & 3) + 4 * (x >> 3); }

// Cross-half (lane ^ 32) reductions without LDS: v_permlane32_swap gives every lane both halves' values.
__device__ __forceinline__ float xhalf_max(float x) {
    const unsigned v = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const unsigned v = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// async global -> LDS, 16 B per lane, LDS destination = wave-uniform base + lane*16
template <int AUX = 0>      // cache policy bits of the load (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else static_assert(N < 0, "unsupported vmcnt");
}
