// hat_common.h — device-side helpers shared by the gfx950 kernels (wave64, MFMA 16x16 tiles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hat_mi355x.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // one 16-byte piece

#define HAT_LDS_MAX 163840  // 160 KiB per workgroup on gfx950

// ---------------------------------------------------------------------------------------------
// MT<T>: one "k-step" of 32 along K on a 16x16 output tile.
//   bf16: one v_mfma_f32_16x16x32_bf16; lane l holds A[row l&15][k = 8*(l>>4)+j], j = 0..7
//   f32 : eight v_mfma_f32_16x16x4_f32; element j of every lane feeds MFMA j, which contracts
//         over k = 8*g + j, g = l>>4 = 0..3 — the same (g, j) -> k map as bf16, so LDS layouts
//         and fragment addressing are shared by both instantiations.  The f32 form is bit-wise
//         an fp32 fmaf chain (exact-fp32 parity path), 1/16 of the bf16 rate.
//   D layout (both): col = l & 15 (B's column), row = 4*(l>>4) + reg (A's row).
// ---------------------------------------------------------------------------------------------
template <typename T> struct MT;

template <> struct MT<bf16_t> {
    static constexpr int VEC = 8;   // elements per 16 bytes
    static constexpr int KC = 64;   // K elements per staged weight chunk (128 B per row)
    typedef bf16_t frag_t __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ frag_t load(const bf16_t* p) { return *reinterpret_cast<const frag_t*>(p); }
    static __device__ __forceinline__ frag_t zero() {
        frag_t z;
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = (bf16_t)0.0f;
        return z;
    }
    static __device__ __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    // half k-step (16 deep): v_mfma_f32_16x16x16_bf16, lane l holds A[row l&15][k = 4*(l>>4)+j], j = 0..3
    typedef short half_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ half_t load_half(const bf16_t* p) { return *reinterpret_cast<const half_t*>(p); }
    static __device__ __forceinline__ f32x4 mma_half(half_t a, half_t b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    }
};

template <> struct MT<float> {
    static constexpr int VEC = 4;
    static constexpr int KC = 32;
    typedef float frag_t __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ frag_t load(const float* p) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
        frag_t r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    }
    static __device__ __forceinline__ frag_t zero() {
        frag_t z;
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = 0.0f;
        return z;
    }
    static __device__ __forceinline__ f32x4 mma(frag_t a, frag_t b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
    // half k-step (16 deep): four 16x16x4 MFMAs, element j <-> k = 4*(l>>4)+j
    typedef f32x4 half_t;
    static __device__ __forceinline__ half_t load_half(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ f32x4 mma_half(half_t a, half_t b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
};

// conversions ---------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T to_T(float v);
template <> __device__ __forceinline__ float to_T<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t to_T<bf16_t>(float v) { return (bf16_t)v; }  // RNE, v_cvt_pk_bf16_f32
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }

// 4 consecutive T values <-> 4 floats
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    typedef f32x4 raw_t;  // load_raw / cvt split a load from its conversion, so a batch of loads can stay in flight
    static __device__ __forceinline__ raw_t load_raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ f32x4 cvt(raw_t r) { return r; }
    static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<bf16_t> {
    typedef bf16_t v4 __attribute__((ext_vector_type(4)));
    typedef v4 raw_t;
    static __device__ __forceinline__ raw_t load_raw(const bf16_t* p) { return *reinterpret_cast<const v4*>(p); }
    static __device__ __forceinline__ f32x4 cvt(raw_t h) {
        f32x4 r;
        r[0] = (float)h[0]; r[1] = (float)h[1]; r[2] = (float)h[2]; r[3] = (float)h[3];
        return r;
    }
    static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
        const v4 h = *reinterpret_cast<const v4*>(p);
        f32x4 r;
        r[0] = (float)h[0]; r[1] = (float)h[1]; r[2] = (float)h[2]; r[3] = (float)h[3];
        return r;
    }
    static __device__ __forceinline__ void store(bf16_t* p, f32x4 v) {
        v4 h;
        h[0] = (bf16_t)v[0]; h[1] = (bf16_t)v[1]; h[2] = (bf16_t)v[2]; h[3] = (bf16_t)v[3];
        *reinterpret_cast<v4*>(p) = h;
    }
};

// The value a stored T holds: what a pool over a stored map must add (the ESC pool, esc_arch.py:96,121, is a mean of the very
// tensor the 13x13 conv then reads — here the T-rounded LayerNorm output; pooling the unrounded fp32 values instead makes the
// pool depend on WHO computes it: a band-sharded frame pools the stored rows).
template <typename T> __device__ __forceinline__ f32x4 as_stored(f32x4 v) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)(bf16_t)v[r];
    }
    return v;
}

// Two adjacent n-tiles' accumulators of one pixel -> ONE 16-byte bf16 store per lane.  In the MFMA D layout lane group g
// holds channels 4g..4g+3 of every n-tile, so a plain store is 8 bytes per lane and 32 contiguous bytes per pixel.
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second: afterwards an
// even lane group holds 8 consecutive channels of tile `nt`, an odd one 8 consecutive channels of tile `nt + 1`, and one
// store instruction writes 64 contiguous bytes per pixel.  All 64 lanes must be active.  `row` = the pixel's output row,
// n0 = first channel of tile nt; needs 16-byte aligned rows.
__device__ __forceinline__ void store_pair_bf16(bf16_t* row, int n0, int g, f32x4 a, f32x4 b) {
    typedef bf16_t v4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const v4 ha = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
    const v4 hb = {(bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    const u32x2 ua = __builtin_bit_cast(u32x2, ha), ub = __builtin_bit_cast(u32x2, hb);
    const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
    const int n = (g & 1) ? n0 + 16 + 4 * (g - 1) : n0 + 4 * g;
    *reinterpret_cast<u32x4*>(row + n) = v;
}
// the same with a per-lane store predicate (the lane exchange itself always runs on all 64 lanes)
__device__ __forceinline__ void store_pair_bf16_if(bf16_t* row, int n0, int g, f32x4 a, f32x4 b, bool pred) {
    typedef bf16_t v4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const v4 ha = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
    const v4 hb = {(bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    const u32x2 ua = __builtin_bit_cast(u32x2, ha), ub = __builtin_bit_cast(u32x2, hb);
    const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
    const int n = (g & 1) ? n0 + 16 + 4 * (g - 1) : n0 + 4 * g;
    if (pred) *reinterpret_cast<u32x4*>(row + n) = v;
}

// ... and for FP16 rows (the 16-bit residual stream, hat_hab_tail3): round to nearest even, clamped to the finite FP16 range
__device__ __forceinline__ void store_pair_f16_if(_Float16* row, int n0, int g, f32x4 a, f32x4 b, bool pred) {
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto c = [](float x) { return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); };
    const v4 ha = {c(a[0]), c(a[1]), c(a[2]), c(a[3])};
    const v4 hb = {c(b[0]), c(b[1]), c(b[2]), c(b[3])};
    const u32x2 ua = __builtin_bit_cast(u32x2, ha), ub = __builtin_bit_cast(u32x2, hb);
    const auto r0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
    const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
    const int n = (g & 1) ? n0 + 16 + 4 * (g - 1) : n0 + 4 * g;
    if (pred) *reinterpret_cast<u32x4*>(row + n) = v;
}

// LDS row stride (in elements) for rows of `n` elements of size `es`, for MFMA operand images read by ds_read_b128
// with lane (c16, g) -> row base + c16, 16-byte slot k0 + g (bf16) or k0 + 2g (+1) (f32).  The instruction is serviced
// in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,...}: rows 0-3 and 12-15 of lane group g together
// with rows 4-11 of lane group g^1.  Enumerating all base alignments (tools: DESIGN.md 4.1): with 2-byte elements a
// stride of S slots is conflict-free exactly when S % 4 == 2 (odd strides, the textbook padding, are 2-way on every
// read); with 4-byte elements nothing beats an odd stride (2-way).
__host__ __device__ constexpr int lds_row_elems(int n, int es) {
    int slots = (n * es + 15) / 16;
    if (es == 2) slots += (6 - slots % 4) % 4;    // -> slots % 4 == 2
    else if ((slots & 1) == 0) slots += 1;
    return slots * 16 / es;
}

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in every lane: four v_add_f32 with row_ror
// modifiers instead of four ds_bpermute round trips (__shfl_xor).  Deterministic; order differs from a butterfly.
__device__ __forceinline__ float row_sum16(float s) {
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x128, 0xf, 0xf, false));
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x124, 0xf, 0xf, false));
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x122, 0xf, 0xf, false));
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x121, 0xf, 0xf, false));
    return s;
}

// Workgroup barrier for LDS hand-offs that leaves global loads in flight.  __syncthreads() also drains vmcnt, which
// turns every prefetch issued before it into a full memory round trip at the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-GELU for T-typed (bf16) outputs: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below a bf16 ulp) with
// raw v_exp/v_rcp: ~16 VALU instructions instead of libdevice erff's ~50, which made the 288-wide GELU linear VALU-bound.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = fmaf(-p * t, e, 1.0f);          // erf(|x| / sqrt 2)
    return 0.5f * x + 0.5f * fabsf(x) * erf_abs;           // 0.5 x (1 + sign(x) erf_abs)
}
template <typename T> __device__ __forceinline__ float gelu_act(float x) {
    if constexpr (sizeof(T) == 2) return gelu_erf_fast(x);
    else return gelu_erf(x);
}

// Launch status.  Every launch site first drains hipGetLastError(): the value is per-thread state that
// other libraries in the process (e.g. a failed probe inside the framework) may have left set.
#define HAT_LAUNCH(...)                      \
    do {                                     \
        (void)hipGetLastError();             \
        hipLaunchKernelGGL(__VA_ARGS__);     \
    } while (0)

static inline int hat_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
