// hat_mlp.hip — the OCAB's MLP (hat_arch.py:309-313 with the residual of :391) as ONE launch for embed_dim 144, hidden 288, bf16:
//     out = r1 + fc2( GELU( fc1(x) ) ),      x = LayerNorm2 output (T rows), r1 = the fp32 residual stream
// (contract: hat_ocab_mlp in include/hat_mi355x.h).  As two hat_linear launches the 288-wide hidden tensor made a round trip
// through HBM (576 B/px written, 576 B/px read: 0.21 + 0.285 ms at 720p, both at the HBM rate); fused, a pixel costs
// 288 + 576 read and 288 (T rows) or 576 (fp32) written.
//
// Weight-stationary like hat_linear's kernel: one workgroup of eight waves per CU keeps BOTH weight matrices in LDS as MFMA A
// fragments and the waves stream over 16-pixel tiles without any barrier.  The two matrices are 162 KB of fragments, 2 KB
// more than the LDS: fc1's K = 144 is 4 full k-steps + one 16-deep half step (v_mfma_f32_16x16x16_bf16: no padding to
// 160), and four of fc2's 81 fragments live in registers (16 VGPRs per wave, loaded once).
// The hidden activations never leave the registers: GELU of fc1's accumulators (MFMA D layout: lane group g holds hidden units
// 4g..4g+3 of a 16-unit tile) IS the B operand of fc2 when fc2's k-slot (g, j) is defined as unit 4g + j of tile 2kk
// (j < 4) or of tile 2kk + 1 (j >= 4) — ops.pack_ocab_mlp orders fc2's fragments that way (the trick of the attention
// kernel's P fragment and of hat_ffn2's gate).
#include <cstdlib>

#include "hat_common.h"

namespace {

// LDS fragment address for a compile-time byte offset, from one of three per-lane bases 60 KiB apart: a DS instruction's
// immediate offset is 16 bits, and with ONE base the compiler re-creates a base register (a v_add) for every fragment beyond
// 64 KiB — 88 of the 1 090 vector instructions per tile in the MLP kernel, whose fragments fill the LDS.
struct LdsBases { unsigned b0, b1, b2; };
constexpr unsigned LB_STEP = 61440;
__device__ __forceinline__ LdsBases lds_bases(const char* lane_base) {
    typedef __attribute__((address_space(3))) char lds_char;
    LdsBases r;
    r.b0 = (unsigned)(uintptr_t)(lds_char*)lane_base;
    r.b1 = r.b0 + LB_STEP;
    r.b2 = r.b0 + 2 * LB_STEP;
    asm volatile("" : "+v"(r.b0), "+v"(r.b1), "+v"(r.b2));   // (opaque: three registers, not one base + re-materialised sums)
    return r;
}
template <typename V> __device__ __forceinline__ V lds_at(const LdsBases& lb, unsigned off) {
    const unsigned base = off < LB_STEP ? lb.b0 : (off < 2 * LB_STEP ? lb.b1 : lb.b2);
    const unsigned rel = off < LB_STEP ? off : (off < 2 * LB_STEP ? off - LB_STEP : off - 2 * LB_STEP);
    return *(__attribute__((address_space(3))) const V*)(uintptr_t)(base + rel);
}

constexpr int ML_C = 144, ML_HID = 288, ML_NT1 = ML_HID / 16, ML_NT2 = ML_C / 16, ML_KK = ML_HID / 32;   // 18, 9, 9
constexpr int ML_W1F = ML_NT1 * 4 * 1024;          // fc1 full fragments  [nt][ks 0..3][64 lanes][8]      73728
constexpr int ML_W1H = ML_NT1 * 512;               // fc1 half fragments  [nt][64 lanes][4] (channels 128..143)  9216
constexpr int ML_NREG = 4;                         // fc2 fragments kept in registers: (nt2 = 8, kk = 5..8)
constexpr int ML_W2F = (ML_NT2 * ML_KK - ML_NREG) * 1024;   // 78848
constexpr int ML_OFF_W1H = ML_W1F, ML_OFF_W2 = ML_W1F + ML_W1H, ML_OFF_B1 = ML_OFF_W2 + ML_W2F, ML_OFF_B2 = ML_OFF_B1 + ML_HID * 4;
constexpr int ML_LDS = ML_OFF_B2 + ML_C * 4;       // 163520 <= 163840

template <bool OUTF32, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ocab_mlp_kernel(const HatMlpDesc d, long npix, long tiles) {
    constexpr int ML_NTHR = WAVES * 64, ML_WAVES = WAVES;
    using M = MT<bf16_t>;
    typedef bf16_t T;
    typedef M::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = tid >> 6;
    {   // weights and biases -> LDS (plain 16-byte copies: the packed images are the LDS images)
        const char* w1 = reinterpret_cast<const char*>(d.w1f);
        for (int i = tid; i < (ML_W1F + ML_W1H) / 16; i += ML_NTHR)
            *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(w1 + (size_t)i * 16);
        const char* w2 = reinterpret_cast<const char*>(d.w2f);
        for (int i = tid; i < ML_W2F / 16; i += ML_NTHR)
            *reinterpret_cast<u32x4*>(smem + ML_OFF_W2 + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(w2 + (size_t)i * 16);
        float* b1 = reinterpret_cast<float*>(smem + ML_OFF_B1);
        for (int i = tid; i < ML_HID; i += ML_NTHR) b1[i] = d.b1[i];
        float* b2 = reinterpret_cast<float*>(smem + ML_OFF_B2);
        for (int i = tid; i < ML_C; i += ML_NTHR) b2[i] = d.b2[i];
    }
    frag_t w2r[ML_NREG];   // fc2 fragments (nt2 = 8, kk = 5 + i): the last four of the packed order
#pragma unroll
    for (int i = 0; i < ML_NREG; ++i)
        w2r[i] = M::load(reinterpret_cast<const T*>(d.w2f) + (size_t)(ML_NT2 * ML_KK - ML_NREG + i) * 512 + lane * 8);
    __syncthreads();

    const T* xg = reinterpret_cast<const T*>(d.x);
    const float* b1l = reinterpret_cast<const float*>(smem + ML_OFF_B1);
    const float* b2l = reinterpret_cast<const float*>(smem + ML_OFF_B2);
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    // one tile's B operand of fc1: 4 x 16 bytes (channels 32 ks + 8 g ..) + 8 bytes (channels 128 + 4 g ..) of pixel c16
    struct XB { frag_t f[4]; s16x4 h; };
    auto load_x = [&](long tile, XB& xb) {
        long p = tile * 16 + c16;
        p = p < npix ? p : npix - 1;
        const T* row = xg + p * d.ldx;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb.f[ks] = M::load(row + ks * 32 + 8 * g);
        xb.h = *reinterpret_cast<const s16x4*>(row + 128 + 4 * g);
    };
    auto load_r = [&](long tile, f32x4 (&r)[ML_NT2]) {
        long p = tile * 16 + c16;
        p = p < npix ? p : npix - 1;
#pragma unroll
        for (int nt = 0; nt < ML_NT2; ++nt) r[nt] = *reinterpret_cast<const f32x4*>(d.r1 + p * d.ldr1 + nt * 16 + 4 * g);
    };

    const long stride = (long)gridDim.x * ML_WAVES;
    long tile = (long)blockIdx.x * ML_WAVES + wave;
    XB xcur, xnxt;
    f32x4 r1v[ML_NT2];
    load_x(tile, xcur);
    load_x(tile + stride, xnxt);
    load_r(tile, r1v);
    for (; tile < tiles; tile += stride) {
        const long p = tile * 16 + c16;
        const long pc = p < npix ? p : npix - 1;
        // the weight fragments are re-read from LDS for every tile: an opaque per-iteration offset keeps the compiler from
        // hoisting (and spilling) them
        int wofs = lane * 16;
        asm volatile("" : "+v"(wofs));
        const char* wl = smem + wofs;
        const char* wl8 = smem + ML_OFF_W1H + (wofs >> 1);   // the half fragments: 8 bytes per lane
        const LdsBases lb = lds_bases(wl), lb8 = lds_bases(smem + (wofs >> 1));
        // Weight fragments are requested ahead of their MFMAs into their own registers — fc1: the five fragments + bias of channel
        // tile nt + 1 while tile nt's five MFMAs run; fc2: a four-deep ring, three fragments ahead — and the scheduler may not move
        // anything across a step (sched_barrier).  Left to itself it reads each fragment right before its MFMA and drains the LDS
        // queue (s_waitcnt lgkmcnt(0)) 140 times for the 171 MFMAs of a tile: the kernel ran at the pace of that chain, 0.33 ms,
        // not of its 1.06 GB (hat_conv.hip's K loop had the same disease).
        // ---- fc1 + GELU -> the B fragments of fc2 -------------------------------------------------------------------------
        frag_t pf[ML_KK];
        {
            struct W1 { s16x4 h; f32x4 b; frag_t f[4]; };
            W1 ws[2];
            auto rd1 = [&](int nt, W1& w) {
                w.h = lds_at<s16x4>(lb8, ML_OFF_W1H + nt * 512);
                w.b = *reinterpret_cast<const f32x4*>(b1l + nt * 16 + 4 * g);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) w.f[ks] = lds_at<frag_t>(lb, (nt * 4 + ks) * 1024);
            };
            rd1(0, ws[0]);
            f32x4 a[2];
#pragma unroll
            for (int nt = 0; nt < ML_NT1; ++nt) {
                if (nt + 1 < ML_NT1) rd1(nt + 1, ws[(nt + 1) & 1]);
                const W1& w = ws[nt & 1];
                // The 16-deep tail accumulates SEPARATELY (bias as its C operand) and is added on the VALU: chained behind the
                // fourth 16x16x32 MFMA as its C operand, the 16x16x16 MFMA read the accumulator before that result had landed
                // (the k-step 3 contribution was lost: hipcc places no wait states between the two MFMA shapes on gfx950).
                const f32x4 tail = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.h, xcur.h, w.b, 0, 0, 0);
                f32x4 ai = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ai = M::mma(w.f[ks], xcur.f[ks], ai);
                a[nt & 1] = ai + tail;
                if (nt & 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pf[nt >> 1][j] = (bf16_t)gelu_act<T>(a[0][j]);
                        pf[nt >> 1][4 + j] = (bf16_t)gelu_act<T>(a[1][j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- fc2 + bias + residual ----------------------------------------------------------------------------------------
        f32x4 acc[ML_NT2];
        {
            constexpr int NF = ML_NT2 * ML_KK - ML_NREG, AD = 3;   // fragments that live in LDS; ring depth
            frag_t ring[AD + 1];
            auto rd2 = [&](int f) { ring[f % (AD + 1)] = lds_at<frag_t>(lb, ML_OFF_W2 + f * 1024); };
#pragma unroll
            for (int f = 0; f < AD; ++f) rd2(f);
#pragma unroll
            for (int nt = 0; nt < ML_NT2; ++nt) {
                acc[nt] = *reinterpret_cast<const f32x4*>(b2l + nt * 16 + 4 * g) + r1v[nt];
#pragma unroll
                for (int kk = 0; kk < ML_KK; ++kk) {
                    const int f = nt * ML_KK + kk;
                    if (f + AD < NF) rd2(f + AD);
                    if (f < NF)
                        acc[nt] = M::mma(ring[f % (AD + 1)], pf[kk], acc[nt]);
                    else
                        acc[nt] = M::mma(w2r[f - NF], pf[kk], acc[nt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // the next tile's operands have arrived; rotate, then store (every lane stores: lanes past the last pixel re-store it)
        xcur = xnxt;
        if constexpr (OUTF32) {
            float* o = reinterpret_cast<float*>(d.out) + pc * d.ldo;
#pragma unroll
            for (int nt = 0; nt < ML_NT2; ++nt) *reinterpret_cast<f32x4*>(o + nt * 16 + 4 * g) = acc[nt];
        } else {
            bf16_t* o = reinterpret_cast<bf16_t*>(d.out) + pc * d.ldo;
#pragma unroll
            for (int nt = 0; nt + 1 < ML_NT2; nt += 2) store_pair_bf16(o, nt * 16, g, acc[nt], acc[nt + 1]);
            Vec4<T>::store(o + (ML_NT2 - 1) * 16 + 4 * g, acc[ML_NT2 - 1]);
        }
        load_x(tile + 2 * stride, xnxt);
        load_r(tile + stride, r1v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// hat_ocab_qkv: the OCAB's q and kv projections (hat_arch.py:347, :350) as ONE weight-stationary launch for embed_dim 144,
// bf16: out[p] = [q | k | v] (432 channels, q pre-scaled) = W . x[p] + b.  Two hat_linear launches read the LayerNorm
// output twice (and share the machine on two streams); here 27 channel tiles x (4 k-steps + the 16-deep tail) = 121.5 KB of A
// fragments stay in LDS and every 16-pixel tile is read once.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int QK_NT = 27, QK_N = QK_NT * 16;
constexpr int QK_WF = QK_NT * 4 * 1024, QK_WH = QK_NT * 512, QK_OFF_B = QK_WF + QK_WH, QK_LDS = QK_OFF_B + QK_N * 4;   // 126144 B

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ocab_qkv_kernel(const HatMlpDesc d, long npix, long tiles) {
    constexpr int ML_NTHR = WAVES * 64, ML_WAVES = WAVES;
    using M = MT<bf16_t>;
    typedef bf16_t T;
    typedef M::frag_t frag_t;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = tid >> 6;
    {
        const char* w1 = reinterpret_cast<const char*>(d.w1f);
        for (int i = tid; i < (QK_WF + QK_WH) / 16; i += ML_NTHR)
            *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(w1 + (size_t)i * 16);
        float* bl = reinterpret_cast<float*>(smem + QK_OFF_B);
        for (int i = tid; i < QK_N; i += ML_NTHR) bl[i] = d.b1[i];
    }
    __syncthreads();
    const T* xg = reinterpret_cast<const T*>(d.x);
    const float* bl = reinterpret_cast<const float*>(smem + QK_OFF_B);
    struct XB { frag_t f[4]; s16x4 h; };
    auto load_x = [&](long tile, XB& xb) {
        long p = tile * 16 + c16;
        p = p < npix ? p : npix - 1;
        const T* row = xg + p * d.ldx;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb.f[ks] = M::load(row + ks * 32 + 8 * g);
        xb.h = *reinterpret_cast<const s16x4*>(row + 128 + 4 * g);
    };
    const long stride = (long)gridDim.x * ML_WAVES;
    long tile = (long)blockIdx.x * ML_WAVES + wave;
    XB xcur, xnxt;
    load_x(tile, xcur);
    load_x(tile + stride, xnxt);
    for (; tile < tiles; tile += stride) {
        const long p = tile * 16 + c16;
        const long pc = p < npix ? p : npix - 1;
        int wofs = lane * 16;
        asm volatile("" : "+v"(wofs));   // (keeps the loop-invariant fragment reads inside the loop)
        const char* wl = smem + wofs;
        const char* wl8 = smem + QK_WF + (wofs >> 1);
        const LdsBases lb = lds_bases(wl), lb8 = lds_bases(smem + (wofs >> 1));
        bf16_t* o = reinterpret_cast<bf16_t*>(d.out) + pc * d.ldo;
        // channel tiles in pairs: results leave as 16-byte stores while the next pair's MFMAs run.  The five weight fragments +
        // bias of channel tile nt + 1 are requested while tile nt's five MFMAs run, order pinned (see ocab_mlp_kernel).
        f32x4 last = {0.f, 0.f, 0.f, 0.f};
        struct W1 { s16x4 h; f32x4 b; frag_t f[4]; };
        W1 ws[2];
        auto rd1 = [&](int nt, W1& w) {
            w.h = lds_at<s16x4>(lb8, QK_WF + nt * 512);
            w.b = *reinterpret_cast<const f32x4*>(bl + nt * 16 + 4 * g);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) w.f[ks] = lds_at<frag_t>(lb, (nt * 4 + ks) * 1024);
        };
        rd1(0, ws[0]);
        f32x4 a[2];
#pragma unroll
        for (int nt = 0; nt < QK_NT; ++nt) {
            if (nt + 1 < QK_NT) rd1(nt + 1, ws[(nt + 1) & 1]);
            const W1& w = ws[nt & 1];
            // (the 16-deep tail accumulates separately: see ocab_mlp_kernel)
            const f32x4 tail = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w.h, xcur.h, w.b, 0, 0, 0);
            f32x4 ai = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ai = M::mma(w.f[ks], xcur.f[ks], ai);
            a[nt & 1] = ai + tail;
            if (nt & 1) store_pair_bf16(o, (nt - 1) * 16, g, a[0], a[1]);
            else if (nt == QK_NT - 1) last = a[0];
            __builtin_amdgcn_sched_barrier(0);
        }
        Vec4<T>::store(o + (QK_NT - 1) * 16 + 4 * g, last);
        xcur = xnxt;
        load_x(tile + 2 * stride, xnxt);
    }
}

}  // namespace

extern "C" int hat_ocab_qkv(const HatMlpDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatMlpDesc& d = *dp;
    if (!d.x || !d.w1f || !d.b1 || !d.out || d.B < 1 || d.H < 1 || d.W < 1) return HAT_EINVAL;
    if (d.C != ML_C || d.hidden != QK_N || d.dtype != HAT_BF16) return HAT_EUNSUPPORTED;
    if (d.ldx < ML_C || d.ldx % 8 || d.ldo < QK_N || d.ldo % 8 || reinterpret_cast<uintptr_t>(d.out) % 16) return HAT_EINVAL;
    const long npix = (long)d.B * d.H * d.W, tiles = (npix + 15) / 16;
    static const int waves = getenv("HAT_QKV_WAVES") ? atoi(getenv("HAT_QKV_WAVES")) : 8;    // (A/B switch: 8 or 16 waves per workgroup; no difference measured)
    const int nw = waves == 8 ? 8 : 16;
    int gx = 256;
    if ((long)gx * nw > tiles) gx = (int)((tiles + nw - 1) / nw);
    auto kern = nw == 8 ? ocab_qkv_kernel<8> : ocab_qkv_kernel<16>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, QK_LDS);
    if (e != hipSuccess) return (int)e;
    HAT_LAUNCH(kern, dim3(gx), dim3(nw * 64), QK_LDS, reinterpret_cast<hipStream_t>(stream), d, npix, tiles);
    return hat_check_launch();
}

extern "C" int hat_ocab_mlp(const HatMlpDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatMlpDesc& d = *dp;
    if (!d.x || !d.w1f || !d.b1 || !d.w2f || !d.b2 || !d.r1 || !d.out) return HAT_EINVAL;
    if (d.B < 1 || d.H < 1 || d.W < 1) return HAT_EINVAL;
    if (d.C != ML_C || d.hidden != ML_HID || d.dtype != HAT_BF16) return HAT_EUNSUPPORTED;
    if (d.ldx < ML_C || d.ldx % 8 || d.ldr1 < ML_C || d.ldr1 % 4 || d.ldo < ML_C) return HAT_EINVAL;
    if (d.out_f32 ? d.ldo % 4 : (d.ldo % 8 || reinterpret_cast<uintptr_t>(d.out) % 16)) return HAT_EINVAL;
    const long npix = (long)d.B * d.H * d.W, tiles = (npix + 15) / 16;
    static const int waves = getenv("HAT_MLP_WAVES") ? atoi(getenv("HAT_MLP_WAVES")) : 8;    // (A/B switch; 16 waves spill at 128 registers and measured slower)
    const int nw = waves == 8 ? 8 : 16;
    int gx = 256;
    if ((long)gx * nw > tiles) gx = (int)((tiles + nw - 1) / nw);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    void (*kern)(const HatMlpDesc, long, long) = d.out_f32 ? (nw == 8 ? ocab_mlp_kernel<true, 8> : ocab_mlp_kernel<true, 16>)
                                                            : (nw == 8 ? ocab_mlp_kernel<false, 8> : ocab_mlp_kernel<false, 16>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ML_LDS);
    if (e != hipSuccess) return (int)e;
    HAT_LAUNCH(kern, dim3(gx), dim3(nw * 64), ML_LDS, s, d, npix, tiles);
    return hat_check_launch();
}
