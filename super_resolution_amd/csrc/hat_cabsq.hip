// hat_cabsq.hip — CAB squeeze conv (3x3, C -> mid <= 8 channels, + bias + erf-GELU) without LDS operand traffic.
// Contract: include/hat_mi355x.h (hat_cab_squeeze); reference: hat/archs/hat_arch.py:84-85 (cab.0 + GELU).
//
// With one n-tile of outputs the implicit-GEMM kernel (hat_conv) does one MFMA per 1 KB activation fragment read from
// LDS: it is bound by LDS operand traffic (6 KB per pixel), not by the 288 bytes per pixel it needs from HBM.  Here a
// wave sweeps a 16-pixel-wide strip top to bottom.  For every INPUT row it loads the row's three dx-shifted activation
// fragments straight from global memory (L1/L2 absorb the overlap) and multiplies them with weights that live in
// registers for the whole kernel; the three taps of a kernel column go to DIFFERENT output rows, and that routing is
// done by the choice of accumulator: the MFMA C operand is one of three rolling output-row accumulators.
//   tile T1(dx): A rows 0-7  = tap (ky = 0, kx) -> output row r + 1        lanes g = 0, 1 (channels 4g .. 4g+3)
//                A rows 8-15 = tap (ky = 1, kx) -> output row r            lanes g = 2, 3
//   tile T2(dx): A rows 0-7  = tap (ky = 2, kx) -> output row r - 1        (rows 8-15 are zero)
// so after input row r the lower lane half of ring slot (r-1)%3 holds the ky = 0 and ky = 2 terms of output row r - 1 and
// the upper lane half of slot r%3 its ky = 1 terms: one cross-half add finishes the row.  No LDS, no barriers.
#include <cstdlib>
#include <type_traits>

#include "hat_common.h"

__device__ __attribute__((aligned(16))) unsigned hat_cabsq_zero_page[80] = {};   // 160 bf16 channels of zeros

namespace {

// KS = k-steps (C padded to 32 KS).  NCHW = false: bias + GELU -> (B,H,W,8) bf16 + per-unit channel sums (CAB squeeze);
// NCHW = true: (acc + bias) * out_scale + mean[ch] -> (B,nst,H,W) fp32 planes (conv_last, hat_arch.py:856-858).
struct SweepEpi { float out_scale; float mean[4]; int nst; };

template <int KS, bool NCHW>
__global__ __launch_bounds__(256, KS > 2 ? 2 : 3) void cab_squeeze_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wpk,
                                                             const float* __restrict__ bias, void* __restrict__ outv,
                                                             float* __restrict__ colsum, int H, int W, int C, int ldx,
                                                             int rows, int strips, int units, SweepEpi epi) {
    using M = MT<bf16_t>;
    using frag_t = M::frag_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int u = blockIdx.x * 4 + wave;          // wave unit: (band, strip), strips of one band adjacent
    const int b = blockIdx.y;
    if (u >= units) return;                        // whole wave; the kernel has no barriers
    const int band = u / strips, strip = u - band * strips;
    const int x0 = strip * 14 - 1, y0 = band * rows, y1 = min(y0 + rows, H);   // x0: the pixel column of lane c16 = 0 (a halo column)
    const bf16_t* xb = x + (size_t)b * H * W * ldx;
    bf16_t* ob = reinterpret_cast<bf16_t*>(outv) + (size_t)b * H * W * 8;
    float* of = reinterpret_cast<float*>(outv) + (size_t)b * epi.nst * H * W;

    frag_t A[6][KS];   // [2 * kx + (0: T1, 1: T2)][k-step], fragment-packed on the host: one coalesced 1 KB load each
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) A[t][ks] = M::load(wpk + ((size_t)(t * KS + ks) * 64 + lane) * 8);
    const f32x4 bs = *reinterpret_cast<const f32x4*>(bias + 4 * (g & 1));

    // ONE fragment set per input row: lane c16 loads pixel x0 + c16 of the row, x0 = 14 strip - 1.  The dx = -1 / +1 fragments
    // are that set moved one lane up / down the 16-lane row (DPP row_shr:1 / row_shl:1), so lanes 1..14 have all three taps
    // of a kernel row and a strip yields 14 output columns; lanes 0 and 15 only carry their neighbours' halo pixels (their
    // own MFMA columns are computed on a stale value and never stored).  Round 2 loaded all three shifted fragments from
    // memory — 15 KB per row and wave through the L1 / L2 path for 5 KB of new data — and that path, not HBM, set the
    // kernel's time (30 MB in flight, 2.7 us per row); the 16/14 overlap costs a seventh more MFMAs, which were idle.
    // A pixel outside the image is read from a zero page (row stride 0): that IS the conv's zero padding in x.
    const int xx = x0 + c16;
    const bool xin = xx >= 0 && xx < W;
    const bf16_t* cbase = xin ? xb + (size_t)xx * ldx : reinterpret_cast<const bf16_t*>(hat_cabsq_zero_page);
    const unsigned cstride = xin ? (unsigned)W * (unsigned)ldx : 0u;
    // channel offset of this lane's 8 channels in k-step ks: 32 ks + 8 g; the last k-step's groups past C re-read a
    // valid group (their weights are zero)
    int coff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) coff[ks] = (32 * ks + 8 * g + 8 <= C) ? 32 * ks + 8 * g : C - 8;

    // activation fragments of NB input rows in flight
    constexpr int NB = 3;
    frag_t Bc[NB][KS];
    auto load_row = [&](int r, frag_t (&C_)[KS]) {   // fragments of input row r (unconditional, clamped)
        const int rc = min(max(r, 0), H - 1);
        const bf16_t* pc = cbase + (size_t)rc * cstride;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) C_[ks] = M::load(pc + coff[ks]);
    };
    auto shifted = [&](const frag_t& c, auto left_tag) -> frag_t {
        constexpr bool LEFT = decltype(left_tag)::value;
        const u32x4 cu = __builtin_bit_cast(u32x4, c);
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = (unsigned)__builtin_amdgcn_update_dpp((int)cu[i], (int)cu[i], LEFT ? 0x111 : 0x101, 0xf, 0xf, false);
        return __builtin_bit_cast(frag_t, o);
    };
    const bool oin = c16 >= 1 && c16 <= 14 && xx < W;   // this lane owns an output column
    f32x4 ring[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};

    // one input row: S1 = slot of output row r + 1, S2 = slot of r - 1, S0 = slot of r (static indices: unrolled by 3)
    auto step = [&](int r, f32x4& S1, f32x4& S2, f32x4& S0, frag_t (&C_)[KS]) {
        if (r >= 0 && r < H) {   // (uniform; a row outside the image contributes nothing: zero padding in y)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const frag_t bl = shifted(C_[ks], std::true_type{});
                const frag_t br = shifted(C_[ks], std::false_type{});
                S1 = M::mma(A[0][ks], bl, S1);
                S2 = M::mma(A[1][ks], bl, S2);
                S1 = M::mma(A[2][ks], C_[ks], S1);
                S2 = M::mma(A[3][ks], C_[ks], S2);
                S1 = M::mma(A[4][ks], br, S1);
                S2 = M::mma(A[5][ks], br, S2);
            }
        }
        load_row(r + NB, C_);   // the buffer's next row (r + NB) goes out as soon as this one is consumed
        // output row y = r - 1: lower lane half of S2 + upper lane half of S0
        const int y = r - 1;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = S2[i] + __shfl_xor(S0[i], 32);
        if (y >= y0 && y < y1 && g < 2 && oin) {
            v += bs;
            if constexpr (NCHW) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * g + i < epi.nst) of[((size_t)(4 * g + i) * H + y) * W + xx] = v[i] * epi.out_scale + epi.mean[(4 * g + i) & 3];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = gelu_erf_fast(v[i]);
                Vec4<bf16_t>::store(ob + ((size_t)y * W + xx) * 8 + 4 * g, v);
                csum += as_stored<bf16_t>(v);
            }
        }
        S2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) S0[i] = g < 2 ? S0[i] : 0.f;
    };

#pragma unroll
    for (int nb = 0; nb < NB; ++nb) load_row(y0 - 1 + nb, Bc[nb]);
    // rows y0 - 1 .. y1, three per iteration so that ring slots and row buffers are compile-time registers: row r uses S1 = ring[(r + 1) % 3], S2 = ring[(r - 1) % 3], S0 = ring[r % 3], r counted from
    // y0 - 1 = "0" (the last group may run past y1: the loads are clamped and those output rows are >= y1, i.e. not stored —
    // cheaper than guards between the steps, at whose joins hipcc merges its s_waitcnt bookkeeping)
    for (int r = y0 - 1; r <= y1; r += 3) {
        step(r, ring[1], ring[2], ring[0], Bc[0]);
        step(r + 1, ring[2], ring[0], ring[1], Bc[1]);
        step(r + 2, ring[0], ring[1], ring[2], Bc[2]);
    }
    if (!NCHW && colsum != nullptr) {   // per-unit channel sums of the stored values (hat_cab_fold's ECA pooling)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = row_sum16(csum[i]);
            if (c16 == 0) colsum[((size_t)b * units + u) * 16 + 4 * g + i] = g < 2 ? s : 0.f;
        }
    }
}

}  // namespace

static int sweep_units(int H, int W, int slots, int* rows_out, int* units_out) {
    const int strips = (W + 13) / 14;   // a strip is 16 loaded columns = 14 output columns + one halo column each side
    // one round of wave units on the chip's wave slots when the frame allows it; never fewer than 8 rows per band (2 halo rows each)
    int bands = slots / strips;
    bands = bands < 1 ? 1 : bands;
    int rows = (H + bands - 1) / bands;
    rows = rows < 8 ? 8 : rows;
    *rows_out = rows;
    *units_out = strips * ((H + rows - 1) / rows);
    return 0;
}

extern "C" int hat_cab_squeeze_units(int32_t H, int32_t W, int32_t* rows_out, int32_t* units_out) {
    if (H < 1 || W < 16 || W % 16 || !rows_out || !units_out) return HAT_EINVAL;
    // 256 CUs x 4 SIMDs x 2 waves.  (One wave per SIMD — which lets the 13x13 ESC conv's workgroups start beside this kernel
    // instead of after it, 240 + 2 x 128 registers — was measured: both kernels then take nearly twice as long, 128 and 92 us,
    // and the pair finishes when it does run back to back, ~140 us after the tail.  HAT_SQUEEZE_SLOTS=1024 repeats it.)
    static const int slots = [] { const char* e = getenv("HAT_SQUEEZE_SLOTS"); const int v = e ? atoi(e) : 0; return v >= 256 ? v : 2048; }();
    return sweep_units(H, W, slots, rows_out, units_out);
}

extern "C" int hat_cab_squeeze(const void* x, const void* wpk, const float* bias, void* out, float* colsum, int32_t B,
                               int32_t H, int32_t W, int32_t C, int32_t ldx, int32_t dtype, void* stream) {
    if (!x || !wpk || !bias || !out || B < 1 || H < 1) return HAT_EINVAL;
    if (dtype != HAT_BF16) return HAT_EUNSUPPORTED;   // the fp32 parity path uses hat_conv
    if (C % 8 || C > 160 || C <= 128 || ldx < C || ldx % 8) return HAT_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpk)) % 16 || reinterpret_cast<uintptr_t>(out) % 8) return HAT_EINVAL;
    int32_t rows = 0, units = 0;
    const int rc = hat_cab_squeeze_units(H, W, &rows, &units);
    if (rc) return rc;
    HAT_LAUNCH((cab_squeeze_kernel<5, false>), dim3((units + 3) / 4, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
               reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(wpk), bias, out, colsum, H, W, C, ldx, rows,
               (W + 13) / 14, units, SweepEpi{1.0f, {0.f, 0.f, 0.f, 0.f}, 8});
    return hat_check_launch();
}

extern "C" int hat_conv3x3_to_planes(const void* x, const void* wpk, const float* bias, float* out, int32_t B, int32_t H,
                                     int32_t W, int32_t C, int32_t ldx, int32_t n_out, float out_scale, const float* mean4,
                                     int32_t dtype, void* stream) {
    if (!x || !wpk || !bias || !out || !mean4 || B < 1 || H < 1 || W < 16 || W % 16 || n_out < 1 || n_out > 8) return HAT_EINVAL;
    if (dtype != HAT_BF16) return HAT_EUNSUPPORTED;
    if (C != 64 || ldx < C || ldx % 8) return HAT_EUNSUPPORTED;   // conv_last: num_feat = 64 (hat_arch.py:656)
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wpk)) % 16) return HAT_EINVAL;
    int rows = 0, units = 0;
    sweep_units(H, W, 3072, &rows, &units);                 // 3 waves per SIMD at 2 k-steps
    HAT_LAUNCH((cab_squeeze_kernel<2, true>), dim3((units + 3) / 4, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
               reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(wpk), bias, out, nullptr, H, W, C, ldx, rows,
               (W + 13) / 14, units, SweepEpi{out_scale, {mean4[0], mean4[1], mean4[2], mean4[3]}, n_out});
    return hat_check_launch();
}
