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
#include "hat_common.h"

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
    const int x0 = strip * 16, y0 = band * rows, y1 = min(y0 + rows, H);
    const bf16_t* xb = x + (size_t)b * H * W * ldx;
    bf16_t* ob = reinterpret_cast<bf16_t*>(outv) + (size_t)b * H * W * 8;
    float* of = reinterpret_cast<float*>(outv) + (size_t)b * epi.nst * H * W;

    frag_t A[6][KS];   // [2 * kx + (0: T1, 1: T2)][k-step], fragment-packed on the host: one coalesced 1 KB load each
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) A[t][ks] = M::load(wpk + ((size_t)(t * KS + ks) * 64 + lane) * 8);
    const f32x4 bs = *reinterpret_cast<const f32x4*>(bias + 4 * (g & 1));

    // this lane's pixel column per dx, clamped, and whether it is inside the image
    int xc[3];
    bool xok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int xx = x0 + c16 + d - 1;
        xok[d] = xx >= 0 && xx < W;
        xc[d] = min(max(xx, 0), W - 1);
    }
    // channel offset of this lane's 8 channels in k-step ks: 32 ks + 8 g; the last k-step's groups past C re-read a
    // valid group (their weights are zero)
    int coff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) coff[ks] = (32 * ks + 8 * g + 8 <= C) ? 32 * ks + 8 * g : C - 8;

    // activation fragments of NB input rows in flight: with 2 k-steps there are registers for a second row, which
    // hides twice the load latency behind the (short) MFMA phase of a row
    constexpr int NB = KS <= 2 ? 2 : 1;
    frag_t Bf[NB][3][KS];
    auto load_row = [&](int r, int d, frag_t (&B)[3][KS]) {   // fragments of input row r shifted by dx = d - 1 (unconditional, clamped)
        const int rc = min(max(r, 0), H - 1);
        const bf16_t* p = xb + ((size_t)rc * W + xc[d]) * ldx;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) B[d][ks] = M::load(p + coff[ks]);
    };
    f32x4 ring[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};

    // one input row: S1 = slot of output row r + 1, S2 = slot of r - 1, S0 = slot of r (static indices: unrolled by 3)
    auto step = [&](int r, f32x4& S1, f32x4& S2, f32x4& S0, frag_t (&B)[3][KS]) {
        const bool rok = r >= 0 && r < H;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const bool ok = rok && xok[d];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const frag_t bf = ok ? B[d][ks] : M::zero();
                S1 = M::mma(A[2 * d][ks], bf, S1);
                S2 = M::mma(A[2 * d + 1][ks], bf, S2);
            }
            load_row(r + NB, d, B);   // the buffer's next row (r + NB), this dx, goes out as soon as this one is consumed
        }
        // output row y = r - 1: lower lane half of S2 + upper lane half of S0
        const int y = r - 1;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = S2[i] + __shfl_xor(S0[i], 32);
        if (y >= y0 && y < y1 && g < 2) {
            v += bs;
            if constexpr (NCHW) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * g + i < epi.nst) of[((size_t)(4 * g + i) * H + y) * W + x0 + c16] = v[i] * epi.out_scale + epi.mean[(4 * g + i) & 3];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = gelu_erf_fast(v[i]);
                Vec4<bf16_t>::store(ob + ((size_t)y * W + x0 + c16) * 8 + 4 * g, v);
                csum += as_stored<bf16_t>(v);
            }
        }
        S2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) S0[i] = g < 2 ? S0[i] : 0.f;
    };

#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int d = 0; d < 3; ++d) load_row(y0 - 1 + nb, d, Bf[nb]);
    // rows y0 - 1 .. y1, three (six with two row buffers) per iteration so that ring slots and buffers are compile-time
    // registers: row r uses S1 = ring[(r + 1) % 3], S2 = ring[(r - 1) % 3], S0 = ring[r % 3], r counted from y0 - 1 = "0"
    // (the last group may run up to 3 NB - 1 rows past y1: their loads are clamped and their output rows are >= y1, i.e.
    // not stored — cheaper than guards between the steps, at whose joins hipcc merges its s_waitcnt bookkeeping)
    for (int r = y0 - 1; r <= y1; r += 3 * NB) {
        step(r, ring[1], ring[2], ring[0], Bf[0]);
        step(r + 1, ring[2], ring[0], ring[1], Bf[1 % NB]);
        step(r + 2, ring[0], ring[1], ring[2], Bf[0]);
        if constexpr (NB == 2) {
            step(r + 3, ring[1], ring[2], ring[0], Bf[1]);
            step(r + 4, ring[2], ring[0], ring[1], Bf[0]);
            step(r + 5, ring[0], ring[1], ring[2], Bf[1]);
        }
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
    const int strips = W / 16;
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
    return sweep_units(H, W, 2048, rows_out, units_out);   // 256 CUs x 4 SIMDs x 2 waves (248 registers)
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
               W / 16, units, SweepEpi{1.0f, {0.f, 0.f, 0.f, 0.f}, 8});
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
               W / 16, units, SweepEpi{out_scale, {mean4[0], mean4[1], mean4[2], mean4[3]}, n_out});
    return hat_check_launch();
}
