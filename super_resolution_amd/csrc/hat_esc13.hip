// hat_esc13.hip — the ESC large-kernel conv (13x13, 16 -> 16 channels, per-sample weights: static filter + dynamic depthwise
// 3x3 on its centre; esc_arch.py:121-123) as a dedicated kernel (contract: hat_esc_conv13 in include/hat_mi355x.h).
//
// hat_conv runs this layer as a one-n-tile implicit GEMM: a fresh 1 KB activation fragment AND half a weight fragment per
// MFMA from LDS, weight chunks streamed through LDS behind a barrier each — 0.19 ms at 720p for 80 GFLOP, four times its
// own LDS / MFMA floors, and it sits on every HAB's critical path.  Here:
//   * ALL weights are resident in LDS for the life of a (persistent) workgroup: 91 A fragments = 7 column pairs x 13 tap rows
//     (the 13 taps of a row are paired as (0,1) .. (10,11), (12, zero)), copied in once by LDS-DMA straight from the
//     [16][169 * 16] rows hat_esc_weights writes (the fragment order is a gather of 16-byte pieces; the zero tap comes
//     from a zero page);
//   * the haloed input tile (44 x 44 pixels x 16 channels, 32 bytes per pixel: lane (p, g) -> pixel p + tap column, half
//     g & 1 is conflict-free for ds_read_b128) is resident too: no barrier inside the K loop;
//   * a wave owns RW output rows x 32 columns and sweeps the tap ROWS for a fixed column pair: the input-row fragment that
//     output row j + 1 uses for tap row dy is the one output row j needs for tap row dy + 1, so RW + 12 row fragments (x 2
//     column tiles) and 13 weight fragments feed 13 * RW * 2 MFMAs.  RW = 2 (sixteen waves, round 2): 0.79 KB of LDS reads
//     per MFMA, the LDS array 75 % busy at the MFMA rate; RW = 4 (eight waves, two per SIMD): 0.43 KB per MFMA.
#include <cstdlib>

#include "hat_common.h"

namespace {

constexpr int E_TR = 32, E_TC = 32, E_HALO = 6;
constexpr int E_HR = E_TR + 2 * E_HALO, E_HC = E_TC + 2 * E_HALO;      // 44 x 44 haloed pixels
constexpr int E_NFRAG = 7 * 13;
constexpr int E_X_OFF = E_NFRAG * 1024;                                  // 93184
constexpr int E_LDS = E_X_OFF + E_HR * E_HC * 32 + 64;                   // 155200 bytes (one workgroup per CU)

__device__ __attribute__((aligned(16))) unsigned hat_esc13_zero_page[4] = {0, 0, 0, 0};

template <int RW>
__global__ __launch_bounds__((E_TR / RW) * 64) void esc13_kernel(const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ wp, int Kpad,
                                                             bf16_t* __restrict__ y16, int H, int W, int tiles_x, int ntiles) {
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    constexpr int E_WAVES = E_TR / RW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const bf16_t* xb = x + (size_t)b * H * W * ldx;
    bf16_t* yb = y16 + (size_t)b * H * W * 16;
    // ---- weights -> LDS, once: fragment s = dxp * 13 + dy holds taps (dy, 2 dxp) and (dy, 2 dxp + 1) -----------------------
    {
        const bf16_t* wrow = wp + ((size_t)b * 16 + c16) * Kpad;
        for (int s = wave; s < E_NFRAG; s += E_WAVES) {
            const int dxp = s / 13, dy = s - 13 * dxp, dx = 2 * dxp + (g >> 1);
            const char* src = dx <= 12 ? reinterpret_cast<const char*>(wrow + (dy * 13 + dx) * 16 + 8 * (g & 1))
                                       : reinterpret_cast<const char*>(hat_esc13_zero_page);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + s * 1024), 16, 0, 0);
        }
        if (tid < 4) *reinterpret_cast<u32x4*>(smem + E_LDS - 64 + tid * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    const int r0 = RW * wave;
    // lane (p, g): pixel column c16 + (g >> 1) (the second tap of a pair is one column further), channel half g & 1
    // (opaque, and TWO bases for the 91 KiB of weight fragments: a DS instruction's immediate offset is 16 bits; derived from one
    // visible base the compiler re-created an address with a v_add for 268 of a tile's 323 reads)
    unsigned xbase = lds0 + E_X_OFF + (unsigned)((r0 * E_HC + c16 + (g >> 1)) * 32 + (g & 1) * 16);
    unsigned abase = lds0 + (unsigned)lane * 16u, abase1 = abase + 60u * 1024u;
    asm volatile("" : "+v"(xbase), "+v"(abase), "+v"(abase1));

    // The next tile's haloed input is fetched into registers BEFORE this tile's K loop and written to LDS after it: with one
    // workgroup per CU nothing else would cover that latency.
    constexpr int NPIECE = E_HR * E_HC * 2, NIT = (NPIECE + E_WAVES * 64 - 1) / (E_WAVES * 64);
    u32x4 pv[NIT];
    auto fetch = [&](int t) {
        const int tc = min(t, ntiles - 1);
        const int ty0 = (tc / tiles_x) * E_TR, tx0 = (tc - (tc / tiles_x) * tiles_x) * E_TC;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = min(tid + it * E_WAVES * 64, NPIECE - 1);
            const int hp = i >> 1, half = i & 1;
            const int hy = hp / E_HC, hx = hp - hy * E_HC;
            const int y = ty0 - E_HALO + hy, xx = tx0 - E_HALO + hx;
            const bool in = y >= 0 && y < H && xx >= 0 && xx < W;
            const u32x4 v = *reinterpret_cast<const u32x4*>(xb + ((size_t)min(max(y, 0), H - 1) * W + min(max(xx, 0), W - 1)) * ldx + half * 8);
            pv[it] = in ? v : u32x4{0u, 0u, 0u, 0u};
        }
    };
    fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ty0 = (t / tiles_x) * E_TR, tx0 = (t - (t / tiles_x) * tiles_x) * E_TC;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * E_WAVES * 64;
            if (i < NPIECE) *reinterpret_cast<u32x4*>(smem + E_X_OFF + i * 16) = pv[it];
        }
        __syncthreads();   // (first tile: also drains the weight copy)
        fetch(t + gridDim.x);   // in flight during the K loop (clamped to a valid tile past the end)

        f32x4 acc[RW][2];
#pragma unroll
        for (int pt = 0; pt < RW; ++pt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[pt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto ldb = [&](int rr, int dxp, int ct) {
            return __builtin_bit_cast(bf8, *(__attribute__((address_space(3))) const u32x4*)(uintptr_t)(xbase + (unsigned)((rr * E_HC) * 32 + dxp * 64 + ct * 512)));
        };
        // Operand reads run ONE tap row ahead of their MFMAs and the scheduler may not move anything across a tap row
        // (sched_barrier): left alone it hoists a column pair's 45 reads above the first MFMA — 144 registers for RW = 4,
        // where 128 is what lets two of these waves share a SIMD with a wave of the CAB squeeze conv (hat_cabsq.hip).
        auto lda = [&](int dxp, int dy) {
            const int sfr = dxp * 13 + dy;
            return __builtin_bit_cast(bf8, *(__attribute__((address_space(3))) const u32x4*)(uintptr_t)(sfr < 60 ? abase + (unsigned)(sfr * 1024) : abase1 + (unsigned)((sfr - 60) * 1024)));
        };
        bf8 rowf[7][RW + 12][2];   // [column pair][input row r0 + i][column tile] (fully unrolled: RW + 1 rows live at a time)
        bf8 afr[7][13];
        constexpr int PRE = RW < 2 ? RW : 2;   // rows of the next column pair requested during the last tap row of this one
        afr[0][0] = lda(0, 0);
#pragma unroll
        for (int j = 0; j < PRE; ++j) { rowf[0][j][0] = ldb(j, 0, 0); rowf[0][j][1] = ldb(j, 0, 1); }
#pragma unroll
        for (int dxp = 0; dxp < 7; ++dxp) {
#pragma unroll
            for (int dy = 0; dy < 13; ++dy) {
                if (dy == 0) {   // the column pair's remaining first rows (used by the LAST MFMAs of this tap row)
#pragma unroll
                    for (int j = PRE; j < RW; ++j) { rowf[dxp][j][0] = ldb(j, dxp, 0); rowf[dxp][j][1] = ldb(j, dxp, 1); }
                }
                if (dy < 12) {
                    afr[dxp][dy + 1] = lda(dxp, dy + 1);
                    rowf[dxp][dy + RW][0] = ldb(dy + RW, dxp, 0);
                    rowf[dxp][dy + RW][1] = ldb(dy + RW, dxp, 1);
                } else if (dxp < 6) {
                    afr[dxp + 1][0] = lda(dxp + 1, 0);
#pragma unroll
                    for (int j = 0; j < PRE; ++j) { rowf[dxp + 1][j][0] = ldb(j, dxp + 1, 0); rowf[dxp + 1][j][1] = ldb(j, dxp + 1, 1); }
                }
#pragma unroll
                for (int j = 0; j < RW; ++j)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[j][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[dxp][dy], rowf[dxp][dy + j][ct], acc[j][ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- store: lane holds channels 4g..4g+3 of pixel c16 of each of its 2 RW 16-pixel tiles ---------------------------
#pragma unroll
        for (int pt = 0; pt < RW; ++pt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int y = ty0 + r0 + pt, xx = tx0 + ct * 16 + c16;
                if (y < H && xx < W) Vec4<bf16_t>::store(yb + ((size_t)y * W + xx) * 16 + 4 * g, acc[pt][ct]);
            }
        __syncthreads();   // every wave is done with the input tile before the next one overwrites it
    }
}

}  // namespace

extern "C" int hat_esc_conv13(const void* x, int32_t ldx, const void* wp, int32_t Kpad, void* y16, int32_t B, int32_t H, int32_t W,
                              int32_t dtype, void* stream) {
    if (!x || !wp || !y16 || B < 1 || H < 1 || W < 1 || ldx < 16 || ldx % 8 || Kpad < 169 * 16 || Kpad % 8) return HAT_EINVAL;
    if (dtype != HAT_BF16) return HAT_EUNSUPPORTED;
    const int tiles_x = (W + E_TC - 1) / E_TC, tiles_y = (H + E_TR - 1) / E_TR, ntiles = tiles_x * tiles_y;
    int gx = 256 / (B < 2 ? 1 : (B < 4 ? 2 : 4));
    if (gx > ntiles) gx = ntiles;
    static const int rw = [] { const char* e = getenv("HAT_ESC13_RW"); return e && atoi(e) == 2 ? 2 : 4; }();   // 2: round 2's sixteen-wave shape (A/B)
    auto launch = [&](auto kern, int waves) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS);
        if (e != hipSuccess) return (int)e;
        HAT_LAUNCH(kern, dim3(gx, B), dim3(waves * 64), E_LDS, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const bf16_t*>(x), ldx,
                   reinterpret_cast<const bf16_t*>(wp), Kpad, reinterpret_cast<bf16_t*>(y16), H, W, tiles_x, ntiles);
        return hat_check_launch();
    };
    return rw == 2 ? launch(esc13_kernel<2>, 16) : launch(esc13_kernel<4>, 8);
}
