// hat_conv64r.hip — 3x3 convolution of a 64-channel map in bf16 with the weights RESIDENT in LDS: the Upsample convs
// (hat_arch.py:598, :601: 64 -> 64 r^2, nn.PixelShuffle folded into the store).  Same contract and weight layout as hat_conv
// (HatConvDesc, include/hat_mi355x.h), which dispatches here.
//
// hat_conv's kernel streams the weight slice through LDS in chunks behind two barriers each and reaches 23 % of the MFMA
// peak on this layer (1.9 ms for 1.09 TFLOP at 1440 x 2560); the round-2 experiments on the wider group conv
// (tools/experiments/hat_conv3_slab.hip.txt) showed why: per-chunk fixed costs and the weight stream from L2, not the
// MFMAs.  With 64 input channels the weights of 64 output channels are 72 KB: they fit.  So, as hat_esc13.hip does:
//   * a persistent workgroup owns ONE 64-channel output slice, copies its 72 A fragments (4 channel tiles x 18 k-steps) into
//     LDS once — LDS-DMA, a gather of 16-byte pieces out of hat_conv's row-major weights — and walks the 16 x 16 tiles;
//   * the haloed input tile (18 x 18 pixels x 64 channels, rows of 10 sixteen-byte slots: conflict-free fragment reads) is
//     resident too, the next tile's is fetched into registers during the K loop: NO barrier inside the K loop;
//   * a wave owns two tile rows x all four channel tiles and sweeps the tap ROWS for a fixed (tap column, channel half): the
//     input-row fragment tile row 1 uses for tap row dy is the one tile row 0 needs for dy + 1 (4 row fragments feed 6
//     (tap row, tile row) pairs: 16 fragment reads per 24 MFMAs).
#include <cstdlib>

#include "hat_common.h"

namespace {

constexpr int R_T = 16, R_WAVES = 8, R_HW = R_T + 2, R_NPH = R_HW * R_HW;   // 16 x 16 tile, 18 x 18 haloed pixels
constexpr int R_ROWB = 160;                                                  // bytes per haloed pixel in LDS (64 channels + pad)
constexpr int R_NFRAG = 4 * 18, R_X_OFF = R_NFRAG * 1024;                    // 72 KB of weights, then the input tile
constexpr int R_LDS = R_X_OFF + R_NPH * R_ROWB;                              // 125568 bytes: one workgroup per CU
constexpr int R_NTHR = R_WAVES * 64;

__global__ __launch_bounds__(R_NTHR) void conv64r_kernel(const HatConvDesc d, int nsl, int nwps, int tiles_x, int tiles_y) {
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    typedef bf16_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups are dealt round-robin to the 8 XCDs: the nsl slice-workgroups that walk the same tiles sit on one XCD
    // (consecutive values of blockIdx.x >> 3), so the input tile comes from HBM once and from that XCD's L2 nsl - 1 times
    const int slice = ((int)blockIdx.x >> 3) % nsl;
    const int wgi = (((int)blockIdx.x >> 3) / nsl) * 8 + ((int)blockIdx.x & 7);
    const int H = d.H, W = d.W;
    const int ntiles = tiles_x * tiles_y * d.B;
    if (wgi >= ntiles) return;   // (whole workgroup, before the weight copy is issued)
    const T* xg = reinterpret_cast<const T*>(d.x);

    // ---- weights -> LDS, once: fragment f = ks * 4 + nt = rows 64 slice + 16 nt + c16, k = 32 ks + 8 g of the [N][Kpad] rows ----
    {
        const T* wrow = reinterpret_cast<const T*>(d.w) + (size_t)(slice * 64 + c16) * d.Kpad + 8 * g;
        for (int f = wave; f < R_NFRAG; f += R_WAVES) {
            const int ks = f >> 2, nt = f & 3;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow + (size_t)nt * 16 * d.Kpad + 32 * ks),
                                             (__attribute__((address_space(3))) void*)(smem + f * 1024), 16, 0, 0);
        }
    }
    const int r0 = 2 * wave;
    // B fragment of (haloed row r0 + rr, tap column dx, channel half): lane (c16, g) -> pixel column c16 + dx, channels 32 half + 8 g
    const unsigned xbase = lds0 + R_X_OFF + (unsigned)((r0 * R_HW + c16) * R_ROWB + g * 16);
    const unsigned abase = lds0 + (unsigned)lane * 16u;

    // The next tile's haloed input is fetched into registers BEFORE this tile's K loop and written to LDS after it.
    constexpr int NPIECE = R_NPH * 8, NIT = (NPIECE + R_NTHR - 1) / R_NTHR;
    u32x4 pv[NIT];
    auto fetch = [&](int t) {
        const int tc = min(t, ntiles - 1);
        const int bb = tc / (tiles_x * tiles_y), tr = tc - bb * tiles_x * tiles_y;
        const int ty0 = (tr / tiles_x) * R_T, tx0 = (tr - (tr / tiles_x) * tiles_x) * R_T;
        const T* xb = xg + (size_t)bb * H * W * d.ldx;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = min(tid + it * R_NTHR, NPIECE - 1);
            const int hp = i >> 3, piece = i & 7;
            const int hy = hp / R_HW, hx = hp - hy * R_HW;
            const int y = ty0 - 1 + hy, xx = tx0 - 1 + hx;
            const bool in = y >= 0 && y < H && xx >= 0 && xx < W;
            const u32x4 v = *reinterpret_cast<const u32x4*>(xb + ((size_t)min(max(y, 0), H - 1) * W + min(max(xx, 0), W - 1)) * d.ldx + piece * 8);
            pv[it] = in ? v : u32x4{0u, 0u, 0u, 0u};
        }
    };
    f32x4 biasv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) biasv[nt] = *reinterpret_cast<const f32x4*>(d.bias + slice * 64 + nt * 16 + 4 * g);

    fetch(wgi);
    for (int t = wgi; t < ntiles; t += nwps) {
        const int bb = t / (tiles_x * tiles_y), tr = t - bb * tiles_x * tiles_y;
        const int ty0 = (tr / tiles_x) * R_T, tx0 = (tr - (tr / tiles_x) * tiles_x) * R_T;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * R_NTHR;
            if (i < NPIECE) *reinterpret_cast<u32x4*>(smem + R_X_OFF + (i >> 3) * R_ROWB + (i & 7) * 16) = pv[it];
        }
        __syncthreads();   // (first tile: also drains the weight copy)
        fetch(t + nwps);   // in flight during the K loop (clamped to a valid tile past the end)

        f32x4 acc[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[nt][pt] = biasv[nt];
        auto ldb = [&](int rr, int dx, int half) {
            return __builtin_bit_cast(bf8, *(__attribute__((address_space(3))) const u32x4*)(uintptr_t)(xbase + (unsigned)((rr * R_HW + dx) * R_ROWB + half * 64)));
        };
        auto lda = [&](int ks, int nt) {
            return __builtin_bit_cast(bf8, *(__attribute__((address_space(3))) const u32x4*)(uintptr_t)(abase + (unsigned)((ks * 4 + nt) * 1024)));
        };
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                bf8 prev = ldb(0, dx, half);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const bf8 nx = ldb(dy + 1, dx, half);
                    const int ks = (dy * 3 + dx) * 2 + half;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const bf8 a = lda(ks, nt);
                        acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, prev, acc[nt][0], 0, 0, 0);
                        acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, nx, acc[nt][1], 0, 0, 0);
                    }
                    prev = nx;
                }
            }
        }
        // ---- epilogue: lane holds channels 64 slice + 16 nt + 4g .. +3 of pixel (r0 + pt, c16) -----------------------------
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int y = ty0 + r0 + pt, xx = tx0 + c16;
            if (y < H && xx < W) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = slice * 64 + nt * 16 + 4 * g;
                    f32x4 v = acc[nt][pt];
                    if (d.act == HAT_ACT_LRELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
                    }
                    if (d.out_mode == HAT_O_PIXSHUF_T) {
                        const int r_ = d.ps_r, cps = d.n_store / (r_ * r_);
                        const int ij = n / cps, cc = n - ij * cps;
                        const int i_ = ij / r_, j_ = ij - i_ * r_;
                        const size_t opix = ((size_t)bb * H * r_ + (size_t)y * r_ + i_) * ((size_t)W * r_) + (size_t)xx * r_ + j_;
                        Vec4<T>::store(reinterpret_cast<T*>(d.out) + opix * d.ldo + cc, v);
                    } else {
                        Vec4<T>::store(reinterpret_cast<T*>(d.out) + (((size_t)bb * H + y) * W + xx) * d.ldo + n, v);
                    }
                }
            }
        }
        __syncthreads();   // every wave is done with the input tile before the next one overwrites it
    }
}

}  // namespace

// Does hat_conv route `d` here?  bf16, 3x3, exactly 64 input channels as T rows, a whole number of 64-channel output slices,
// every one stored, plain epilogue (bias, optional LeakyReLU), NHWC or PixelShuffle output.
bool hat_conv64r_can_launch(const HatConvDesc& d) {
    static const bool off = getenv("HAT_NO_CONV64R") != nullptr;
    if (off || d.dtype != HAT_BF16 || d.ksize != 3 || d.Cin != 64 || d.x_mode != HAT_X_NHWC_T || d.x0) return false;
    const int n = d.n_slices * d.nt * 16;
    if (n % 64 || d.n_store != n || d.Kpad < 576 || d.w_bstride) return false;
    if (d.r1 || d.r2 || d.colsum || d.ln_out || d.gap_out || (d.act != HAT_ACT_NONE && d.act != HAT_ACT_LRELU)) return false;
    if (d.out_mode != HAT_O_NHWC_T && d.out_mode != HAT_O_PIXSHUF_T) return false;
    return d.ldx % 8 == 0;
}

int hat_conv64r_launch(const HatConvDesc& d, hipStream_t s) {
    auto kern = conv64r_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
    if (e != hipSuccess) return (int)e;
    const int nsl = d.n_slices * d.nt * 16 / 64;
    const int tiles_x = (d.W + R_T - 1) / R_T, tiles_y = (d.H + R_T - 1) / R_T, ntiles = tiles_x * tiles_y * d.B;
    int m = 32 / nsl;                       // groups of 8 workgroups per slice: 256 workgroups for 4 slices
    if (m < 1) m = 1;
    while (m > 1 && 8 * (m - 1) >= ntiles) --m;   // (small frames: no idle workgroups)
    const int nwps = 8 * m;
    HAT_LAUNCH(kern, dim3(8 * nsl * m), dim3(R_NTHR), R_LDS, s, d, nsl, nwps, tiles_x, tiles_y);
    return hat_check_launch();
}
