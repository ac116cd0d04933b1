// hat_tail3l.hip — hat_hab_tail3 for embed_dim 180 (HAT / HAT-L: BASELINE configs 3, 4, 5), bf16 storage.  Same structure as
// hat_tail3.hip (activation-stationary fc1: LayerNorm2(tB) of the wave's three 16-pixel tiles stays in its registers as B
// fragments; the weights of the current 32 (+32 gate) hidden units are copied into LDS once per workgroup by LDS-DMA and
// shared by the four waves; pipelined stage 0; fp16 hidden tensor, packed-fp16 depthwise conv and gate, fp16 fc2 MFMA), for
//     tB    = t + W_aggr . [y16 | n[16:]] + r2scale * c2 + bias          (hat_arch.py:233-236 with esc_arch.py:123)
//     t_out = tB + fc2( a * SiLU(g) ),   [a | g] = dwconv3x3( fc1( LayerNorm2(tB) ) )      (hat_arch.py:237, :107-119)
// What differs from the embed_dim-144 kernel:
//   * 12 channel tiles (192: channels 180..191 are dead lanes — lane groups 1..3 of tile 11 — masked out of the LayerNorm
//     statistics and of every store), K = 180 in 6 k-steps, whose slot k = 180 carries the fc1 bias;
//   * the CAB of these models squeezes to 60 channels, so its expand conv cannot be folded into this kernel's stage 0 the way
//     HAT-S's 6-channel one is (K = 540): c2 = conv3x3(c1) arrives as a map and enters with its per-sample ECA scale as a
//     vector term of stage 0 (c2 and t in the MFMA D layout, bias / scale / fc2 bias from three 1 KiB LDS records);
//   * the hidden width 360 is padded to 384 = 12 chunks (zero fc1 rows and fc2 columns: u = 0, a * SiLU(g) = 0);
//   * registers: 96 fc2 accumulators + 72 B-fragment registers persist (144: 72 + 60), so the fc1 fragments of a chunk are
//     single-buffered in LDS (the next chunk's are copied in after the barrier that ends phase A: 66 KB instead of 91) and
//     the depthwise phase walks the 3x3 window column by column (3 tap rows of weights and 2 of U live instead of 6 and 4).
#include "hat_common.h"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int L_C = 180, L_NT = 12, L_KS = 6, L_WAVES = 4, L_ROWS = 8, L_HW = 18;
constexpr int L_NPH = (L_ROWS + 2) * L_HW;            // 180 haloed pixels
// LDS of the chunk loop:  [0, 28800) Us (180 rows x 160 B) | W1 24 KiB (4 n-tiles x 6 k-steps) | W2 12 KiB | Wd 2 KiB = 67 712 B
// stage 0:               [0, 73728) the 72 KiB of aggregation A fragments | three 1 KiB records: fc2 bias, aggregation bias, ECA scale
constexpr int L_US_ROWB = 160;
constexpr int L_W1_OFF = L_NPH * L_US_ROWB;           // 28800
constexpr int L_W1_BYTES = 4 * L_KS * 1024;           // 24576
constexpr int L_W2_OFF = L_W1_OFF + L_W1_BYTES;       // 53376
constexpr int L_WD_OFF = L_W2_OFF + L_NT * 1024;      // 65664
constexpr int L_WD_REC = 2048;
constexpr int L_B2_OFF = 73728, L_BIAS_OFF = 74752, L_SCALE_OFF = 75776;
constexpr int L_LDS = 76800;

typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ u32x4 lds_rd16(unsigned addr) { return *(__attribute__((address_space(3))) const u32x4*)(uintptr_t)addr; }
__device__ __forceinline__ h2 as_h2(unsigned v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ unsigned as_u(h2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ void dma1k(const char* src, unsigned lane_off, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane_off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

struct LAggr {
    const bf16_t* n;        // (B,H,W,ldn) LayerNorm1 output
    const bf16_t* y16;      // (B,H,W,16)  ESC large-kernel conv output: replaces channels [0, 16) of n
    const bf16_t* c2;       // (B,H,W,ldc2) CAB expand conv output
    const char* wl;         // aggregation weights, fragment packed [12][6][64][8] bf16
    const float* bias;      // [256] aggregation bias (zero padded 1 KiB record)
    const float* scale;     // [B][scale_bstride] conv_scale * ECA (>= 192 valid floats per sample, 1 KiB readable)
    int ldn, ldc2, scale_bstride;
};

template <int DBG>
__global__ __launch_bounds__(256, 2) void tail3l_kernel(const HatFfnDesc d, const LAggr ag) {
    constexpr int C = L_C, NT = L_NT, KS = L_KS, HALO_W = L_HW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * L_ROWS;
    const int H = d.H, W = d.W;
    const float* tin = d.t_in + (size_t)b * H * W * C;
    const unsigned lane16 = (unsigned)lane * 16u;
    const char* w1g = reinterpret_cast<const char*>(d.w1f);    // [chunk][24 KiB]
    const char* w2g = reinterpret_cast<const char*>(d.w2f);    // [chunk][12 KiB]
    const char* wdg = reinterpret_cast<const char*>(d.dww);    // [chunk][2 KiB]
    // lane groups 1..3 of channel tile 11 hold channels 180..191: not part of the model
    auto live = [&](int nt) { return nt < NT - 1 || g == 0; };

    auto dma_fc1 = [&](int chunk) {    // 24 pieces, 6 per wave
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int f = wave * 6 + j;
            dma1k(w1g + ((size_t)chunk * L_W1_BYTES + f * 1024), lane16, smem + L_W1_OFF + f * 1024);
        }
    };
    auto dma_rest = [&](int chunk) {   // 12 fc2 + 2 depthwise pieces (+ the depthwise ones once more): 4 per wave
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = wave * 4 + j;
            const int fd = (f & 1) ? 1024 : 0;
            const char* src = f < 12 ? w2g + ((size_t)chunk * (NT * 1024) + f * 1024) : wdg + ((size_t)chunk * L_WD_REC + fd);
            char* dst = f < 12 ? smem + L_W2_OFF + f * 1024 : smem + L_WD_OFF + fd;
            dma1k(src, lane16, dst);
        }
    };

    // ------------------------------------------------ stage 0 ------------------------------------------------
    f32x4 acc2[NT][2];   // persistent fc2 accumulators: this wave's two tile rows x 12 channel tiles
    bf8 mb[3][KS];       // LayerNorm2(tB) of this wave's three pixel tiles as fc1 B fragments (k = 180: the ones column)
    int hp[3];
    {
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            const int f = wave * 18 + j;
            dma1k(ag.wl + (size_t)f * 1024, lane16, smem + f * 1024);
        }
        dma1k(reinterpret_cast<const char*>(d.b2), lane16, smem + L_B2_OFF);
        dma1k(reinterpret_cast<const char*>(ag.bias), lane16, smem + L_BIAS_OFF);
        dma1k(reinterpret_cast<const char*>(ag.scale + (size_t)b * ag.scale_bstride), lane16, smem + L_SCALE_OFF);
        __builtin_amdgcn_sched_barrier(0);   // the 21 copies are this wave's OLDEST memory operations
        bool ins[3];
        bf8 bfr[3][KS];
        u32x2 c2r[NT];       // c2 of the tile whose LayerNorm comes next (D layout, bf16 pairs)
        f32x4 accx[NT];
        int pixs[3];
        const bf16_t* nb = ag.n + (size_t)b * H * W * ag.ldn;
        const bf16_t* yb = ag.y16 + (size_t)b * H * W * 16;
        const bf16_t* cb = ag.c2 + (size_t)b * H * W * ag.ldc2;
        auto place = [&](int t) {
            int hy, hx;
            if (t < 2) { hy = 2 * wave + 1 + t; hx = 1 + c16; }
            else if (wave < 2) { hy = wave * (L_ROWS + 1); hx = 1 + c16; }
            else { const int jj = wave == 2 ? c16 : (c16 & 3); hy = (wave == 2 ? 0 : 8) + (jj >> 1); hx = (jj & 1) * (HALO_W - 1); }
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            ins[t] = y >= 0 && y < H && x >= 0 && x < W;
            hp[t] = hy * HALO_W + hx;
            pixs[t] = min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
        };
        auto issue_b = [&](int t) {      // 6 B fragments
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c = ks * 32 + 8 * g;
                // (channels past 180 meet zero weight rows: the clamped read only has to be finite — it is real data)
                const bf16_t* src = c < 16 ? yb + (size_t)pixs[t] * 16 + c : nb + (size_t)pixs[t] * ag.ldn + min(c, ag.ldn - 8);
                bfr[t][ks] = MT<bf16_t>::load(src);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto issue_t = [&](int t) {      // the residual stream straight into the accumulators (MFMA D layout): 12 loads
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch = min(nt * 16 + 4 * g, C - 4);   // dead lanes re-read channels 176..179: finite, masked below
                const f32x4 tv = *reinterpret_cast<const f32x4*>(tin + (size_t)pixs[t] * C + ch);
                if (t < 2) acc2[nt][t] = tv; else accx[nt] = tv;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto issue_c2 = [&](int t) {     // 12 loads of 8 bytes
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                c2r[nt] = *reinterpret_cast<const u32x2*>(cb + (size_t)pixs[t] * ag.ldc2 + min(nt * 16 + 4 * g, C - 4));
            __builtin_amdgcn_sched_barrier(0);
        };
        // The own rows' accumulators exist for the whole kernel, so their residual loads cost no extra registers and are all
        // issued up front; the halo tile (slot 2) goes first through the pipeline, each tile's B fragments are requested one
        // tile ahead and its c2 while its MFMAs run.
        place(2); place(0); place(1);
        issue_b(2); issue_t(2); issue_c2(2);
        issue_t(0); issue_b(0);
        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");   // 69 operations issued: the 21 oldest (the copies) have landed
        lds_barrier();
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const int t = o == 0 ? 2 : o - 1;
            f32x4 v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = t == 2 ? accx[nt] : acc2[nt][t];
            constexpr int RD = 3;
            u32x4 ar[RD];
#pragma unroll
            for (int i = 0; i < RD; ++i) ar[i] = lds_rd16(lds0 + (unsigned)(i * 1024) + lane16);
#pragma unroll
            for (int i = 0; i < NT * KS; ++i) {
                const int nt = i / KS, ks = i - nt * KS;
                v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ar[i % RD]), bfr[t][ks], v[nt], 0, 0, 0);
                if (i + RD < NT * KS) ar[i % RD] = lds_rd16(lds0 + (unsigned)((i + RD) * 1024) + lane16);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // + bias + (conv_scale * ECA) * c2; dead lanes -> 0
            float s = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 bs = __builtin_bit_cast(f32x4, lds_rd16(lds0 + L_BIAS_OFF + (unsigned)(nt * 64 + g * 16)));
                const f32x4 sc = __builtin_bit_cast(f32x4, lds_rd16(lds0 + L_SCALE_OFF + (unsigned)(nt * 64 + g * 16)));
                const u32x2 cr = c2r[nt];
                const f32x4 cv = {__builtin_bit_cast(float, cr[0] << 16), __builtin_bit_cast(float, cr[0] & 0xFFFF0000u),
                                  __builtin_bit_cast(float, cr[1] << 16), __builtin_bit_cast(float, cr[1] & 0xFFFF0000u)};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[nt][r] = live(nt) ? v[nt][r] + bs[r] + sc[r] * cv[r] : 0.f;
                s += (v[nt][0] + v[nt][1]) + (v[nt][2] + v[nt][3]);
                __builtin_amdgcn_sched_barrier(0);   // (or every bias / scale read of the tile is hoisted: 96 registers)
            }
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / (float)C);
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dl = v[nt][r] - mean; q += live(nt) ? dl * dl : 0.f; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            const float rstd = ins[t] ? __builtin_amdgcn_rsqf(q * (1.0f / (float)C) + 1e-5f) : 0.f;
            const float nmr = -mean * rstd;
            typedef bf16_t v4b __attribute__((ext_vector_type(4)));
            u32x2 pk[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const v4b hb = {(bf16_t)(v[nt][0] * rstd + nmr), (bf16_t)(v[nt][1] * rstd + nmr), (bf16_t)(v[nt][2] * rstd + nmr),
                                (bf16_t)(v[nt][3] * rstd + nmr)};
                pk[nt] = __builtin_bit_cast(u32x2, hb);
                if (t < 2) acc2[nt][t] = v[nt] + __builtin_bit_cast(f32x4, lds_rd16(lds0 + L_B2_OFF + (unsigned)(nt * 64 + g * 16)));
                __builtin_amdgcn_sched_barrier(0);
            }
            // channels 180..191: [1.0 (the fc1 bias column, k = 180) if the pixel is inside the image, 0 ...]
            if (g >= 1) pk[NT - 1] = u32x2{(g == 1 && ins[t]) ? 0x00003F80u : 0u, 0u};
            // D layout -> B fragments in registers (hat_tail3.hip): v_permlane32_swap + v_permlane16_swap per n-tile pair
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * ks][0], pk[2 * ks + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * ks][1], pk[2 * ks + 1][1], false, false);
                const auto s0 = __builtin_amdgcn_permlane16_swap(r0[0], r0[1], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(r1[0], r1[1], false, false);
                mb[t][ks] = __builtin_bit_cast(bf8, u32x4{s0[0], s1[0], s0[1], s1[1]});
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next tile in pipeline order is slot o (2 -> 0 -> 1): its c2 now; after the halo tile also the second own
            // row's residual and B fragments (their registers were the halo tile's until here)
            if (o == 0) { issue_t(1); issue_b(1); }
            if (o < 2) issue_c2(o);
        }
        lds_barrier();     // every wave is done with the aggregation weights and the three records
        dma_fc1(0);
    }

    unsigned ust[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) ust[t] = lds0 + (unsigned)(hp[t] * L_US_ROWB + g * 16);
    const unsigned ubase = lds0 + (unsigned)((2 * wave * HALO_W + c16) * L_US_ROWB + g * 16);
    unsigned wdad = lds0 + L_WD_OFF + (unsigned)g * 320u;
    unsigned w2ad = lds0 + L_W2_OFF + lane16;
    unsigned w1b = lds0 + L_W1_OFF + lane16;
    asm volatile("" : "+v"(wdad), "+v"(w2ad), "+v"(w1b));   // (keep the region offsets out of the 16-bit DS offset fields' way)

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();

    for (int chunk = 0; chunk < d.chunks; ++chunk) {
        dma_rest(chunk);
        // ================================ phase A: fc1 -> Us (fp16) ====================================
        {
            f32x4 acc[2][2][3];
            auto store_u = [&](int p) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    u32x4 pkk;
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        pkk[2 * ii] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(acc[p][ii][t][0], acc[p][ii][t][1]));
                        pkk[2 * ii + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(acc[p][ii][t][2], acc[p][ii][t][3]));
                    }
                    *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(ust[t] + (unsigned)(p * 64)) = pkk;
                }
            };
            constexpr int NSTEP = 2 * KS;
            u32x4 ar[2][2];
            auto rd_a = [&](int step) {
                const int p1 = step / KS, k1 = step % KS;
                ar[step & 1][0] = lds_rd16(w1b + (unsigned)(((2 * p1) * KS + k1) * 1024));
                ar[step & 1][1] = lds_rd16(w1b + (unsigned)(((2 * p1 + 1) * KS + k1) * 1024));
            };
            rd_a(0);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int step = p * KS + ks;
                    if (step + 1 < NSTEP) rd_a(step + 1);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const f32x4 c0 = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[p][ii][t];
                            acc[p][ii][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ar[step & 1][ii]), mb[t][ks], c0, 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (p == 1) store_u(0);
            }
            store_u(1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // fc2 fragments and depthwise taps of this chunk have landed
        lds_barrier();     // ... everybody's; Us complete; every wave is done with this chunk's fc1 fragments
        dma_fc1(min(chunk + 1, d.chunks - 1));   // the next chunk's, in flight during phases B and C (the last chunk re-copies itself)

        // ====================== phase B: depthwise 3x3 in packed fp16, one window column at a time ======================
        h2 da[2][4], dg[2][4];
        {
            const u32x4 ba = lds_rd16(wdad + 9 * 32), bg = lds_rd16(wdad + 9 * 32 + 16);   // "tap 9" = depthwise bias
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int k = 0; k < 4; ++k) { da[pt][k] = as_h2(ba[k]); dg[pt][k] = as_h2(bg[k]); }
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            u32x4 wa[3], wg[3], ua[2], ug[2];
#pragma unroll
            for (int tr = 0; tr < 3; ++tr) {
                wa[tr] = lds_rd16(wdad + (unsigned)((tr * 3 + dx) * 32));
                wg[tr] = lds_rd16(wdad + (unsigned)((tr * 3 + dx) * 32 + 16));
            }
            ua[0] = lds_rd16(ubase + (unsigned)(dx * L_US_ROWB));
            ug[0] = lds_rd16(ubase + (unsigned)(dx * L_US_ROWB + 64));
#pragma unroll
            for (int hr = 0; hr < 4; ++hr) {
                if (hr + 1 < 4) {
                    ua[(hr + 1) & 1] = lds_rd16(ubase + (unsigned)(((hr + 1) * HALO_W + dx) * L_US_ROWB));
                    ug[(hr + 1) & 1] = lds_rd16(ubase + (unsigned)(((hr + 1) * HALO_W + dx) * L_US_ROWB + 64));
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 cua = ua[hr & 1], cug = ug[hr & 1];
                if (hr < 3) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        da[0][k] = as_h2(wa[hr][k]) * as_h2(cua[k]) + da[0][k];
                        dg[0][k] = as_h2(wg[hr][k]) * as_h2(cug[k]) + dg[0][k];
                    }
                }
                if (hr >= 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        da[1][k] = as_h2(wa[hr - 1][k]) * as_h2(cua[k]) + da[1][k];
                        dg[1][k] = as_h2(wg[hr - 1][k]) * as_h2(cug[k]) + dg[1][k];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ================================ phase C: gate + fc2 ===================================
        h8 a2[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a2[nt] = __builtin_bit_cast(h8, lds_rd16(w2ad + (unsigned)(nt * 1024)));
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            u32x4 gu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const h2 x = dg[pt][k];
                const h2 tt = x * (h2){(_Float16)-1.4426950408889634f, (_Float16)-1.4426950408889634f};
                h2 e = {(_Float16)__builtin_exp2f16(tt[0]), (_Float16)__builtin_exp2f16(tt[1])};
                e = e + (h2){(_Float16)1.0f, (_Float16)1.0f};
                const h2 r = {(_Float16)__builtin_amdgcn_rcph(e[0]), (_Float16)__builtin_amdgcn_rcph(e[1])};
                gu[k] = as_u(da[pt][k] * (x * r));
            }
            const h8 gf = __builtin_bit_cast(h8, gu);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc2[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[nt], gf, acc2[nt][pt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next chunk's fc1 fragments has landed
        lds_barrier();
    }

    // ----------------------------------- epilogue ------------------------------------------------
    float* tout = d.t_out + (size_t)b * H * W * C;
    const bool do_ln = d.ln1_g != nullptr;
    f32x4 gapv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int y = y0 + 2 * wave + pt, x = x0 + c16;
        const bool valid = y < H && x < W;
        const size_t pix = (size_t)y * W + x;
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = acc2[nt][pt];
            if (valid && live(nt)) *reinterpret_cast<f32x4*>(tout + pix * C + nt * 16 + 4 * g) = v;
            s += live(nt) ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
        }
        if (do_ln) {
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s / (float)C;
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dl = acc2[nt][pt][r] - mean; q += live(nt) ? dl * dl : 0.f; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
            bf16_t* nout = reinterpret_cast<bf16_t*>(d.n_out) + ((size_t)b * H * W + pix) * d.ldn;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch = min(nt * 16 + 4 * g, C - 4);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(d.ln1_g + ch), bt1 = *reinterpret_cast<const f32x4*>(d.ln1_b + ch);
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (acc2[nt][pt][r] - mean) * rstd * g1[r] + bt1[r];
                if (valid && live(nt)) {
                    Vec4<bf16_t>::store(nout + nt * 16 + 4 * g, o);
                    if (nt == 0 && 4 * g < d.gap_c) gapv += as_stored<bf16_t>(o);
                    if (nt == 0 && d.n16_out != nullptr)
                        Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(d.n16_out) + ((size_t)b * H * W + pix) * 16 + 4 * g, o);
                }
            }
        }
    }
    if (do_ln && d.gap_out != nullptr) {
        float* red = reinterpret_cast<float*>(smem + L_WD_OFF);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = gapv[r];
            s = row_sum16(s);
            if (c16 == 0) red[wave * 16 + 4 * g + r] = s;
        }
        __syncthreads();
        if (tid < 16) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < L_WAVES; ++w) s += red[w * 16 + tid];
            const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            d.gap_out[((size_t)b * gridDim.x * gridDim.y + tile) * 16 + tid] = tid < d.gap_c ? s : 0.f;
        }
    }
}

}  // namespace

// hat_hab_tail3 for embed_dim 180 (called from hat_tail3.hip's entry point)
int hat_tail3_launch_c180(const HatHabTailDesc& h, void* stream) {
    const HatFfnDesc& d = h.ffn;
    if (!h.n || !h.y16 || !h.r2 || !h.r2scale || !h.w_aggr || !h.bias_b || h.ldn_in < L_C || h.ldn_in % 8 || h.ldr2 < L_C || h.ldr2 % 4 ||
        h.r2scale_bstride < 192)
        return HAT_EINVAL;
    if (d.ln1_g && (!d.ln1_b || !d.n_out || d.ldn < d.C || d.ldn % 4 || d.gap_c < 0 || d.gap_c > 16 || d.gap_c % 4)) return HAT_EINVAL;
    const LAggr ag{reinterpret_cast<const bf16_t*>(h.n), reinterpret_cast<const bf16_t*>(h.y16), reinterpret_cast<const bf16_t*>(h.r2),
                   reinterpret_cast<const char*>(h.w_aggr), h.bias_b, h.r2scale, h.ldn_in, h.ldr2, h.r2scale_bstride};
    auto kern = tail3l_kernel<0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L_LDS);
    if (e != hipSuccess) return (int)e;
    dim3 grid((d.W + 15) / 16, (d.H + L_ROWS - 1) / L_ROWS, d.B);
    HAT_LAUNCH(kern, grid, dim3(256), L_LDS, reinterpret_cast<hipStream_t>(stream), d, ag);
    return hat_check_launch();
}
