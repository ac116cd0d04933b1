// hat_tail3.hip — third-generation fused HAB tail for embed_dim 144, bf16 storage (contract: hat_hab_tail3 in
// include/hat_mi355x.h).  Same arithmetic, tile geometry (8 x 16 pixels, four waves, two workgroups per CU) and outputs as
// hat_hab_tail (hat_ffn2.hip):
//     tB    = t + W_aggr . [y16 | n[16:]] + wf . im2col3x3(c1) + bias_b
//     t_out = tB + fc2( a * SiLU(g) ),   [a | g] = dwconv3x3( fc1( LayerNorm2(tB) ) )
// reference: hat/archs/hat_arch.py:233-237 with GatedDconvFFN.forward :107-119 and esc_arch.py:123.
//
// What changed against the second generation, and why (its waves streamed every fc1 fragment twice and
// every fc2 fragment four times per tile through L1/L2 — 1.46 MB per tile, 10.5 GB per launch):
//   * ACTIVATION-stationary fc1.  LayerNorm2(tB) of the haloed tile no longer lives in LDS (51.8 KB of "Ms") but in the
//     registers of the wave that computed it in stage 0: every wave keeps the B fragments of ITS three 16-pixel tiles (its
//     two tile rows + one of the four halo tiles: 3 x 5 k-steps x 4 registers = 60) for all nine chunks.  The LDS that
//     frees holds the WEIGHTS of the current chunk instead, copied in ONCE per workgroup by LDS-DMA (no registers, no
//     vmcnt-ordered register loads in the loop) and read by all four waves: fc1 20 KB (double buffered: chunk c + 1 lands
//     while chunk c computes), fc2 9 KB and the depthwise taps 1.25 KB.  Per tile the waves now pull 9 x 30.25 KB + 72 KB
//     = 344 KB of weights through L2 instead of 1 460 KB.
//   * One A-fragment read (ds_read_b128) feeds THREE MFMAs (the wave's three pixel tiles) — 20 + 9 operand reads per chunk
//     and wave where the second generation read 30 B fragments from Ms and 19 fragments from global memory.
//   * The fc1 bias is a k-slot again (k = 144, "ones column" of the B fragment; the fifth k-step was half empty anyway):
//     a pixel outside the image has an all-zero B row INCLUDING that slot, so its U is exactly zero — what the
//     depthwise conv's zero padding needs (hat_arch.py:112-114 pads u AFTER the bias) — without the 32 selects per chunk
//     that forced U to zero, and without a bias operand.
// Everything downstream of U (packed-fp16 depthwise conv, gate, fp16 fc2 MFMA, epilogue with the next LayerNorm, its GAP
// partials and compact 16-channel copy) is the second generation's.
#include "hat_common.h"

// compile-time experiment switches (tools/ubench_tail3.hip builds the variants; the library builds the defaults)
#ifndef T3_ISSUE_ALL
#define T3_ISSUE_ALL 0      // 1: all three tiles' stage-0 loads up front (0: the third tile's after the first tile's MFMAs)
#endif
#ifndef T3_APF
#define T3_APF 1            // fc1 A-fragment prefetch distance in k-steps (two fragments per step)
#endif
#ifndef T3_S0RD
#define T3_S0RD 8           // stage 0: A-fragment read distance in MFMAs
#endif
#ifndef T3_ABL
#define T3_ABL 0            // timing ablations for tools/ubench_tail3.hip (results wrong): 1 no weight copies in the chunk loop,
#endif                      // 2 no U stores, 4 fc1 A fragments read once per chunk, 8 no MFMAs in phase A
#ifndef T3_USTORE_IL
#define T3_USTORE_IL 1      // 1: pair 0's U stores interleaved with pair 1's MFMAs (0: after them)
#endif
#ifndef T3_DMA_LATE
#define T3_DMA_LATE 0       // 1: the next chunk's fc1 copies are issued after the barrier that ends phase A instead of at the chunk top
#endif

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr int T3_C = 144, T3_NT = 9, T3_KS = 5, T3_WAVES = 4, T3_ROWS = 8, T3_HW = 18;
constexpr int T3_NPH = (T3_ROWS + 2) * T3_HW;        // 180 haloed pixels
// LDS map of the chunk loop (81 024 B, two workgroups per CU):
//   [0, 28800)        Us   180 rows x 160 B: 64 fp16 channels [a 0..31 | gate 0..31] + 32 B pad (10 slots = 2 mod 4:
//                          conflict-free depthwise operand reads)
//   [28800, 69760)    W1   2 x 20 480 B: fc1 A fragments [4 n-tiles][5 k-steps][1 KiB] of chunk c (buffer c & 1)
//   [69760, 78976)    W2   fc2 A fragments [9 n-tiles][1 KiB] of the current chunk
//   [78976, 81024)    Wd   depthwise taps of the current chunk [g][tap 0..8, bias][a-units 8 | gate-units 8] fp16 (1 280 B
//                          of a 2 KiB record); the epilogue's GAP reduction scratch afterwards
// Stage 0 borrows [0, 73728) for the 72 KiB of aggregation / folded-CAB A fragments, as in the second generation.
constexpr int T3_US_ROWB = 160;
constexpr int T3_W1_OFF = T3_NPH * T3_US_ROWB;        // 28800
constexpr int T3_W1_BYTES = 4 * T3_KS * 1024;         // 20480
constexpr int T3_W2_OFF = T3_W1_OFF + 2 * T3_W1_BYTES;   // 69760
constexpr int T3_WD_OFF = T3_W2_OFF + T3_NT * 1024;   // 78976
constexpr int T3_WD_REC = 2048;                       // bytes per chunk of the depthwise record in global memory
constexpr int T3_LDS = T3_WD_OFF + T3_WD_REC;         // 81024
constexpr int T3_B2_OFF = 73728;                      // stage 0 only: 1 KiB record of the fc2 bias behind the 72 KiB of A operands

typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ u32x4 lds_rd16(unsigned addr) { return *(__attribute__((address_space(3))) const u32x4*)(uintptr_t)addr; }
__device__ __forceinline__ h2 as_h2(unsigned v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ unsigned as_u(h2 v) { return __builtin_bit_cast(unsigned, v); }
// one LDS-DMA piece: 64 lanes x 16 B from src (wave-uniform) + lane_off -> 1 KiB at dst (wave-uniform)
__device__ __forceinline__ void dma1k(const char* src, unsigned lane_off, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane_off),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

struct T3Aggr {
    const bf16_t* n;       // (B,H,W,ldn) LayerNorm1 output
    const bf16_t* y16;     // (B,H,W,16)  ESC large-kernel conv output: replaces channels [0, 16) of n
    const bf16_t* c1;      // (B,H,W,8)   CAB squeeze conv output
    const char* wl;        // aggregation weights, fragment packed [9][5][64][8] bf16
    const char* wf;        // per-sample folded CAB expand weights [B][9][3][64][8] bf16 (hat_cab_fold)
    const float* bias_b;   // per-sample bias [B][144]
    int ldn;
};
__device__ __attribute__((aligned(16))) unsigned hat_tail3_zero_page[4] = {0, 0, 0, 0};
__device__ __attribute__((aligned(16))) unsigned hat_tail3_ones_page[4] = {0x3F803F80u, 0, 0, 0};   // bf16 {1, 1, 0 ...}

// DBG 64: per-phase s_memtime totals of every wave -> gap_out[wg][wave][12] (tools/ubench_tail3.hip; the library builds 0)
// TH: the residual stream's type — bit 0: t_in is FP16 rows, bit 1: t_out is FP16 rows (else fp32).  Between two HABs of a group
// the stream is read and written by this kernel only; 16 bits there take 4C of a pixel's 12.3C bytes away at 0.1 dB
// (profiles/r03_residual16_emulation.txt).  FP16, not bf16 (2.9 dB); values are clamped to the finite FP16 range on the way out.
template <int DBG, int TH = 0>
__global__ __launch_bounds__(256, 2) void tail3_kernel(const HatFfnDesc d, const T3Aggr ag) {
    constexpr int C = T3_C, NT = T3_NT, KS = T3_KS, NPH = T3_NPH, HALO_W = T3_HW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * T3_ROWS;
    const int H = d.H, W = d.W;
    const float* tin = d.t_in + (size_t)b * H * W * C;
    const _Float16* tinh = reinterpret_cast<const _Float16*>(d.t_in) + (size_t)b * H * W * C;

    long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = 0;
    auto stamp = [&](int slot) {
        if constexpr (DBG & 64) {
            const long long now = (long long)__builtin_amdgcn_s_memtime();
            tph[slot] += now - tlast;
            tlast = now;
        }
    };
    if constexpr (DBG & 64) tlast = (long long)__builtin_amdgcn_s_memtime();

    // weight records of a chunk in global memory (uniform bases; the lane's 16-byte piece is the only per-lane part of a
    // copy's address, so the copies use the SGPR-base + VGPR-offset form and cost no 64-bit vector address arithmetic)
    const char* w1g = reinterpret_cast<const char*>(d.w1f);    // [chunk][20 KiB]
    const char* w2g = reinterpret_cast<const char*>(d.w2f);    // [chunk][9 KiB]
    const char* wdg = reinterpret_cast<const char*>(d.dww);    // [chunk][2 KiB]
    const unsigned lane16 = (unsigned)lane * 16u;
    // this wave's share of the weight copies: 5 of the 20 fc1 pieces, 3 of the 12 others (9 fc2 + 2 depthwise + the first
    // depthwise piece once more: every wave issues the SAME number of copies per chunk, so one vmcnt immediate serves all)
    auto dma_fc1 = [&](int chunk, int buf) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int f = wave * 5 + j;
            dma1k(w1g + ((size_t)chunk * T3_W1_BYTES + f * 1024), lane16, smem + T3_W1_OFF + buf * T3_W1_BYTES + f * 1024);
        }
    };
    auto dma_rest = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int f = wave * 3 + j;   // 0..8 fc2, 9..10 depthwise, 11 = 9 again
            const int fd = f == 10 ? 1024 : 0;
            const char* src = f < 9 ? w2g + ((size_t)chunk * (NT * 1024) + f * 1024) : wdg + ((size_t)chunk * T3_WD_REC + fd);
            char* dst = f < 9 ? smem + T3_W2_OFF + f * 1024 : smem + T3_WD_OFF + fd;
            dma1k(src, lane16, dst);
        }
    };

    // ---------------- stage 0: aggregation + folded CAB + both residual terms + LayerNorm2 -> B fragments in registers ----------------
    // A three-deep pipeline over the wave's three pixel tiles.  The second generation issued every load, waited for all of
    // them, ran 216 MFMAs and then three LayerNorms: its stamps (profiles/r03_ubench_ffn2_stage0.txt) showed the ISSUE of
    // the ~70 KB per wave taking 21.6 k of the stage's 45 k cycles — the CU's memory path at its fair share of HBM — with
    // nothing else running, then 11.4 k for MFMAs that waited on one LDS fragment read each, then 9.6 k of LayerNorm.
    // Here the loads of tiles 1 and 2 are still landing while tile 0's MFMAs and LayerNorm run, and so on.
    f32x4 acc2[NT][2];   // persistent fc2 accumulators: this wave's two tile rows x 9 channel tiles
    bf8 mb[3][KS];       // LayerNorm2(tB) of this wave's three pixel tiles as fc1 B fragments (k = 144: the ones column)
    int hp[3];
    {
        // All 80 KiB of LDS are free here: the 72 KiB of A operands (9 n-tiles x 8 k-steps: 5 of the 144-wide aggregation,
        // 3 of the 72-deep im2col of c1 — whose k-slots 72, 73 carry the per-sample bias, hat_cab_fold) are copied in once
        // per workgroup by LDS-DMA, + 1 KiB of fc2 bias (every wave copies it: same count of copies per wave).
        {
            const char* wfb = ag.wf + (size_t)b * NT * 3 * 1024;
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                const int f = wave * 18 + j, nt = f >> 3, ks = f & 7;
                const char* src = ks < 5 ? ag.wl + (size_t)(nt * 5 + ks) * 1024 : wfb + (size_t)(nt * 3 + ks - 5) * 1024;
                dma1k(src, lane16, smem + f * 1024);
            }
            dma1k(reinterpret_cast<const char*>(d.b2), lane16, smem + T3_B2_OFF);
        }
        __builtin_amdgcn_sched_barrier(0);   // the 19 copies are this wave's OLDEST memory operations (vmcnt below counts on it)
        bool ins[3];
        bf8 bfr[3][8];
        f32x4 accx[NT];
        typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
        u32x2h thr[3][TH & 1 ? NT : 1];   // TH & 1: the FP16 residual rows of the wave's three pixel tiles, raw
        const bf16_t* nb = ag.n + (size_t)b * H * W * ag.ldn;
        const bf16_t* yb = ag.y16 + (size_t)b * H * W * 16;
        const bf16_t* cb = ag.c1 + (size_t)b * H * W * 8;
        // Tile order of the pipeline: the halo tile first (slot t = 2), then the two own rows.  17 loads per tile: 8 B
        // fragments and the residual t straight into the accumulators in the MFMA D layout — no VALU touches them before the
        // MFMAs, so nothing waits for them early.  Two tiles' loads are in flight at a time (the third tile's are issued when
        // the first tile's MFMAs have consumed its B fragments: all three at once is 204 live registers and spills).
        auto issue_tile = [&](int t) {
            int hy, hx;
            if (t < 2) { hy = 2 * wave + 1 + t; hx = 1 + c16; }
            else if (wave < 2) { hy = wave * (T3_ROWS + 1); hx = 1 + c16; }
            else { const int jj = wave == 2 ? c16 : (c16 & 3); hy = (wave == 2 ? 0 : 8) + (jj >> 1); hx = (jj & 1) * (HALO_W - 1); }
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            ins[t] = y >= 0 && y < H && x >= 0 && x < W;
            const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
            hp[t] = hy * HALO_W + hx;
            const int pix = yc * W + xc;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int c = ks * 32 + 8 * g;
                const bf16_t* src = c < 16 ? yb + (size_t)pix * 16 + c : nb + (size_t)pix * ag.ldn + min(c, C - 8);
                bfr[t][ks] = MT<bf16_t>::load(src);
            }
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) {
                const int tap = 4 * kc + g;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int yy = yc + dy, xx = xc + dx;
                const bool inb = tap < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W;
                // "tap 9" (k = 72..79): 1.0 in the two slots that hold the bias head and remainder
                const bf16_t* cst = reinterpret_cast<const bf16_t*>(tap == 9 ? hat_tail3_ones_page : hat_tail3_zero_page);
                bfr[t][5 + kc] = MT<bf16_t>::load(inb ? cb + ((size_t)yy * W + xx) * 8 : cst);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (TH & 1) {   // 4 halves: they wait as raw bits until their tile's turn (converted in front of its MFMAs)
                    thr[t][nt] = *reinterpret_cast<const u32x2h*>(tinh + (size_t)pix * C + nt * 16 + 4 * g);
                } else {
                    const f32x4 tv = *reinterpret_cast<const f32x4*>(tin + (size_t)pix * C + nt * 16 + 4 * g);
                    if (t < 2) acc2[nt][t] = tv; else accx[nt] = tv;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        issue_tile(2);
        issue_tile(0);
        if constexpr (T3_ISSUE_ALL) issue_tile(1);
        stamp(8);
        // 53 (70) operations issued: the 19 oldest (the copies) have landed
        if constexpr (T3_ISSUE_ALL) asm volatile("s_waitcnt vmcnt(51)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
        lds_barrier();     // ... and everybody's have
        stamp(9);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const int t = o == 0 ? 2 : o - 1;
            // 72 MFMAs against the resident A operands, fragment reads three ahead
            f32x4 v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (TH & 1) {
                    const h2 lo = as_h2(thr[t][nt][0]), hi = as_h2(thr[t][nt][1]);
                    v[nt] = f32x4{(float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]};
                } else {
                    v[nt] = t == 2 ? accx[nt] : acc2[nt][t];
                }
            }
            // (A fragment reads T3_S0RD MFMAs ahead of their use, the order pinned: left to the scheduler each read is sunk to
            // just before its MFMA — one LDS latency per MFMA, 22 k cycles for the 216 of them in the first build)
            constexpr int RD = T3_S0RD;
            u32x4 ar[RD];
#pragma unroll
            for (int i = 0; i < RD; ++i) ar[i] = lds_rd16(lds0 + (unsigned)(i * 1024) + lane16);
#pragma unroll
            for (int i = 0; i < NT * 8; ++i) {
                const int nt = i >> 3, ks = i & 7;
                v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ar[i % RD]), bfr[t][ks], v[nt], 0, 0, 0);
                if (i + RD < NT * 8) ar[i % RD] = lds_rd16(lds0 + (unsigned)((i + RD) * 1024) + lane16);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (o == 0 && !T3_ISSUE_ALL) issue_tile(1);   // (into the registers the halo tile's B fragments just left)
            // LayerNorm2 WITHOUT its affine part (ops.pack_ffn3 folds gamma into the fc1 columns and W1.beta into the fc1
            // bias): fp32 statistics over the 4 lane groups of a pixel
            float s = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) s += (v[nt][0] + v[nt][1]) + (v[nt][2] + v[nt][3]);
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / (float)C);
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dl = v[nt][r] - mean; q += dl * dl; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            // a pixel outside the image: rstd = 0 -> an all-zero row (and no ones column below)
            const float rstd = ins[t] ? __builtin_amdgcn_rsqf(q * (1.0f / (float)C) + 1e-5f) : 0.f;
            const float nmr = -mean * rstd;
            typedef bf16_t v4b __attribute__((ext_vector_type(4)));
            u32x2 pk[NT + 1];   // the normalised row as packed bf16 in the D layout: lane (c16, g) <-> channels 16 nt + 4 g ..+3
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const v4b hb = {(bf16_t)(v[nt][0] * rstd + nmr), (bf16_t)(v[nt][1] * rstd + nmr), (bf16_t)(v[nt][2] * rstd + nmr),
                                (bf16_t)(v[nt][3] * rstd + nmr)};
                pk[nt] = __builtin_bit_cast(u32x2, hb);
                // the wave's own rows: tB + fc2 bias is where the fc2 accumulation starts
                if (t < 2) acc2[nt][t] = v[nt] + __builtin_bit_cast(f32x4, lds_rd16(lds0 + T3_B2_OFF + (unsigned)(nt * 64 + g * 16)));
            }
            // "n-tile 9" = channels 144..159: [1.0 (the fc1 bias column) if the pixel is inside the image, 0 ...]
            pk[NT] = u32x2{(g == 0 && ins[t]) ? 0x00003F80u : 0u, 0u};
            // D layout -> B fragments (lane (c16, g'): channels 32 ks + 8 g' ..+7) entirely in registers: for the n-tile pair
            // (E, O) = (2 ks, 2 ks + 1), v_permlane32_swap moves O's lower half-wave under E's upper one, v_permlane16_swap
            // then interleaves the 16-lane rows — 4 instructions per k-step instead of a round trip through LDS, and no
            // scratch space: the layout change happens here, inside the pipeline, while the next tile's loads are landing.
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * ks][0], pk[2 * ks + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * ks][1], pk[2 * ks + 1][1], false, false);
                const auto s0 = __builtin_amdgcn_permlane16_swap(r0[0], r0[1], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(r1[0], r1[1], false, false);
                mb[t][ks] = __builtin_bit_cast(bf8, u32x4{s0[0], s1[0], s0[1], s1[1]});
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(10);
        lds_barrier();     // every wave is done with the aggregation weights: the LDS map of the chunk loop takes over
        stamp(11);
        dma_fc1(0, 0);     // chunk 0's fc1 fragments
    }

    // Us store position of pixel tile t, pair p (n-tiles 2p, 2p + 1): ONE 16-byte store holds the lane's 4 + 4 results, so
    // fc1's output row (tile ii, 4g + r) is hidden unit 8g + 4ii + r of the half (ops.pack_ffn3 orders the fc1 rows so)
    unsigned ust[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) ust[t] = lds0 + (unsigned)(hp[t] * T3_US_ROWB + g * 16);
    // depthwise operands: haloed rows 2*wave + hr (hr = 0..3), columns c16 + dx; a-slot g, gate-slot 4 + g
    const unsigned ubase = lds0 + (unsigned)((2 * wave * HALO_W + c16) * T3_US_ROWB + g * 16);
    unsigned wdad = lds0 + T3_WD_OFF + (unsigned)g * 320u;   // this lane group's [tap][32 bytes]
    unsigned w2ad = lds0 + T3_W2_OFF + lane16;
    // (opaque to the optimiser: it would otherwise re-derive every address as (lane part) + a constant beyond the 16-bit
    // offset field of the DS instructions — one v_add_u32 per read, 29 per chunk)
    asm volatile("" : "+v"(wdad), "+v"(w2ad));

    stamp(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of chunk 0's fc1 fragments has landed
    lds_barrier();     // ... everybody's has
    stamp(1);

    for (int chunk = 0; chunk < d.chunks; ++chunk) {
        // weights: fc2 + depthwise taps of THIS chunk (needed after the next barrier), fc1 of the next chunk (needed after the
        // barrier that ends this one).  Their buffers' last readers passed the barrier that ended the previous chunk.
        if constexpr (!(T3_ABL & 33)) dma_rest(chunk);
        if constexpr (!T3_DMA_LATE && !(T3_ABL & 33)) dma_fc1(min(chunk + 1, d.chunks - 1), (chunk + 1) & 1);
        // ================================ phase A: fc1 -> Us (fp16) ====================================
        {
            const unsigned w1b = lds0 + T3_W1_OFF + (unsigned)((chunk & 1) * T3_W1_BYTES) + lane16;
            f32x4 acc[2][2][3] = {};
            auto store_u1 = [&](int p, int t) {
                u32x4 pk;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    pk[2 * ii] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(acc[p][ii][t][0], acc[p][ii][t][1]));
                    pk[2 * ii + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(acc[p][ii][t][2], acc[p][ii][t][3]));
                }
                *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(ust[t] + (unsigned)(p * 64)) = pk;
            };
            auto store_u = [&](int p) {
#pragma unroll
                for (int t = 0; t < 3; ++t) store_u1(p, t);
            };
            // A fragments of step s = (pair p, k-step ks): two per step, requested T3_APF steps ahead of their MFMAs
            constexpr int APF = T3_APF, NSTEP = 2 * KS;
            u32x4 ar[APF + 1][2];
            auto rd_a = [&](int step) {
                const int p1 = step / KS, k1 = step % KS;
                ar[step % (APF + 1)][0] = lds_rd16(w1b + (unsigned)(((2 * p1) * KS + k1) * 1024));
                ar[step % (APF + 1)][1] = lds_rd16(w1b + (unsigned)(((2 * p1 + 1) * KS + k1) * 1024));
            };
#pragma unroll
            for (int i = 0; i < APF; ++i) rd_a(i);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int step = p * KS + ks;
                    if (step + APF < NSTEP && !((T3_ABL & 4) && step + APF >= 1 + APF)) rd_a(step + APF);
                    if constexpr (!(T3_ABL & 8))
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const f32x4 c0 = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[p][ii][t];
                            acc[p][ii][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ar[step % (APF + 1)][ii]), mb[t][ks], c0, 0, 0, 0);
                        }
                        // pair 0's results leave (4 conversions + one 16-byte LDS store per pixel tile) in the shadow of pair 1's
                        // MFMAs, one pixel tile per k-step: an MFMA holds the vector issue for half of its 16 cycles only
                        if (T3_USTORE_IL && p == 1 && ii == 0 && ks >= 1 && ks <= 3 && !(T3_ABL & 18)) store_u1(0, ks - 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!T3_USTORE_IL && p == 1 && !(T3_ABL & 18)) store_u(0);
            }
            if constexpr (!(T3_ABL & 18)) store_u(1);
            if constexpr ((T3_ABL & 16) != 0) {
#pragma unroll
                for (int q = 0; q < 12; ++q) asm volatile("" :: "v"(acc[q / 6][(q / 3) & 1][q % 3]));
            }
        }
        stamp(2);
        // all but the 5 youngest copies (next chunk's fc1) have landed
        if constexpr (T3_DMA_LATE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        lds_barrier();     // Us, fc2 fragments and depthwise taps complete
        stamp(3);
        if constexpr (T3_DMA_LATE) dma_fc1(min(chunk + 1, d.chunks - 1), (chunk + 1) & 1);
        if constexpr ((T3_ABL & 32) != 0) { dma_rest(chunk); dma_fc1(min(chunk + 1, d.chunks - 1), (chunk + 1) & 1); }

        // ====================== phase B: depthwise 3x3 in packed fp16 (this wave's two rows) ======================
        h2 da[2][4], dg[2][4];   // [tile row][dword]: a-units / gate-units 8g..8g+7, two per dword
        {
            const u32x4 ba = lds_rd16(wdad + 9 * 32), bg = lds_rd16(wdad + 9 * 32 + 16);   // "tap 9" = depthwise bias
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int k = 0; k < 4; ++k) { da[pt][k] = as_h2(ba[k]); dg[pt][k] = as_h2(bg[k]); }
        }
        {
            // 12 positions (haloed row hr = 0..3, column offset dx = 0..2), software-pipelined by hand one row ahead: left to
            // itself the scheduler hoists all 44 LDS reads (176 registers) above the first FMA and spills.  Position (hr, dx)
            // feeds tile row 0 with tap (hr, dx) and tile row 1 with tap (hr - 1, dx); the weights of tap (hr + 1, dx) replace
            // those of (hr - 1, dx) as soon as this position is done.
            constexpr int PD = 3;
            u32x4 wa[2][3], wg[2][3], ua[PD + 1], ug[PD + 1];
            auto rd_w = [&](int tr, int dx) {
                wa[tr & 1][dx] = lds_rd16(wdad + (unsigned)((tr * 3 + dx) * 32));
                wg[tr & 1][dx] = lds_rd16(wdad + (unsigned)((tr * 3 + dx) * 32 + 16));
            };
            auto rd_u = [&](int pos) {
                const int hr = pos / 3, dx = pos - 3 * hr;
                ua[pos % (PD + 1)] = lds_rd16(ubase + (unsigned)((hr * HALO_W + dx) * T3_US_ROWB));
                ug[pos % (PD + 1)] = lds_rd16(ubase + (unsigned)((hr * HALO_W + dx) * T3_US_ROWB + 64));
            };
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) rd_w(0, dx);
#pragma unroll
            for (int p = 0; p < PD; ++p) rd_u(p);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) rd_w(1, dx);
#pragma unroll
            for (int pos = 0; pos < 12; ++pos) {
                const int hr = pos / 3, dx = pos - 3 * hr;
                if (pos + PD < 12) rd_u(pos + PD);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 cua = ua[pos % (PD + 1)], cug = ug[pos % (PD + 1)];
                if (hr < 3) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        da[0][k] = as_h2(wa[hr & 1][dx][k]) * as_h2(cua[k]) + da[0][k];
                        dg[0][k] = as_h2(wg[hr & 1][dx][k]) * as_h2(cug[k]) + dg[0][k];
                    }
                }
                if (hr >= 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        da[1][k] = as_h2(wa[(hr - 1) & 1][dx][k]) * as_h2(cua[k]) + da[1][k];
                        dg[1][k] = as_h2(wg[(hr - 1) & 1][dx][k]) * as_h2(cug[k]) + dg[1][k];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (hr >= 1 && hr + 1 < 3) rd_w(hr + 1, dx);
            }
        }
        stamp(4);
        // ================================ phase C: gate + fc2 ===================================
        h8 a2[NT];  // fc2 fragments of the chunk: requested now, consumed after the gate math
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
                gu[k] = as_u(da[pt][k] * (x * r));                  // a * g * sigmoid(g)
            }
            const h8 gf = __builtin_bit_cast(h8, gu);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc2[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[nt], gf, acc2[nt][pt], 0, 0, 0);
        }
        stamp(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next chunk's fc1 fragments has landed
        lds_barrier();  // every wave is done reading Us / W2 / Wd before the next chunk overwrites them
        stamp(6);
    }

    // ----------------------------------- epilogue ------------------------------------------------
    float* tout = d.t_out + (size_t)b * H * W * C;
    const bool do_ln = d.ln1_g != nullptr;
    f32x4 gapv = {0.f, 0.f, 0.f, 0.f};
    f32x4 g1v[NT], bt1v[NT];
    if (do_ln) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            g1v[nt] = *reinterpret_cast<const f32x4*>(d.ln1_g + nt * 16 + 4 * g);
            bt1v[nt] = *reinterpret_cast<const f32x4*>(d.ln1_b + nt * 16 + 4 * g);
        }
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int y = y0 + 2 * wave + pt, x = x0 + c16;
        const bool valid = y < H && x < W;
        const size_t pix = (size_t)y * W + x;
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = acc2[nt][pt];
            if constexpr (!(TH & 2)) { if (valid) *reinterpret_cast<f32x4*>(tout + pix * C + nt * 16 + 4 * g) = v; }
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        if constexpr ((TH & 2) != 0) {   // FP16 rows: n-tile pairs as 16-byte stores, the last tile as an 8-byte one
            _Float16* trow = reinterpret_cast<_Float16*>(d.t_out) + ((size_t)b * H * W + pix) * C;
#pragma unroll
            for (int nt = 0; nt + 1 < NT; nt += 2) store_pair_f16_if(trow, nt * 16, g, acc2[nt][pt], acc2[nt + 1][pt], valid);
            if (valid) {
                typedef _Float16 v4h __attribute__((ext_vector_type(4)));
                const f32x4 v = acc2[NT - 1][pt];
                auto c = [](float x) { return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); };
                *reinterpret_cast<v4h*>(trow + (NT - 1) * 16 + 4 * g) = v4h{c(v[0]), c(v[1]), c(v[2]), c(v[3])};
            }
        }
        if (do_ln) {  // LayerNorm (eps 1e-5) of the finished pixel for the next block; 4 lane groups share a pixel
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s / (float)C;
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dl = acc2[nt][pt][r] - mean; q += dl * dl; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
            bf16_t* nout = reinterpret_cast<bf16_t*>(d.n_out) + ((size_t)b * H * W + pix) * d.ldn;
            auto lnv = [&](int nt) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (acc2[nt][pt][r] - mean) * rstd * g1v[nt][r] + bt1v[nt][r];
                return o;
            };
            // the bf16 rows leave as 16-byte stores of n-tile pairs (store_pair_bf16: 64 contiguous bytes per pixel and
            // instruction instead of 32) when the rows are 16-byte aligned; tile 8 and the compact copy as 8-byte stores
            const bool pair16 = d.ldn % 8 == 0 && reinterpret_cast<uintptr_t>(d.n_out) % 16 == 0;   // (uniform)
            {
                const f32x4 o0 = lnv(0);
                if (valid) {
                    if (4 * g < d.gap_c) gapv += as_stored<bf16_t>(o0);
                    if (d.n16_out != nullptr)   // compact copy of channels 0..15 for the next block's ESC conv
                        Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(d.n16_out) + ((size_t)b * H * W + pix) * 16 + 4 * g, o0);
                }
                if (pair16) {
                    store_pair_bf16_if(nout, 0, g, o0, lnv(1), valid);
#pragma unroll
                    for (int nt = 2; nt + 1 < NT; nt += 2) store_pair_bf16_if(nout, nt * 16, g, lnv(nt), lnv(nt + 1), valid);
                    if (valid) Vec4<bf16_t>::store(nout + (NT - 1) * 16 + 4 * g, lnv(NT - 1));
                } else if (valid) {
                    Vec4<bf16_t>::store(nout + 4 * g, o0);
#pragma unroll
                    for (int nt = 1; nt < NT; ++nt) Vec4<bf16_t>::store(nout + nt * 16 + 4 * g, lnv(nt));
                }
            }
        }
    }
    if constexpr (DBG & 64) {
        stamp(7);
        if (lane == 0) {
            const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int i = 0; i < 12; ++i) d.gap_out[(wg * T3_WAVES + wave) * 12 + i] = (float)tph[i];
        }
        return;
    }
    if (do_ln && d.gap_out != nullptr) {
        float* red = reinterpret_cast<float*>(smem + T3_WD_OFF);
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
            for (int w = 0; w < T3_WAVES; ++w) s += red[w * 16 + tid];
            const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            d.gap_out[((size_t)b * gridDim.x * gridDim.y + tile) * 16 + tid] = tid < d.gap_c ? s : 0.f;
        }
    }
}

}  // namespace

#ifndef HAT_TAIL3_NO_ENTRY
int hat_tail3_launch_c180(const HatHabTailDesc& h, void* stream);   // hat_tail3l.hip

extern "C" int hat_hab_tail3(const HatHabTailDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatHabTailDesc& h = *dp;
    const HatFfnDesc& d = h.ffn;
    if (!d.t_in || !d.t_out || d.t_in == d.t_out || !d.w1f || !d.dww || !d.w2f || !d.b2) return HAT_EINVAL;
    if (d.B < 1 || d.H < 1 || d.W < 1 || d.chunks < 1 || d.m_in) return HAT_EINVAL;
    if (d.dtype != HAT_BF16) return HAT_EUNSUPPORTED;
    if (d.C == 180) return (d.chunks == 12 && h.reserved1 == 0) ? hat_tail3_launch_c180(h, stream) : HAT_EINVAL;
    if (d.C != T3_C) return HAT_EUNSUPPORTED;
    if (d.ln1_g && (!d.ln1_b || !d.n_out || d.ldn < d.C || d.ldn % 4 || d.gap_c < 0 || d.gap_c > 16 || d.gap_c % 4)) return HAT_EINVAL;
    if (!h.n || !h.y16 || !h.c1 || !h.w_aggr || !h.wf || !h.bias_b || h.ldn_in < T3_C || h.ldn_in % 8) return HAT_EINVAL;
    const T3Aggr ag{reinterpret_cast<const bf16_t*>(h.n), reinterpret_cast<const bf16_t*>(h.y16), reinterpret_cast<const bf16_t*>(h.c1),
                    reinterpret_cast<const char*>(h.w_aggr), reinterpret_cast<const char*>(h.wf), h.bias_b, h.ldn_in};
    // reserved1 bit 0 / bit 1: t_in / t_out are FP16 rows (8-byte aligned for the loads, 16-byte for the paired stores)
    const int th = h.reserved1 & 3;
    if ((h.reserved1 & ~3) != 0) return HAT_EINVAL;
    if ((th & 1) && reinterpret_cast<uintptr_t>(d.t_in) % 8) return HAT_EINVAL;
    if ((th & 2) && reinterpret_cast<uintptr_t>(d.t_out) % 16) return HAT_EINVAL;
    void (*kern)(const HatFfnDesc, const T3Aggr) = th == 0 ? tail3_kernel<0, 0> : th == 1 ? tail3_kernel<0, 1> : th == 2 ? tail3_kernel<0, 2> : tail3_kernel<0, 3>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T3_LDS);
    if (e != hipSuccess) return (int)e;
    dim3 grid((d.W + 15) / 16, (d.H + T3_ROWS - 1) / T3_ROWS, d.B);
    HAT_LAUNCH(kern, grid, dim3(256), T3_LDS, reinterpret_cast<hipStream_t>(stream), d, ag);
    return hat_check_launch();
}
#endif
