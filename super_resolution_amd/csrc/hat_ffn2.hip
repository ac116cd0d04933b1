// hat_ffn2.hip — second-generation fused HAB feed-forward half for embed_dim 144, bf16 storage (contract: hat_ffn2 in
// include/hat_mi355x.h):
//     t_out = t_in + fc2( a * SiLU(g) ),   [a | g] = dwconv3x3( fc1( LayerNorm2(t_in) ) )
// reference: hat/archs/hat_arch.py:237 (x + mlp(norm2(x))) with GatedDconvFFN.forward :107-119.
//
// Same tiling as hat_ffn (8 x 16 pixel tile, four waves, two workgroups per CU, 32 (+32 gate) hidden channels per
// chunk), but the depthwise 3x3 no longer runs as 1/16-dense "diagonal" MFMAs (a third of the matrix instructions of
// hat_ffn for 4 % of the useful FLOPs, plus three VALU instructions each to build their operands):
//   * fc1 (bf16 MFMA, fp32 accumulate, bias as the C operand) writes the haloed hidden tile U to LDS as fp16 —
//     11 significant bits against bf16's 8, converted with round-toward-zero so that it saturates at 65504 instead of
//     producing an infinity;
//   * the depthwise conv is packed fp16 VALU (v_pk_fma_f16, two channels per lane per instruction): lane (pixel p,
//     group g) owns hidden units 8g..8g+7 of the chunk for its pixel in the wave's two tile rows, reads the 3x3
//     neighbourhood's a- and gate-slots (16 bytes each) straight from U, weights broadcast from LDS;
//   * gate a * SiLU(g) in packed fp16 (v_exp_f16 / v_rcp_f16); its result IS the B operand of the fc2 MFMA (fp16 MFMA,
//     same rate as bf16) in natural k order.
// Per chunk and wave: 78 MFMAs instead of 118 and ~250 VALU instead of ~300; LDS reads 24 + 20 instead of 40.
// Ms rows are exactly 144 bf16 = 18 sixteen-byte slots (18 = 2 mod 4: conflict-free ds_read_b128 operand reads,
// hat_common.h lds_row_elems) — the fc1 bias left the K dimension, so the image is 51.8 KB instead of 57.6 KB and the
// depthwise weights (1.3 KB per chunk) fit beside it.
#include "hat_common.h"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr int F2_C = 144, F2_NT = 9, F2_KS = 5, F2_WAVES = 4, F2_ROWS = 8, F2_HW = 18;
constexpr int F2_NPH = (F2_ROWS + 2) * F2_HW;      // 180 haloed pixels
constexpr int F2_NPTW = 6;                          // fc1 pixel tiles per wave (12 tiles of 16 over the flattened halo tile)
constexpr int F2_MS_ROWB = 288;                     // bytes per Ms row
// LDS map (exactly 80 KiB, two workgroups per CU):
//   [0, 51840)       Ms   180 rows x 288 B (bf16)
//   [51840, 53120)   Wd   depthwise weights of the current chunk [g][tap 0..8, bias][a-units 8 | gate-units 8] fp16.
//                         It doubles as (a) the dummy row that stage 0's partial last pass stores into — which also makes
//                         its first 288 bytes finite before anything reads them: the k >= 144 half of the last fc1 k-step
//                         of Ms row 179 reads the first 32 bytes of the "next row", i.e. of this region, against zero
//                         weights — and (b) the GAP reduction scratch of the epilogue
//   [53120, 81920)   Us   180 rows x 160 B: 64 fp16 channels [a 0..31 | gate 0..31] + 32 B pad.  10 slots = 2 (mod 4):
//                         the depthwise operand reads (lane (p, g) -> row p + const, slot g) are conflict-free with a plain
//                         base + immediate address (hat_common.h lds_row_elems)
constexpr int F2_WD_OFF = F2_NPH * F2_MS_ROWB;      // 51840
constexpr int F2_WD_BYTES = 4 * 10 * 32;            // 1280
constexpr int F2_RED_OFF = F2_WD_OFF;
constexpr int F2_US_OFF = F2_WD_OFF + F2_WD_BYTES;  // 53120
constexpr int F2_US_ROWB = 160;
constexpr int F2_LDS = F2_US_OFF + F2_NPH * F2_US_ROWB;   // 81920

typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ u32x4 lds_read16(unsigned addr) { return *(__attribute__((address_space(3))) const u32x4*)(uintptr_t)addr; }
__device__ __forceinline__ h2 as_h2(unsigned v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ unsigned as_u(h2 v) { return __builtin_bit_cast(unsigned, v); }

// Inputs of the fused aggregation stage (hat_hab_tail): stage 0 then computes tB = t + W_aggr . [y16 | n[16:]] +
// wf . im2col3x3(c1) + bias_b itself (what hat_aggr_cab writes to HBM as fp32 and hat_ffn reads back with a halo) and
// t_in is the residual stream BEFORE the aggregation.
struct F2Aggr {
    const bf16_t* n;       // (B,H,W,ldn) LayerNorm1 output
    const bf16_t* y16;     // (B,H,W,16)  ESC large-kernel conv output: replaces channels [0, 16) of n
    const bf16_t* c1;      // (B,H,W,8)   CAB squeeze conv output
    const char* wl;        // aggregation weights, fragment packed [9][5][64][8] bf16
    const char* wf;        // per-sample folded CAB expand weights [B][9][3][64][8] bf16 (hat_cab_fold)
    const float* bias_b;   // per-sample bias [B][144]
    int ldn;
};
__device__ __attribute__((aligned(16))) unsigned hat_ffn2_zero_page[4] = {0, 0, 0, 0};

// DBG: timing-ablation mask for tools/ubench_ffn2.hip (the library instantiates 0):
//   1 skip LN stage, 2 skip fc1 MFMAs, 4 skip the depthwise FMAs, 8 skip gate math, 16 skip fc2 MFMAs, 32 skip weight loads,
//   64 per-phase s_memtime totals of every wave -> gap_out[wg][wave][8]
template <int DBG, bool AGGR = false>
__global__ __launch_bounds__(256, 2) void ffn2_kernel(const HatFfnDesc d, const F2Aggr ag) {
    constexpr int C = F2_C, NT = F2_NT, KS = F2_KS, NTHR = 256, NPH = F2_NPH, HALO_W = F2_HW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_char*)smem;
    bf16_t* Ms = reinterpret_cast<bf16_t*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * F2_ROWS;
    const int H = d.H, W = d.W;
    const float* tin = d.t_in + (size_t)b * H * W * C;
    // any haloed pixel outside the image?  (uniform)  Only then must U be forced to zero there: the depthwise conv
    // zero-pads u AFTER the fc1 bias (hat_arch.py:112-114), and the bias enters as the MFMA C operand.
    const bool edge = x0 == 0 || y0 == 0 || x0 + 16 >= W || y0 + F2_ROWS >= H;

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

    // ------------------------------ stage 0: LayerNorm2 of the haloed tile -> Ms (bf16) --------------------------------
    // All global loads unconditional (clamped address, result discarded by a select) and issued before the first use.
    f32x4 acc2[NT][2];   // persistent fc2 accumulators: this wave's two tile rows x 9 channel tiles
    if constexpr (AGGR && !(DBG & 1)) {
        // ---------------- stage 0, fused: aggregation + folded CAB + both residual terms + LayerNorm2 -> Ms ----------------
        // All 80 KiB of LDS are free here, so the 72 KiB of A operands (9 n-tiles x 8 k-steps: 5 of the 144-wide aggregation,
        // 3 of the 72-deep im2col of c1) are copied in ONCE per workgroup by LDS-DMA (fragment-contiguous: 1 KiB per wave
        // instruction, no registers) and every wave multiplies three 16-pixel tiles against them: its own two tile rows —
        // whose fp32 results stay in registers as the initial value of the fc2 accumulators (tB never exists in memory) — and
        // one of the four halo tiles (top row, bottom row, left/right columns of rows 0-7, of rows 8-9).
        {
            const char* wfb = ag.wf + (size_t)b * NT * 3 * 1024;
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                const int f = wave * 18 + j, nt = f >> 3, ks = f & 7;
                const char* src = (ks < 5 ? ag.wl + (size_t)(nt * 5 + ks) * 1024 : wfb + (size_t)(nt * 3 + ks - 5) * 1024) + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + f * 1024), 16, 0, 0);
            }
        }
        int hp[3], pix[3];
        bool ins[3];
        bf8 bfr[3][8];
        const bf16_t* nb = ag.n + (size_t)b * H * W * ag.ldn;
        const bf16_t* yb = ag.y16 + (size_t)b * H * W * 16;
        const bf16_t* cb = ag.c1 + (size_t)b * H * W * 8;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            int hy, hx;
            if (t < 2) { hy = 2 * wave + 1 + t; hx = 1 + c16; }
            else if (wave < 2) { hy = wave * (F2_ROWS + 1); hx = 1 + c16; }
            else { const int jj = wave == 2 ? c16 : (c16 & 3); hy = (wave == 2 ? 0 : 8) + (jj >> 1); hx = (jj & 1) * (HALO_W - 1); }
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            ins[t] = y >= 0 && y < H && x >= 0 && x < W;
            const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
            hp[t] = hy * HALO_W + hx;
            pix[t] = yc * W + xc;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int c = ks * 32 + 8 * g;
                const bf16_t* src = c < 16 ? yb + (size_t)pix[t] * 16 + c : nb + (size_t)pix[t] * ag.ldn + min(c, C - 8);
                bfr[t][ks] = MT<bf16_t>::load(src);
            }
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) {
                const int tap = 4 * kc + g;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int yy = yc + dy, xx = xc + dx;
                const bool inb = tap < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W;
                bfr[t][5 + kc] = MT<bf16_t>::load(inb ? cb + ((size_t)yy * W + xx) * 8 : reinterpret_cast<const bf16_t*>(hat_ffn2_zero_page));
            }
        }
        // The accumulators START as residual + bias (t and the per-sample bias, loaded straight into them in the MFMA D
        // layout): the loads fly during the weight copy, cost no registers beyond the accumulators themselves and there is
        // no residual pass after the GEMM.
        f32x4 accx[NT];
        {
            const float* bbp = ag.bias_b + (size_t)b * NT * 16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bbp + nt * 16 + 4 * g);
                acc2[nt][0] = *reinterpret_cast<const f32x4*>(tin + (size_t)pix[0] * C + nt * 16 + 4 * g) + bb;
                acc2[nt][1] = *reinterpret_cast<const f32x4*>(tin + (size_t)pix[1] * C + nt * 16 + 4 * g) + bb;
                accx[nt] = *reinterpret_cast<const f32x4*>(tin + (size_t)pix[2] * C + nt * 16 + 4 * g) + bb;
            }
        }
        stamp(8);
        __syncthreads();   // (drains vmcnt: the LDS-DMA copies of every wave have landed)
        stamp(9);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf8 a = __builtin_bit_cast(bf8, lds_read16(lds0 + (unsigned)((nt * 8 + ks) * 1024) + (unsigned)lane * 16u));
                acc2[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[0][ks], acc2[nt][0], 0, 0, 0);
                acc2[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[1][ks], acc2[nt][1], 0, 0, 0);
                accx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[2][ks], accx[nt], 0, 0, 0);
            }
        }
        // LayerNorm2 parameters: requested now, in flight across the barrier
        f32x4 gmv[NT], btv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            gmv[nt] = *reinterpret_cast<const f32x4*>(d.ln_g + nt * 16 + 4 * g);
            btv[nt] = *reinterpret_cast<const f32x4*>(d.ln_b + nt * 16 + 4 * g);
        }
        stamp(10);
        __syncthreads();   // every wave is done with the weights: Ms may be written
        stamp(11);
        // LayerNorm2 (fp32 statistics over the 4 lane groups of a pixel) -> bf16 rows of Ms
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            f32x4 v[NT];
            float s = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                v[nt] = t == 0 ? acc2[nt][0] : (t == 1 ? acc2[nt][1] : accx[nt]);
                s += (v[nt][0] + v[nt][1]) + (v[nt][2] + v[nt][3]);
            }
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / (float)C);
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dl = v[nt][r] - mean; q += dl * dl; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            const float rstd = __builtin_amdgcn_rsqf(q * (1.0f / (float)C) + 1e-5f);
            char* rowp = smem + hp[t] * F2_MS_ROWB + 8 * g;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = ins[t] ? (v[nt][r] - mean) * rstd * gmv[nt][r] + btv[nt][r] : 0.f;
                Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(rowp + nt * 32), o);
                // the wave's own rows: tB + fc2 bias is where the fc2 accumulation starts
                if (t < 2) acc2[nt][t] = v[nt] + *reinterpret_cast<const f32x4*>(d.b2 + nt * 16 + 4 * g);
            }
        }
    } else {
        if constexpr (!(DBG & 1)) {
            const int j = tid & 15, grp = tid >> 4;
            constexpr int NGRP = NTHR / 16, LNB = (NPH + NGRP - 1) / NGRP;   // 12 pixels per 16-lane group, one pass
            const float invC = 1.0f / (float)C;
            f32x4 gmv[3], btv[3];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int c = min(4 * j + 64 * v, C - 4);
                gmv[v] = *reinterpret_cast<const f32x4*>(d.ln_g + c);
                btv[v] = *reinterpret_cast<const f32x4*>(d.ln_b + c);
            }
            f32x4 xv[LNB][3];
            bool inside[LNB];
#pragma unroll
            for (int u = 0; u < LNB; ++u) {
                const int hp = u * NGRP + grp;
                const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                inside[u] = hp < NPH && y >= 0 && y < H && x >= 0 && x < W;
                const float* src = tin + ((size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * C;
#pragma unroll
                for (int v = 0; v < 3; ++v) xv[u][v] = *reinterpret_cast<const f32x4*>(src + min(4 * j + 64 * v, C - 4));
            }
#pragma unroll
            for (int u = 0; u < LNB; ++u) {
                const int hp = u * NGRP + grp;
                // lanes j >= 4 of the third vector loaded channels 140..143 again (clamped address): they take no part in the
                // statistics but normalise and store the same values to the same place as lane 3 — no branch, no lane mask
                const f32x4 x2raw = xv[u][2];
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    if (4 * j + 64 * v >= C) xv[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                    s += (xv[u][v][0] + xv[u][v][1]) + (xv[u][v][2] + xv[u][v][3]);
                }
                s = row_sum16(s);
                const float mean = s * invC;
                float q = 0.f;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float dl = xv[u][v][r] - mean; q += (4 * j + 64 * v < C) ? dl * dl : 0.f; }
                }
                q = row_sum16(q);
                const float rstd = __builtin_amdgcn_rsqf(q * invC + 1e-5f);
                // pixels past the end of the haloed tile (the last pass is partial) store into a scratch line instead
                char* rowp = smem + (hp < NPH ? hp * F2_MS_ROWB : F2_RED_OFF);
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const int c = min(4 * j + 64 * v, C - 4);
                    const f32x4 xs = v == 2 ? x2raw : xv[u][v];
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = inside[u] ? (xs[r] - mean) * rstd * gmv[v][r] + btv[v][r] : 0.f;
                    Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(rowp) + c, o);
                }
            }
        }

        __builtin_amdgcn_sched_barrier(0);   // (stage 0 is one basic block: keep the loads below out of its 144 live registers)
        // fc2 accumulators initialised with t_in + b2
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const size_t pixc = (size_t)min(y0 + 2 * wave + pt, H - 1) * W + min(x0 + c16, W - 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 16 + 4 * g;
                acc2[nt][pt] = *reinterpret_cast<const f32x4*>(d.b2 + n) + *reinterpret_cast<const f32x4*>(tin + pixc * C + n);
            }
        }

    }

    const bf16_t* w1f = reinterpret_cast<const bf16_t*>(d.w1f);
    const _Float16* w2f = reinterpret_cast<const _Float16*>(d.w2f);
    const int nt2 = wave & 1, pg = wave >> 1;

    // fc1 A fragments + bias of the CURRENT chunk (loaded one chunk ahead)
    bf8 a1[2][KS];
    f32x4 b1v[2];
    auto load_a1 = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                a1[i][ks] = (DBG & 32) ? MT<bf16_t>::zero()
                                       : MT<bf16_t>::load(w1f + ((((size_t)chunk * 4 + (2 * nt2 + i)) * KS + ks) * 64 + lane) * 8);
            b1v[i] = *reinterpret_cast<const f32x4*>(d.b1 + (size_t)chunk * 64 + (2 * nt2 + i) * 16 + 4 * g);
        }
    };
    // fc1 B operand: pixel tile (pg + 2i), lane (c16, g) reads slot ks*4 + g of row hpa + 32 i
    const int hpa = pg * 16 + c16;
    const unsigned mbase = lds0 + (unsigned)hpa * F2_MS_ROWB + (unsigned)g * 16u;
    auto load_b = [&](int i, bf8 (&bf)[KS]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            bf[ks] = __builtin_bit_cast(bf8, lds_read16(mbase + (unsigned)(i * 32 * F2_MS_ROWB + ks * 64)));
    };
    // this lane's Us store position of its first pixel tile (tile i is 32 rows further): ONE 16-byte store holds the
    // lane's 4 + 4 results of n-tiles 2*nt2 and 2*nt2+1, so fc1's output row (tile ii, 4g + r) is hidden unit
    // 8g + 4ii + r of the half (ops.pack_ffn2 orders the fc1 rows that way).  Two 8-byte stores at the natural
    // positions were 4-way bank conflicts (16 rows x 160 B = 32 (mod 128) bytes apart): 145 k of the 167 k conflict
    // cycles per CU of the r02 profile.
    const unsigned ust = lds0 + F2_US_OFF + (unsigned)hpa * F2_US_ROWB + (unsigned)(nt2 * 64 + g * 16);
    // bit i: pixel (tile i, column c16) of the flattened haloed tile lies inside the image
    unsigned inmask = 0;
#pragma unroll
    for (int i = 0; i < F2_NPTW; ++i) {
        const int hp = hpa + 32 * i;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        inmask |= (hp < NPH && y >= 0 && y < H && x >= 0 && x < W) ? (1u << i) : 0u;
    }
    // depthwise operands: haloed rows 2*wave + hr (hr = 0..3), columns c16 + dx; a-slot g, gate-slot 4 + g
    const unsigned ubase = lds0 + F2_US_OFF + (unsigned)((2 * wave * HALO_W + c16) * F2_US_ROWB + g * 16);
    const unsigned wdad = lds0 + F2_WD_OFF + (unsigned)g * 320u;   // this lane group's [tap][32 bytes]

    load_a1(0);
    u32x4 wdreg = {0u, 0u, 0u, 0u};   // next chunk's depthwise weights on their way to LDS (threads < 80)
    const char* dwg = reinterpret_cast<const char*>(d.dww);
    if (tid < F2_WD_BYTES / 16) wdreg = *reinterpret_cast<const u32x4*>(dwg + tid * 16);
    stamp(0);
    __syncthreads();  // Ms complete
    stamp(1);

    for (int chunk = 0; chunk < d.chunks; ++chunk) {
        // ================================ phase A: fc1 -> Us (fp16) ====================================
        if (tid < F2_WD_BYTES / 16) {   // (the previous chunk's readers passed the barrier that ended that chunk)
            *reinterpret_cast<u32x4*>(smem + F2_WD_OFF + tid * 16) = wdreg;
            const int nc = min(chunk + 1, d.chunks - 1);
            wdreg = *reinterpret_cast<const u32x4*>(dwg + (size_t)nc * F2_WD_BYTES + tid * 16);
        }
        {
            // The conversion + store of tile i is issued after the MFMAs of tile i + 1 (two accumulator sets): the MFMA
            // results are then long complete and the matrix pipe never waits for the epilogue VALU work.
            bf8 bcur[KS], bnxt[KS];
            f32x4 acc[2][2];
            load_b(0, bcur);
            auto store_u = [&](int i, const f32x4 (&av)[2]) {
                if (i < F2_NPTW - 1 || hpa + 32 * i < NPH) {
                    u32x4 pk;
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        pk[2 * ii] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(av[ii][0], av[ii][1]));
                        pk[2 * ii + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(av[ii][2], av[ii][3]));
                    }
                    if (edge) {
                        const bool in = (inmask >> i) & 1u;
#pragma unroll
                        for (int k = 0; k < 4; ++k) pk[k] = in ? pk[k] : 0u;
                    }
                    *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(ust + (unsigned)(i * 32 * F2_US_ROWB)) = pk;
                }
            };
#pragma unroll
            for (int i = 0; i < F2_NPTW; ++i) {
                if (i + 1 < F2_NPTW) load_b(i + 1, bnxt);
                acc[i & 1][0] = b1v[0];
                acc[i & 1][1] = b1v[1];
                if constexpr (!(DBG & 2)) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
                            acc[i & 1][ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ii][ks], bcur[ks], acc[i & 1][ii], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i > 0) store_u(i - 1, acc[(i - 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < F2_NPTW) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnxt[ks];
                }
            }
            store_u(F2_NPTW - 1, acc[(F2_NPTW - 1) & 1]);
        }
        stamp(2);
        __syncthreads();  // Us and the depthwise weights complete
        stamp(3);

        // ====================== phase B: depthwise 3x3 in packed fp16 (this wave's two rows) ======================
        h8 a2[NT];  // fc2 weights: issued half-way through the depthwise FMAs, consumed after the gate math
        auto load_a2 = [&]() {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                a2[nt] = (DBG & 32) ? (h8)(_Float16)0 : *reinterpret_cast<const h8*>(w2f + (((size_t)chunk * NT + nt) * 64 + lane) * 8);
        };
        if constexpr (DBG & 4) load_a2();
        h2 da[2][4], dg[2][4];   // [tile row][dword]: a-units / gate-units 8g..8g+7, two per dword
        {
            const u32x4 ba = lds_read16(wdad + 9 * 32), bg = lds_read16(wdad + 9 * 32 + 16);   // "tap 9" = depthwise bias
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int k = 0; k < 4; ++k) { da[pt][k] = as_h2(ba[k]); dg[pt][k] = as_h2(bg[k]); }
        }
        if constexpr (!(DBG & 4)) {
            // 12 positions (haloed row hr = 0..3, column offset dx = 0..2), software-pipelined by hand one position ahead:
            // left to itself the scheduler hoists all 44 LDS reads (176 registers) above the first FMA and spills.
            // Position (hr, dx) feeds tile row 0 with tap (hr, dx) and tile row 1 with tap (hr - 1, dx); the weights of
            // tap (hr + 1, dx) replace those of (hr - 1, dx) as soon as this position is done.
            constexpr int PD = 3;   // U operands in flight: a whole haloed row ahead of the FMAs that consume them
            u32x4 wa[2][3], wg[2][3], ua[PD + 1], ug[PD + 1];
            auto rd_w = [&](int tr, int dx) {
                wa[tr & 1][dx] = lds_read16(wdad + (unsigned)((tr * 3 + dx) * 32));
                wg[tr & 1][dx] = lds_read16(wdad + (unsigned)((tr * 3 + dx) * 32 + 16));
            };
            auto rd_u = [&](int pos) {
                const int hr = pos / 3, dx = pos - 3 * hr;
                ua[pos % (PD + 1)] = lds_read16(ubase + (unsigned)((hr * HALO_W + dx) * F2_US_ROWB));
                ug[pos % (PD + 1)] = lds_read16(ubase + (unsigned)((hr * HALO_W + dx) * F2_US_ROWB + 64));
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
                if (pos == 6) load_a2();
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
                // tap (hr + 1, dx) into the registers tap (hr - 1, dx) just left (tap row 1 was read up front)
                if (hr >= 1 && hr + 1 < 3) rd_w(hr + 1, dx);
            }
        }
        stamp(4);
        // ================================ phase C: gate + fc2 ===================================
        // next chunk's fc1 weights: in flight during gate + fc2.  Unconditional (the last chunk re-loads itself): a branch
        // here splits the block and the depthwise FMAs get sunk below it, away from their 44 LDS reads (176 live registers)
        load_a1(min(chunk + 1, d.chunks - 1));
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            u32x4 gu;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h2 v = da[pt][k];
                if constexpr (!(DBG & 8)) {
                    const h2 x = dg[pt][k];
                    const h2 t = x * (h2){(_Float16)-1.4426950408889634f, (_Float16)-1.4426950408889634f};
                    h2 e = {(_Float16)__builtin_exp2f16(t[0]), (_Float16)__builtin_exp2f16(t[1])};
                    e = e + (h2){(_Float16)1.0f, (_Float16)1.0f};
                    const h2 r = {(_Float16)__builtin_amdgcn_rcph(e[0]), (_Float16)__builtin_amdgcn_rcph(e[1])};
                    v = v * (x * r);                  // a * g * sigmoid(g)
                }
                gu[k] = as_u(v);
            }
            const h8 gf = __builtin_bit_cast(h8, gu);
            if constexpr (!(DBG & 16)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc2[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[nt], gf, acc2[nt][pt], 0, 0, 0);
            }
        }
        stamp(5);
        __syncthreads();  // every wave is done reading Us / the depthwise weights before the next chunk overwrites them
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
            if (valid) *reinterpret_cast<f32x4*>(tout + pix * C + nt * 16 + 4 * g) = v;
            s += (v[0] + v[1]) + (v[2] + v[3]);
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
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (acc2[nt][pt][r] - mean) * rstd * g1v[nt][r] + bt1v[nt][r];
                if (valid) {
                    Vec4<bf16_t>::store(nout + nt * 16 + 4 * g, o);
                    if (nt == 0 && 4 * g < d.gap_c) gapv += as_stored<bf16_t>(o);
                    if (nt == 0 && d.n16_out != nullptr)   // compact copy of channels 0..15 for the next block's ESC conv
                        Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(d.n16_out) + ((size_t)b * H * W + pix) * 16 + 4 * g, o);
                }
            }
        }
    }
    if constexpr (DBG & 64) {
        stamp(7);
        if (lane == 0) {
            const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int i = 0; i < 12; ++i) d.gap_out[(wg * F2_WAVES + wave) * 12 + i] = (float)tph[i];
        }
        return;
    }
    if (do_ln && d.gap_out != nullptr) {
        float* red = reinterpret_cast<float*>(smem + F2_RED_OFF);
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
            for (int w = 0; w < F2_WAVES; ++w) s += red[w * 16 + tid];
            const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            d.gap_out[((size_t)b * gridDim.x * gridDim.y + tile) * 16 + tid] = tid < d.gap_c ? s : 0.f;
        }
    }
}

}  // namespace

#ifndef HAT_FFN2_NO_ENTRY
namespace {
int ffn2_check(const HatFfnDesc& d) {
    if (!d.t_in || !d.t_out || d.t_in == d.t_out || !d.ln_g || !d.ln_b || !d.w1f || !d.b1 || !d.dww || !d.w2f || !d.b2) return HAT_EINVAL;
    if (d.B < 1 || d.H < 1 || d.W < 1 || d.chunks < 1 || d.m_in) return HAT_EINVAL;
    if (d.C != F2_C || d.dtype != HAT_BF16) return HAT_EUNSUPPORTED;
    if (d.ln1_g && (!d.ln1_b || !d.n_out || d.ldn < d.C || d.ldn % 4 || d.gap_c < 0 || d.gap_c > 16 || d.gap_c % 4)) return HAT_EINVAL;
    return 0;
}
template <bool AGGR> int ffn2_launch(const HatFfnDesc& d, const F2Aggr& ag, void* stream) {
    auto kern = ffn2_kernel<0, AGGR>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS);
    if (e != hipSuccess) return (int)e;
    dim3 grid((d.W + 15) / 16, (d.H + F2_ROWS - 1) / F2_ROWS, d.B);
    HAT_LAUNCH(kern, grid, dim3(256), F2_LDS, reinterpret_cast<hipStream_t>(stream), d, ag);
    return hat_check_launch();
}
}  // namespace

extern "C" int hat_ffn2(const HatFfnDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    if (int rc = ffn2_check(*dp)) return rc;
    return ffn2_launch<false>(*dp, F2Aggr{}, stream);
}

extern "C" int hat_hab_tail(const HatHabTailDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatHabTailDesc& h = *dp;
    if (int rc = ffn2_check(h.ffn)) return rc;
    if (!h.n || !h.y16 || !h.c1 || !h.w_aggr || !h.wf || !h.bias_b || h.ldn_in < F2_C || h.ldn_in % 8) return HAT_EINVAL;
    const F2Aggr ag{reinterpret_cast<const bf16_t*>(h.n), reinterpret_cast<const bf16_t*>(h.y16), reinterpret_cast<const bf16_t*>(h.c1),
                    reinterpret_cast<const char*>(h.w_aggr), reinterpret_cast<const char*>(h.wf), h.bias_b, h.ldn_in};
    return ffn2_launch<true>(h.ffn, ag, stream);
}
#endif
