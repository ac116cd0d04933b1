// hat_pw.hip — pointwise linear layers (nn.Linear / 1x1 conv on channel-last tokens) as a weight-stationary,
// barrier-free streaming GEMM on gfx950.  Contract: hat_linear in include/hat_mi355x.h (it takes the same
// HatConvDesc as hat_conv with ksize == 1, but the weights are FRAGMENT packed).
//
// These layers (ESC aggr, OCAB q/kv/proj/MLP; hat_arch.py:309-313,347,350,391, esc_arch.py:144) have K, N <= 360
// and 1e6 pixels: they are HBM-bound (70-115 FLOP/B vs a ridge of ~400), so the structure is chosen for bytes in
// flight, not MFMA rate:
//   * one n-slice of the weight matrix (<= 192 x K) lives in LDS for the whole life of the workgroup, already in
//     MFMA A-fragment order (a fragment is 1 KiB read as base + lane*16: conflict free);
//   * every wave walks its own 16-pixel tiles (flat pixel order, persistent grid): B fragments come straight
//     from global memory into registers (one tile ahead), no LDS staging of activations, NO barrier in the loop,
//     so waves free-run and 12-16 of them per CU keep HBM requests in flight;
//   * the epilogue (bias, GELU, fp32 residual, scaled T residual) is applied on the accumulators: a lane owns
//     4 consecutive channels of one pixel.
#include "hat_common.h"

namespace {

template <bool SPEC_, bool R1_, bool R2_, bool OUTF32_, bool LN_, bool PAIR_ = false> struct PwTag {
    // pair: bf16 rows (out / ln_out) leave as 16-byte stores of n-tile pairs — needs every n-tile full and aligned rows
    static constexpr bool spec = SPEC_, r1 = R1_, r2 = R2_, outf32 = OUTF32_, ln = LN_, pair = PAIR_;
};

// RES: the launch has residual operands (r1 and/or r2): their loads are batched ahead of the MFMAs (+54 registers)
template <typename T, int NT, int KS, int WAVES, int MINW, bool RES>
__global__ __launch_bounds__(WAVES * 64, MINW) void pw_kernel(const HatConvDesc d, long npix_total, int tiles, int scale_in_lds) {
    using M = MT<T>;
    using frag_t = typename M::frag_t;
    constexpr int KSMAX = KS;
    constexpr int ks_total = KS;
    constexpr int nthr = WAVES * 64, nwaves = WAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Wl = reinterpret_cast<T*>(smem);  // [NT][ks][64 lanes][8]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = tid >> 6;
    const int slice = blockIdx.y;
    const int nbase = slice * NT * 16;
    const T* wg = reinterpret_cast<const T*>(d.w) + (size_t)slice * NT * ks_total * 512;  // 512 elements per fragment
    for (int i = tid; i < NT * ks_total * 64 * (int)sizeof(T) / 2; i += nthr)  // 16 bytes per item
        *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(wg) + (size_t)i * 16);
    // bias: [NT*16] fp32 in LDS right behind the weights (read per tile in the epilogue: NT*4 fewer live registers)
    float* bsl = reinterpret_cast<float*>(smem + (size_t)NT * KS * 64 * 8 * sizeof(T));
    for (int i = tid; i < NT * 16; i += nthr) bsl[i] = d.bias[nbase + i];
    // r2's per-(batch, channel) scale: a [B][NT*16] table in LDS behind that (when the launcher found room)
    float* sct = bsl + NT * 16;
    if (RES && scale_in_lds) {
        for (int i = tid; i < d.B * NT * 16; i += nthr) {
            const int bb = i / (NT * 16), n = nbase + (i - bb * NT * 16);
            sct[i] = n < d.n_store ? d.r2scale[(size_t)bb * d.r2scale_bstride + n] : 0.f;
        }
    }
    // fused LayerNorm of the finished pixel (RES variant only): gamma | beta behind the scale table
    const bool emit_ln = RES && d.ln_out != nullptr;
    float* lnp = sct + (scale_in_lds ? d.B * NT * 16 : 0);
    if (emit_ln) {
        for (int i = tid; i < NT * 16; i += nthr) {
            lnp[i] = i < d.n_store ? d.ln_g[i] : 0.f;
            lnp[NT * 16 + i] = i < d.n_store ? d.ln_b[i] : 0.f;
        }
    }
    __syncthreads();

    const T* xg = reinterpret_cast<const T*>(d.x);
    const T* xg0 = reinterpret_cast<const T*>(d.x0);
    const int Cin = d.Cin;
    // B fragments of one 16-pixel tile.  bf16: every load is unconditional — k groups at or beyond Cin re-read the
    // row's last 8-channel group (finite activations against zero weight columns) — because a load under a lane mask
    // that merges with a default is waited for where it is issued, which would make this one-tile-ahead prefetch
    // synchronous.
    auto load_b = [&](long tile, frag_t (&bf)[KSMAX]) {
        long p = tile * 16 + c16;
        p = p < npix_total ? p : npix_total - 1;
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            if (ks < ks_total) {
                if constexpr (sizeof(T) == 2) {
                    const int c = min(ks * 32 + 8 * g, ((Cin + 7) & ~7) - 8);
                    const T* src = (xg0 != nullptr && c < d.c_split) ? xg0 + p * d.ldx0 + c : xg + p * d.ldx + c;
                    bf[ks] = M::load(src);
                } else {
                    const int c = ks * 32 + 8 * g;
                    if (c + 8 <= Cin) {
                        const T* src = (xg0 != nullptr && c < d.c_split) ? xg0 + p * d.ldx0 + c : xg + p * d.ldx + c;
                        bf[ks] = M::load(src);
                    } else if (c < Cin) {                                // f32 tail: 4 valid channels
                        const f32x4 lo = *reinterpret_cast<const f32x4*>(xg + p * d.ldx + c);
                        frag_t t = M::zero();
                        t[0] = to_T<T>(lo[0]); t[1] = to_T<T>(lo[1]); t[2] = to_T<T>(lo[2]); t[3] = to_T<T>(lo[3]);
                        bf[ks] = t;
                    } else {
                        bf[ks] = M::zero();
                    }
                }
            }
        }
    };

    const long stride = (long)gridDim.x * nwaves;
    const long tile0 = (long)blockIdx.x * nwaves + wave;

    // The tile loop exists in several copies selected ONCE per launch.  A specialised copy (Tag::spec) has no control
    // flow around its memory operations — operand presence, output type and "every channel is stored" are compile-time,
    // lanes past the last pixel recompute and re-store the last pixel — because hipcc merges its s_waitcnt bookkeeping
    // conservatively at every CFG join: with conditional loads/stores in the body each wait became vmcnt(0), i.e. the
    // prefetch of the next tile and the stores of the previous one were drained on every tile.
    auto tile_loop = [&](auto tag) {
        using Tag = decltype(tag);
        constexpr bool SPEC = Tag::spec;
        const bool has_r1 = SPEC ? Tag::r1 : (RES && d.r1 != nullptr);
        const bool has_r2 = SPEC ? Tag::r2 : (RES && d.r2 != nullptr);
        const bool out_f32 = SPEC ? Tag::outf32 : (d.out_mode != HAT_O_NHWC_T);
        const bool do_ln = SPEC ? Tag::ln : emit_ln;
        const bool sc_lds = SPEC ? true : (scale_in_lds != 0);
        // The last n-tile may be partial (C = 180 = 11.25 tiles).  The specialised copies stay free of control flow: a lane
        // group past the last channel loads its residual from, and re-stores, ITS OWN channels of the previous n-tile (same
        // address, same value as the store it already made), chosen with selects.  Only the fp32-output residual copies
        // are entered with a partial tile (`tail_ok` below).
        const bool lastok = !SPEC || 4 * g < d.n_store - (NT - 1) * 16;
        long tile = tile0;
        // Software pipeline, ordered around how hipcc places its waits.  Its s_waitcnt bookkeeping is merged at the loop
        // header with the (store-free) loop entry state, so every in-loop wait on a load degenerates to "wait for
        // everything issued so far"; the loop therefore issues its loads LAST (B operands of the tile after next, residual
        // operands of the next tile) and consumes them only after the next tile's MFMAs: by then they, and the stores
        // issued just before them, have had a whole MFMA phase to complete.
        auto load_r = [&](long t, f32x4 (&r1v)[RES ? NT : 1], typename Vec4<T>::raw_t (&r2v)[RES ? NT : 1]) {
            if constexpr (RES) {
                long pp = t * 16 + c16;
                pp = pp < npix_total ? pp : npix_total - 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = SPEC ? (nt == NT - 1 && NT > 1 && !lastok ? (nt - 1) * 16 + 4 * g : nt * 16 + 4 * g)
                                       : min(nbase + nt * 16 + 4 * g, d.n_store - 4);
                    if (has_r1) r1v[nt] = *reinterpret_cast<const f32x4*>(d.r1 + pp * d.ldr1 + n);
                    if (has_r2) r2v[nt] = Vec4<T>::load_raw(reinterpret_cast<const T*>(d.r2) + pp * d.ldr2 + n);
                }
            }
        };
        // (without residual registers and with a short K there is room for a third set: B operands two tiles ahead)
        constexpr bool DEEP = !RES && KS <= 5;
        frag_t bcur[KSMAX], bnxt[KSMAX], bnx2[DEEP ? KSMAX : 1];
        f32x4 r1v[RES ? NT : 1];
        typename Vec4<T>::raw_t r2v[RES ? NT : 1];
        load_b(tile, bcur);
        load_b(tile + stride, bnxt);
        if constexpr (DEEP) load_b(tile + 2 * stride, bnx2);
        load_r(tile, r1v, r2v);
        for (; tile < tiles; tile += stride) {
            const long p = tile * 16 + c16;
            const long pc = p < npix_total ? p : npix_total - 1;
            const long bidx = pc / ((long)d.H * d.W);
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // The weight fragments are re-read from LDS for every tile: make the address opaque per iteration, or the
            // compiler hoists all NT*KS loop-invariant fragments into registers and spills.
            int wofs = lane * 8;
            asm volatile("" : "+v"(wofs));
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) {
                if (ks < ks_total) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const frag_t af = M::load(Wl + (size_t)(nt * ks_total + ks) * 512 + wofs);
                        acc[nt] = M::mma(af, bcur[ks], acc[nt]);
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {  // finish the values in place (the LayerNorm below needs the whole pixel)
                f32x4 v = acc[nt] + *reinterpret_cast<const f32x4*>(bsl + nt * 16 + 4 * g);
                if (d.act == HAT_ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_act<T>(v[r]);
                } else if (d.act == HAT_ACT_LRELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
                }
                if constexpr (RES) {
                    if (has_r1) v += r1v[nt];
                    if (has_r2) {
                        const int n = min(nbase + nt * 16 + 4 * g, d.n_store - 4);
                        const f32x4 sc = sc_lds ? *reinterpret_cast<const f32x4*>(sct + bidx * (NT * 16) + nt * 16 + 4 * g)
                                                : *reinterpret_cast<const f32x4*>(d.r2scale + bidx * d.r2scale_bstride + n);
                        v += sc * Vec4<T>::cvt(r2v[nt]);
                    }
                }
                acc[nt] = v;
            }
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) {   // (the next tile's operands have arrived)
                bcur[ks] = bnxt[ks];
                if constexpr (DEEP) bnxt[ks] = bnx2[ks];
            }
            if constexpr (SPEC && !Tag::outf32 && sizeof(T) == 2 && Tag::pair) {
                // full tiles, bf16 rows: n-tiles leave in pairs, 16 bytes per lane (store_pair_bf16)
                bf16_t* orow = reinterpret_cast<bf16_t*>(d.out) + pc * d.ldo;
#pragma unroll
                for (int nt = 0; nt + 1 < NT; nt += 2) store_pair_bf16(orow, nt * 16, g, acc[nt], acc[nt + 1]);
                if constexpr (NT & 1) Vec4<T>::store(reinterpret_cast<T*>(d.out) + pc * d.ldo + (NT - 1) * 16 + 4 * g, acc[NT - 1]);
            } else if (SPEC || p < npix_total) {
                const long ps = SPEC ? pc : p;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    int n = nbase + nt * 16 + 4 * g;
                    f32x4 v = acc[nt];
                    if constexpr (SPEC && NT > 1) {
                        if (nt == NT - 1) {   // partial last tile: re-store the previous tile's value (selects, no branch)
                            n = lastok ? n : n - 16;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = lastok ? v[r] : acc[NT - 2][r];
                        }
                    }
                    if (SPEC || n < d.n_store) {
                        if (!out_f32) Vec4<T>::store(reinterpret_cast<T*>(d.out) + ps * d.ldo + n, v);
                        else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.out) + ps * d.ldo + n) = v;
                    }
                }
            }
            if constexpr (RES) {
                if (do_ln) {  // wave-uniform; the four lane groups of a pixel hold its NT*16 channels
                    const int C = d.n_store;
                    float sm = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (!SPEC && nt * 16 + 4 * g >= C) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (SPEC && NT > 1) {   // partial last tile: zero by select
                            if (nt == NT - 1) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[nt][r] = lastok ? acc[nt][r] : 0.f;
                            }
                        }
                        sm += (acc[nt][0] + acc[nt][1]) + (acc[nt][2] + acc[nt][3]);
                    }
                    sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
                    const float mean = sm / (float)C;
                    float qq = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (SPEC || nt * 16 + 4 * g < C) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float dl = acc[nt][r] - mean;
                                if (SPEC && NT > 1 && nt == NT - 1) dl = lastok ? dl : 0.f;
                                qq += dl * dl;
                            }
                        }
                    }
                    qq += __shfl_xor(qq, 16); qq += __shfl_xor(qq, 32);
                    const float rstd = 1.0f / sqrtf(qq / (float)C + 1e-5f);
                    auto ln_val = [&](int nt) {
                        const int n = nt * 16 + 4 * g;
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(lnp + n);
                        const f32x4 bt = *reinterpret_cast<const f32x4*>(lnp + NT * 16 + n);
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (acc[nt][r] - mean) * rstd * gm[r] + bt[r];
                        return o;
                    };
                    if constexpr (SPEC && sizeof(T) == 2 && Tag::pair) {   // full tiles: 16-byte stores of n-tile pairs
                        bf16_t* lo = reinterpret_cast<bf16_t*>(d.ln_out) + pc * d.ld_ln;
#pragma unroll
                        for (int nt = 0; nt + 1 < NT; nt += 2) store_pair_bf16(lo, nt * 16, g, ln_val(nt), ln_val(nt + 1));
                        if constexpr (NT & 1) Vec4<T>::store(reinterpret_cast<T*>(d.ln_out) + pc * d.ld_ln + (NT - 1) * 16 + 4 * g, ln_val(NT - 1));
                    } else if (SPEC || p < npix_total) {
                        T* lo = reinterpret_cast<T*>(d.ln_out) + (SPEC ? pc : p) * d.ld_ln;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            int n = nt * 16 + 4 * g;
                            if (SPEC || n < C) {
                                f32x4 o = ln_val(nt);
                                if constexpr (SPEC && NT > 1) {
                                    if (nt == NT - 1) {   // partial last tile: re-store the previous tile's value
                                        const f32x4 op = ln_val(NT - 2);
                                        n = lastok ? n : n - 16;
#pragma unroll
                                        for (int r = 0; r < 4; ++r) o[r] = lastok ? o[r] : op[r];
                                    }
                                }
                                Vec4<T>::store(lo + n, o);
                            }
                        }
                        if constexpr (!SPEC) {
                            if (d.ln_ones) {  // [C] = 1.0 (the consumer's bias column), zeros up to ld_ln
                                for (int n = C + 4 * g; n < d.ld_ln; n += 16)
                                    Vec4<T>::store(lo + n, f32x4{n == C ? 1.0f : 0.f, 0.f, 0.f, 0.f});
                            }
                        }
                    }
                }
            }
            if constexpr (DEEP) load_b(tile + 3 * stride, bnx2);
            else load_b(tile + 2 * stride, bnxt);
            load_r(tile + stride, r1v, r2v);
        }
    };

    // specialised copies need: one slice that stores all NT*16 channels, no split source oddities beyond load_b's, and
    // (with r2) the scale table in LDS
    const bool full = d.n_slices == 1 && d.n_store == NT * 16;
    const bool tail_ok = d.n_slices == 1 && NT > 1 && d.n_store > (NT - 1) * 16 && d.n_store <= NT * 16;  // partial last tile
    // T-typed rows leave the specialised copies in 16-byte pieces (store_pair_bf16)
    const bool out16 = d.ldo % 8 == 0 && reinterpret_cast<uintptr_t>(d.out) % 16 == 0;
    const bool ln16 = d.ld_ln % 8 == 0 && reinterpret_cast<uintptr_t>(d.ln_out) % 16 == 0;
    if constexpr (RES) {
        const bool f32o = d.out_mode == HAT_O_NHWC_F32;
        if (tail_ok && f32o && d.r1 && d.r2 && scale_in_lds && !emit_ln) tile_loop(PwTag<true, true, true, true, false>{});
        else if (full && f32o && d.r1 && !d.r2 && emit_ln && !d.ln_ones && ln16) tile_loop(PwTag<true, true, false, true, true, true>{});
        else if (tail_ok && f32o && d.r1 && !d.r2 && emit_ln && !d.ln_ones) tile_loop(PwTag<true, true, false, true, true, false>{});
        else if (tail_ok && f32o && d.r1 && !d.r2 && !emit_ln) tile_loop(PwTag<true, true, false, true, false>{});
        // residual added in fp32, result stored as T rows only (the OCAB's last linear when a 3x3 conv consumes it)
        else if (full && !f32o && d.r1 && !d.r2 && !emit_ln && out16) tile_loop(PwTag<true, true, false, false, false, true>{});
        else tile_loop(PwTag<false, false, false, false, false>{});
    } else {
        if (full && d.out_mode == HAT_O_NHWC_T && out16) tile_loop(PwTag<true, false, false, false, false, true>{});
        else if (tail_ok && d.out_mode == HAT_O_NHWC_T) tile_loop(PwTag<true, false, false, false, false, false>{});
        else tile_loop(PwTag<false, false, false, false, false>{});
    }
}

template <typename T, int NT, int KS, int WAVES, int MINW, bool RES>
int launch_pw_cfg(const HatConvDesc& d, hipStream_t s, size_t lds, int wgs_per_cu, int scale_in_lds) {
    const long npix = (long)d.B * d.H * d.W;
    const int tiles = (int)((npix + 15) / 16);
    int gx = 256 * wgs_per_cu;
    if (gx > (tiles + WAVES - 1) / WAVES) gx = (tiles + WAVES - 1) / WAVES;
    auto kern = pw_kernel<T, NT, KS, WAVES, MINW, RES>;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    HAT_LAUNCH(kern, dim3(gx, d.n_slices, 1), dim3(WAVES * 64), lds, s, d, npix, tiles, scale_in_lds);
    return hat_check_launch();
}

template <typename T, int NT, int KS>
int launch_pw(const HatConvDesc& d, hipStream_t s) {
    size_t lds = (size_t)NT * KS * 64 * 8 * sizeof(T) + (size_t)NT * 16 * sizeof(float);  // fragments (64 lanes x 8) + bias
    if (lds > HAT_LDS_MAX) return HAT_EUNSUPPORTED;
    const bool res = d.r1 != nullptr || d.r2 != nullptr;
    int scale_in_lds = 0;
    if (d.r2 != nullptr) {
        const size_t tbl = (size_t)d.B * NT * 16 * sizeof(float);
        if (tbl <= 16384 && lds + tbl <= HAT_LDS_MAX) { scale_in_lds = 1; lds += tbl; }
    }
    if (d.ln_out != nullptr) {
        if (!res) return HAT_EUNSUPPORTED;  // the fused LayerNorm lives in the residual variant (both users have r1)
        lds += (size_t)2 * NT * 16 * sizeof(float);
        if (lds > HAT_LDS_MAX) return HAT_EUNSUPPORTED;
    }
    int wgs_per_cu = (int)(HAT_LDS_MAX / lds);
    // waves per SIMD the register allocation is sized for: 3 workgroups x 4 waves -> 3, 2 x 4 -> 2, 1 x 8 -> 2; the
    // residual variant keeps a tile's residual operands in registers and is sized for 2
    wgs_per_cu = wgs_per_cu > (res ? 2 : 3) ? (res ? 2 : 3) : wgs_per_cu;
    if (wgs_per_cu < 2) return res ? launch_pw_cfg<T, NT, KS, 8, 2, true>(d, s, lds, 1, scale_in_lds) : launch_pw_cfg<T, NT, KS, 8, 2, false>(d, s, lds, 1, 0);
    if (wgs_per_cu == 2) return res ? launch_pw_cfg<T, NT, KS, 4, 2, true>(d, s, lds, 2, scale_in_lds) : launch_pw_cfg<T, NT, KS, 4, 2, false>(d, s, lds, 2, 0);
    return launch_pw_cfg<T, NT, KS, 4, 3, false>(d, s, lds, 3, 0);
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 convolutions whose whole weight slice fits in LDS (CAB: C -> C/cr and C/cr -> C, hat_arch.py:84,86), same
// free-running structure: the B fragment of k-step ks is the 16-byte group ci..ci+7 of the NEIGHBOUR pixel of
// tap (32 ks + 8 g) / Cin_p, gathered straight from global memory (L1/L2 serve the 9-fold re-reads); neighbours
// outside the image read a zero page.  Optional per-workgroup column sums (ECA pooling, hat_arch.py:73).
// ---------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned hat_zero_page[16] = {0};

template <typename T, int NT, int KS, int CINP, int WAVES, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void tap3_kernel(const HatConvDesc d, int tiles) {
    using M = MT<T>;
    using frag_t = typename M::frag_t;
    constexpr int nthr = WAVES * 64;
    constexpr int GRP = 8;  // B fragments gathered before their MFMAs are issued
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Wl = reinterpret_cast<T*>(smem);  // [NT][KS][64 lanes][8]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = tid >> 6;
    const int b = blockIdx.z;
    const T* wg = reinterpret_cast<const T*>(d.w);
    for (int i = tid; i < NT * KS * 64 * (int)sizeof(T) / 2; i += nthr)
        *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(wg) + (size_t)i * 16);
    __syncthreads();

    const int H = d.H, W = d.W, Cin = d.Cin;
    const long HW = (long)H * W;
    const T* xb = reinterpret_cast<const T*>(d.x) + (size_t)b * HW * d.ldx;
    const T* zero = reinterpret_cast<const T*>(hat_zero_page);
    f32x4 csum[NT], bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        csum[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bias[nt] = *reinterpret_cast<const f32x4*>(d.bias + nt * 16 + 4 * g);   // once, not one dependent load per tile
    }

    const int stride = gridDim.x * WAVES;
    for (int tile = blockIdx.x * WAVES + wave; tile < tiles; tile += stride) {
        const long p = (long)tile * 16 + c16;
        const bool pv = p < HW;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        int wofs = lane * 8;
        asm volatile("" : "+v"(wofs));  // keep the loop-invariant LDS weight reads inside the loop (no hoist + spill)
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += GRP) {
            frag_t bf[GRP];
#pragma unroll
            for (int kk = 0; kk < GRP; ++kk) {
                const int ks = k0 + kk;
                if (ks < KS) {
                    const int k = 32 * ks + 8 * g;
                    const int tap = k / CINP, ci = k - tap * CINP;
                    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                    const int yy = y + dy, xx = x + dx;
                    const bool inb = pv && tap < 9 && ci < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W;
                    const T* src = inb ? xb + ((size_t)yy * W + xx) * d.ldx + ci : zero;
                    bf[kk] = M::load(src);
                }
            }
#pragma unroll
            for (int kk = 0; kk < GRP; ++kk) {
                const int ks = k0 + kk;
                if (ks < KS) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const frag_t af = M::load(Wl + (size_t)(nt * KS + ks) * 512 + wofs);
                        acc[nt] = M::mma(af, bf[kk], acc[nt]);
                    }
                }
            }
        }
        if (pv) {
            const size_t pix = (size_t)b * HW + p;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 16 + 4 * g;
                if (n < d.n_store) {
                    f32x4 v = acc[nt] + bias[nt];
                    if (d.act == HAT_ACT_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_act<T>(v[r]);
                    } else if (d.act == HAT_ACT_LRELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
                    }
                    csum[nt] += d.out_mode == HAT_O_NHWC_T ? as_stored<T>(v) : v;   // (the pool of the STORED map)
                    if (d.out_mode == HAT_O_NHWC_T) Vec4<T>::store(reinterpret_cast<T*>(d.out) + pix * d.ldo + n, v);
                    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.out) + pix * d.ldo + n) = v;
                }
            }
        }
    }
    if (d.colsum != nullptr) {  // per-workgroup column sums, fixed reduction order: lanes (pixels), then waves
        __syncthreads();        // every wave is done with the weights: reuse LDS as scratch
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sv = csum[nt][r];
                sv = row_sum16(sv);
                if (c16 == 0) red[wave * (NT * 16) + nt * 16 + 4 * g + r] = sv;
            }
        }
        __syncthreads();
        for (int n = tid; n < NT * 16; n += nthr) {
            float sv = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) sv += red[w * (NT * 16) + n];
            d.colsum[((size_t)b * gridDim.x + blockIdx.x) * (NT * 16) + n] = sv;
        }
    }
}

template <typename T, int NT, int KS, int CINP>
int launch_tap3(const HatConvDesc& d, hipStream_t s, int32_t* groups_out) {
    const size_t lds = (size_t)NT * KS * 64 * 8 * sizeof(T);
    if (lds > HAT_LDS_MAX / 2) return HAT_EUNSUPPORTED;
    const long hw = (long)d.H * d.W;
    const int tiles = (int)((hw + 15) / 16);
    const int wgs_per_cu = (int)(HAT_LDS_MAX / lds) >= 3 ? 3 : 2;
    int gx = 256 * wgs_per_cu;
    if (gx > (tiles + 3) / 4) gx = (tiles + 3) / 4;
    if (groups_out) { *groups_out = gx; return 0; }
    dim3 grid(gx, 1, d.B);
    if (wgs_per_cu == 3) {
        auto kern = tap3_kernel<T, NT, KS, CINP, 4, 3>;
        HAT_LAUNCH(kern, grid, dim3(256), lds, s, d, tiles);
    } else {
        auto kern = tap3_kernel<T, NT, KS, CINP, 4, 2>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        HAT_LAUNCH(kern, grid, dim3(256), lds, s, d, tiles);
    }
    return hat_check_launch();
}

int tap3_dispatch(const HatConvDesc& d, hipStream_t s, int32_t* groups_out) {
    if (d.ksize != 3 || d.n_slices != 1 || d.x_mode != HAT_X_NHWC_T || d.x0 || d.r1 || d.r2) return HAT_EUNSUPPORTED;
    if (d.out_mode != HAT_O_NHWC_T && d.out_mode != HAT_O_NHWC_F32) return HAT_EUNSUPPORTED;
    const int cinp = (d.Cin + 7) & ~7;
#define HAT_T3(TT)                                                                                            \
    if (d.nt == 1 && cinp == 144) return launch_tap3<TT, 1, 41, 144>(d, s, groups_out);  /* CAB conv 144 -> 6 */   \
    if (d.nt == 9 && cinp == 8) return launch_tap3<TT, 9, 3, 8>(d, s, groups_out);       /* CAB conv 6 -> 144 */   \
    if (d.nt == 1 && cinp == 24) return launch_tap3<TT, 1, 7, 24>(d, s, groups_out);     /* tiny: 24 -> 8 */       \
    if (d.nt == 4 && cinp == 8) return launch_tap3<TT, 4, 3, 8>(d, s, groups_out);       /* tiny: 8 -> 24 */
    if (d.dtype == HAT_BF16) { HAT_T3(bf16_t) }
    else if (d.dtype == HAT_F32) { HAT_T3(float) }
#undef HAT_T3
    return HAT_EUNSUPPORTED;
}

}  // namespace

extern "C" int hat_linear(const HatConvDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatConvDesc& d = *dp;
    if (!d.x || !d.w || !d.bias || !d.out || d.ksize != 1 || d.B < 1 || d.H < 1 || d.W < 1 || d.Cin < 4) return HAT_EINVAL;
    if (d.x_mode != HAT_X_NHWC_T || (d.out_mode != HAT_O_NHWC_T && d.out_mode != HAT_O_NHWC_F32)) return HAT_EINVAL;
    if (d.n_store % 4 || d.ldo % 4 || d.n_store > d.n_slices * d.nt * 16 || d.colsum) return HAT_EINVAL;
    const int vec = d.dtype == HAT_BF16 ? 8 : 4;
    if (d.ldx % vec || d.Cin % 4 || (d.x0 && (d.ldx0 % vec || d.c_split % vec || d.c_split > d.Cin))) return HAT_EINVAL;
    if ((d.r1 && d.ldr1 % 4) || (d.r2 && (d.ldr2 % 4 || !d.r2scale))) return HAT_EINVAL;
    if (d.ln_out && (!d.ln_g || !d.ln_b || d.n_slices != 1 || d.ld_ln % 4 || d.ld_ln < d.n_store + (d.ln_ones ? 4 : 0))) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ks = (d.Cin + 31) / 32;
#define HAT_PW_CASE(TT)                                                   \
    if (d.nt == 9 && ks == 5) return launch_pw<TT, 9, 5>(d, s);           \
    if (d.nt == 18 && ks == 5) return launch_pw<TT, 18, 5>(d, s);         \
    if (d.nt == 23 && ks == 6) { if constexpr (sizeof(TT) == 2) return launch_pw<TT, 23, 6>(d, s); }  /* 180 -> 360, one slice */ \
    if (d.nt == 9 && ks == 9) return launch_pw<TT, 9, 9>(d, s);           \
    if (d.nt == 12 && ks == 6) return launch_pw<TT, 12, 6>(d, s);         \
    if (d.nt == 12 && ks == 12) return launch_pw<TT, 12, 12>(d, s);       \
    if (d.nt == 4 && ks == 1) return launch_pw<TT, 4, 1>(d, s);           \
    if (d.nt == 4 && ks == 2) return launch_pw<TT, 4, 2>(d, s);
    if (d.dtype == HAT_BF16) { HAT_PW_CASE(bf16_t) }
    else if (d.dtype == HAT_F32) { HAT_PW_CASE(float) }
    else return HAT_EINVAL;
#undef HAT_PW_CASE
    return HAT_EUNSUPPORTED;  // shapes not instantiated here: use hat_conv (ksize 1)
}

// ---------------------------------------------------------------------------------------------------------
// hat_aggr_cab (include/hat_mi355x.h): the ESC aggregation 1x1 with the CAB expand conv folded in as three more
// k-steps whose B operands are the 3x3 neighbours of c1 (8 channels = 16 bytes per tap, gathered like tap3_kernel)
// and whose A operands are the per-sample (ECA-scaled) expand weights.  Same streaming structure and software
// pipeline as pw_kernel's specialised loop; one grid row per sample (the folded weights are per sample).
// ---------------------------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(256, 2) void aggr_cab_kernel(const HatAggrCabDesc dd, int tiles) {
    using M = MT<T>;
    using frag_t = typename M::frag_t;
    constexpr int NT = 9, KS = 5, KC = 3, KT = KS + KC, nthr = 256, WAVES = 4;
    const HatConvDesc& d = dd.lin;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Wl = reinterpret_cast<T*>(smem);                       // [NT][KS][64][8] aggregation weights
    T* Wc = Wl + (size_t)NT * KS * 512;                       // [NT][KC][64][8] this sample's folded expand weights
    float* bsl = reinterpret_cast<float*>(Wc + (size_t)NT * KC * 512);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15, wave = tid >> 6;
    const int b = blockIdx.y;
    {
        const char* w1 = reinterpret_cast<const char*>(d.w);
        const char* w2 = reinterpret_cast<const char*>(dd.wf) + (size_t)b * NT * KC * 512 * sizeof(T);
        for (int i = tid; i < NT * KS * 512 * (int)sizeof(T) / 16; i += nthr)
            *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(w1 + (size_t)i * 16);
        for (int i = tid; i < NT * KC * 512 * (int)sizeof(T) / 16; i += nthr)
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(Wc) + (size_t)i * 16) = *reinterpret_cast<const u32x4*>(w2 + (size_t)i * 16);
        for (int i = tid; i < NT * 16; i += nthr) bsl[i] = dd.bias_b[(size_t)b * NT * 16 + i];
    }
    __syncthreads();
    const int H = d.H, W = d.W, Cin = d.Cin;
    const long HW = (long)H * W;
    const T* xg = reinterpret_cast<const T*>(d.x) + (size_t)b * HW * d.ldx;
    const T* xg0 = d.x0 ? reinterpret_cast<const T*>(d.x0) + (size_t)b * HW * d.ldx0 : nullptr;
    const T* c1 = reinterpret_cast<const T*>(dd.c1) + (size_t)b * HW * 8;
    const T* zero = reinterpret_cast<const T*>(hat_zero_page);
    const float* r1 = d.r1 + (size_t)b * HW * d.ldr1;
    float* out = reinterpret_cast<float*>(d.out) + (size_t)b * HW * d.ldo;

    auto load_b = [&](long t, frag_t (&bf)[KT]) {
        long p = t * 16 + c16;
        p = p < HW ? p : HW - 1;
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = min(ks * 32 + 8 * g, ((Cin + 7) & ~7) - 8);
            bf[ks] = M::load((xg0 != nullptr && c < d.c_split) ? xg0 + p * d.ldx0 + c : xg + p * d.ldx + c);
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int tap = 4 * kc + g;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int yy = y + dy, xx = x + dx;
            const bool inb = tap < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W;
            bf[KS + kc] = M::load(inb ? c1 + ((size_t)yy * W + xx) * 8 : zero);
        }
    };
    auto load_r = [&](long t, f32x4 (&rv)[NT]) {
        long p = t * 16 + c16;
        p = p < HW ? p : HW - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rv[nt] = *reinterpret_cast<const f32x4*>(r1 + p * d.ldr1 + nt * 16 + 4 * g);
    };
    const long stride = (long)gridDim.x * WAVES;
    long tile = (long)blockIdx.x * WAVES + wave;
    frag_t bcur[KT], bnxt[KT];
    f32x4 rv[NT];
    load_b(tile, bcur);
    load_b(tile + stride, bnxt);
    load_r(tile, rv);
    for (; tile < tiles; tile += stride) {
        long p = tile * 16 + c16;
        p = p < HW ? p : HW - 1;   // lanes past the last pixel recompute and re-store it
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        int wofs = lane * 8;
        asm volatile("" : "+v"(wofs));
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const T* wp = ks < KS ? Wl + (size_t)(nt * KS + ks) * 512 : Wc + (size_t)(nt * KC + (ks - KS)) * 512;
                acc[nt] = M::mma(M::load(wp + wofs), bcur[ks], acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] += *reinterpret_cast<const f32x4*>(bsl + nt * 16 + 4 * g) + rv[nt];
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) bcur[ks] = bnxt[ks];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(out + p * d.ldo + nt * 16 + 4 * g) = acc[nt];
        load_b(tile + 2 * stride, bnxt);
        load_r(tile + stride, rv);
    }
}
}  // namespace

/* 3x3 convolution with the whole weight slice resident in LDS (see tap3_kernel); weights fragment packed like
 * hat_linear with K index = tap * Cin_p + ci.  colsum (optional) is [B][groups][nt*16] with groups from
 * hat_conv3x3_small_groups().  HAT_EUNSUPPORTED for shapes that are not instantiated: use hat_conv. */
extern "C" int hat_conv3x3_small_groups(const HatConvDesc* dp, int32_t* groups_out) {
    if (!dp || !groups_out) return HAT_EINVAL;
    return tap3_dispatch(*dp, nullptr, groups_out);
}

extern "C" int hat_conv3x3_small(const HatConvDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatConvDesc& d = *dp;
    if (!d.x || !d.w || !d.bias || !d.out || d.B < 1 || d.H < 1 || d.W < 1) return HAT_EINVAL;
    const int vec = d.dtype == HAT_BF16 ? 8 : 4;
    if (d.ldx % vec || d.n_store % 4 || d.ldo % 4 || d.n_store > d.nt * 16) return HAT_EINVAL;
    return tap3_dispatch(d, reinterpret_cast<hipStream_t>(stream), nullptr);
}

extern "C" int hat_aggr_cab(const HatAggrCabDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatConvDesc& d = dp->lin;
    if (!d.x || !d.w || !d.out || !d.r1 || !dp->c1 || !dp->wf || !dp->bias_b) return HAT_EINVAL;
    if (d.B < 1 || d.H < 1 || d.W < 1 || d.ksize != 1 || d.x_mode != HAT_X_NHWC_T || d.out_mode != HAT_O_NHWC_F32) return HAT_EINVAL;
    if (d.nt != 9 || d.n_slices != 1 || d.n_store != 144 || d.Cin != 144) return HAT_EUNSUPPORTED;
    const int vec = d.dtype == HAT_BF16 ? 8 : 4;
    if (d.ldx % vec || d.ldx < 144 || d.ldo % 4 || d.ldo < 144 || d.ldr1 % 4 || d.ldr1 < 144) return HAT_EINVAL;
    if (d.x0 && (d.ldx0 % vec || d.c_split % vec || d.c_split > d.Cin)) return HAT_EINVAL;
    if (d.dtype != HAT_BF16) return HAT_EUNSUPPORTED;  // (the fp32 image of the weights does not leave room for two workgroups)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)9 * (5 + 3) * 512 * sizeof(bf16_t) + (size_t)9 * 16 * sizeof(float);
    const long HW = (long)d.H * d.W;
    const int tiles = (int)((HW + 15) / 16);
    int gx = 512 / (d.B < 2 ? 1 : 2);
    if (gx > (tiles + 3) / 4) gx = (tiles + 3) / 4;
    auto kern = aggr_cab_kernel<bf16_t>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    HAT_LAUNCH(kern, dim3(gx, d.B, 1), dim3(256), lds, s, *dp, tiles);
    return hat_check_launch();
}
