// hat_conv.hip — implicit-GEMM convolution / pointwise linear for channel-last activations on
// gfx950, MFMA 16x16 tiles, fused epilogues.  See include/hat_mi355x.h (HatConvDesc) for the
// contract and the reference lines each use replaces.
//
// Workgroup = WAVES waves; output tile = (WAVES*PT) rows x 16 columns of pixels x (NT*16) channels
// per n-slice.  Orientation: MFMA A operand = packed weights (rows = output channel),
// B operand = activations (columns = 16 consecutive pixels of one tile row), so a lane's four
// accumulator registers are four CONSECUTIVE CHANNELS of one pixel (vector epilogue stores).
//
//   LDS:  Xs[(rows+2h) * (16+2h)][Cin_p (+pad)]   the haloed input tile, ALL input channels, staged once
//         Ws[NT*16][KC (+pad)]                    one K-chunk of the weight slice, register-prefetched
//   K is flat: k = tap * Cin_p + ci; a lane's 8-element k group never straddles a tap because
//   Cin_p % 8 == 0, so each lane tracks its own (tap, ci) and reads its B fragment from the
//   shifted pixel: no im2col buffer exists anywhere.
#include <cstdlib>
#include <type_traits>

#include "hat_common.h"

// hat_conv64r.hip: resident-weight kernel for the 3x3 convs of a 64-channel map (the Upsample convs)
bool hat_conv64r_can_launch(const HatConvDesc& d);
int hat_conv64r_launch(const HatConvDesc& d, hipStream_t s);

#ifndef HAT_CONV_PIPE
#define HAT_CONV_PIPE 1   // 0: round 2's K loop (operand reads left to the scheduler), for A/B builds
#endif

namespace {

struct TileCfg { int waves, pt; };

__host__ inline size_t conv_lds_bytes(const HatConvDesc& d, int waves, int pt, int es, int kc) {
    const int hl = d.ksize / 2;
    const int cin_p = (d.Cin + 7) & ~7;
    const size_t xs = (size_t)(waves * pt + 2 * hl) * (16 + 2 * hl) * lds_row_elems(cin_p, es) * es;
    const size_t ws = (size_t)d.nt * 16 * lds_row_elems(kc, es) * es;
    const size_t koff = ((size_t)(d.Kpad / 32) * 16 + 15) & ~(size_t)15;  // B-operand offset table: [k-step][4 lane groups]
    return xs + ws + koff;
}

__host__ inline bool conv_pick(const HatConvDesc& d, TileCfg* out, size_t* lds) {
    const int es = d.dtype == HAT_BF16 ? 2 : 4;
    const int kc = (d.dtype == HAT_BF16 ? 64 : 32) * (d.nt == 1 ? 3 : (d.nt <= 4 ? 2 : 1));
    const TileCfg cands[3] = {{8, 2}, {4, 2}, {4, 1}};
    // first choice: the largest tile that still lets TWO workgroups share a CU (LDS <= 80 KiB each), so one
    // workgroup's load / store phases overlap the other's MFMA phase — only when the weight slice is cheap to
    // re-stream per tile (smaller tiles re-read it more often); otherwise the largest tile that fits
    const size_t wbytes = (size_t)d.nt * 16 * d.ksize * d.ksize * ((d.Cin + 7) & ~7) * es;
    // (measured at 720p: 144 -> 64, 166 KB of weights, 0.278 ms with one 16-row tile per CU and 0.232 with two 8-row tiles;
    // 144 -> 144, 373 KB: 0.58 against 0.98)
    for (int pass = (wbytes <= 180000 ? 0 : 1); pass < 2; ++pass) {
        const size_t limit = pass == 0 ? HAT_LDS_MAX / 2 : HAT_LDS_MAX;
        for (int i = 0; i < 3; ++i) {
            const int rows = cands[i].waves * cands[i].pt;
            if (i < 2 && d.H * 2 <= rows) continue;  // do not waste most of a tall tile on a short image
            if (pass == 0 && i == 2) continue;       // a 1-row-per-wave tile is not worth it just for occupancy
            const size_t b = conv_lds_bytes(d, cands[i].waves, cands[i].pt, es, kc);
            if (b <= limit) { *out = cands[i]; *lds = b; return true; }
        }
    }
    return false;
}

template <typename T, int WAVES, int PT, int NT>
__global__ __launch_bounds__(WAVES * 64) void conv_kernel(const HatConvDesc d) {
    using M = MT<T>;
    constexpr int NTHR = WAVES * 64;
    constexpr int TROWS = WAVES * PT;
    // one-n-tile layers (CAB conv 144->6, the 13x13 ESC conv, conv_last) do 1 MFMA per k-step per pixel tile: their
    // chunk is 3x longer so the two barriers per chunk are amortised over 3x the work; up to 4 n-tiles (CAB squeeze of
    // the C = 180 models, conv_before_upsample) 2x.  Wider layers keep 64: the longer chunk's fetch registers spill there.
    constexpr int KC = M::KC * (NT == 1 ? 3 : (NT <= 4 ? 2 : 1));
    constexpr int KS = KC / 32;
    constexpr int PPR = KC * (int)sizeof(T) / 16;  // 16-byte pieces per weight-chunk row
    constexpr int VEC = M::VEC;
    constexpr int WPIECES = NT * 16 * PPR;  // 16-byte pieces per weight chunk
    constexpr int WPT = (WPIECES + NTHR - 1) / NTHR;

    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef HAT_CONV_STAMPS
    const long long st0 = (long long)__builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    if (d.reserved0 > 0) {
        // One workgroup per CU and equal tiles: all 256 CUs would move through staging (HBM reads), K loop (no HBM traffic) and
        // epilogue (HBM reads + writes) in lockstep, the memory system idle for half of every round and saturated for the
        // rest.  The first round's workgroups start in four groups, d.reserved0 x 8128 cycles apart (hat_conv sets it: about a
        // fifth of a tile time); later workgroups inherit the offsets because each starts when its CU's previous one ends.
        // Group conv at 720p: 0.600 -> 0.580 ms.
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (lin < 256u) {
            const int n = (int)((lin >> 3) & 3u) * d.reserved0;
            for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
    const int ks_ = d.ksize;
    const int hl = ks_ >> 1, TWH = 16 + 2 * hl, THH = TROWS + 2 * hl;
    const int Cin = d.Cin, Cin_p = (Cin + 7) & ~7;
    const int ldxs = lds_row_elems(Cin_p, sizeof(T));
    const int ldws = lds_row_elems(KC, sizeof(T));
    T* Xs = reinterpret_cast<T*>(smem);
    T* Ws = Xs + (size_t)THH * TWH * ldxs;
    int* koff = reinterpret_cast<int*>(Ws + NT * 16 * ldws);
    float* cs = reinterpret_cast<float*>(Ws);  // column-sum scratch, reuses Ws after the K loop

    const int b = blockIdx.z;
    const int x0 = blockIdx.x * 16, y0 = blockIdx.y * TROWS;
    const int H = d.H, W = d.W;
    const int ntaps = ks_ * ks_;

    // koff[k-step][g]: element offset (inside Xs, relative to the output pixel) of the 8-channel group that lane group g
    // feeds to that k-step: k = 32 * kstep + 8 * g = tap * Cin_p + ci  ->  (tap's row * TWH + tap's column) * ldxs + ci.
    // K-padding steps past the last tap point at the last tap (their weights are zero).  One LDS read per k-step replaces
    // per-lane (tap, ci) tracking, which cost more VALU time than the MFMAs of the one-n-tile layers.
    for (int i = tid; i < (d.Kpad / 32) * 4; i += NTHR) {
        const int k = 32 * (i >> 2) + 8 * (i & 3);
        int tp = k / Cin_p;
        const int cc = k - tp * Cin_p;
        tp = tp < ntaps ? tp : ntaps - 1;
        koff[i] = ((tp / ks_) * TWH + (tp % ks_)) * ldxs + cc;
    }

    // ---------------- stage the haloed input tile (all channels), zero filled ------------------
    const int npixh = THH * TWH;
    if (d.x_mode == HAT_X_NHWC_T) {
        const T* xg = reinterpret_cast<const T*>(d.x);
        const T* xg0 = reinterpret_cast<const T*>(d.x0);
        const int ppp = Cin_p / VEC;
        const int total = npixh * ppp;
        // SU pieces per thread per pass, every load unconditional (clamped address, zero selected afterwards): a load
        // under a lane mask is waited for where it is issued, which made this loop one memory round trip per piece
        // (13 in flight when the tile needs more than 6 per thread: the CAB squeeze conv's tile is 13 pieces per thread and
        // its K loop is short, so each extra round trip of staging showed directly in its time)
        // piece i = (haloed pixel q = (qy, qx), 16-byte piece r of its row): a thread's pieces are NTHR apart, so (q, r) and
        // (qy, qx) advance by fixed amounts with a carry — four integer divisions per thread in all.  (Two divisions by run-time
        // divisors per piece, done again for the store, were ~2 000 vector instructions per thread: more than half of this phase.)
        const int dq = NTHR / ppp, dr = NTHR - dq * ppp, dqy = dq / TWH, dqx = dq - dqy * TWH;
        auto stage = [&](auto su_tag) {
            constexpr int SU = decltype(su_tag)::value;
            int q = tid / ppp, r = tid - q * ppp;
            int qy = q / TWH, qx = q - qy * TWH;
            for (int i0 = tid; i0 < total; i0 += NTHR * SU) {
                u32x4 v[SU];
                bool ok[SU];
                int lo[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int c = r * VEC;
                    const int y = y0 - hl + qy, x = x0 - hl + qx;
                    ok[u] = y >= 0 && y < H && x >= 0 && x < W && c < Cin;
                    lo[u] = q * ldxs + c;
                    const size_t pix = ((size_t)b * H + min(max(y, 0), H - 1)) * W + min(max(x, 0), W - 1);
                    const T* src = (xg0 != nullptr && c < d.c_split) ? xg0 + pix * d.ldx0 + c : xg + pix * d.ldx + c;
                    v[u] = *reinterpret_cast<const u32x4*>(src);   // (pieces past the tile re-read a clamped pixel: never stored)
                    r += dr; q += dq; qx += dqx; qy += dqy;
                    if (r >= ppp) { r -= ppp; q += 1; qx += 1; }
                    if (qx >= TWH) { qx -= TWH; qy += 1; }
                }
#pragma unroll
                for (int u = 0; u < SU; ++u)
                    if (i0 + u * NTHR < total) *reinterpret_cast<u32x4*>(Xs + lo[u]) = ok[u] ? v[u] : u32x4{0u, 0u, 0u, 0u};
            }
        };
        if (total > NTHR * 6) stage(std::integral_constant<int, 13>{});
        else stage(std::integral_constant<int, 6>{});
    } else if (d.x_mode == HAT_X_NHWC_F32) {
        const float* xg = reinterpret_cast<const float*>(d.x);
        const int gpp = Cin_p / 4;
        const int total = npixh * gpp;
        auto stage = [&](auto su_tag) {   // see the bf16/T branch above
            constexpr int SU = decltype(su_tag)::value;
            for (int i0 = tid; i0 < total; i0 += NTHR * SU) {
                f32x4 v[SU];
                bool ok[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = min(i0 + u * NTHR, total - 1);
                    const int q = i / gpp, c = (i - q * gpp) * 4;
                    const int qy = q / TWH, qx = q - qy * TWH;
                    const int y = y0 - hl + qy, x = x0 - hl + qx;
                    ok[u] = y >= 0 && y < H && x >= 0 && x < W && c < Cin;
                    const size_t pix = ((size_t)b * H + min(max(y, 0), H - 1)) * W + min(max(x, 0), W - 1);
                    v[u] = *reinterpret_cast<const f32x4*>(xg + pix * d.ldx + min(c, Cin - 4));
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int i = i0 + u * NTHR;
                    if (i < total) {
                        const int q = i / gpp, c = (i - q * gpp) * 4;
                        Vec4<T>::store(Xs + (size_t)q * ldxs + c, ok[u] ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f});
                    }
                }
            }
        };
        if (total > NTHR * 6) stage(std::integral_constant<int, 12>{});
        else stage(std::integral_constant<int, 6>{});
    } else {  // HAT_X_NCHW_F32_MEAN: (x - mean[c]) * in_scale, zero padding applied AFTER the shift
        const float* xg = reinterpret_cast<const float*>(d.x);
        const int total = npixh * Cin_p;
        for (int i = tid; i < total; i += NTHR) {
            const int q = i / Cin_p, c = i - q * Cin_p;
            const int qy = q / TWH, qx = q - qy * TWH;
            const int y = y0 - hl + qy, x = x0 - hl + qx;
            float v = 0.f;
            if (y >= 0 && y < H && x >= 0 && x < W && c < Cin)
                v = (xg[(((size_t)b * Cin + c) * H + y) * W + x] - d.mean[c & 3]) * d.in_scale;
            Xs[(size_t)q * ldxs + c] = to_T<T>(v);
        }
    }

#ifdef HAT_CONV_STAMPS   // tools/conv_phases.py: per-wave cycle counts of the three phases -> gap_out[wg][wave][4]
    __syncthreads();
    const long long st1 = (long long)__builtin_amdgcn_s_memtime();
#endif
    const T* wg = reinterpret_cast<const T*>(d.w) + (size_t)b * d.w_bstride;
    const int nchunks = d.Kpad / KC;
    const int Npad = d.n_slices * NT * 16;
    const size_t tile_id = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const size_t ntiles = (size_t)gridDim.x * gridDim.y;

    for (int slice = 0; slice < d.n_slices; ++slice) {
        f32x4 acc[NT][PT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

        const T* wslice = wg + (size_t)slice * NT * 16 * d.Kpad;
        // Weight chunks go global -> registers -> LDS, fetched TWO chunks ahead of their commit (one chunk of MFMAs is
        // shorter than an L2 round trip) into two register sets used alternately; the loads are unconditional (clamped
        // piece / chunk) and the barriers are LDS-only, so the fetches stay in flight across them.
        u32x4 wra[WPT], wrb[WPT];
        auto w_fetch = [&](int chunk, u32x4 (&wr)[WPT]) {
            const int ch = chunk < nchunks ? chunk : nchunks - 1;
#pragma unroll
            for (int j = 0; j < WPT; ++j) {
                const int p_ = min(tid + j * NTHR, WPIECES - 1);
                wr[j] = *reinterpret_cast<const u32x4*>(wslice + (size_t)(p_ / PPR) * d.Kpad + (size_t)ch * KC + (p_ % PPR) * VEC);
            }
        };
        auto w_commit = [&](const u32x4 (&wr)[WPT]) {
#pragma unroll
            for (int j = 0; j < WPT; ++j) {
                const int p_ = tid + j * NTHR;
                if (WPIECES % NTHR == 0 || p_ < WPIECES) *reinterpret_cast<u32x4*>(Ws + (p_ / PPR) * ldws + (p_ % PPR) * VEC) = wr[j];
            }
        };

        const T* xrow = Xs + (size_t)(wave * PT * TWH + c16) * ldxs;  // this lane's pixel of the wave's first tile row
        auto k_chunk = [&](int chunk) {
            int ko[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ko[ks] = koff[(chunk * KS + ks) * 4 + g];
            if constexpr (HAT_CONV_PIPE && NT <= 9 && sizeof(T) == 2) {
                // Weight fragments are requested AD n-tiles ahead of their MFMAs into a small register ring, the activation
                // fragments of the next k-step while this one's last n-tiles run, and the scheduler may not move anything across
                // an n-tile (sched_barrier).  Left to itself it allocates ONE register quad for the weight fragment and runs
                // "read, wait, two MFMAs" nine times per k-step — the LDS latency nine times in the open, half the K loop's
                // time in the 3x3 convs of the wide layers (group conv at 720p: 0.658 -> 0.606 ms; a ring of 2, 4, 5: 0.602,
                // 0.614, 0.618 against 0.600 for 3 on one box).
                constexpr int AD = 3, TOT = KS * NT;
                typename M::frag_t af[AD + 1], bf[2][PT];
                auto rd_a = [&](int i) { af[i % (AD + 1)] = M::load(Ws + ((i % NT) * 16 + c16) * ldws + (i / NT) * 32 + 8 * g); };
                auto rd_b = [&](int ks) {
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) bf[ks & 1][pt] = M::load(xrow + (size_t)pt * TWH * ldxs + ko[ks]);
                };
                rd_b(0);
#pragma unroll
                for (int i = 0; i < AD; ++i) rd_a(i);
#pragma unroll
                for (int i = 0; i < TOT; ++i) {
                    const int ks = i / NT, nt = i % NT;
                    if (i + AD < TOT) rd_a(i + AD);
                    if (nt == (NT > AD ? NT - AD : 0) && ks + 1 < KS) rd_b(ks + 1);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = M::mma(af[i % (AD + 1)], bf[ks & 1][pt], acc[nt][pt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    typename M::frag_t bf[PT];
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) bf[pt] = M::load(xrow + (size_t)pt * TWH * ldxs + ko[ks]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const typename M::frag_t af = M::load(Ws + (nt * 16 + c16) * ldws + ks * 32 + 8 * g);
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = M::mma(af, bf[pt], acc[nt][pt]);
                    }
                }
            }
        };

        w_fetch(0, wra);
        w_fetch(1, wrb);
        for (int chunk = 0; chunk < nchunks; chunk += 2) {
            w_commit(wra);
            lds_barrier();  // Ws (and, first time, Xs / koff) visible
            w_fetch(chunk + 2, wra);
            k_chunk(chunk);
            lds_barrier();  // all waves done with Ws before the next commit / cs reuse
            if (chunk + 1 < nchunks) {
                w_commit(wrb);
                lds_barrier();
                w_fetch(chunk + 3, wrb);
                k_chunk(chunk + 1);
                lds_barrier();
            }
        }

#ifdef HAT_CONV_STAMPS
        const long long st2 = (long long)__builtin_amdgcn_s_memtime();
#endif
        // ---------------------------------- epilogue ------------------------------------------
        const int nbase = slice * NT * 16;
        const bool want_cs = d.colsum != nullptr;
        constexpr int NG = NT % 3 == 0 ? 3 : (NT % 2 == 0 ? 2 : 1);
        const bool has_r1 = d.r1 != nullptr, has_r2 = d.r2 != nullptr;
        // bf16 layers: Pass A — every load of the epilogue, no store: bias, activation and the residual terms go into the
        // accumulators group by group.  Pass B — the stores.  vmcnt counts loads and stores in issue order, so a load issued after
        // stores is waited for together with every one of those stores' acknowledgements; interleaved (round 2, and still the
        // fp32 instantiations, whose registers do not allow the split) that happens four to five times per tile — each later
        // residual batch behind the previous batch's stores, the LayerNorm's gamma / beta behind the fp32 stores of each pixel
        // row.  The LayerNorm parameters of the fused form come from an LDS table (behind the column-sum scratch in the weight
        // chunk's buffer, which every wave is done with), filled here.
        constexpr bool TWO_PASS = sizeof(T) == 2;
        float* lnt = cs + 2048;
        float lpg = 0.f, lpb = 0.f;
        const bool ln_tab = TWO_PASS && NT >= 8 && NT * 16 <= NTHR && d.ln_out != nullptr;
        if (ln_tab && tid < NT * 16) { lpg = d.ln_g[tid]; lpb = d.ln_b[tid]; }
        // store of one finished (n-tile, pixel row) by out_mode
        auto put = [&](int nt, int pt, const f32x4& v, f32x4& csum) {
            const int n = nbase + nt * 16 + 4 * g;  // first of this lane's 4 consecutive channels
            const int y = y0 + wave * PT + pt, x = x0 + c16;
            const bool valid = (y < H) && (x < W);
            if (valid && n < d.n_store) {
                const size_t pix = ((size_t)b * H + y) * W + x;
                if (d.out_mode == HAT_O_NHWC_T) {
                    Vec4<T>::store(reinterpret_cast<T*>(d.out) + pix * d.ldo + n, v);
                } else if (d.out_mode == HAT_O_NHWC_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.out) + pix * d.ldo + n) = v;
                } else if (d.out_mode == HAT_O_PIXSHUF_T) {
                    const int r_ = d.ps_r, cps = d.n_store / (r_ * r_);
                    const int ij = n / cps, cc = n - ij * cps;
                    const int i_ = ij / r_, j_ = ij - i_ * r_;
                    const size_t opix = ((size_t)b * H * r_ + (size_t)y * r_ + i_) * ((size_t)W * r_) + (size_t)x * r_ + j_;
                    Vec4<T>::store(reinterpret_cast<T*>(d.out) + opix * d.ldo + cc, v);
                } else {  // HAT_O_NCHW_F32
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < d.n_store)
                            reinterpret_cast<float*>(d.out)[(((size_t)b * d.n_store + n + r) * H + y) * W + x] =
                                v[r] * d.out_scale + d.mean[(n + r) & 3];
                }
            }
            if (want_cs && valid) csum += d.out_mode == HAT_O_NHWC_T ? as_stored<T>(v) : v;   // (the pool of the STORED map)
        };
        auto put_cs = [&](int nt, const f32x4& csum) {
            if (want_cs) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = csum[r];
                    s = row_sum16(s);
                    if (c16 == 0) cs[wave * (NT * 16) + nt * 16 + 4 * g + r] = s;
                }
            }
        };
#pragma unroll
        for (int ng = 0; ng < NT; ng += NG) {
            f32x4 r1v[NG][PT], scv[NG], biasv[NG];
            typename Vec4<T>::raw_t r2v[NG][PT];
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int n = nbase + (ng + i) * 16 + 4 * g;
                const int nc = min(n, d.n_store - 4) & ~3;
                biasv[i] = *reinterpret_cast<const f32x4*>(d.bias + n);
                if (has_r2) scv[i] = *reinterpret_cast<const f32x4*>(d.r2scale + (size_t)b * d.r2scale_bstride + nc);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const size_t pixc = ((size_t)b * H + min(y0 + wave * PT + pt, H - 1)) * W + min(x0 + c16, W - 1);
                    if (has_r1) r1v[i][pt] = *reinterpret_cast<const f32x4*>(d.r1 + pixc * d.ldr1 + nc);
                    if (has_r2) r2v[i][pt] = Vec4<T>::load_raw(reinterpret_cast<const T*>(d.r2) + pixc * d.ldr2 + nc);
                }
            }
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int nt = ng + i;
                f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    f32x4 v = acc[nt][pt] + biasv[i];
                    if (d.act == HAT_ACT_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_act<T>(v[r]);
                    } else if (d.act == HAT_ACT_LRELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : 0.01f * v[r];
                    }
                    if (has_r1) v += r1v[i][pt];
                    if (has_r2) v += scv[i] * Vec4<T>::cvt(r2v[i][pt]);
                    acc[nt][pt] = v;   // (pass B and the fused LayerNorm below need the finished pixel)
                    if constexpr (!TWO_PASS) put(nt, pt, v, csum);
                }
                if constexpr (!TWO_PASS) put_cs(nt, csum);
            }
        }
        if (ln_tab) {
            if (tid < NT * 16) { lnt[tid] = lpg; lnt[NT * 16 + tid] = lpb; }
            lds_barrier();
        }
        if constexpr (TWO_PASS) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) put(nt, pt, acc[nt][pt], csum);
                put_cs(nt, csum);
            }
        }
        if (d.ln_out != nullptr) {
            // LayerNorm (eps 1e-5) of the finished pixels for the consumer of this conv's output (host checks: one slice,
            // every channel stored): a pixel's NT*16 channels sit in the four lane groups of its column.
            const float invC = 1.0f / (float)(NT * 16);
            f32x4 gapv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int y = y0 + wave * PT + pt, x = x0 + c16;
                const bool valid = (y < H) && (x < W);
                const size_t pix = ((size_t)b * H + min(y, H - 1)) * W + min(x, W - 1);
                float sm = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) sm += (acc[nt][pt][0] + acc[nt][pt][1]) + (acc[nt][pt][2] + acc[nt][pt][3]);
                sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
                const float mean = sm * invC;
                float qq = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float dl = acc[nt][pt][r] - mean; qq += dl * dl; }
                qq += __shfl_xor(qq, 16); qq += __shfl_xor(qq, 32);
                const float rstd = 1.0f / sqrtf(qq * invC + 1e-5f);
                T* lo = reinterpret_cast<T*>(d.ln_out) + pix * d.ld_ln;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = nt * 16 + 4 * g;
                    const f32x4 gm = ln_tab ? *reinterpret_cast<const f32x4*>(lnt + n) : *reinterpret_cast<const f32x4*>(d.ln_g + n);
                    const f32x4 bt = ln_tab ? *reinterpret_cast<const f32x4*>(lnt + NT * 16 + n) : *reinterpret_cast<const f32x4*>(d.ln_b + n);
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (acc[nt][pt][r] - mean) * rstd * gm[r] + bt[r];
                    if (valid) {
                        Vec4<T>::store(lo + n, o);
                        if (nt == 0) {
                            if (d.n16_out != nullptr) Vec4<T>::store(reinterpret_cast<T*>(d.n16_out) + pix * 16 + n, o);
                            if (n < d.gap_c) gapv += as_stored<T>(o);
                        }
                    }
                }
            }
            if (d.gap_out != nullptr) {
                lds_barrier();   // (cs aliases the weight chunk: every wave is past its last read)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s_ = row_sum16(gapv[r]);
                    if (c16 == 0) cs[wave * 16 + 4 * g + r] = s_;
                }
                __syncthreads();
                if (tid < 16) {
                    float s_ = 0.f;
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) s_ += cs[w * 16 + tid];
                    d.gap_out[((size_t)b * ntiles + tile_id) * 16 + tid] = tid < d.gap_c ? s_ : 0.f;
                }
                __syncthreads();
            }
        }
#ifdef HAT_CONV_STAMPS
        if (d.ln_out == nullptr && d.gap_out != nullptr) {
            __builtin_amdgcn_s_waitcnt(0);
            const long long st3 = (long long)__builtin_amdgcn_s_memtime();
            if (lane == 0) {
                float* o = d.gap_out + ((size_t)tile_id * WAVES + wave) * 8;
                o[0] = (float)(st1 - st0); o[1] = (float)(st2 - st1); o[2] = (float)(st3 - st2); o[3] = (float)(st0 & 0xffffff);
                o[4] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffff);    // HW_ID
                o[5] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xffff);     // XCC_ID
                o[6] = (float)((st0 >> 24) & 0xffffff);
            }
        }
#endif
        if (want_cs) {
            __syncthreads();
            for (int n = tid; n < NT * 16; n += NTHR) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) s += cs[w * (NT * 16) + n];
                d.colsum[((size_t)b * ntiles + tile_id) * Npad + nbase + n] = s;
            }
            __syncthreads();
        }
    }
}

int* g_occ_query = nullptr;   // hat_conv_occupancy: when set, launch_conv reports workgroups per CU instead of launching

template <typename T, int WAVES, int PT, int NT>
int launch_conv(const HatConvDesc& d, size_t lds, hipStream_t stream) {
    auto kern = conv_kernel<T, WAVES, PT, NT>;
    if (g_occ_query != nullptr) {
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(g_occ_query, kern, WAVES * 64, lds);
    }
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((d.W + 15) / 16, (d.H + WAVES * PT - 1) / (WAVES * PT), d.B);
    HatConvDesc dd = d;
    // stagger (kernel side: see conv_kernel): only for launches that keep every CU busy for several rounds of ONE workgroup each
    static const int stag = [] { const char* e = getenv("HAT_CONV_STAGGER"); return e ? atoi(e) : 2; }();
    dd.reserved0 = (lds > HAT_LDS_MAX / 2 && (size_t)grid.x * grid.y * grid.z >= 1024) ? stag : 0;
    HAT_LAUNCH(kern, grid, dim3(WAVES * 64), lds, stream, dd);
    return hat_check_launch();
}

template <typename T, int WAVES, int PT>
int dispatch_nt(const HatConvDesc& d, size_t lds, hipStream_t s) {
    switch (d.nt) {
        case 1: return launch_conv<T, WAVES, PT, 1>(d, lds, s);
        case 4: return launch_conv<T, WAVES, PT, 4>(d, lds, s);
        case 8: return launch_conv<T, WAVES, PT, 8>(d, lds, s);   // 256 outputs = 2 x 8 tiles exactly (the upsampler convs)
        case 9: return launch_conv<T, WAVES, PT, 9>(d, lds, s);
        case 12: return launch_conv<T, WAVES, PT, 12>(d, lds, s);
        default: return HAT_EINVAL;
    }
}

template <typename T>
int dispatch_tile(const HatConvDesc& d, TileCfg tc, size_t lds, hipStream_t s) {
    if (tc.waves == 8 && tc.pt == 2) return dispatch_nt<T, 8, 2>(d, lds, s);
    if (tc.waves == 4 && tc.pt == 2) return dispatch_nt<T, 4, 2>(d, lds, s);
    return dispatch_nt<T, 4, 1>(d, lds, s);
}

int conv_validate(const HatConvDesc& d) {
    if (!d.x || !d.w || !d.bias || !d.out) return HAT_EINVAL;
    if (d.B < 1 || d.H < 1 || d.W < 1 || d.Cin < 1) return HAT_EINVAL;
    if (d.ksize < 1 || (d.ksize & 1) == 0 || d.ksize > 17) return HAT_EINVAL;   // (17: the OCAB-ESC kernel of the HATX training config)
    if (d.dtype != HAT_F32 && d.dtype != HAT_BF16) return HAT_EINVAL;
    const int kc = (d.dtype == HAT_BF16 ? 64 : 32) * (d.nt == 1 ? 3 : (d.nt <= 4 ? 2 : 1)), vec = d.dtype == HAT_BF16 ? 8 : 4;
    const int cin_p = (d.Cin + 7) & ~7;
    if (d.Kpad % kc || d.Kpad < d.ksize * d.ksize * cin_p) return HAT_EINVAL;
    if (d.n_slices < 1 || d.n_store < 1 || d.n_store > d.n_slices * d.nt * 16) return HAT_EINVAL;
    if (d.x_mode == HAT_X_NHWC_T) {
        if (d.ldx % vec || d.ldx < cin_p) return HAT_EINVAL;  // 16-byte rows; pad channels up to Cin_p must exist (zeros)
        if (d.x0 && (d.ldx0 % vec || d.c_split % vec || d.c_split > d.Cin)) return HAT_EINVAL;
    } else if (d.x_mode == HAT_X_NHWC_F32) {
        if (d.ldx % 4 || d.Cin % 4 || d.x0) return HAT_EINVAL;
    } else if (d.x_mode == HAT_X_NCHW_F32_MEAN) {
        if (d.Cin > 4 || d.x0) return HAT_EINVAL;
    } else {
        return HAT_EINVAL;
    }
    if (d.out_mode == HAT_O_NCHW_F32) {
        if (d.n_store > 4 || d.r1 || d.r2) return HAT_EINVAL;
    } else {
        if (d.n_store % 4 || d.ldo % 4) return HAT_EINVAL;
        if (d.out_mode == HAT_O_PIXSHUF_T) {
            if (d.ps_r < 2 || d.n_store % (d.ps_r * d.ps_r) || (d.n_store / (d.ps_r * d.ps_r)) % 4 || d.r1 || d.r2) return HAT_EINVAL;
        } else if (d.out_mode != HAT_O_NHWC_T && d.out_mode != HAT_O_NHWC_F32) {
            return HAT_EINVAL;
        }
    }
    if (d.r1 && d.ldr1 % 4) return HAT_EINVAL;
    if (d.r2 && (d.ldr2 % 4 || !d.r2scale)) return HAT_EINVAL;
    if (d.ln_out) {   // fused LayerNorm of the output: whole pixels in one slice, NHWC rows
        if (!d.ln_g || !d.ln_b || d.ln_ones || d.n_slices != 1 || d.n_store != d.nt * 16 || d.ld_ln % 4 || d.ld_ln < d.n_store) return HAT_EINVAL;
        if (d.out_mode != HAT_O_NHWC_T && d.out_mode != HAT_O_NHWC_F32) return HAT_EINVAL;
        if (d.colsum || d.gap_c < 0 || d.gap_c > 16 || d.gap_c % 4) return HAT_EINVAL;
    } else if (d.gap_out || d.n16_out) {
#ifndef HAT_CONV_STAMPS
        return HAT_EINVAL;
#endif
    }
    return 0;
}

}  // namespace

extern "C" int hat_conv_tiles(const HatConvDesc* d, int32_t* tiles_out) {
    if (!d || !tiles_out) return HAT_EINVAL;
    TileCfg tc; size_t lds;
    if (!conv_pick(*d, &tc, &lds)) return HAT_ELDS;
    *tiles_out = ((d->W + 15) / 16) * ((d->H + tc.waves * tc.pt - 1) / (tc.waves * tc.pt));
    return 0;
}

extern "C" int hat_conv_plan(const HatConvDesc* d, int32_t* waves, int32_t* rows_per_wave, int32_t* tiles, int64_t* lds_bytes) {
    if (!d || !waves || !rows_per_wave || !tiles || !lds_bytes) return HAT_EINVAL;
    TileCfg tc; size_t lds;
    if (!conv_pick(*d, &tc, &lds)) return HAT_ELDS;
    *waves = tc.waves; *rows_per_wave = tc.pt; *lds_bytes = (int64_t)lds;
    *tiles = ((d->W + 15) / 16) * ((d->H + tc.waves * tc.pt - 1) / (tc.waves * tc.pt));
    return 0;
}

/* Debug query (not thread safe): workgroups of hat_conv's kernel for `d` that fit one CU (registers, LDS). */
extern "C" int hat_conv_occupancy(const HatConvDesc* dp, int32_t* wgs_per_cu) {
    if (!dp || !wgs_per_cu) return HAT_EINVAL;
    TileCfg tc; size_t lds;
    if (!conv_pick(*dp, &tc, &lds)) return HAT_ELDS;
    int v = 0;
    g_occ_query = &v;
    const int rc = dp->dtype == HAT_BF16 ? dispatch_tile<bf16_t>(*dp, tc, lds, nullptr) : dispatch_tile<float>(*dp, tc, lds, nullptr);
    g_occ_query = nullptr;
    *wgs_per_cu = v;
    return rc;
}

extern "C" int hat_conv(const HatConvDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatConvDesc& d = *dp;
    int rc = conv_validate(d);
    if (rc) return rc;
    TileCfg tc; size_t lds;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hat_conv64r_can_launch(d)) return hat_conv64r_launch(d, s);
    if (!conv_pick(d, &tc, &lds)) return HAT_ELDS;
    if (d.dtype == HAT_BF16) return dispatch_tile<bf16_t>(d, tc, lds, s);
    return dispatch_tile<float>(d, tc, lds, s);
}
