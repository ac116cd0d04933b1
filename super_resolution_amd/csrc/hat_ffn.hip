// hat_ffn.hip — fused HAB feed-forward half on gfx950 (contract: HatFfnDesc in include/hat_mi355x.h):
//     t_out = t_in + fc2( a * SiLU(g) ),   [a | g] = dwconv3x3( fc1( LayerNorm2(t_in) ) )
// reference: hat/archs/hat_arch.py:237 (x + mlp(norm2(x))) with GatedDconvFFN.forward :107-119.
//
// One workgroup = WAVES waves = a (2*WAVES) x 16 pixel tile.  The 4C-wide hidden tensor (the
// reference materialises it twice in HBM: 1-6 GB) lives only on chip, 32 (+32 gate) channels at a time:
//
//   stage 0   LN2 of the haloed (rows+2) x 18 input tile, fp32 stats; column C of every in-image
//             row is 1.0 so that the fc1 bias is one more k (out-of-image rows are all zero, hence
//             U = 0 there: the depthwise conv zero-pads u AFTER the fc1 bias)       -> Ms[pix][Kp]  (T, LDS)
//   per chunk c of 32 hidden channels (a-part) + their 32 gate channels:
//     fc1     U = Ms . [W1[chunk] | b1]^T on ALL haloed pixels (MFMA)               -> Us[pix][64]  (T, LDS)
//     dw      depthwise 3x3 as MFMA: D[ch][pix] += diag(w_tap) . U_tap, two taps per 32-deep k-step
//             (A = [diag(w_t0) | diag(w_t1)] built in registers from one per-lane weight, B = the
//             shifted pixels' 16 channels read from Us); the 10th "tap" multiplies a constant 1 by
//             the depthwise bias.                                                          (registers)
//     gate    a * SiLU(g) on the accumulators; because D has channels on the accumulator-row index,
//             the product is already the B operand of fc2 (k order permuted identically in W2)
//     fc2     acc[C][pix] += W2[:, chunk] . G   (MFMA; accumulators persist across chunks)
//   epilogue  t_out = t_in + acc + b2 ; optionally the NEXT LayerNorm of t_out and its GAP partials.
//
// Weights never touch LDS: they are fragment-packed on the host, so a wave's A operand is ONE
// coalesced 1 KiB global load (L1/L2 resident), issued a stage ahead of its use.  Each wave runs dw,
// gate and fc2 on its OWN two tile rows, so only U crosses waves (2 barriers per chunk).
#include "hat_common.h"

#include <type_traits>

namespace {

// source of out-of-image halo rows on the pre-normalised (m_in) path
__device__ __attribute__((aligned(16))) unsigned hat_ffn_zero_page[4] = {0, 0, 0, 0};

constexpr int CH = 32;        // hidden channels per chunk (a-part); the chunk also carries CH gate channels
constexpr int HALO_W = 18;    // 16 + 2
constexpr int NPAIR = 5;      // 9 taps, two per k-step

// Swizzled 16-byte slot inside the [rows][NS slots] Us array, NS a power of two <= 16: slot ^ (row & (NS-1)).
// For the 8-slot (128-byte, bf16) rows this is conflict-free for the depthwise operand reads — ds_read_b128's lane
// groups mix rows 0-3 / 12-15 of one k-half with rows 4-11 of the other, for every tap alignment — and 2-way on the
// 8-byte fc1 stores (enumerated with the lane groups of MI355X_MICROARCH.md; the earlier (row/2)&7 read 1.75x slower).
template <int NS> __device__ __forceinline__ int swz_slot(int row, int slot) { return slot ^ (row & (NS - 1)); }

// K of fc1 padded to a multiple of 32 with room for the bias column at k = C
__host__ __device__ inline int ffn_kp(int C) { return (C + 1 + 31) & ~31; }
// Ms row stride: MSWZ = unpadded rows + XOR swizzle (rows of 4 (mod 8) slots, e.g. 160 bf16), else odd-slot padding
template <typename T, bool MSWZ> __host__ __device__ inline int ffn_ldm(int C) {
    return MSWZ ? ffn_kp(C) : lds_row_elems(ffn_kp(C), sizeof(T));
}

template <typename T, int WAVES, bool MSWZ, int RPW = 2>
__host__ __device__ inline size_t ffn_lds_bytes(int C) {
    const int nph = (RPW * WAVES + 2) * HALO_W;
    const size_t ms = ((size_t)nph * ffn_ldm<T, MSWZ>(C) * sizeof(T) + 255) & ~(size_t)255;  // Us starts 256-byte aligned
    const size_t us = (size_t)(nph + 1) * 2 * CH * sizeof(T);  // + the constant-one row (depthwise bias)
    return ms + us;
}

__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_exact(float x) { return x / (1.0f + expf(-x)); }

// DBG is a timing-ablation mask used only by tools/ubench_ffn.hip (the library instantiates DBG = 0):
//   1 skip LN stage, 2 skip fc1 MFMA loop, 4 skip the dw MFMAs, 8 skip gate math, 16 skip fc2 MFMAs,
//   32 skip all weight-fragment loads, 64 per-phase s_memtime totals of every wave -> gap_out[wg][wave][8]
// MINW = waves per SIMD the register allocation must allow (2 when two workgroups' LDS fit one CU)
// RPW = tile rows per wave in the depthwise / gate / fc2 phases (2, or 1 when eight waves share an 8-row tile because the
// LDS image is too large for two workgroups per CU: C = 180)
template <typename T, int WAVES, int NT, int KS, bool MSWZ, int MINW, int DBG = 0, int RPW = 2>
__global__ __launch_bounds__(WAVES * 64, MINW) void ffn_kernel(const HatFfnDesc d) {
    using M = MT<T>;
    using frag_t = typename M::frag_t;
    constexpr int NTHR = WAVES * 64;
    constexpr int TROWS = RPW * WAVES;
    constexpr int NPH = (TROWS + 2) * HALO_W;     // haloed pixels
    constexpr int NPT = (NPH + 15) / 16;          // fc1 pixel tiles over the flattened halo tile
    constexpr int VECN = M::VEC;                  // elements per 16 bytes
    constexpr int NSU = 2 * CH / VECN;            // 16-byte slots per Us row
    constexpr int PG = WAVES / 2;                 // fc1 pixel groups (each handled by 2 waves: n-tile pairs)
    constexpr bool BF = sizeof(T) == 2;
    // pixels per 16-lane group kept in flight in the LN stage: bf16 takes the whole haloed tile in ONE pass (nothing else
    // is live yet), because every pass costs a full HBM round trip
    constexpr int LNB = BF ? (NPH + NTHR / 16 - 1) / (NTHR / 16) : 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = d.C;
    constexpr int Kp = KS * 32;                   // == ffn_kp(C): checked by the launcher
    constexpr int ldm = MSWZ ? Kp : lds_row_elems(Kp, sizeof(T));
    // 16-byte slot -> element offset inside an Ms row (swizzled when MSWZ: rows of 4 (mod 8) slots collide 4 apart)
    // MSWZ rows are 20 slots (4 mod 16), so rows r and r+4 start on the same banks; ds_read_b128 is serviced in the lane
    // groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... i.e. rows 0-3 and 12-15 of lane group g TOGETHER WITH rows 4-11 of
    // g^1.  XOR-ing the slot with (-(row>>2)) & 3 = {0,3,2,1} makes all 16 reads of every such group hit distinct banks
    // (with (row>>2)&3 they were exactly 2-way: SQ_LDS_BANK_CONFLICT).
    auto ms_slot = [](int row, int slot) { return (MSWZ ? (slot ^ ((0 - (row >> 2)) & 3)) : slot) * VECN; };
    T* Ms = reinterpret_cast<T*>(smem);
    T* Us = reinterpret_cast<T*>(smem + (((size_t)NPH * ldm * sizeof(T) + 255) & ~(size_t)255));  // 256-byte aligned: see uoff

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * TROWS;
    const int H = d.H, W = d.W;
    const float* tin = d.t_in + (size_t)b * H * W * C;
    const int hid_p = d.chunks * CH;

    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = 0;
    auto stamp = [&](int slot) {
        if constexpr (DBG & 64) {
            const long long now = (long long)__builtin_amdgcn_s_memtime();
            tph[slot] += now - tlast;
            tlast = now;
        }
    };
    if constexpr (DBG & 64) tlast = (long long)__builtin_amdgcn_s_memtime();
    // ------------------------------ stage 0: LayerNorm2 -> Ms --------------------------------
    // Every global load here is unconditional (clamped address, result discarded by a select): a load under a lane
    // mask whose result merges with a default makes the compiler wait for it right where it is issued, which turned
    // this stage into one memory round trip per load.  gamma/beta are read once, not per pixel.
    if (d.m_in != nullptr) {
        // The producer (hat_linear with ln_out / ln_ones) already wrote LayerNorm2(t_in) as rows [LN (C) | 1.0 | 0...]:
        // stage 0 is a copy of the haloed tile, 16 bytes per item, all loads issued before the first LDS store;
        // out-of-image rows come from a zero page.
        constexpr int SPR = Kp / VECN;                  // 16-byte slots per Ms row
        constexpr int NIT = (NPH * SPR + NTHR - 1) / NTHR;
        const T* mg = reinterpret_cast<const T*>(d.m_in) + (size_t)b * H * W * d.ldm_in;
        u32x4 cv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = min(tid + it * NTHR, NPH * SPR - 1);
            const int hp = i / SPR, sl = i - hp * SPR;
            const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool inside = y >= 0 && y < H && x >= 0 && x < W;
            const T* src = inside ? mg + ((size_t)y * W + x) * d.ldm_in + sl * VECN : reinterpret_cast<const T*>(hat_ffn_zero_page);
            cv[it] = *reinterpret_cast<const u32x4*>(src);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * NTHR;
            if (i < NPH * SPR) {
                const int hp = i / SPR, sl = i - hp * SPR;
                *reinterpret_cast<u32x4*>(Ms + (size_t)hp * ldm + ms_slot(hp, sl)) = cv[it];
            }
        }
    } else if constexpr (!(DBG & 1)) {
        const int j = tid & 15, grp = tid >> 4;
        constexpr int NGRP = NTHR / 16;
        const float invC = 1.0f / (float)C;
        f32x4 gmv[3], btv[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int c = min(4 * j + 64 * v, C - 4);
            gmv[v] = *reinterpret_cast<const f32x4*>(d.ln_g + c);
            btv[v] = *reinterpret_cast<const f32x4*>(d.ln_b + c);
        }
        for (int base = 0; base < NPH; base += NGRP * LNB) {
            f32x4 xv[LNB][3];
            bool inside[LNB];
#pragma unroll
            for (int u = 0; u < LNB; ++u) {  // issue all loads of LNB pixels first (latency), then reduce
                const int hp = base + u * NGRP + grp;
                const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                inside[u] = hp < NPH && y >= 0 && y < H && x >= 0 && x < W;
                const float* src = tin + ((size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * C;
#pragma unroll
                for (int v = 0; v < 3; ++v) xv[u][v] = *reinterpret_cast<const f32x4*>(src + min(4 * j + 64 * v, C - 4));
            }
#pragma unroll
            for (int u = 0; u < LNB; ++u) {
                const int hp = base + u * NGRP + grp;
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    if (!(inside[u] && 4 * j + 64 * v < C)) xv[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
                    s += (xv[u][v][0] + xv[u][v][1]) + (xv[u][v][2] + xv[u][v][3]);
                }
                s = row_sum16(s);
                const float mean = s * invC;
                float q = 0.f;
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    if (4 * j + 64 * v < C) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float dl = xv[u][v][r] - mean; q += dl * dl; }
                    }
                }
                q = row_sum16(q);
                const float rstd = BF ? __builtin_amdgcn_rsqf(q * invC + 1e-5f) : 1.0f / sqrtf(q * invC + 1e-5f);
                if (hp < NPH) {
#pragma unroll
                    for (int v = 0; v < 3; ++v) {   // selects, not branches; lanes past the row's end rewrite its last (zero) group
                        const int c = 4 * j + 64 * v;
                        if (64 * v < Kp) {
                            const bool live = inside[u] && c < C;
                            f32x4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = live ? (xv[u][v][r] - mean) * rstd * gmv[v][r] + btv[v][r] : 0.f;
                            const int cs = min(c, Kp - 4);           // (cs >= C whenever it differs from c)
                            if (inside[u] && cs == C) o[0] = 1.0f;   // bias column (C % 4 == 0)
                            Vec4<T>::store(Ms + (size_t)hp * ldm + ms_slot(hp, cs / VECN) + (cs % VECN), o);
                        }
                    }
                }
            }
        }
    }

    // persistent fc2 accumulators: this wave's two tile rows x all NT channel tiles, initialised with the residual
    // t_in + b2 (clamped, unconditional loads that complete during chunk 0; out-of-image pixels and pad channels hold
    // values that are never stored)
    f32x4 acc2[NT][RPW];
#pragma unroll
    for (int pt = 0; pt < RPW; ++pt) {
        const size_t pixc = (size_t)min(y0 + RPW * wave + pt, H - 1) * W + min(x0 + c16, W - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + 4 * g;
            acc2[nt][pt] = *reinterpret_cast<const f32x4*>(d.b2 + n) + *reinterpret_cast<const f32x4*>(tin + pixc * C + min(n, C - 4));
        }
    }

    const T* w1f = reinterpret_cast<const T*>(d.w1f);
    const T* w2f = reinterpret_cast<const T*>(d.w2f);
    const int nt2 = wave & 1, pg = wave >> 1;
    const int jstar = c16 & 7;                          // position of this lane's channel inside its 8-wide k group
    // bf16: dword masks that place the (duplicated) 16-bit weight at element jstar of an otherwise zero fragment
    unsigned dmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dmask[i] = (i == (jstar >> 1)) ? ((jstar & 1) ? 0xFFFF0000u : 0x0000FFFFu) : 0u;
    using wdw_t = typename std::conditional<sizeof(T) == 2, unsigned, float>::type;
    const wdw_t* dwl = reinterpret_cast<const wdw_t*>(d.dww);  // [chunk][lane][4 groups * NPAIR]

    // fc1 weight fragments of the CURRENT chunk (loaded one chunk ahead, during the previous fc2)
    frag_t a1[2][KS];
    auto load_a1 = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                a1[i][ks] = (DBG & 32) ? M::zero() : M::load(w1f + ((((size_t)chunk * 4 + (2 * nt2 + i)) * KS + ks) * 64 + lane) * 8);
    };
    // B operand of fc1: the LayerNorm'ed pixels of one 16-pixel tile, all K.  A wave's tiles are PG * 16 rows
    // apart, which leaves both swizzles unchanged, so every read/write of phase A is ONE per-lane base plus a
    // compile-time offset.  Tiles may run past NPH (into Us): those columns are never stored.
    const int hpa = pg * 16 + c16;
    const T* mbase = Ms + (size_t)hpa * ldm + ms_slot(hpa, BF ? g : 2 * g);
    auto load_b = [&](int i, frag_t (&bf)[KS]) {
        const T* mrow = mbase + (size_t)i * (PG * 16 * ldm);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (BF) {
                bf[ks] = M::load(mrow + ks * 32);
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(mrow + ks * 32);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(mrow + ks * 32 + 4);
                bf[ks][0] = lo[0]; bf[ks][1] = lo[1]; bf[ks][2] = lo[2]; bf[ks][3] = lo[3];
                bf[ks][4] = hi[0]; bf[ks][5] = hi[1]; bf[ks][6] = hi[2]; bf[ks][7] = hi[3];
            }
        }
    };
    // this lane's two Us store positions (n-tiles 2*nt2, 2*nt2+1) for its first tile
    T* ust[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int nl = (2 * nt2 + ii) * 16 + 4 * g;  // chunk-local channel: [0,32) a-part, [32,64) gate
        ust[ii] = Us + (size_t)hpa * 2 * CH + swz_slot<NSU>(hpa, nl / VECN) * VECN + (nl % VECN);
    }
    // Depthwise B operands (the shifted pixels' 16 channels, read from Us).  Lanes g < 2 carry tap 2*pr, lanes
    // g >= 2 tap 2*pr + 1; lanes g >= 2 of pair 4 carry the BIAS "tap" and read the constant-one row NPH.
    // uoff[pr][row] = LDS byte address of this lane's group-0 fragment; the Us rows are ROWB (a power of two)
    // bytes and Us is ROWB-aligned, so group gi's fragment is at uoff ^ (gi * ROWB / 4): one v_xor per read,
    // and the ten addresses are chunk-invariant.
    constexpr unsigned ROWB = 2 * CH * sizeof(T);
    unsigned uoff[NPAIR][RPW];
    {
        const unsigned usb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)Us;
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            const int tapr = 2 * pr + (g >> 1);
            const int tap = tapr < 9 ? tapr : 8;
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
            const int hp0 = (RPW * wave + dy) * HALO_W + c16 + dx;   // row 0 of this wave; row 1 is one halo row below
#pragma unroll
            for (int pt = 0; pt < RPW; ++pt) {
                const int hp = tapr < 9 ? hp0 + pt * HALO_W : NPH;
                uoff[pr][pt] = usb + (unsigned)hp * ROWB + (unsigned)swz_slot<NSU>(hp, (BF ? 1 : 2) * (g & 1)) * 16u;
            }
        }
    }
    auto ldu = [&](int pr, int pt, int gi) -> frag_t {
        const unsigned a = uoff[pr][pt] ^ ((unsigned)gi * (ROWB / 4));
        if constexpr (BF) {
            return *(__attribute__((address_space(3))) const frag_t*)(uintptr_t)(a);
        } else {
            const f32x4 lo = *(__attribute__((address_space(3))) const f32x4*)(uintptr_t)(a);
            const f32x4 hi = *(__attribute__((address_space(3))) const f32x4*)(uintptr_t)(a ^ 16u);
            return frag_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    constexpr int NPTW = (NPT + PG - 1) / PG;  // fc1 pixel tiles per wave
    using wdw4_t = wdw_t __attribute__((ext_vector_type(4)));

    // constant-one row of Us (the depthwise bias is "tap 9" times 1)
    for (int i = tid; i < 2 * CH; i += NTHR) Us[(size_t)NPH * 2 * CH + i] = to_T<T>(1.0f);
    load_a1(0);
    stamp(0);
    __syncthreads();  // Ms complete
    stamp(1);

    for (int chunk = 0; chunk < d.chunks; ++chunk) {
        // ================================ phase A: fc1 -> Us ====================================
        wdw_t wdw[4 * NPAIR];  // depthwise weights of this chunk: in flight during fc1
        {
            const wdw4_t* wp = reinterpret_cast<const wdw4_t*>(dwl + ((size_t)chunk * 64 + lane) * (4 * NPAIR));
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) {
                const wdw4_t v = wp[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) wdw[4 * i + k] = (DBG & 32) ? wdw_t(0) : v[k];
            }
        }
        {
            frag_t bcur[KS], bnxt[KS];
            load_b(0, bcur);
#pragma unroll
            for (int i = 0; i < NPTW; ++i) {
                if (i + 1 < NPTW) load_b(i + 1, bnxt);  // next tile's operands are in flight during these MFMAs
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < ((DBG & 2) ? 0 : KS); ++ks) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) acc[ii] = M::mma(a1[ii][ks], bcur[ks], acc[ii]);
                }
                // every tile of every wave is whole except possibly the last ones
                const bool whole = ((i + 1) * PG * 16) <= NPH;
                if (whole || hpa + i * PG * 16 < NPH) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) Vec4<T>::store(ust[ii] + (size_t)i * (PG * 16 * 2 * CH), acc[ii]);
                }
                if (i + 1 < NPTW) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnxt[ks];
                }
            }
        }
        stamp(2);
        __syncthreads();  // Us complete
        stamp(3);

        // ============ phase B: depthwise 3x3 as MFMA with diagonal weights (this wave's two rows) ============
        f32x4 dacc[4][RPW];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int pt = 0; pt < RPW; ++pt) dacc[gi][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        frag_t a2[NT];  // fc2 weights: issued half-way through the depthwise steps, consumed after the gate math
        auto load_a2 = [&]() {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                a2[nt] = (DBG & 32) ? M::zero() : M::load(w2f + (((size_t)chunk * NT + nt) * 64 + lane) * 8);
        };
        if constexpr (DBG & 4) load_a2();
        if constexpr (!(DBG & 4)) {
            // 20 steps (tap pair, channel group), two MFMAs each (the wave's two rows); operands two steps ahead.
            // (Pinning the reads further ahead with sched_barrier measured no faster — the other wave on the SIMD
            // already covers the LDS latency — and cost 84 bytes/lane of scratch.)
            constexpr int NSTEP = NPAIR * 4, AHEAD = 2, DEPTH = AHEAD + 1;
            frag_t ub[DEPTH][RPW];
            auto issue = [&](int st) {
#pragma unroll
                for (int pt = 0; pt < RPW; ++pt) ub[st % DEPTH][pt] = ldu(st >> 2, pt, st & 3);
            };
#pragma unroll
            for (int st = 0; st < AHEAD; ++st) issue(st);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + AHEAD < NSTEP) issue(st + AHEAD);
                if (st == NSTEP / 2) load_a2();
                const int pr = st >> 2, gi = st & 3;
                // A = [diag(w_tap0) | diag(w_tap1)]: this lane's row (channel c16 of the group) has ONE non-zero
                // element, at position jstar of its 8-wide k group (the host zeroes the weight in lanes whose k
                // group does not hold channel c16)
                frag_t af;
                if constexpr (BF) {
                    u32x4 aw;
#pragma unroll
                    for (int i = 0; i < 4; ++i) aw[i] = wdw[gi * NPAIR + pr] & dmask[i];
                    af = __builtin_bit_cast(frag_t, aw);
                } else {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) af[jj] = (jj == jstar) ? wdw[gi * NPAIR + pr] : 0.f;
                }
#pragma unroll
                for (int pt = 0; pt < RPW; ++pt) dacc[gi][pt] = M::mma(af, ub[st % DEPTH][pt], dacc[gi][pt]);
            }
        }
        stamp(4);
        // ================================ phase C: gate + fc2 ===================================
        if (chunk + 1 < d.chunks) load_a1(chunk + 1);  // next chunk's fc1 weights: in flight during gate + fc2
#pragma unroll
        for (int pt = 0; pt < RPW; ++pt) {
            // gate: G = a * SiLU(g); element (g, j<4) <- a-group 0 channel 4g+j, (g, j>=4) <- a-group 1 channel 4g+j-4
            frag_t gf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v0 = dacc[0][pt][r], v1 = dacc[1][pt][r];
                if constexpr (!(DBG & 8)) {
                    v0 *= BF ? silu_fast(dacc[2][pt][r]) : silu_exact(dacc[2][pt][r]);
                    v1 *= BF ? silu_fast(dacc[3][pt][r]) : silu_exact(dacc[3][pt][r]);
                }
                gf[r] = to_T<T>(v0);
                gf[4 + r] = to_T<T>(v1);
            }
            if constexpr (!(DBG & 16)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc2[nt][pt] = M::mma(a2[nt], gf, acc2[nt][pt]);
            }
        }
        stamp(5);
        __syncthreads();  // every wave is done reading Us before the next chunk's fc1 overwrites it
        stamp(6);
    }

    // ----------------------------------- epilogue ------------------------------------------------
    float* tout = d.t_out + (size_t)b * H * W * C;
    const bool do_ln = d.ln1_g != nullptr;
    f32x4 gapv = {0.f, 0.f, 0.f, 0.f};
    f32x4 g1v[NT], b1v[NT];  // next LayerNorm's gamma/beta for this lane's channels: one batch of loads, not 2 per n-tile
    if (do_ln) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = min(nt * 16 + 4 * g, C - 4);
            g1v[nt] = *reinterpret_cast<const f32x4*>(d.ln1_g + n);
            b1v[nt] = *reinterpret_cast<const f32x4*>(d.ln1_b + n);
        }
    }
#pragma unroll
    for (int pt = 0; pt < RPW; ++pt) {
        const int y = y0 + RPW * wave + pt, x = x0 + c16;
        const bool valid = y < H && x < W;
        const size_t pix = (size_t)y * W + x;
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + 4 * g;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (valid && n < C) {
                v = acc2[nt][pt];
                *reinterpret_cast<f32x4*>(tout + pix * C + n) = v;
            }
            acc2[nt][pt] = v;
            s += (v[0] + v[1]) + (v[2] + v[3]);
        }
        if (do_ln) {  // LayerNorm (eps 1e-5) of the finished pixel for the next block; 4 lane groups share a pixel
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s / (float)C;
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt * 16 + 4 * g < C) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float dl = acc2[nt][pt][r] - mean; q += dl * dl; }
                }
            }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
            T* nout = reinterpret_cast<T*>(d.n_out) + ((size_t)b * H * W + pix) * d.ldn;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 16 + 4 * g;
                if (valid && n < C) {
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (acc2[nt][pt][r] - mean) * rstd * g1v[nt][r] + b1v[nt][r];
                    Vec4<T>::store(nout + n, o);
                    if (nt == 0 && n < d.gap_c) gapv += as_stored<T>(o);
                }
            }
        }
    }
    if constexpr (DBG & 64) {
        stamp(7);
        if (lane == 0) {
            const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int i = 0; i < 8; ++i) d.gap_out[(wg * WAVES + wave) * 8 + i] = (float)tph[i];
        }
        return;
    }
    if (do_ln && d.gap_out != nullptr) {
        float* red = reinterpret_cast<float*>(Us);  // Us is free after the last barrier of the chunk loop
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
            for (int w = 0; w < WAVES; ++w) s += red[w * 16 + tid];
            const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            d.gap_out[((size_t)b * gridDim.x * gridDim.y + tile) * 16 + tid] = tid < d.gap_c ? s : 0.f;
        }
    }
}

template <typename T, int WAVES, int NT, int KS, bool MSWZ, int MINW, int RPW = 2>
int launch_ffn(const HatFfnDesc& d, hipStream_t s) {
    if (ffn_kp(d.C) != KS * 32 || (d.C + 15) / 16 != NT) return HAT_EUNSUPPORTED;
    const size_t lds = ffn_lds_bytes<T, WAVES, MSWZ, RPW>(d.C);
    if (lds > HAT_LDS_MAX) return HAT_ELDS;
    auto kern = ffn_kernel<T, WAVES, NT, KS, MSWZ, MINW, 0, RPW>;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((d.W + 15) / 16, (d.H + RPW * WAVES - 1) / (RPW * WAVES), d.B);
    HAT_LAUNCH(kern, grid, dim3(WAVES * 64), lds, s, d);
    return hat_check_launch();
}

// tile rows used for (C, dtype); 0 if the shape is not instantiated
int ffn_rows(const HatFfnDesc& d) {
    const bool small = d.C <= 32 && d.C % 32 != 16;  // one zero-padded 32-deep k-step
    if (d.dtype == HAT_BF16) return (d.C == 144 || d.C == 180 || small) ? 8 : 0;
    if (d.dtype == HAT_F32) return (d.C == 144 || d.C == 180 || small) ? 4 : 0;
    return 0;
}

}  // namespace

extern "C" int hat_ffn_tiles(const HatFfnDesc* d, int32_t* tiles_out) {
    if (!d || !tiles_out) return HAT_EINVAL;
    const int rows = ffn_rows(*d);
    if (!rows) return HAT_EUNSUPPORTED;
    *tiles_out = ((d->W + 15) / 16) * ((d->H + rows - 1) / rows);
    return 0;
}

extern "C" int hat_ffn(const HatFfnDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatFfnDesc& d = *dp;
    if (!d.t_in || !d.t_out || d.t_in == d.t_out || ((!d.ln_g || !d.ln_b) && !d.m_in) || !d.w1f || !d.b1 || !d.dww || !d.dwb || !d.w2f || !d.b2)
        return HAT_EINVAL;
    if (d.B < 1 || d.H < 1 || d.W < 1 || d.C < 8 || d.C % 4 || d.chunks < 1) return HAT_EINVAL;
    if (d.m_in && (d.ldm_in < ffn_kp(d.C) || d.ldm_in % (d.dtype == HAT_BF16 ? 8 : 4))) return HAT_EINVAL;
    if (d.ln1_g && (!d.ln1_b || !d.n_out || d.ldn < d.C || d.ldn % 4 || d.gap_c < 0 || d.gap_c > 16 || d.gap_c % 4)) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool small = d.C <= 32 && d.C % 32 != 16;
    if (d.dtype == HAT_BF16) {
        if (d.C == 144) return launch_ffn<bf16_t, 4, 9, 5, true, 2>(d, s);
        if (d.C == 180) return launch_ffn<bf16_t, 8, 12, 6, false, 2, 1>(d, s);  // 98 KB of LDS: one workgroup of eight waves per CU
        if (small) return launch_ffn<bf16_t, 4, 2, 1, true, 2>(d, s);
    } else if (d.dtype == HAT_F32) {
        if (d.C == 144) return launch_ffn<float, 2, 9, 5, false, 1>(d, s);
        if (d.C == 180) return launch_ffn<float, 2, 12, 6, false, 1>(d, s);
        if (small) return launch_ffn<float, 2, 2, 1, false, 1>(d, s);
    } else {
        return HAT_EINVAL;
    }
    return HAT_EUNSUPPORTED;  // embed_dim other than 144 / 180 / <=32: the host falls back to the unfused kernels
}
