// hat_attn.hip — overlapping cross-attention (OCAB) core on gfx950.   Contract: include/hat_mi355x.h
// (hat_ocab_attention); reference: hat/archs/hat_arch.py:353-388.
//
// One workgroup (4 waves) per (window, head).  K (nk x d, zero padded to 32 channels) and V
// (transposed: 32 x nk) of the (wse x wse) key window live in LDS, out-of-image keys are exact
// zeros (the reference unfolds AFTER the biased projection with zero padding and does not mask).
// Each wave owns 16-query tiles: S^T = K.Q^T via MFMA 16x16 (A = keys, B = queries) so one lane
// holds, for ONE query, 4 keys of every key tile; softmax statistics are per-lane loops plus two
// cross-lane shuffles; P^T feeds the second MFMA (O^T = V^T.P^T) straight from the accumulator
// registers — the contraction index (key) ordering inside a 32-wide k-step is permuted
// identically for both operands, so no LDS round trip or lane movement is needed for P.
#include <cstdlib>
#include <type_traits>

#include "hat_common.h"

namespace {

template <typename T> struct alignas(4 * sizeof(T)) Q4 { T v[4]; };  // 4 consecutive elements, one LDS read

template <typename T> __device__ __forceinline__ float exp_t(float x);
template <> __device__ __forceinline__ float exp_t<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ float exp_t<bf16_t>(float x) { return __expf(x); }

// SELF = true turns the same kernel into (shifted-)window self-attention, (S)W-MSA of swinir_arch.py:140-172, 291-317:
// the key window is the query window (wse == ws), window coordinates live on the cyclically shifted frame (pixel
// (Y, X) of the shifted frame is pixel ((Y + shift) % H, (X + shift) % W) of the map, so the two torch.roll calls become
// addressing), and pairs whose positions fall in different bands of the shift mask (:262-280) get -100 added after the
// bias, exactly as the reference adds its mask tensor.
// ODD: wse % 4 != 0 (HATX: window 16, overlap 0.6 -> 25 x 25 keys).  The key window is stored as NKT * 16 >= wse * wse keys: the
// keys past wse * wse are dead (zero K / V, logit -3e38 so that they vanish from the softmax), and a lane's four keys of a
// tile may sit in two key rows, so each looks up its own bias entry.
template <typename T, int NKT, int KCH, bool SELF, bool ODD = false>
__global__ __launch_bounds__(256) void ocab_attn_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                        const float* __restrict__ bias_rot, T* __restrict__ out, int H, int W,
                                                        int C, int heads, int ws, int wse, int ldq, int ldkv, int ldo,
                                                        int shift, const float* __restrict__ kb, int pad) {
    using M = MT<T>;
    constexpr int NK = NKT * 16;
    static_assert(NKT % KCH == 0, "key tiles must split evenly into chunks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = C / heads;
    const int dk8 = (d + 7) & ~7;                      // K channels kept in LDS (zero padded d..dk8)
    const int dv = (d + 1) & ~1;                        // V^T rows kept in LDS
    const int ldk = lds_row_elems(dk8, sizeof(T));
    const int ldv = lds_row_elems(NK, sizeof(T));
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vt = Ks + (size_t)NK * ldk;
    float* tab = reinterpret_cast<float*>(smem + (((size_t)NK * ldk + (size_t)dv * ldv) * sizeof(T) + 15) / 16 * 16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int wx = blockIdx.x, wy = blockIdx.y;
    const int b = blockIdx.z / heads, h = blockIdx.z - b * heads;
    const int Mr = ws + wse - 1;
    const size_t img = (size_t)b * H * W;
    // HATX key bias / prune mask of this window (hat_ocab_keybias): NK floats, -inf = pruned key
    const float* kbw = kb ? kb + (((size_t)b * gridDim.y + wy) * gridDim.x + wx) * NK : nullptr;

    for (int i = tid; i < Mr * Mr; i += 256) tab[i] = bias_rot[(size_t)h * Mr * Mr + i];

    // ---- stage K [key][32] and V^T [32][key]; 2 channels per work item -----------------------
    const int cpk = dk8 / 2;  // channel pairs per key
    for (int i = tid; i < NK * cpk; i += 256) {
        const int key = i / cpk, c = (i - key * cpk) * 2;
        const int kh = key / wse, kw = key - kh * wse;
        int y = wy * ws - pad + kh, x = wx * ws - pad + kw;
        if (SELF) {
            y += shift; if (y >= H) y -= H;
            x += shift; if (x >= W) x -= W;
        }
        T k0 = to_T<T>(0.f), k1 = k0, v0 = k0, v1 = k0;
        if (c < d && (!ODD || key < wse * wse) && y >= 0 && y < H && x >= 0 && x < W) {
            const T* p = kv + (img + (size_t)y * W + x) * ldkv + h * d + c;
            k0 = p[0]; k1 = p[1];
            v0 = p[C]; v1 = p[C + 1];
        }
        Ks[key * ldk + c] = k0; Ks[key * ldk + c + 1] = k1;
        if (c < dv) { Vt[c * ldv + key] = v0; Vt[(c + 1) * ldv + key] = v1; }
    }
    __syncthreads();

    const int nqt = (ws * ws) / 16;
    for (int qt = wave; qt < nqt; qt += 4) {
        // ---- Q fragment (B operand): this lane's query, channels 8g..8g+7 --------------------
        const int qi = qt * 16 + c16;
        const int qy = qi / ws, qx = qi - qy * ws;
        int qyo = wy * ws + qy, qxo = wx * ws + qx;   // position on the (shifted) frame -> pixel of the map
        auto band = [&](int v, int n) { return v < n - ws ? 0 : (v < n - shift ? 1 : 2); };
        const int rid_q = SELF ? 3 * band(qyo, H) + band(qxo, W) : 0;
        if (SELF) {
            qyo += shift; if (qyo >= H) qyo -= H;
            qxo += shift; if (qxo >= W) qxo -= W;
        }
        const size_t qpix = img + (size_t)qyo * W + qxo;
        typename M::frag_t qf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * g + j;
            qf[j] = c < d ? q[qpix * ldq + h * d + c] : to_T<T>(0.f);
        }
        // ---- online softmax over chunks of KCH key tiles (O is only d <= 32 wide, so the rescale
        //      is 8 registers; this keeps the live S tile set at KCH*4 registers) ------------------
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        float mrun = -3.0e38f, l = 0.f;
        int kh = (4 * g) / wse, kw = 4 * g - kh * wse;  // the lane's 4 keys of a tile share one key row (wse % 4 == 0)
#pragma unroll 1
        for (int ch = 0; ch < NKT / KCH; ++ch) {
            const int kt0 = ch * KCH;
            // S^T tiles: s[t][r] = S[key = 16 (kt0+t) + 4g + r][query c16]
            f32x4 s[KCH];
#pragma unroll
            for (int t = 0; t < KCH; ++t) {
                // channel groups beyond dk8 are clamped: they meet a zero Q fragment, so contribute 0
                const typename M::frag_t kf = M::load(Ks + ((kt0 + t) * 16 + c16) * ldk + (8 * g < dk8 ? 8 * g : dk8 - 8));
                s[t] = M::mma(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f});
            }
            if (kbw != nullptr) {   // hatx_arch.py:421-449: + focus bias per key; a pruned key's logit is REPLACED by -1e4
#pragma unroll
                for (int t = 0; t < KCH; ++t) {
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(kbw + (kt0 + t) * 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t][r] = k4[r] == -INFINITY ? -1.0e4f : s[t][r] + k4[r];
                }
            }
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < KCH; ++t) {
                const float* tb = tab + (kh - qy + ws - 1) * Mr + (kw - qx + ws - 1);
                // the lane's 4 keys share a mask band: shift % 4 == 0 and kw % 4 == 0
                const float madd = (SELF && shift > 0 && 3 * band(wy * ws + kh, H) + band(wx * ws + kw, W) != rid_q) ? -100.0f : 0.0f;
                if constexpr (ODD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = (kt0 + t) * 16 + 4 * g + r;
                        const int khr = key / wse, kwr = key - khr * wse;
                        const int ti = min((khr - qy + ws - 1) * Mr + (kwr - qx + ws - 1), Mr * Mr - 1);
                        s[t][r] = key < wse * wse ? s[t][r] + tab[ti] : -3.0e38f;
                        mx = fmaxf(mx, s[t][r]);
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[t][r] += tb[r];
                    if (SELF) s[t][r] += madd;
                    mx = fmaxf(mx, s[t][r]);
                }
                kw += 16;
                while (kw >= wse) { kw -= wse; ++kh; }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun, mx);
            const float alpha = exp_t<T>(mrun - mnew);
            mrun = mnew;
            float lsum = 0.f;
#pragma unroll
            for (int t = 0; t < KCH; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = exp_t<T>(s[t][r] - mnew);
                    s[t][r] = p;
                    lsum += p;
                }
            }
            l = l * alpha + lsum;  // per-lane partial; reduced across the 4 lane groups after the loop
            o[0] *= alpha;
            o[1] *= alpha;
            // O^T += V^T . P^T : k-slot (g, j) = key 32 kk + 4g + j (j<4) | 32 kk + 16 + 4g + (j-4)
#pragma unroll
            for (int kk = 0; kk < (KCH + 1) / 2; ++kk) {
                const bool has_b = (2 * kk + 1 < KCH);
                const int kb = has_b ? 2 * kk + 1 : 2 * kk;
                typename M::frag_t pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pf[j] = to_T<T>(s[2 * kk][j]);
                    pf[j + 4] = has_b ? to_T<T>(s[kb][j]) : to_T<T>(0.f);
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    // rows >= d are clamped: their outputs are never stored
                    const int vrow = (ct * 16 + c16 < dv) ? ct * 16 + c16 : dv - 1;
                    const T* vr = Vt + (size_t)vrow * ldv + (kt0 + 2 * kk) * 16 + 4 * g;
                    const Q4<T> va = *reinterpret_cast<const Q4<T>*>(vr);
                    const Q4<T> vb = *reinterpret_cast<const Q4<T>*>(vr + (has_b ? 16 : 0));
                    typename M::frag_t vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vf[j] = va.v[j];
                        vf[j + 4] = has_b ? vb.v[j] : to_T<T>(0.f);
                    }
                    o[ct] = M::mma(vf, pf, o[ct]);
                }
            }
        }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        // ---- store: lane holds channels ct*16 + 4g + r of query c16 -----------------------------
        const float inv = 1.0f / l;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = ct * 16 + 4 * g + r;
                if (c < d) out[qpix * ldo + h * d + c] = to_T<T>(o[ct][r] * inv);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Key-STREAMING form of the generic kernel, for key windows whose K / V image does not fit in LDS — the exact-fp32 path at
// the fork's live HATX shape (hatx_arch.py:289-465 with overlap_ratio 0.6: 25 x 25 keys in 40 tiles x 30-channel heads in
// fp32 = 169 KB).  The online softmax already consumed the keys in chunks of KCH tiles; here only ONE chunk of K and V^T is
// resident (KCH = 10: 43 KB), staged by the whole workgroup between two barriers, and every wave carries the softmax state
// (O, running max, partial denominator) of ALL its NQW query tiles across the chunks instead of finishing one query tile
// after the other.  Same arithmetic, same order of the key chunks, same ODD handling (dead keys past wse^2, per-key bias
// lookup) and the same optional HATX key bias as ocab_attn_kernel.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int NKT, int KCH, int NQW>
__global__ __launch_bounds__(256) void ocab_attn_stream_kernel(const T* __restrict__ q, const T* __restrict__ kv,
                                                               const float* __restrict__ bias_rot, T* __restrict__ out, int H, int W,
                                                               int C, int heads, int ws, int wse, int ldq, int ldkv, int ldo,
                                                               const float* __restrict__ kb, int pad) {
    using M = MT<T>;
    constexpr int NK = NKT * 16, CK = KCH * 16;
    static_assert(NKT % KCH == 0, "key tiles must split evenly into chunks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = C / heads;
    const int dk8 = (d + 7) & ~7, dv = (d + 1) & ~1;
    const int ldk = lds_row_elems(dk8, sizeof(T)), ldv = lds_row_elems(CK, sizeof(T));
    T* Ks = reinterpret_cast<T*>(smem);                       // [CK][ldk]   one chunk of keys
    T* Vt = Ks + (size_t)CK * ldk;                            // [dv][ldv]   the same chunk of V, transposed
    float* tab = reinterpret_cast<float*>(smem + (((size_t)CK * ldk + (size_t)dv * ldv) * sizeof(T) + 15) / 16 * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int wx = blockIdx.x, wy = blockIdx.y;
    const int b = blockIdx.z / heads, h = blockIdx.z - b * heads;
    const int Mr = ws + wse - 1;
    const size_t img = (size_t)b * H * W;
    const float* kbw = kb ? kb + (((size_t)b * gridDim.y + wy) * gridDim.x + wx) * NK : nullptr;
    for (int i = tid; i < Mr * Mr; i += 256) tab[i] = bias_rot[(size_t)h * Mr * Mr + i];

    typename M::frag_t qf[NQW];
    f32x4 o[NQW][2];
    float mrun[NQW], l[NQW];
    int qy[NQW], qx[NQW];
    size_t qpix[NQW];
#pragma unroll
    for (int i = 0; i < NQW; ++i) {
        const int qi = (wave + 4 * i) * 16 + c16;
        qy[i] = qi / ws; qx[i] = qi - qy[i] * ws;
        qpix[i] = img + (size_t)(wy * ws + qy[i]) * W + wx * ws + qx[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * g + j;
            qf[i][j] = c < d ? q[qpix[i] * ldq + h * d + c] : to_T<T>(0.f);
        }
        o[i][0] = o[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        mrun[i] = -3.0e38f; l[i] = 0.f;
    }
    const int cpk = dk8 / 2;
#pragma unroll 1
    for (int ch = 0; ch < NKT / KCH; ++ch) {
        const int kt0 = ch * KCH;
        __syncthreads();   // every wave is done with the previous chunk (first pass: the bias table is complete)
        for (int i = tid; i < CK * cpk; i += 256) {
            const int kl = i / cpk, c = (i - kl * cpk) * 2, key = kt0 * 16 + kl;
            const int kh = key / wse, kw = key - kh * wse;
            const int y = wy * ws - pad + kh, x = wx * ws - pad + kw;
            T k0 = to_T<T>(0.f), k1 = k0, v0 = k0, v1 = k0;
            if (c < d && key < wse * wse && y >= 0 && y < H && x >= 0 && x < W) {
                const T* p = kv + (img + (size_t)y * W + x) * ldkv + h * d + c;
                k0 = p[0]; k1 = p[1];
                v0 = p[C]; v1 = p[C + 1];
            }
            Ks[kl * ldk + c] = k0; Ks[kl * ldk + c + 1] = k1;
            if (c < dv) { Vt[c * ldv + kl] = v0; Vt[(c + 1) * ldv + kl] = v1; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NQW; ++i) {
            f32x4 s[KCH];
#pragma unroll
            for (int t = 0; t < KCH; ++t) {
                const typename M::frag_t kf = M::load(Ks + (t * 16 + c16) * ldk + (8 * g < dk8 ? 8 * g : dk8 - 8));
                s[t] = M::mma(kf, qf[i], f32x4{0.f, 0.f, 0.f, 0.f});
            }
            if (kbw != nullptr) {   // hatx_arch.py:421-449: + focus bias per key; a pruned key's logit is REPLACED by -1e4
#pragma unroll
                for (int t = 0; t < KCH; ++t) {
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(kbw + (kt0 + t) * 16 + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t][r] = k4[r] == -INFINITY ? -1.0e4f : s[t][r] + k4[r];
                }
            }
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < KCH; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = (kt0 + t) * 16 + 4 * g + r;
                    const int khr = key / wse, kwr = key - khr * wse;
                    const int ti = min((khr - qy[i] + ws - 1) * Mr + (kwr - qx[i] + ws - 1), Mr * Mr - 1);
                    s[t][r] = key < wse * wse ? s[t][r] + tab[ti] : -3.0e38f;
                    mx = fmaxf(mx, s[t][r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mnew = fmaxf(mrun[i], mx);
            const float alpha = exp_t<T>(mrun[i] - mnew);
            mrun[i] = mnew;
            float lsum = 0.f;
#pragma unroll
            for (int t = 0; t < KCH; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pe = exp_t<T>(s[t][r] - mnew);
                    s[t][r] = pe;
                    lsum += pe;
                }
            }
            l[i] = l[i] * alpha + lsum;
            o[i][0] *= alpha;
            o[i][1] *= alpha;
#pragma unroll
            for (int kk = 0; kk < (KCH + 1) / 2; ++kk) {
                const bool has_b = (2 * kk + 1 < KCH);
                const int kb2 = has_b ? 2 * kk + 1 : 2 * kk;
                typename M::frag_t pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pf[j] = to_T<T>(s[2 * kk][j]);
                    pf[j + 4] = has_b ? to_T<T>(s[kb2][j]) : to_T<T>(0.f);
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int vrow = (ct * 16 + c16 < dv) ? ct * 16 + c16 : dv - 1;
                    const T* vr = Vt + (size_t)vrow * ldv + (2 * kk) * 16 + 4 * g;
                    const Q4<T> va = *reinterpret_cast<const Q4<T>*>(vr);
                    const Q4<T> vb = *reinterpret_cast<const Q4<T>*>(vr + (has_b ? 16 : 0));
                    typename M::frag_t vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vf[j] = va.v[j];
                        vf[j + 4] = has_b ? vb.v[j] : to_T<T>(0.f);
                    }
                    o[i][ct] = M::mma(vf, pf, o[i][ct]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NQW; ++i) {
        float ls = l[i];
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        const float inv = 1.0f / ls;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = ct * 16 + 4 * g + r;
                if (c < d) out[qpix[i] * ldo + h * d + c] = to_T<T>(o[i][ct][r] * inv);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fast path: bf16, 16x16 query windows, 24x24 key windows, head_dim 24 (HAT-S and every C = 144 model).
// Differences from the generic kernel above, all aimed at the VALU (softmax) and LDS-store bottlenecks:
//   * K and V are staged with 16-byte pieces, V row-major [key][32] (no 2-byte transposing stores); the A
//     operand V^T of the second MFMA is read with ds_read_b64_tr_b16 (hardware transpose);
//   * V carries a constant-1 column at channel 24, so the softmax denominator is output row 24 of the same
//     MFMA (it is rescaled with O for free in the online-softmax update): no VALU row sums;
//   * the relative-position bias enters as the C operand of the S MFMA: no VALU bias adds;
//   * p = exp2(fma(s, log2e, -m*log2e)): one FMA + one exp per score.
// ---------------------------------------------------------------------------------------------------------
// v_max3_f32 without the per-input canonicalising v_max the compiler adds for fmaxf (scores are never NaN-signalling)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max(|a|, |b|, |c|) in one instruction (VOP3 abs modifiers)
__device__ __forceinline__ float max3_abs(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// D = 24 (C = 144) or 30 (C = 180, six heads): D = 30 pads K / Q to 32 channels (K rows of 4 XOR-swizzled slots), stages
// and stores with 4-byte accesses (a head's 60-byte slice is only 4-byte aligned) and keeps the ones column at V[30].
// WSE = 24, SELF = false: OCAB (zero-padded 24 x 24 key window around the query window).  WSE = 16, SELF = true: (shifted-)
// window self-attention for hat_window_attention — the key window is the query window, all coordinates live on the frame
// shifted cyclically by `shift` (so every key is a real pixel), and in the last window row / column the -100 of the
// shift mask is added to the scores of pairs in different mask bands.
// NTH: threads per workgroup.  512 (the OCAB of the C = 144 models) = eight waves with two query rows each and key chunks of 6
// tiles (half the score registers: 80 VGPRs): the K / V image of a 24 x 24 key window (64.5 KB) limits a CU to two workgroups,
// and sixteen waves instead of eight hide more of the softmax's latencies (0.995 -> 0.955 ms at 720p; sixteen-wave workgroups
// measured the same as eight).
template <int D, int WSE, bool SELF, int NTH = 256, bool QLOG2 = false>
__global__ __launch_bounds__(NTH, WSE == 24 ? 2 : 4) void ocab_attn_fast_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv,
                                                                const float* __restrict__ bias_rot, bf16_t* __restrict__ out,
                                                                int B, int H, int W, int C, int heads, int ldq, int ldkv,
                                                                int ldo, int shift) {
    using M = MT<bf16_t>;
    using frag_t = M::frag_t;
    constexpr int WS = 16, NK = WSE * WSE, MR = WS + WSE - 1, NKT = NK / 16, KCH = WSE == 24 ? (NTH >= 512 ? 6 : 12) : 8, PAD = (WSE - WS) / 2;
    constexpr int CH_ROWS = KCH * 16 / WSE;  // key rows per chunk
    static_assert(KCH % 2 == 0 && KCH * 16 % WSE == 0, "a chunk is whole key rows and whole pairs of key tiles");
    static_assert(!SELF || WSE == WS, "self-attention windows coincide with the query windows");
    // OFFS (the OCAB of the embed_dim-144 models): the softmax offset rides in the spare k-slots of the QK^T MFMA.  K rows
    // are 32 channels wide with [1.0, 0 ...] in channels 24..31 of EVERY key, the query fragment carries -m (the running
    // offset, a bf16 value, in log2 units) in channel 24, Q and the bias table are pre-multiplied by log2(e): the MFMA
    // delivers s * log2e + bias * log2e - m, and p = exp2 of that with NO per-score FMA, no per-chunk rescale of O and no
    // cross-lane max — while every |score - m| of a chunk stays below 64 (checked with 12 v_max3 |.| and one ballot); a chunk
    // that leaves that range takes the classic path (re-centre, rescale O) and moves m.
    constexpr bool OFFS = QLOG2 && D == 24 && WSE == 24 && !SELF;
    constexpr int KR = (D == 24 && !OFFS) ? 24 : 32;   // K row length in LDS (elements)
    constexpr int ONE = D == 24 ? 24 : 30;  // V column that holds 1.0 (the softmax denominator row of O^T)
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);                 // [NK][KR]
    bf16_t* Vs = Ks + NK * KR;                                    // [NK][32]: D channels, 1.0, zeros
    float* tab = reinterpret_cast<float*>(Vs + NK * 32);          // [MR*MR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  A head reads a 2D-byte slice of every q / k / v
    // row, i.e. part of a cache line that the window's other heads need too: block L -> XCD L % 8, and inside an XCD
    // consecutive blocks are the heads of ONE window, so the lines are fetched from HBM once and hit in that XCD's L2
    // (with the head as the slowest grid index every head re-fetched them: 0.82 -> 0.43 ms for W-MSA, 0.97 -> 0.88 ms for
    // OCAB at 720p; giving each XCD a contiguous range of windows instead of every 8th one measured no better).
    const int nwx = W / WS, nwy = H / WS;
    const int xj = (int)blockIdx.x >> 3;
    const int widx = (xj / heads) * 8 + ((int)blockIdx.x & 7), h = xj % heads;
    if (widx >= B * nwx * nwy) return;   // whole workgroup, before any barrier
    const int b = widx / (nwx * nwy), wrem = widx - b * nwx * nwy;
    const int wy = wrem / nwx, wx = wrem - wy * nwx;
    const size_t img = (size_t)b * H * W;
    // window position (row, col) -> pixel of the map: identity for OCAB queries, cyclic shift for SELF
    auto qpixel = [&](int row, int col) {
        int y = wy * WS + row, x = wx * WS + col;
        if (SELF) {
            y += shift; if (y >= H) y -= H;
            x += shift; if (x >= W) x -= W;
        }
        return img + (size_t)y * W + x;
    };
    // key (kh, kw) of the key window -> clamped pixel and whether it is a real one (SELF: always, after the wrap)
    auto kpixel = [&](int kh, int kw, bool& inb) {
        if (SELF) { inb = true; return qpixel(kh, kw); }
        const int y = wy * WS - PAD + kh, x = wx * WS - PAD + kw;
        inb = y >= 0 && y < H && x >= 0 && x < W;
        return img + (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
    };

    constexpr int NWV = NTH / 64, QPW = 16 / NWV;   // waves, query rows (tiles) per wave
    frag_t qfr[QPW];   // this wave's query tiles: query tile qt = window row qt, lane c16 = window column
    constexpr int NTAB = (MR * MR + NTH - 1) / NTH;
    float tv[NTAB];
#pragma unroll
    for (int it = 0; it < NTAB; ++it) {
        const int i = tid + it * NTH;
        tv[it] = bias_rot[(size_t)h * MR * MR + (i < MR * MR ? i : MR * MR - 1)] * (OFFS ? LOG2E : 1.0f);
    }
    if constexpr (D == 24) {
        // Every global load of the staging phase is issued before the first LDS store (branch-free: out-of-image keys
        // load a clamped pixel and are zeroed by a select), so the phase costs one memory latency, not nine.
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const size_t qpix = qpixel(wave + NWV * i, c16);
            qfr[i] = M::load(q + qpix * ldq + h * D + 8 * (g < 3 ? g : 2));
            if (g == 3) qfr[i] = M::zero();
            // (OFFS: q arrives multiplied by head_dim^-1/2 * log2(e) — the factor is folded into the projection's weights
            // before they are rounded, hat_ocab_attention_log2 — so the scores are in log2 units without another rounding)
        }
        constexpr int NIT = (NK * 4 + NTH - 1) / NTH;   // (the last pass is partial with 512 threads: clamped loads, masked stores)
        u32x4 kq[NIT], vq[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = min(tid + it * NTH, NK * 4 - 1);
            const int key = i >> 2, c = i & 3;
            const int kh = key / WSE, kw = key - kh * WSE;
            bool inb;
            const bf16_t* p = kv + kpixel(kh, kw, inb) * ldkv + h * D + 8 * (c < 3 ? c : 2);
            kq[it] = *reinterpret_cast<const u32x4*>(p);
            vq[it] = *reinterpret_cast<const u32x4*>(p + C);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * NTH;
            const int key = i >> 2, c = i & 3;
            const int kh = key / WSE, kw = key - kh * WSE;
            bool inb;
            (void)kpixel(kh, kw, inb);
            const int vsw = (key >> 1) & 2;  // rows 4..7 of every 8 swap their 32-byte halves: transposed reads stay conflict-free
            const u32x4 zero = {0u, 0u, 0u, 0u};
            // channel 24 = 1.0 for EVERY key (out-of-image keys still count in the softmax denominator)
            const u32x4 vval = c == 3 ? u32x4{0x00003F80u, 0u, 0u, 0u} : (inb ? vq[it] : zero);
            if (NIT * NTH == NK * 4 || i < NK * 4) {
                *reinterpret_cast<u32x4*>(Vs + key * 32 + 8 * (c ^ vsw)) = vval;
                if constexpr (OFFS) {   // 4 slots per row, XOR-swizzled by the row (conflict-free ds_read_b128); slot 3 = [1.0, 0 ...]
                    const u32x4 kval = c == 3 ? u32x4{0x00003F80u, 0u, 0u, 0u} : (inb ? kq[it] : zero);
                    *reinterpret_cast<u32x4*>(Ks + key * KR + 8 * (c ^ ((key >> 1) & 3))) = kval;
                } else if (c < 3) {
                    *reinterpret_cast<u32x4*>(Ks + key * KR + 8 * c) = inb ? kq[it] : zero;
                }
            }
        }
    } else {
        // 4-byte staging: dword dw (two channels) of key `key`; dword 15 is the pad of K and [1.0, 0] of V
#pragma unroll
        for (int i = 0; i < QPW; ++i) {
            const size_t qpix = qpixel(wave + NWV * i, c16);
            const unsigned* qp = reinterpret_cast<const unsigned*>(q + qpix * ldq + h * D) + 4 * g;
            u32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = qp[min(4 * g + j, D / 2 - 1) - 4 * g];
            if (g == 3) v[3] = 0u;   // channels 30, 31
            qfr[i] = __builtin_bit_cast(frag_t, v);
        }
        constexpr int BATCH = WSE == 24 ? (NTH == 512 ? 9 : 12) : 8, NPASS = NK * 16 / NTH / BATCH;
        static_assert(NPASS * BATCH * NTH == NK * 16, "staging passes must cover the key window exactly");
        unsigned* Kd = reinterpret_cast<unsigned*>(Ks);
        unsigned* Vd = reinterpret_cast<unsigned*>(Vs);
        for (int ps = 0; ps < NPASS; ++ps) {
            unsigned kq[BATCH], vq[BATCH];
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                const int i = tid + (ps * BATCH + it) * NTH;
                const int key = i >> 4, dw = i & 15;
                const int kh = key / WSE, kw = key - kh * WSE;
                bool inb;
                const unsigned* p = reinterpret_cast<const unsigned*>(kv + kpixel(kh, kw, inb) * ldkv + h * D) + min(dw, D / 2 - 1);
                kq[it] = p[0];
                vq[it] = p[C / 2];
            }
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                const int i = tid + (ps * BATCH + it) * NTH;
                const int key = i >> 4, dw = i & 15;
                const int kh = key / WSE, kw = key - kh * WSE;
                bool inb;
                (void)kpixel(kh, kw, inb);
                const int slot = dw >> 2, vsw = (key >> 1) & 2, ksw = (key >> 1) & 3;
                Kd[key * 16 + 4 * (slot ^ ksw) + (dw & 3)] = (inb && dw < D / 2) ? kq[it] : 0u;
                Vd[key * 16 + 4 * (slot ^ vsw) + (dw & 3)] = dw == 15 ? 0x00003F80u : (inb ? vq[it] : 0u);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NTAB; ++it) {
        const int i = tid + it * NTH;
        if (i < MR * MR) tab[i] = tv[it];
    }
    __syncthreads();

    const int trq = c16 >> 2, trp = c16 & 3;  // this lane's address role inside its 16-lane transpose group
    // Bias-table offsets of this lane's 4 keys (rows 4g..4g+3 of key tile t; WSE % 4 == 0 keeps them in one key
    // row) against query column c16 of window row 0.  A chunk of KCH key tiles is exactly 8 key rows and a query
    // tile is exactly one window row, so chunk ch / query tile qt only add the uniform (8 * ch - qt) * MR.
    int toff[KCH];
#pragma unroll
    for (int t = 0; t < KCH; ++t) {
        const int key = t * 16 + 4 * g;
        const int kh = key / WSE, kw = key - kh * WSE;
        toff[t] = (kh + WS - 1) * MR + (kw - c16 + WS - 1);
    }
#pragma unroll
    for (int qi4 = 0; qi4 < QPW; ++qi4) {
        const int qt = wave + NWV * qi4;
        const size_t qpix = qpixel(qt, c16);
        frag_t qf = qfr[qi4];
        // D = 24: lanes g == 3 meet a zero Q fragment and re-read group 2; D = 30: 4 slots per row, XOR-swizzled by the row
        const bf16_t* krow = (D == 24 && !OFFS) ? Ks + c16 * KR + (g < 3 ? 8 * g : 16) : Ks + c16 * KR + 8 * (g ^ ((c16 >> 1) & 3));
        f32x4 o[2];
        // OFFS: every LDS address of the (fully unrolled) chunk loop is a per-lane base plus an immediate; rolled up, 23 of the
        // 76 vector instructions per 12 MFMAs were address additions.  The bias reads (four consecutive floats at a 4-byte-
        // aligned address) get their per-lane bases once per query tile — 8 ch key rows further down the table is an
        // immediate — and are `volatile`: left alone the load/store optimizer pairs them into ds_read2_b32, whose 8-bit
        // offsets do not reach, and re-creates a base register with a v_add for every pair.
        typedef const volatile __attribute__((address_space(3))) float* lds_vf_p;
        lds_vf_p tq[KCH];
        if constexpr (OFFS) {
#pragma unroll
            for (int t = 0; t < KCH; ++t) tq[t] = (lds_vf_p)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(tab + toff[t] - qt * MR);
        }
        // One pass over the key window.  OFFS has two forms of it:
        //   CHECKED = false (always tried first): the first chunk centres the offset on its row maximum and every later chunk
        //     trusts it — no range check at all (it was 12 v_max3 |.| + a ballot per chunk, a sixth of the loop's VALU time).  A
        //     later score far ABOVE the offset only makes p large (exp2 of up to ~100 is an ordinary float and bf16 has the same
        //     exponent range); what cannot be represented shows in the denominator, and then
        //   CHECKED = true: the pass is repeated with the range check and the re-centring step in every chunk (rare: it takes a
        //     score 2^100 times the first chunk's maximum).
        auto pass = [&](auto checked_tag) {
            constexpr bool CHECKED = decltype(checked_tag)::value;
            float moff = 0.f;   // OFFS: the offset currently riding in the query fragment's channel 24 (log2 units, a bf16 value)
            if constexpr (OFFS) { if (g == 3) qf[0] = (bf16_t)0.f; }
            o[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            o[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            float mrun = -3.0e38f;
            constexpr int UNR = (OFFS && !CHECKED) ? NKT / KCH : 1;
#pragma unroll UNR
            for (int ch = 0; ch < NKT / KCH; ++ch) {
                const int kt0 = ch * KCH;
                const float* tbase = tab + (CH_ROWS * ch - qt) * MR;
                f32x4 s[KCH];
#pragma unroll
                for (int t = 0; t < KCH; ++t) {
                    f32x4 bias4;
                    if constexpr (OFFS && !CHECKED) {
                        lds_vf_p tb = tq[t] + CH_ROWS * ch * MR;
                        bias4 = f32x4{tb[0], tb[1], tb[2], tb[3]};
                    } else {
                        const float* tb = tbase + toff[t];
                        bias4 = f32x4{tb[0], tb[1], tb[2], tb[3]};
                    }
                    const frag_t kf = M::load(krow + (kt0 + t) * 16 * KR);
                    s[t] = M::mma(kf, qf, bias4);
                }
                if constexpr (SELF) {
                    // Shift mask (swinir_arch.py:262-280): bands [0, n-ws), [n-ws, n-shift), [n-shift, n) of the shifted frame,
                    // so only the last window row / column mixes bands.  Key tile t is key row kt0 + t, the query tile is
                    // query row qt (both wave-uniform); the lane's 4 key columns 4g.. share a band (shift % 4 == 0).
                    const bool lastr = wy == nwy - 1, lastc = wx == nwx - 1;
                    if (shift > 0 && (lastr || lastc)) {
                        const float xm = (lastc && ((4 * g >= WS - shift) != (c16 >= WS - shift))) ? -100.0f : 0.0f;
#pragma unroll
                        for (int t = 0; t < KCH; ++t) {
                            const bool ydiff = lastr && ((kt0 + t >= WS - shift) != (qt >= WS - shift));
                            const float madd = ydiff ? -100.0f : xm;
                            s[t][0] += madd; s[t][1] += madd; s[t][2] += madd; s[t][3] += madd;
                        }
                    }
                }
                float c2 = 0.f;      // exponent offset applied on the VALU (classic path only)
                if constexpr (OFFS) {
                    bool plain = ch > 0;   // (uniform) the offset the MFMA applied is good enough for this chunk
                    if constexpr (CHECKED) {
                        float am = 0.f;
#pragma unroll
                        for (int t = 0; t < KCH; ++t) am = max3_abs(max3_abs(am, s[t][0], s[t][1]), s[t][2], s[t][3]);
                        plain = ch > 0 && __builtin_amdgcn_ballot_w64(am > 64.0f) == 0ull;
                    }
                    // The first chunk always centres on its row maximum (any sign: nothing is accumulated yet, and a row whose every
                    // logit lies far below zero must not underflow as a whole); later chunks (CHECKED) re-centre only UPWARDS, when
                    // a score is more than 2^64 above the offset (a score far below it just contributes nothing).
                    if (!plain) {
                        float mx = -3.0e38f;
#pragma unroll
                        for (int t = 0; t < KCH; ++t) mx = max3_raw(max3_raw(mx, s[t][0], s[t][1]), s[t][2], s[t][3]);
                        mx = fmaxf(mx, __shfl_xor(mx, 16));
                        mx = fmaxf(mx, __shfl_xor(mx, 32));
                        const float mnew = (float)(bf16_t)(moff + (ch == 0 ? mx : fmaxf(mx, 0.f)));   // what channel 24 can hold
                        const float delta = mnew - moff;                           // (exact: both are short; >= 0 after the first chunk)
                        if (ch > 0) {
                            const float alpha = __builtin_amdgcn_exp2f(-delta);    // <= 1
                            o[0] *= alpha;
                            o[1] *= alpha;
                        }
#pragma unroll
                        for (int t = 0; t < KCH; ++t) s[t] -= f32x4{delta, delta, delta, delta};
                        moff = mnew;
                        if (g == 3) qf[0] = (bf16_t)(-moff);
                    }
                } else {
                    float mx = -3.0e38f;
#pragma unroll
                    for (int t = 0; t < KCH; ++t) mx = max3_raw(max3_raw(mx, s[t][0], s[t][1]), s[t][2], s[t][3]);
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float mnew = fmaxf(mrun, mx);
                    const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * LOG2E);  // raw v_exp_f32: underflow flushes to 0
                    mrun = mnew;
                    c2 = -mnew * LOG2E;
                    o[0] *= alpha;
                    o[1] *= alpha;
                }
#pragma unroll
                for (int kk = 0; kk < KCH / 2; ++kk) {
                    frag_t pf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (OFFS) {
                            pf[j] = (bf16_t)__builtin_amdgcn_exp2f(s[2 * kk][j]);
                            pf[4 + j] = (bf16_t)__builtin_amdgcn_exp2f(s[2 * kk + 1][j]);
                        } else {
                            pf[j] = (bf16_t)__builtin_amdgcn_exp2f(fmaf(s[2 * kk][j], LOG2E, c2));
                            pf[4 + j] = (bf16_t)__builtin_amdgcn_exp2f(fmaf(s[2 * kk + 1][j], LOG2E, c2));
                        }
                    }
                    const int key0 = (kt0 + 2 * kk) * 16 + 4 * g;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        typedef short s16x4 __attribute__((ext_vector_type(4)));
                        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
                        const bf16_t* va = Vs + (key0 + trq) * 32 + (ct ^ (g & 1)) * 16 + 4 * trp;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(va + 16 * 32));
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        o[ct] = M::mma(__builtin_bit_cast(frag_t, both), pf, o[ct]);
                    }
                }
            }
            // row ONE of O^T is the softmax denominator: register (ONE - 16) % 4 of lane group (ONE - 16) / 4, second tile
            return __shfl(o[1][(ONE - 16) & 3], 16 * ((ONE - 16) >> 2) + c16);
        };
        float l = pass(std::false_type{});
        if constexpr (OFFS) {
            // (the query tile's 16 denominators: all finite and far from the top of the range, or the pass is repeated)
            if (__builtin_amdgcn_ballot_w64(!(l < 1.0e30f)) != 0ull) {
                asm volatile("; repeat the pass with range checks (rare path: keep it a branch)" ::: "memory");
                l = pass(std::true_type{});
            }
        }
        const float inv = 1.0f / l;
        bf16_t* op = out + qpix * ldo + h * D + 4 * g;
        if constexpr (D == 24) {
            Vec4<bf16_t>::store(op, o[0] * inv);
            if (g < 2) Vec4<bf16_t>::store(op + 16, o[1] * inv);
        } else {  // 4-byte stores: the head slice is only 4-byte aligned; channels 28, 29 are the last two
            typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
            const f32x4 a0 = o[0] * inv, a1 = o[1] * inv;
            *reinterpret_cast<bf2*>(op) = bf2{(bf16_t)a0[0], (bf16_t)a0[1]};
            *reinterpret_cast<bf2*>(op + 2) = bf2{(bf16_t)a0[2], (bf16_t)a0[3]};
            *reinterpret_cast<bf2*>(op + 16) = bf2{(bf16_t)a1[0], (bf16_t)a1[1]};
            if (g < 3) *reinterpret_cast<bf2*>(op + 18) = bf2{(bf16_t)a1[2], (bf16_t)a1[3]};
        }
    }
}

template <typename T, int NKT, int KCH, bool SELF = false, bool ODD = false>
int launch_attn(const void* q, const void* kv, const float* bias_rot, void* out, int B, int H, int W, int C, int heads,
                int ws, int wse, int ldq, int ldkv, int ldo, hipStream_t s, int shift = 0, const float* kb = nullptr, int pad = -1) {
    if (pad < 0) pad = (wse - ws + 1) / 2;   // (HATX pads odd overlaps with the ceiling, hatx_arch.py:303-305)
    const int es = sizeof(T);
    const int Mr = ws + wse - 1;
    const int d = C / heads, dk8 = (d + 7) & ~7, dv = (d + 1) & ~1;
    const size_t kvb = ((size_t)NKT * 16 * lds_row_elems(dk8, es) + (size_t)dv * lds_row_elems(NKT * 16, es)) * es;
    const size_t lds = (kvb + 15) / 16 * 16 + (size_t)Mr * Mr * 4;
    if (lds > HAT_LDS_MAX) {
        // the key window does not fit: stream it through LDS chunk by chunk (instantiated for 16 x 16 query windows)
        if constexpr (!SELF) {
            if (ws != 16) return HAT_ELDS;
            const size_t cb = ((size_t)KCH * 16 * lds_row_elems(dk8, es) + (size_t)dv * lds_row_elems(KCH * 16, es)) * es;
            const size_t lds2 = (cb + 15) / 16 * 16 + (size_t)Mr * Mr * 4;
            if (lds2 > HAT_LDS_MAX) return HAT_ELDS;
            auto ks = ocab_attn_stream_kernel<T, NKT, KCH, 4>;
            if (lds2 > 65536) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                if (e != hipSuccess) return (int)e;
            }
            HAT_LAUNCH(ks, dim3(W / ws, H / ws, B * heads), dim3(256), lds2, s, reinterpret_cast<const T*>(q), reinterpret_cast<const T*>(kv),
                       bias_rot, reinterpret_cast<T*>(out), H, W, C, heads, ws, wse, ldq, ldkv, ldo, kb, pad);
            return hat_check_launch();
        }
        return HAT_ELDS;
    }
    auto kern = ocab_attn_kernel<T, NKT, KCH, SELF, ODD>;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(W / ws, H / ws, B * heads);
    HAT_LAUNCH(kern, grid, dim3(256), lds, s, reinterpret_cast<const T*>(q), reinterpret_cast<const T*>(kv), bias_rot,
                       reinterpret_cast<T*>(out), H, W, C, heads, ws, wse, ldq, ldkv, ldo, shift, kb, pad);
    return hat_check_launch();
}

// ---------------------------------------------------------------------------------------------------------
// HATX key bias / prune mask (hatx_arch.py:421-449): one workgroup per key window, one thread per key.
//   score = tanh(saliency at the key's pixel)  (0 for the zero-padded keys outside the image: tanh(0)), or ||k||_2 over all
//   channels when there is no focus head; the k_keep keys with the largest score are kept — among EQUAL scores the key with
//   the lower window index (torch.topk leaves that order unspecified: DESIGN.md §7) —, kb = focus score (0 without focus
//   head) for a kept key and -inf for a pruned one.
// ---------------------------------------------------------------------------------------------------------
template <typename T, typename ST = T>
__global__ __launch_bounds__(1024) void keybias_kernel(const ST* __restrict__ sal, int ldsal, const T* __restrict__ kv, int ldkv,
                                                       float* __restrict__ kb, int H, int W, int C, int ws, int wse, int pad, int k_keep) {
    __shared__ float sc[1024];
    const int nk = wse * wse, nkp = (nk + 15) & ~15, key = threadIdx.x;   // rows of nkp floats: the attention kernel's key tiles
    const int wx = blockIdx.x, wy = blockIdx.y, b = blockIdx.z;
    float s = 0.f;
    if (key < nk) {
        const int kh = key / wse, kw = key - kh * wse;
        const int y = wy * ws - pad + kh, x = wx * ws - pad + kw;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const size_t pix = ((size_t)b * H + y) * W + x;
            if (sal != nullptr) {
                s = tanhf(to_f(sal[pix * ldsal]));
            } else {
                float q = 0.f;
                for (int c = 0; c < C; ++c) { const float v = to_f(kv[pix * ldkv + c]); q += v * v; }
                s = sqrtf(q);
            }
        }
    }
    sc[key] = s;
    __syncthreads();
    if (key < nk) {
        bool keep = true;
        if (k_keep < nk) {
            int rank = 0;
            for (int j = 0; j < nk; ++j) rank += (sc[j] > s || (sc[j] == s && j < key)) ? 1 : 0;
            keep = rank < k_keep;
        }
        kb[(((size_t)b * gridDim.y + wy) * gridDim.x + wx) * nkp + key] = keep ? (sal != nullptr ? s : 0.f) : -INFINITY;
    } else if (key < nkp) {   // dead keys of the padded window (odd wse): the attention kernel ignores the value
        kb[(((size_t)b * gridDim.y + wy) * gridDim.x + wx) * nkp + key] = 0.f;
    }
}
}  // namespace

namespace {
// The generic kernel's instantiations by key-window size: 24 / 12 (HAT: window 16 / 8, overlap 0.5), 25 (HATX training
// config: window 16, overlap 0.6; 625 keys in 40 tiles) and 13 (window 8, overlap 0.7: the small model of the goldens).
int attn_dispatch(const void* q, const void* kv, const float* bias_rot, void* out, int B, int H, int W, int C, int heads, int ws, int wse,
                  int ldq, int ldkv, int ldo, int dtype, hipStream_t s, const float* kb, int pad) {
#define HAT_ATTN_CASE(TT, N, K, ODD_) return launch_attn<TT, N, K, false, ODD_>(q, kv, bias_rot, out, B, H, W, C, heads, ws, wse, ldq, ldkv, ldo, s, 0, kb, pad)
    if (dtype != HAT_BF16 && dtype != HAT_F32) return HAT_EINVAL;
    if (dtype == HAT_BF16) {
        if (wse == 24) HAT_ATTN_CASE(bf16_t, 36, 12, false);
        if (wse == 12) HAT_ATTN_CASE(bf16_t, 9, 9, false);
        if (wse == 25) HAT_ATTN_CASE(bf16_t, 40, 10, true);
        if (wse == 13) HAT_ATTN_CASE(bf16_t, 11, 11, true);
    } else {
        if (wse == 24) HAT_ATTN_CASE(float, 36, 12, false);
        if (wse == 12) HAT_ATTN_CASE(float, 9, 9, false);
        if (wse == 25) HAT_ATTN_CASE(float, 40, 10, true);
        if (wse == 13) HAT_ATTN_CASE(float, 11, 11, true);
    }
#undef HAT_ATTN_CASE
    return HAT_EUNSUPPORTED;  // other key-window sizes are not instantiated
}
}  // namespace

static int ocab_attention_impl(const void* q, const void* kv, const float* bias_rot, void* out, int32_t B, int32_t H,
                               int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t wse, int32_t ldq,
                               int32_t ldkv, int32_t ldo, int32_t dtype, void* stream, bool qlog2) {
    if (!q || !kv || !bias_rot || !out || B < 1 || heads < 1 || C % heads) return HAT_EINVAL;
    if (ws < 4 || H % ws || W % ws || wse < ws || (ws * ws) % 16) return HAT_EINVAL;
    const int d = C / heads;
    if (d > 32 || d % 2 || ldq < C || ldkv < 2 * C || ldo < C) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool fast24 = d == 24 && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0 && C % 8 == 0;
    const bool fast30 = d == 30 && ldq % 2 == 0 && ldkv % 2 == 0 && ldo % 2 == 0 && C % 2 == 0;
    if (dtype == HAT_BF16 && ws == 16 && wse == 24 && (fast24 || fast30)) {
        const size_t lds = (size_t)576 * ((fast24 && !qlog2) ? 24 : 32) * 2 + (size_t)576 * 32 * 2 + (size_t)39 * 39 * 4;
        static const bool w4 = getenv("HAT_ATTN_4WAVES") != nullptr;   // (A/B switch: round 1's four-wave workgroups)
        if (qlog2 && !fast24) return HAT_EUNSUPPORTED;
        auto kern = fast24 ? (qlog2 ? ocab_attn_fast_kernel<24, 24, false, 512, true>
                                    : (w4 ? ocab_attn_fast_kernel<24, 24, false> : ocab_attn_fast_kernel<24, 24, false, 512>))
                           : (w4 ? ocab_attn_fast_kernel<30, 24, false> : ocab_attn_fast_kernel<30, 24, false, 512>);
        const int nth = (w4 && !qlog2) ? 256 : 512;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        const int nwin = B * (W / ws) * (H / ws);
        HAT_LAUNCH(kern, dim3((nwin + 7) / 8 * 8 * heads), dim3(nth), lds, s, reinterpret_cast<const bf16_t*>(q),
                   reinterpret_cast<const bf16_t*>(kv), bias_rot, reinterpret_cast<bf16_t*>(out), B, H, W, C, heads, ldq, ldkv, ldo, 0);
        return hat_check_launch();
    }
    if (qlog2) return HAT_EUNSUPPORTED;   // only the tuned kernel takes log2-domain queries
    return attn_dispatch(q, kv, bias_rot, out, B, H, W, C, heads, ws, wse, ldq, ldkv, ldo, dtype, s, nullptr, -1);
}

extern "C" int hat_ocab_attention(const void* q, const void* kv, const float* bias_rot, void* out, int32_t B, int32_t H,
                                  int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t wse, int32_t ldq,
                                  int32_t ldkv, int32_t ldo, int32_t dtype, void* stream) {
    return ocab_attention_impl(q, kv, bias_rot, out, B, H, W, C, heads, ws, wse, ldq, ldkv, ldo, dtype, stream, false);
}

extern "C" int hat_ocab_attention_log2(const void* q, const void* kv, const float* bias_rot, void* out, int32_t B, int32_t H,
                                       int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t wse, int32_t ldq,
                                       int32_t ldkv, int32_t ldo, int32_t dtype, void* stream) {
    return ocab_attention_impl(q, kv, bias_rot, out, B, H, W, C, heads, ws, wse, ldq, ldkv, ldo, dtype, stream, true);
}

extern "C" int hat_ocab_keybias(const void* sal, int32_t ldsal, const void* kv, int32_t ldkv, float* kb, int32_t B, int32_t H, int32_t W,
                                int32_t C, int32_t ws, int32_t wse, int32_t pad, int32_t k_keep, int32_t dtype, void* stream) {
    if (!kb || (!sal && !kv) || B < 1 || ws < 1 || H % ws || W % ws || wse < ws || wse * wse > 1024 || pad < 0 || k_keep < 1) return HAT_EINVAL;
    if ((sal && ldsal == 0) || (sal && ldsal < 0 && dtype != HAT_BF16) || (!sal && ldkv < C)) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(W / ws, H / ws, B);
    if (dtype == HAT_BF16 && sal && ldsal < 0)   // ldsal < 0: the saliency map is FP32 with row stride -ldsal (the keys are ranked on it)
        HAT_LAUNCH((keybias_kernel<bf16_t, float>), grid, dim3(1024), 0, s, reinterpret_cast<const float*>(sal), -ldsal, reinterpret_cast<const bf16_t*>(kv),
                   ldkv, kb, H, W, C, ws, wse, pad, k_keep);
    else if (dtype == HAT_BF16)
        HAT_LAUNCH(keybias_kernel<bf16_t>, grid, dim3(1024), 0, s, reinterpret_cast<const bf16_t*>(sal), ldsal, reinterpret_cast<const bf16_t*>(kv), ldkv, kb,
                   H, W, C, ws, wse, pad, k_keep);
    else if (dtype == HAT_F32)
        HAT_LAUNCH(keybias_kernel<float>, grid, dim3(1024), 0, s, reinterpret_cast<const float*>(sal), ldsal, reinterpret_cast<const float*>(kv), ldkv, kb, H, W,
                   C, ws, wse, pad, k_keep);
    else
        return HAT_EINVAL;
    return hat_check_launch();
}

extern "C" int hat_ocab_attention_kb(const void* q, const void* kv, const float* bias_rot, const float* kb, void* out, int32_t B, int32_t H,
                                     int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t wse, int32_t pad, int32_t ldq, int32_t ldkv,
                                     int32_t ldo, int32_t dtype, void* stream) {
    if (!q || !kv || !bias_rot || !kb || !out || B < 1 || heads < 1 || C % heads) return HAT_EINVAL;
    if (ws < 4 || H % ws || W % ws || wse < ws || (ws * ws) % 16 || pad != (wse - ws + 1) / 2) return HAT_EINVAL;
    const int d = C / heads;
    if (d > 32 || d % 2 || ldq < C || ldkv < 2 * C || ldo < C) return HAT_EINVAL;
    return attn_dispatch(q, kv, bias_rot, out, B, H, W, C, heads, ws, wse, ldq, ldkv, ldo, dtype, reinterpret_cast<hipStream_t>(stream), kb, pad);
}

extern "C" int hat_window_attention(const void* q, const void* kv, const float* bias_flip, void* out, int32_t B, int32_t H,
                                    int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t shift, int32_t ldq,
                                    int32_t ldkv, int32_t ldo, int32_t dtype, void* stream) {
    if (!q || !kv || !bias_flip || !out || B < 1 || heads < 1 || C % heads) return HAT_EINVAL;
    if (ws < 4 || ws % 4 || H % ws || W % ws || shift < 0 || shift >= ws || shift % 4) return HAT_EINVAL;
    const int d = C / heads;
    if (d > 32 || d % 2 || ldq < C || ldkv < 2 * C || ldo < C) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool fast24 = d == 24 && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0 && C % 8 == 0 &&
                        (reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(kv)) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0;
    const bool fast30 = d == 30 && ldq % 2 == 0 && ldkv % 2 == 0 && ldo % 2 == 0 && C % 2 == 0;
    if (dtype == HAT_BF16 && ws == 16 && (fast24 || fast30)) {
        const size_t lds = (size_t)256 * (fast24 ? 24 : 32) * 2 + (size_t)256 * 32 * 2 + (size_t)31 * 31 * 4;
        auto kern = fast24 ? ocab_attn_fast_kernel<24, 16, true> : ocab_attn_fast_kernel<30, 16, true>;
        const int nwin = B * (W / ws) * (H / ws);
        HAT_LAUNCH(kern, dim3((nwin + 7) / 8 * 8 * heads), dim3(256), lds, s, reinterpret_cast<const bf16_t*>(q),
                   reinterpret_cast<const bf16_t*>(kv), bias_flip, reinterpret_cast<bf16_t*>(out), B, H, W, C, heads, ldq, ldkv,
                   ldo, shift);
        return hat_check_launch();
    }
#define HAT_WATTN_CASE(TT, N, K) return launch_attn<TT, N, K, true>(q, kv, bias_flip, out, B, H, W, C, heads, ws, ws, ldq, ldkv, ldo, s, shift)
    if (dtype == HAT_BF16) {
        if (ws == 16) HAT_WATTN_CASE(bf16_t, 16, 8);
        if (ws == 8) HAT_WATTN_CASE(bf16_t, 4, 4);
    } else if (dtype == HAT_F32) {
        if (ws == 16) HAT_WATTN_CASE(float, 16, 8);
        if (ws == 8) HAT_WATTN_CASE(float, 4, 4);
    } else {
        return HAT_EINVAL;
    }
#undef HAT_WATTN_CASE
    return HAT_EUNSUPPORTED;  // window sizes other than 16 and 8 are not instantiated
}
