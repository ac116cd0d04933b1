// hat_ffn.hip — fused HAB feed-forward half on gfx950 (contract: HatFfnDesc in include/hat_mi355x.h):
//     t_out = t_in + fc2( a * SiLU(g) ),   [a | g] = dwconv3x3( fc1( LayerNorm2(t_in) ) )
// reference: hat/archs/hat_arch.py:237 (x + mlp(norm2(x))) with GatedDconvFFN.forward :107-119.
//
// One workgroup = WAVES waves = a (2*WAVES) x 16 pixel tile.  The 4C-wide hidden tensor (the
// reference materialises it twice in HBM: 1-6 GB) lives only in LDS, 32 (+32 gate) channels at a time:
//
//   stage 0   LN2 of the haloed (rows+2) x 18 input tile, fp32 stats            -> Ms[pix][C]     (T)
//   per chunk c of 32 hidden channels (a-part) + their 32 gate channels:
//     fc1     U = Ms . W1[chunk]^T + b1 on ALL haloed pixels (MFMA), 0 outside the image (the
//             depthwise conv zero-pads u, AFTER the fc1 bias)                    -> Us[pix][64]    (T)
//     dw      depthwise 3x3 + bias on the tile's own pixels, a * SiLU(g) (VALU; lane = pixel,
//             wave-uniform channel group => weights come from SGPRs)             -> Gs[pix][32]    (T)
//     fc2     acc[C][pix] += W2[:, chunk] . Gs^T  (MFMA, accumulators live in registers across chunks)
//   epilogue  t_out = t_in + acc + b2 ; optionally the NEXT LayerNorm of t_out and its GAP partials.
//
// Weights never touch LDS: they are fragment-packed on the host, so a wave's A operand is ONE
// coalesced 1 KiB global load (L1/L2 resident).  Us / Gs rows are exact powers of two and are
// XOR-swizzled per 16-byte slot instead of padded (LDS is full: 156 KiB at C = 144, bf16).
#include "hat_common.h"

namespace {

constexpr int CH = 32;        // hidden channels per chunk (a-part); the chunk also carries CH gate channels
constexpr int HALO_W = 18;    // 16 + 2

template <typename T> struct Q16 { static constexpr int N = 16 / sizeof(T); T v[16 / sizeof(T)]; } __attribute__((aligned(16)));

// swizzled element offset inside a [rows][NS slots of 16 bytes] LDS array with NS a power of two <= 16
template <int NS> __device__ __forceinline__ int swz_slot(int row, int slot) {
    constexpr int R = 16 / NS;  // rows per 256-byte bank row
    return slot ^ ((row / R) & (NS - 1));
}

__host__ __device__ inline int ffn_kp(int C) { return (C % 32 == 16) ? C : ((C + 31) & ~31); }

template <typename T, int WAVES>
__host__ __device__ inline size_t ffn_lds_bytes(int C) {
    const int nph = (2 * WAVES + 2) * HALO_W;
    const size_t ms = (size_t)nph * lds_row_elems(ffn_kp(C), sizeof(T)) * sizeof(T);
    const size_t us = (size_t)nph * 2 * CH * sizeof(T);
    const size_t gs = (size_t)2 * WAVES * 16 * CH * sizeof(T);
    return ms + us + gs;
}

template <typename T, int WAVES, int NT, int KS, bool KHALF>
__global__ __launch_bounds__(WAVES * 64) void ffn_kernel(const HatFfnDesc d) {
    using M = MT<T>;
    constexpr int NTHR = WAVES * 64;
    constexpr int TROWS = 2 * WAVES;
    constexpr int NPH = (TROWS + 2) * HALO_W;     // haloed pixels
    constexpr int NPT = (NPH + 15) / 16;          // fc1 pixel tiles over the flattened halo tile
    constexpr int VECN = M::VEC;                  // elements per 16 bytes
    constexpr int NOCT = CH / VECN;               // 16-byte channel groups per half-chunk
    constexpr int NSU = 2 * CH / VECN;            // 16-byte slots per Us row
    constexpr int NSG = CH / VECN;                // 16-byte slots per Gs row
    constexpr int PG = WAVES / 2;                 // fc1 pixel groups (each handled by 2 waves: a / g n-tile pairs)
    constexpr bool BF = sizeof(T) == 2;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = d.C;
    const int Kp = ffn_kp(C);
    const int ldm = lds_row_elems(Kp, sizeof(T));
    T* Ms = reinterpret_cast<T*>(smem);
    T* Us = Ms + (size_t)NPH * ldm;
    T* Gs = Us + (size_t)NPH * 2 * CH;

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * TROWS;
    const int H = d.H, W = d.W;
    const float* tin = d.t_in + (size_t)b * H * W * C;
    const int hid_p = d.chunks * CH;

    // ------------------------------ stage 0: LayerNorm2 -> Ms --------------------------------
    {
        const int j = tid & 15, grp = tid >> 4;
        const float invC = 1.0f / (float)C;
        for (int hp = grp; hp < NPH; hp += NTHR / 16) {
            const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool inside = y >= 0 && y < H && x >= 0 && x < W;
            f32x4 xv[3];
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int c = 4 * j + 64 * v;
                xv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (inside && c < C) xv[v] = *reinterpret_cast<const f32x4*>(tin + ((size_t)y * W + x) * C + c);
                s += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
            }
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            const float mean = s * invC;
            float q = 0.f;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int c = 4 * j + 64 * v;
                if (c < C) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float dl = xv[v][r] - mean; q += dl * dl; }
                }
            }
            q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
            const float rstd = 1.0f / sqrtf(q * invC + 1e-5f);
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int c = 4 * j + 64 * v;
                if (c < Kp) {
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (inside && c < C) {
                        const f32x4 gm = *reinterpret_cast<const f32x4*>(d.ln_g + c);
                        const f32x4 bt = *reinterpret_cast<const f32x4*>(d.ln_b + c);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (xv[v][r] - mean) * rstd * gm[r] + bt[r];
                    }
                    Vec4<T>::store(Ms + (size_t)hp * ldm + c, o);
                }
            }
        }
    }
    __syncthreads();

    // persistent fc2 accumulators: this wave's two tile rows x all NT channel tiles
    f32x4 acc2[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc2[nt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[nt][1] = acc2[nt][0]; }

    const T* w1f = reinterpret_cast<const T*>(d.w1f);
    const T* w2f = reinterpret_cast<const T*>(d.w2f);
    const int nt2 = wave & 1, pg = wave >> 1;

    for (int chunk = 0; chunk < d.chunks; ++chunk) {
        // ------------------------------------ fc1 -> Us ---------------------------------------
        {
            typename M::frag_t a[2][KS];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    a[i][ks] = M::load(w1f + ((((size_t)chunk * 4 + (2 * nt2 + i)) * KS + ks) * 64 + lane) * 8);
            f32x4 bias[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nl = (2 * nt2 + i) * 16 + 4 * g;  // chunk-local channel 0..63: [0,32) a-part, [32,64) gate
                const int gi = nl < CH ? chunk * CH + nl : hid_p + chunk * CH + (nl - CH);
                bias[i] = *reinterpret_cast<const f32x4*>(d.b1 + gi);
            }
            for (int pt = pg; pt < NPT; pt += PG) {
                const int hp = pt * 16 + c16;
                const int hpc = hp < NPH ? hp : NPH - 1;
                const T* mrow = Ms + (size_t)hpc * ldm;
                f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (KHALF && ks == KS - 1) {
                        const typename M::half_t bh = M::load_half(mrow + ks * 32 + 4 * g);
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            typename M::half_t ah;
                            if constexpr (BF) {
                                const typename M::frag_t f = a[i][ks];
                                ah = *reinterpret_cast<const typename M::half_t*>(&f);
                            } else {
                                ah = f32x4{a[i][ks][0], a[i][ks][1], a[i][ks][2], a[i][ks][3]};
                            }
                            acc[i] = M::mma_half(ah, bh, acc[i]);
                        }
                    } else {
                        const typename M::frag_t bf = M::load(mrow + ks * 32 + 8 * g);
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i] = M::mma(a[i][ks], bf, acc[i]);
                    }
                }
                const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
                const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                const bool inside = hp < NPH && y >= 0 && y < H && x >= 0 && x < W;
                if (hp < NPH) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int nl = (2 * nt2 + i) * 16 + 4 * g;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (inside) v = acc[i] + bias[i];
                        const int slot = swz_slot<NSU>(hp, nl / VECN);
                        Vec4<T>::store(Us + (size_t)hp * 2 * CH + slot * VECN + (nl % VECN), v);
                    }
                }
            }
        }
        __syncthreads();

        // --------------------- depthwise 3x3 + bias, a * SiLU(g) -> Gs ------------------------
        {
            constexpr int NRGU = (TROWS + 7) / 8;
            for (int u = wave; u < NOCT * NRGU; u += WAVES) {
                const int oct = u % NOCT, hh = u / NOCT;       // wave-uniform
                const int x = lane & 15, yr = hh * 8 + (lane >> 4) * 2;  // this lane's two output rows yr, yr+1
                if (yr < TROWS) {
                    const int ca = chunk * CH + oct * VECN;    // first a-part channel of this group (global index)
                    float aA[2][VECN], aG[2][VECN];
#pragma unroll
                    for (int c = 0; c < VECN; ++c) {
                        aA[0][c] = aA[1][c] = d.dwb[ca + c];
                        aG[0][c] = aG[1][c] = d.dwb[hid_p + ca + c];
                    }
#pragma unroll
                    for (int ry = 0; ry < 4; ++ry) {
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int hp = (yr + ry) * HALO_W + x + dx;
                            const T* urow = Us + (size_t)hp * 2 * CH;
                            const Q16<T> ua = *reinterpret_cast<const Q16<T>*>(urow + swz_slot<NSU>(hp, oct) * VECN);
                            const Q16<T> ug = *reinterpret_cast<const Q16<T>*>(urow + swz_slot<NSU>(hp, NOCT + oct) * VECN);
#pragma unroll
                            for (int o = 0; o < 2; ++o) {       // output row yr + o uses tap row dy = ry - o
                                const int dy = ry - o;
                                if (dy < 0 || dy > 2) continue;
                                const int tap = dy * 3 + dx;
                                if constexpr (BF) {
                                    typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
                                    const bf2* wq = reinterpret_cast<const bf2*>(d.dww) + (((size_t)chunk * NOCT + oct) * 9 + tap) * 16;
                                    const bf2* pa = reinterpret_cast<const bf2*>(&ua);
                                    const bf2* pgp = reinterpret_cast<const bf2*>(&ug);
#pragma unroll
                                    for (int p = 0; p < 4; ++p) {
                                        aA[o][2 * p] = __builtin_amdgcn_fdot2_f32_bf16(pa[p], wq[2 * p], aA[o][2 * p], false);
                                        aA[o][2 * p + 1] = __builtin_amdgcn_fdot2_f32_bf16(pa[p], wq[2 * p + 1], aA[o][2 * p + 1], false);
                                        aG[o][2 * p] = __builtin_amdgcn_fdot2_f32_bf16(pgp[p], wq[8 + 2 * p], aG[o][2 * p], false);
                                        aG[o][2 * p + 1] = __builtin_amdgcn_fdot2_f32_bf16(pgp[p], wq[8 + 2 * p + 1], aG[o][2 * p + 1], false);
                                    }
                                } else {
                                    const float* wq = reinterpret_cast<const float*>(d.dww) + (((size_t)chunk * NOCT + oct) * 9 + tap) * 8;
#pragma unroll
                                    for (int c = 0; c < VECN; ++c) {
                                        aA[o][c] = fmaf(to_f(ua.v[c]), wq[c], aA[o][c]);
                                        aG[o][c] = fmaf(to_f(ug.v[c]), wq[4 + c], aG[o][c]);
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const int pix = (yr + o) * 16 + x;
                        Q16<T> gv;
#pragma unroll
                        for (int c = 0; c < VECN; ++c) {
                            const float gt = aG[o][c];
                            const float sg = BF ? gt / (1.0f + __expf(-gt)) : gt / (1.0f + expf(-gt));
                            gv.v[c] = to_T<T>(aA[o][c] * sg);
                        }
                        *reinterpret_cast<Q16<T>*>(Gs + (size_t)pix * CH + swz_slot<NSG>(pix, oct) * VECN) = gv;
                    }
                }
            }
        }
        __syncthreads();

        // ------------------------------------ fc2 ---------------------------------------------
        {
            typename M::frag_t bf[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int pix = (2 * wave + pt) * 16 + c16;
                const T* grow = Gs + (size_t)pix * CH;
                if constexpr (BF) {
                    bf[pt] = M::load(grow + swz_slot<NSG>(pix, g) * VECN);
                } else {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(grow + swz_slot<NSG>(pix, 2 * g) * VECN);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(grow + swz_slot<NSG>(pix, 2 * g + 1) * VECN);
                    bf[pt][0] = lo[0]; bf[pt][1] = lo[1]; bf[pt][2] = lo[2]; bf[pt][3] = lo[3];
                    bf[pt][4] = hi[0]; bf[pt][5] = hi[1]; bf[pt][6] = hi[2]; bf[pt][7] = hi[3];
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const typename M::frag_t af = M::load(w2f + (((size_t)chunk * NT + nt) * 64 + lane) * 8);
                acc2[nt][0] = M::mma(af, bf[0], acc2[nt][0]);
                acc2[nt][1] = M::mma(af, bf[1], acc2[nt][1]);
            }
        }
        // no barrier needed here: the next fc1 writes Us (last read before the previous barrier) and the
        // next dw stage writes Gs only after the barrier that follows that fc1
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
            const int n = nt * 16 + 4 * g;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (valid && n < C) {
                v = acc2[nt][pt] + *reinterpret_cast<const f32x4*>(d.b2 + n) + *reinterpret_cast<const f32x4*>(tin + pix * C + n);
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
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(d.ln1_g + n);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(d.ln1_b + n);
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (acc2[nt][pt][r] - mean) * rstd * gm[r] + bt[r];
                    Vec4<T>::store(nout + n, o);
                    if (nt == 0 && n < d.gap_c) gapv += o;
                }
            }
        }
    }
    if (do_ln && d.gap_out != nullptr) {
        __syncthreads();  // Us is free: use it as the cross-wave reduction scratch
        float* red = reinterpret_cast<float*>(Us);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = gapv[r];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
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

template <typename T, int WAVES, int NT, int KS, bool KHALF>
int launch_ffn(const HatFfnDesc& d, hipStream_t s) {
    const size_t lds = ffn_lds_bytes<T, WAVES>(d.C);
    if (lds > HAT_LDS_MAX) return HAT_ELDS;
    auto kern = ffn_kernel<T, WAVES, NT, KS, KHALF>;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((d.W + 15) / 16, (d.H + 2 * WAVES - 1) / (2 * WAVES), d.B);
    HAT_LAUNCH(kern, grid, dim3(WAVES * 64), lds, s, d);
    return hat_check_launch();
}

// tile rows (= 2 * waves) used for (C, dtype); 0 if the shape is not instantiated
int ffn_waves(const HatFfnDesc& d) {
    const bool small = d.C <= 32 && d.C % 32 != 16;  // one zero-padded 32-deep k-step
    if (d.dtype == HAT_BF16) return d.C == 144 ? 8 : (d.C == 180 ? 4 : (small ? 8 : 0));
    if (d.dtype == HAT_F32) return (d.C == 144 || d.C == 180 || small) ? 2 : 0;
    return 0;
}

}  // namespace

extern "C" int hat_ffn_tiles(const HatFfnDesc* d, int32_t* tiles_out) {
    if (!d || !tiles_out) return HAT_EINVAL;
    const int wv = ffn_waves(*d);
    if (!wv) return HAT_EUNSUPPORTED;
    *tiles_out = ((d->W + 15) / 16) * ((d->H + 2 * wv - 1) / (2 * wv));
    return 0;
}

extern "C" int hat_ffn(const HatFfnDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatFfnDesc& d = *dp;
    if (!d.t_in || !d.t_out || d.t_in == d.t_out || !d.ln_g || !d.ln_b || !d.w1f || !d.b1 || !d.dww || !d.dwb || !d.w2f || !d.b2)
        return HAT_EINVAL;
    if (d.B < 1 || d.H < 1 || d.W < 1 || d.C < 8 || d.C % 4 || d.chunks < 1) return HAT_EINVAL;
    if (d.ln1_g && (!d.ln1_b || !d.n_out || d.ldn < d.C || d.ldn % 4 || d.gap_c < 0 || d.gap_c > 16 || d.gap_c % 4)) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool small = d.C <= 32 && d.C % 32 != 16;
    if (d.dtype == HAT_BF16) {
        if (d.C == 144) return launch_ffn<bf16_t, 8, 9, 5, true>(d, s);
        if (d.C == 180) return launch_ffn<bf16_t, 4, 12, 6, false>(d, s);
        if (small) return launch_ffn<bf16_t, 8, 2, 1, false>(d, s);
    } else if (d.dtype == HAT_F32) {
        if (d.C == 144) return launch_ffn<float, 2, 9, 5, true>(d, s);
        if (d.C == 180) return launch_ffn<float, 2, 12, 6, false>(d, s);
        if (small) return launch_ffn<float, 2, 2, 1, false>(d, s);
    } else {
        return HAT_EINVAL;
    }
    return HAT_EUNSUPPORTED;  // embed_dim other than 144 / 180 / <=32: the host falls back to the unfused kernels
}
