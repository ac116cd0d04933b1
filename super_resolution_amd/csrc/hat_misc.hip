// hat_misc.hip — the HBM-bound and tiny kernels of the HAT forward on gfx950:
// LayerNorm (+ESC global-average-pool partials), ESC per-sample weights, ECA scale,
// depthwise-3x3 + gate.  Contracts: include/hat_mi355x.h.
#include "hat_common.h"

namespace {

constexpr int LN_BLOCKS = 1024;  // fixed grid.x: the gap partial layout [B][LN_BLOCKS][16 or 32] is deterministic

// One pixel per 16 lanes; lane j owns channels 4j + 64v (float4 each), v < NV  =>  every wave
// instruction reads 4 pixels x 256 contiguous bytes.  Two-pass statistics in registers.
template <typename OutT, int NV>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, OutT* __restrict__ y,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float* __restrict__ gap_partial, long npix, int C, int ldy, int gap_c) {
    __shared__ float red[16][32];
    const int tid = threadIdx.x, j = tid & 15, grp = tid >> 4;
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * npix * C;
    OutT* yb = y + (size_t)b * npix * ldy;
    f32x4 gm[NV], bt[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = 4 * j + 64 * v;
        gm[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        bt[v] = gm[v];
        if (c < C) {
            gm[v] = *reinterpret_cast<const f32x4*>(gamma + c);
            bt[v] = *reinterpret_cast<const f32x4*>(beta + c);
        }
    }
    f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
    const float invC = 1.0f / (float)C;
    for (long p0 = (long)blockIdx.x * 16; p0 < npix; p0 += (long)gridDim.x * 16) {
        const long p = p0 + grp;
        const bool pv = p < npix;
        f32x4 xv[NV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * j + 64 * v;
            xv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pv && c < C) xv[v] = *reinterpret_cast<const f32x4*>(xb + (size_t)p * C + c);
            s += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
        }
        s = row_sum16(s);
        const float mean = s * invC;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * j + 64 * v;
            if (c < C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float dlt = xv[v][r] - mean; q += dlt * dlt; }
            }
        }
        q = row_sum16(q);
        const float rstd = 1.0f / sqrtf(q * invC + 1e-5f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = 4 * j + 64 * v;
            if (pv && c < C) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (xv[v][r] - mean) * rstd * gm[v][r] + bt[v][r];
                Vec4<OutT>::store(yb + (size_t)p * ldy + c, o);
                if (v == 0 && c < gap_c) gsum += as_stored<OutT>(o);
            }
        }
    }
    if (gap_partial != nullptr) {
        // lanes j < 8 hold channels 4j..4j+3 (< 32): fixed-order reduction over the 16 pixel groups.  Rows of 16 floats, or of
        // 32 when more than 16 channels are pooled (ESC on 24 / 32 channels: the HATX training config)
        const int gs = gap_c > 16 ? 32 : 16;
        if (j < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[grp][4 * j + r] = gsum[r];
        }
        __syncthreads();
        if (tid < gs) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k][tid];
            gap_partial[((size_t)b * gridDim.x + blockIdx.x) * gs + tid] = tid < gap_c ? s : 0.f;
        }
    }
}

template <typename OutT>
int launch_ln(const float* x, void* y, const float* gamma, const float* beta, float* gap, int B, long npix, int C,
              int ldy, int gap_c, hipStream_t s) {
    dim3 grid(LN_BLOCKS, B), block(256);
    const int nv = (C + 63) / 64;
    OutT* yo = reinterpret_cast<OutT*>(y);
    switch (nv) {
        case 1: HAT_LAUNCH((ln_kernel<OutT, 1>), grid, block, 0, s, x, yo, gamma, beta, gap, npix, C, ldy, gap_c); break;
        case 2: HAT_LAUNCH((ln_kernel<OutT, 2>), grid, block, 0, s, x, yo, gamma, beta, gap, npix, C, ldy, gap_c); break;
        case 3: HAT_LAUNCH((ln_kernel<OutT, 3>), grid, block, 0, s, x, yo, gamma, beta, gap, npix, C, ldy, gap_c); break;
        case 4: HAT_LAUNCH((ln_kernel<OutT, 4>), grid, block, 0, s, x, yo, gamma, beta, gap, npix, C, ldy, gap_c); break;
        default: return HAT_EUNSUPPORTED;
    }
    return hat_check_launch();
}

// ---------------------------------------------------------------------------------------------
// ESC per-sample weights: one block per (output channel co, sample b).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void esc_weights_kernel(const float* __restrict__ gap_partial, int nblk, float inv_npix,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          const float* __restrict__ plk, T* __restrict__ w_out,
                                                          int pdim, int ksize, int Kpad) {
    __shared__ float part[1024];
    __shared__ float pmean[32];
    __shared__ float hid[16];
    __shared__ float dk[9];
    const int tid = threadIdx.x, co = blockIdx.x, b = blockIdx.y;
    const int npad = gridDim.x;            // weight rows per sample: 16, or 32 for pdim > 16
    const int gs = pdim > 16 ? 32 : 16;    // floats per GAP partial block (hat_layernorm's layout for gap_c = pdim)
    if (co >= pdim) {  // rows pdim..npad-1 of the MFMA tiles are zero
        T* z = w_out + ((size_t)b * npad + co) * Kpad;
        for (int k = tid; k < Kpad; k += 1024) z[k] = to_T<T>(0.f);
        return;
    }
    const int np = 1024 / gs;              // strided parts of the reduction
    {   // fixed-order two-level reduction of the GAP partials: np strided parts, 8 independent chains each, the 8 loads of
        // an iteration unconditional (clamped + select) so that they are all in flight together — this kernel sits on the
        // critical path in front of the large-kernel conv and is nothing but this latency
        const int ci = tid & (gs - 1), pp = tid / gs;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* gp = gap_partial + (size_t)b * nblk * gs + ci;
        // FOUR iterations' loads (32) are issued before the first add: a 720p frame has 7200 partial blocks = 15 iterations,
        // each one round trip to wherever the tail's workgroups left them (other XCDs' L2 / HBM): 4 round trips instead of 15.
        // The adds keep the order of the one-iteration-at-a-time loop (same bits).
        for (int k0 = pp; k0 < nblk; k0 += np * 8 * 4) {
            float v[4][8];
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int u = 0; u < 8; ++u) v[it][u] = gp[(size_t)min(k0 + np * (8 * it + u), nblk - 1) * gs];
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += (k0 + np * (8 * it + u) < nblk) ? v[it][u] : 0.f;
        }
        part[pp * gs + ci] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    __syncthreads();
    if (tid < gs) {
        float s = 0.f;
        for (int k = 0; k < np; ++k) s += part[k * gs + tid];
        pmean[tid] = s * inv_npix;
    }
    __syncthreads();
    const int hdim = pdim / 2;
    if (tid < hdim) {
        float s = b1[tid];
        for (int ci = 0; ci < pdim; ++ci) s += w1[tid * pdim + ci] * pmean[ci];
        hid[tid] = gelu_erf(s);
    }
    __syncthreads();
    if (tid < 9) {
        const int t = co * 9 + tid;
        float s = b2[t];
        for (int m = 0; m < hdim; ++m) s += w2[t * hdim + m] * hid[m];
        dk[tid] = s;
    }
    __syncthreads();
    const int cin_p = (pdim + 7) & ~7;
    const int ctr = ksize / 2;
    const float* src = plk + (size_t)co * Kpad;
    T* dst = w_out + ((size_t)b * npad + co) * Kpad;
    for (int k = tid; k < Kpad; k += 1024) {
        float v = src[k];
        const int tap = k / cin_p, ci = k - tap * cin_p;
        if (ci == co && tap < ksize * ksize) {
            const int dy = tap / ksize - ctr, dx = tap % ksize - ctr;
            if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) v += dk[(dy + 1) * 3 + (dx + 1)];
        }
        dst[k] = to_T<T>(v);
    }
}

// ---------------------------------------------------------------------------------------------
// ECA: two-stage deterministic reduction of hat_conv's per-tile column sums, conv1d, sigmoid.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eca_reduce_kernel(const float* __restrict__ colsum, int tiles, int ldc,
                                                         float* __restrict__ tmp) {
    // 32 parts x B workgroups; inside a part the 256 threads form nsub = 256 / pow2(ldc) interleaved sub-rows per
    // channel (a 16-channel colsum would otherwise keep 16 lanes busy with hundreds of dependent loads), combined in
    // a fixed order through LDS.
    __shared__ float sub[256];
    int lw = 1;
    while (lw < ldc) lw <<= 1;
    const int nsub = 256 / lw;
    const int c = threadIdx.x % lw, sb = threadIdx.x / lw, part = blockIdx.x, b = blockIdx.y;
    const int per = (tiles + 31) / 32;
    const int t0 = part * per, t1 = min(tiles, t0 + per);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // 8 loads in flight per thread
    if (c < ldc && t1 > t0) {
        for (int t = t0 + sb; t < t1; t += 8 * nsub) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tt = t + u * nsub;
                const float v = colsum[((size_t)b * tiles + min(tt, t1 - 1)) * ldc + c];
                a[u] += tt < t1 ? v : 0.f;
            }
        }
    }
    sub[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (sb == 0 && c < ldc) {
        float s = 0.f;
        for (int i = 0; i < nsub; ++i) s += sub[i * lw + c];
        tmp[((size_t)b * 32 + part) * ldc + c] = s;
    }
}

__global__ __launch_bounds__(256) void eca_scale_kernel(const float* __restrict__ tmp, int ldc, float inv_npix,
                                                        const float* __restrict__ wk, int k, float conv_scale,
                                                        float* __restrict__ scale, int C) {
    __shared__ float mean[256];
    const int c = threadIdx.x, b = blockIdx.x;
    float s = 0.f;
    if (c < C) {
        for (int p = 0; p < 32; ++p) s += tmp[((size_t)b * 32 + p) * ldc + c];
    }
    mean[c] = s * inv_npix;
    __syncthreads();
    if (c < C) {
        float e = 0.f;
        for (int j = 0; j < k; ++j) {
            const int cc = c + j - k / 2;
            if (cc >= 0 && cc < C) e += wk[j] * mean[cc];
        }
        scale[(size_t)b * ldc + c] = conv_scale / (1.0f + expf(-e));
    }
}

// ---------------------------------------------------------------------------------------------
// hat_cab_fold (see include/hat_mi355x.h): one 256-thread workgroup per sample.
// ---------------------------------------------------------------------------------------------
// DIRECT: few tiles — the workgroup sums the per-tile column sums itself (one launch instead of two)
template <typename T, bool DIRECT>
__global__ __launch_bounds__(256) void cab_fold_kernel(const HatCabFoldDesc d) {
    // 6.6 KB of LDS in all, so that the one workgroup fits on a CU beside a workgroup of the 13x13 ESC conv (155 KB of the CU's
    // 160 KB, held for the conv's whole life) instead of waiting for one of them to exit.
    __shared__ float sm[256 * 5 + 256];
    float* red = sm;                                                    // [thread][4 values], rows padded to 5 floats: one of 8 passes
    float (*part)[32] = reinterpret_cast<float (*)[32]>(sm + 256 * 5);   // [8][32]
    float* mean = sm;                                                   // (after the sums are done)
    float* scl = sm + 256;
    __shared__ float bord[4][8];     // first row, last row, first column, last column sums
    __shared__ float S[9][8];        // per-tap sums of c1 over the output pixels the tap contributes to
    const int tid = threadIdx.x, b = blockIdx.x;
    const int H = d.H, W = d.W, C = d.C, mid = d.mid;
    const T* c1 = reinterpret_cast<const T*>(d.c1) + (size_t)b * H * W * d.ld1;
    const float* st = d.stats ? d.stats + (size_t)b * 72 : nullptr;   // (uniform) sums supplied by the caller: band-sharded frames
    if (st != nullptr) {
        if (tid < 32) bord[tid >> 3][tid & 7] = st[8 + tid];
        __syncthreads();
    } else {
    // border sums in a fixed order: every thread strides the four lines, then two levels of partial sums
        float a[4][8];
#pragma unroll
        for (int line = 0; line < 4; ++line)
#pragma unroll
            for (int c = 0; c < 8; ++c) a[line][c] = 0.f;
        const int lmax = W > H ? W : H;
        // BU iterations' loads (4 BU, clamped: unconditional) are in flight together: a load under a lane mask is waited for where
        // it is issued, which made this loop 20 dependent round trips at 720p (40 of the kernel's 56 us; it is ONE workgroup, on
        // the chain squeeze conv -> fold -> tail).  Additions in the order of the one-at-a-time loop.
        constexpr int BU = sizeof(T) == 2 ? 5 : 2;
        for (int i0 = tid; i0 < lmax; i0 += 256 * BU) {
            typename MT<T>::frag_t v[BU][4];
#pragma unroll
            for (int u = 0; u < BU; ++u)
#pragma unroll
                for (int line = 0; line < 4; ++line) {
                    const int len = line < 2 ? W : H;
                    const int i = min(i0 + u * 256, len - 1);
                    const size_t pix = line == 0 ? (size_t)i : line == 1 ? (size_t)(H - 1) * W + i : line == 2 ? (size_t)i * W : (size_t)i * W + (W - 1);
                    v[u][line] = MT<T>::load(c1 + pix * d.ld1);
                }
#pragma unroll
            for (int u = 0; u < BU; ++u)
#pragma unroll
                for (int line = 0; line < 4; ++line) {
                    const bool in = i0 + u * 256 < (line < 2 ? W : H);
#pragma unroll
                    for (int c = 0; c < 8; ++c) a[line][c] += in ? to_f(v[u][line][c]) : 0.f;
                }
        }
        // value v = line * 8 + channel: 32 interleaved partial sums of 32 threads each, then their sum — in eight passes of
        // four values through the small staging buffer (same order of additions as one pass over all 32 values)
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[tid * 5 + k] = a[pass >> 1][(pass & 1) * 4 + k];
            __syncthreads();
            if (tid < 32) {
                const int v4 = tid & 3, pt = tid >> 2;
                float sp = 0.f;
                for (int i = 0; i < 32; ++i) sp += red[(pt * 32 + i) * 5 + v4];
                part[pt][pass * 4 + v4] = sp;
            }
            __syncthreads();
        }
        if (tid < 32) {
            float sp = 0.f;
            for (int i = 0; i < 8; ++i) sp += part[i][tid];
            bord[tid >> 3][tid & 7] = sp;
        }
        __syncthreads();
    }
    if (DIRECT && st == nullptr) {  // totals of the 8 channels: 32 interleaved sub-sums per channel, then a fixed-order sum
        const int c = tid & 7, sb = tid >> 3;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // 8 loads in flight per thread
        for (int t = sb; t < d.tiles; t += 256) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tt = t + 32 * u;
                const float v = d.c1_colsum[((size_t)b * d.tiles + min(tt, d.tiles - 1)) * d.ldcs + c];
                a[u] += tt < d.tiles ? v : 0.f;
            }
        }
        part[c][sb] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __syncthreads();
    }
    if (tid < 72) {
        const int tap = tid >> 3, c = tid & 7;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        float tot = 0.f;
        if (st != nullptr) {
            tot = st[c];
        } else if constexpr (DIRECT) {
            for (int p = 0; p < 32; ++p) tot += part[c][p];
        } else {
            for (int p = 0; p < 32; ++p) tot += d.tmp[((size_t)b * 32 + p) * d.ldcs + c];
        }
        // input pixel q = p + (dy,dx) must lie inside: dy=+1 never reads row 0, dy=-1 never reads row H-1 (same for columns)
        float v = tot;
        const int rl = dy == 1 ? 0 : dy == -1 ? 1 : -1, cl = dx == 1 ? 2 : dx == -1 ? 3 : -1;
        if (rl >= 0) v -= bord[rl][c];
        if (cl >= 0) v -= bord[cl][c];
        if (rl >= 0 && cl >= 0) {
            const size_t pix = (size_t)(rl == 0 ? 0 : H - 1) * W + (cl == 2 ? 0 : W - 1);
            v += st != nullptr ? st[40 + (rl * 2 + (cl - 2)) * 8 + c] : to_f(c1[pix * d.ld1 + c]);
        }
        S[tap][c] = c < mid ? v : 0.f;
    }
    __syncthreads();
    float m = 0.f;
    {   // (every global load of this kernel is unconditional and batched: it is one workgroup, so each dependent load
        // costs a full L2 round trip)
        const int co = tid < C ? tid : C - 1;
        float wv[72];
        if (d.w2f != nullptr) {
            // fragment order: the 8 ci of (co, tap) are contiguous and the 16 co of a tile 32 bytes apart — unit-stride reads where
            // the [C][mid][3][3] layout made every thread walk its own 216-byte row
            const f32x4* wf4 = reinterpret_cast<const f32x4*>(d.w2f) + (size_t)(co >> 4) * (3 * 64 * 2);
            f32x4 w8[9][2];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int e = ((tap >> 2) * 64 + (tap & 3) * 16 + (co & 15)) * 2;
                w8[tap][0] = wf4[e];
                w8[tap][1] = wf4[e + 1];
            }
#pragma unroll
            for (int q = 0; q < 72; ++q) { const int ci = q / 9, tap = q - ci * 9; wv[q] = w8[tap][ci >> 2][ci & 3]; }   // (zero for ci >= mid)
        } else {
#pragma unroll
            for (int q = 0; q < 72; ++q) wv[q] = d.w2[(size_t)co * mid * 9 + min(q, mid * 9 - 1)];
        }
        const float bb = d.b2[co];
#pragma unroll
        for (int q = 0; q < 72; ++q) {
            const int ci = q / 9, tap = q - ci * 9;
            if (q < mid * 9) m += wv[q] * S[tap][ci < 8 ? ci : 7];
        }
        m = tid < C ? m / ((float)H * (float)W) + bb : 0.f;
    }
    mean[tid] = m;
    __syncthreads();
    float sc = 0.f;
    if (tid < C) {
        float e = 0.f;
        for (int j = 0; j < d.k; ++j) {
            const int cc = tid + j - d.k / 2;
            if (cc >= 0 && cc < C) e += d.wk[j] * mean[cc];
        }
        sc = d.conv_scale / (1.0f + expf(-e));
        d.scale[(size_t)b * d.ld_scale + tid] = sc;
    }
    scl[tid] = sc;
    __syncthreads();
    const int nt = (C + 15) / 16;
    for (int n = tid; n < nt * 16; n += 256) d.bias_out[(size_t)b * nt * 16 + n] = n < C ? d.bias_in[n] + scl[n] * d.b2[n] : 0.f;
    T* wf = reinterpret_cast<T*>(d.wf) + (size_t)b * nt * 3 * 512;
    if (d.w2f != nullptr) {
        // four consecutive elements per thread and step (one output channel, four ci), every step's 16-byte load issued before
        // the first product: one round trip for the whole image
        constexpr int MAXS = 16;   // 4 * 256 * 16 elements per pass
        const int total4 = nt * 3 * 128;
        for (int base = 0; base < total4; base += 256 * MAXS) {   // (one pass up to C = 160)
        f32x4 wq[MAXS];
#pragma unroll
        for (int st = 0; st < MAXS; ++st) wq[st] = reinterpret_cast<const f32x4*>(d.w2f)[min(base + tid + st * 256, total4 - 1)];
#pragma unroll
        for (int st = 0; st < MAXS; ++st) {
            const int i4 = base + tid + st * 256;
            if (i4 < total4) {
                const int i = 4 * i4, j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % 3, t = i / (3 * 512);
                const int co = t * 16 + (lane & 15), tap = 4 * ks + (lane >> 4);
                const int cov = min(co, C - 1);
                const float sv = scl[cov];
                f32x4 v = wq[st] * f32x4{sv, sv, sv, sv};
                if (co < C && tap == 9 && j == 0) {   // k = 72, 73: the folded bias, head and remainder (see below)
                    const float bfull = d.bias_in[cov] + sv * d.b2[cov];
                    v[0] = bfull;
                    v[1] = bfull - to_f(to_T<T>(bfull));
                }
                Vec4<T>::store(wf + i, v);
            }
        }
        }
        return;
    }
    constexpr int FU = 18;  // elements per thread per batch (13 824 elements at C = 144: three batches)
    for (int i0 = tid; i0 < nt * 3 * 512; i0 += 256 * FU) {
        float wv[FU], bi[FU], b2v[FU];
        bool ok[FU];
        int cov[FU], bsel[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = min(i0 + u * 256, nt * 3 * 512 - 1);
            const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % 3, t = i / (3 * 512);
            const int co = t * 16 + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
            const int tap = k >> 3, ci = k & 7;
            ok[u] = co < C && tap < 9 && ci < mid;
            // k = 72, 73 (the first two of the 24 unused k-slots): the folded bias itself, split into a bf16 head and the bf16
            // of its remainder.  hat_hab_tail3 multiplies these two slots by 1.0 (its accumulators then start from the
            // residual alone); every other consumer reads zeros there (their im2col has no tap 9), so the columns are inert.
            bsel[u] = (co < C && tap == 9 && ci < 2) ? 1 + ci : 0;
            cov[u] = min(co, C - 1);
            wv[u] = d.w2[((size_t)cov[u] * mid + min(ci, mid - 1)) * 9 + min(tap, 8)];
            bi[u] = d.bias_in[cov[u]];
            b2v[u] = d.b2[cov[u]];
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = i0 + u * 256;
            const float bfull = bi[u] + scl[cov[u]] * b2v[u];
            const float bhead = to_f(to_T<T>(bfull));
            const float v = bsel[u] == 1 ? bfull : (bsel[u] == 2 ? bfull - bhead : (ok[u] ? scl[cov[u]] * wv[u] : 0.f));
            if (i < nt * 3 * 512) wf[i] = to_T<T>(v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// depthwise 3x3 + bias on 2*hid channels, then a * silu(g).  4 channels per thread.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dwgate_kernel(const T* __restrict__ u, const float* __restrict__ wdw,
                                                     const float* __restrict__ bdw, T* __restrict__ out, int B, int H, int W,
                                                     int hid, int ldu, int ldo) {
    const int gpp = hid / 4;
    const size_t total = (size_t)B * H * W * gpp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / gpp;
        const int c = (int)(i - pix * gpp) * 4;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        f32x4 a = *reinterpret_cast<const f32x4*>(bdw + c);
        f32x4 gt = *reinterpret_cast<const f32x4*>(bdw + hid + c);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const T* up = u + (size_t)((long)pix + (long)dy * W + dx) * ldu;
                const int tap = (dy + 1) * 3 + (dx + 1);
                a += Vec4<T>::load(up + c) * *reinterpret_cast<const f32x4*>(wdw + (size_t)tap * 2 * hid + c);
                gt += Vec4<T>::load(up + hid + c) * *reinterpret_cast<const f32x4*>(wdw + (size_t)tap * 2 * hid + hid + c);
            }
        }
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = a[r] * (gt[r] / (1.0f + expf(-gt[r])));
        Vec4<T>::store(out + pix * ldo + c, o);
    }
}

// ---------------------------------------------------------------------------------------------
// Spatial-gate step of HATX's SGFN (hatx_arch.py:165-177): depthwise 3x3 + bias on the FIRST `half` channels of u, the
// second half gates it (SiLU) and is passed on unchanged: out = [dw(a) * silu(b) | b].  4 channels per thread.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sgfn_gate_kernel(const T* __restrict__ u, const float* __restrict__ wdw,
                                                        const float* __restrict__ bdw, T* __restrict__ out, int B, int H, int W,
                                                        int half, int ldu, int ldo) {
    const int gpp = half / 4;
    const size_t total = (size_t)B * H * W * gpp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / gpp;
        const int c = (int)(i - pix * gpp) * 4;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        f32x4 a = *reinterpret_cast<const f32x4*>(bdw + c);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const int tap = (dy + 1) * 3 + (dx + 1);
                a += Vec4<T>::load(u + (size_t)((long)pix + (long)dy * W + dx) * ldu + c) * *reinterpret_cast<const f32x4*>(wdw + (size_t)tap * half + c);
            }
        }
        const f32x4 gt = Vec4<T>::load(u + pix * ldu + half + c);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = a[r] * (gt[r] / (1.0f + expf(-gt[r])));
        Vec4<T>::store(out + pix * ldo + c, o);
        Vec4<T>::store(out + pix * ldo + half + c, gt);
    }
}

}  // namespace

extern "C" int hat_layernorm_blocks(void) { return LN_BLOCKS; }

extern "C" int hat_layernorm(const float* x, void* y, const float* gamma, const float* beta, float* gap_partial,
                             int32_t B, int64_t npix, int32_t C, int32_t ldy, int32_t out_f32, int32_t gap_c,
                             int32_t dtype, void* stream) {
    if (!x || !y || !gamma || !beta || B < 1 || npix < 1 || C < 4 || C % 4 || C > 256 || ldy < C || ldy % 4) return HAT_EINVAL;
    if (gap_c < 0 || gap_c > 32 || gap_c % 4 || (gap_c > 0 && !gap_partial)) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* gp = gap_c > 0 ? gap_partial : nullptr;
    if (out_f32 || dtype == HAT_F32) return launch_ln<float>(x, y, gamma, beta, gp, B, npix, C, ldy, gap_c, s);
    if (dtype == HAT_BF16) return launch_ln<bf16_t>(x, y, gamma, beta, gp, B, npix, C, ldy, gap_c, s);
    return HAT_EINVAL;
}

extern "C" int hat_esc_weights(const float* gap_partial, int32_t nblk, int64_t npix, const float* w1, const float* b1,
                               const float* w2, const float* b2, const float* plk_packed, void* w_out, int32_t B,
                               int32_t pdim, int32_t ksize, int32_t Kpad, int32_t dtype, void* stream) {
    if (!gap_partial || !w1 || !b1 || !w2 || !b2 || !plk_packed || !w_out) return HAT_EINVAL;
    if (pdim < 2 || pdim > 32 || pdim % 4 || ksize < 3 || (ksize & 1) == 0 || B < 1 || nblk < 1) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(pdim > 16 ? 32 : 16, B), block(1024);
    const float inv = 1.0f / (float)npix;
    if (dtype == HAT_BF16)
        HAT_LAUNCH(esc_weights_kernel<bf16_t>, grid, block, 0, s, gap_partial, nblk, inv, w1, b1, w2, b2, plk_packed,
                           reinterpret_cast<bf16_t*>(w_out), pdim, ksize, Kpad);
    else if (dtype == HAT_F32)
        HAT_LAUNCH(esc_weights_kernel<float>, grid, block, 0, s, gap_partial, nblk, inv, w1, b1, w2, b2, plk_packed,
                           reinterpret_cast<float*>(w_out), pdim, ksize, Kpad);
    else
        return HAT_EINVAL;
    return hat_check_launch();
}

extern "C" int hat_eca_scale(const float* colsum, int32_t tiles, int32_t ldc, int64_t npix, const float* wk, int32_t k,
                             float conv_scale, float* tmp, float* scale, int32_t B, int32_t C, void* stream) {
    if (!colsum || !wk || !tmp || !scale || tiles < 1 || ldc < C || ldc > 256 || C < 1 || k < 1 || (k & 1) == 0 || B < 1) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HAT_LAUNCH(eca_reduce_kernel, dim3(32, B), dim3(256), 0, s, colsum, tiles, ldc, tmp);
    int rc = hat_check_launch();
    if (rc) return rc;
    HAT_LAUNCH(eca_scale_kernel, dim3(B), dim3(256), 0, s, tmp, ldc, 1.0f / (float)npix, wk, k, conv_scale, scale, C);
    return hat_check_launch();
}

extern "C" int hat_cab_fold(const HatCabFoldDesc* dp, void* stream) {
    if (!dp) return HAT_EINVAL;
    const HatCabFoldDesc& d = *dp;
    if (!d.w2 || !d.b2 || !d.wk || !d.bias_in || !d.scale || !d.wf || !d.bias_out) return HAT_EINVAL;
    if (!d.stats && (!d.c1 || !d.c1_colsum || !d.tmp)) return HAT_EINVAL;
    if (d.B < 1 || d.H < 2 || d.W < 2 || d.C < 1 || d.C > 256 || d.mid < 1 || d.mid > 8 || d.ld1 != 8 || d.tiles < 1 || d.ldcs < 8 ||
        d.ldcs > 256 || d.k < 1 || (d.k & 1) == 0 || d.ld_scale < d.C)
        return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d.dtype != HAT_BF16 && d.dtype != HAT_F32) return HAT_EINVAL;
    if (d.tiles <= 2048 || d.stats) {
        if (d.dtype == HAT_BF16) HAT_LAUNCH((cab_fold_kernel<bf16_t, true>), dim3(d.B), dim3(256), 0, s, d);
        else HAT_LAUNCH((cab_fold_kernel<float, true>), dim3(d.B), dim3(256), 0, s, d);
        return hat_check_launch();
    }
    HAT_LAUNCH(eca_reduce_kernel, dim3(32, d.B), dim3(256), 0, s, d.c1_colsum, d.tiles, d.ldcs, d.tmp);
    int rc = hat_check_launch();
    if (rc) return rc;
    if (d.dtype == HAT_BF16) HAT_LAUNCH((cab_fold_kernel<bf16_t, false>), dim3(d.B), dim3(256), 0, s, d);
    else HAT_LAUNCH((cab_fold_kernel<float, false>), dim3(d.B), dim3(256), 0, s, d);
    return hat_check_launch();
}

extern "C" int hat_dwconv_gate(const void* u, const float* wdw, const float* bdw, void* out, int32_t B, int32_t H,
                               int32_t W, int32_t hid, int32_t ldu, int32_t ldo, int32_t dtype, void* stream) {
    if (!u || !wdw || !bdw || !out || B < 1 || H < 1 || W < 1 || hid < 4 || hid % 4 || ldu < 2 * hid || ldu % 4 || ldo < hid || ldo % 4) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)B * H * W * (hid / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == HAT_BF16)
        HAT_LAUNCH(dwgate_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(u), wdw, bdw,
                           reinterpret_cast<bf16_t*>(out), B, H, W, hid, ldu, ldo);
    else if (dtype == HAT_F32)
        HAT_LAUNCH(dwgate_kernel<float>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float*>(u), wdw, bdw,
                           reinterpret_cast<float*>(out), B, H, W, hid, ldu, ldo);
    else
        return HAT_EINVAL;
    return hat_check_launch();
}

namespace {
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ c, float* out,
                                                      long n4, long n, long c_bstride) {
    const int b = blockIdx.y;
    const f32x4* av = reinterpret_cast<const f32x4*>(a + (size_t)b * n);
    const f32x4* cv = reinterpret_cast<const f32x4*>(c + (size_t)b * c_bstride);
    f32x4* ov = reinterpret_cast<f32x4*>(out + (size_t)b * n);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) ov[i] = av[i] + cv[i];
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// hat_rect_sum: per-channel sums over a pixel rectangle of a channel-last map, deterministic (fixed reduction order):
// every workgroup sums a contiguous share of the rectangle's pixels, the LAST one to finish adds the shares in order.
// ---------------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(256) void rect_sum_kernel(const T* __restrict__ x, int ld, int C, int W, int r0, int c0, int rw, long npx,
                                                       long bstride, float* __restrict__ out, int ldo, float* __restrict__ tmp,
                                                       unsigned* __restrict__ counter) {
    __shared__ float red[256 * 4];
    __shared__ unsigned last;
    const int tid = threadIdx.x, b = blockIdx.y, nwg = gridDim.x;
    const int cpp = (C + 3) / 4;              // threads per pixel (4 channels each)
    const int ppi = 256 / cpp;                // pixels per pass
    const int slot = tid / cpp, cg = tid - slot * cpp;
    const long share = (npx + nwg - 1) / nwg, p0 = (long)blockIdx.x * share, p1 = p0 + share < npx ? p0 + share : npx;
    const T* xb = x + (size_t)b * bstride;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (slot < ppi) {
        for (long p = p0 + slot; p < p1; p += ppi) {
            const long r = p / rw, c = p - r * rw;
            const T* src = xb + ((size_t)(r0 + r) * W + (c0 + c)) * ld + 4 * cg;
            a += Vec4<T>::load(src);       // (rows are padded to a multiple of 4 channels: pad channels are zero)
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[tid * 4 + k] = slot < ppi ? a[k] : 0.f;
    __syncthreads();
    float* mine = tmp + ((size_t)b * nwg + blockIdx.x) * 256;
    if (tid < 4 * cpp) {                       // channel tid of this workgroup's share: the pixel slots in order
        const int g2 = tid >> 2, k = tid & 3;
        float s = 0.f;
        for (int sl = 0; sl < ppi; ++sl) s += red[(sl * cpp + g2) * 4 + k];
        mine[tid] = s;
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(counter + b, 1u) == (unsigned)(nwg - 1);
    __syncthreads();
    if (last) {
        __threadfence();
        if (tid < 4 * cpp) {
            double s = 0.0;
            for (int g2 = 0; g2 < nwg; ++g2) s += (double)__builtin_nontemporal_load(tmp + ((size_t)b * nwg + g2) * 256 + tid);
            if (tid < C) out[(size_t)b * ldo + tid] = (float)s;
        }
        if (tid == 0) counter[b] = 0u;         // self-resetting: the next launch on this stream finds it zero
    }
}
}  // namespace

extern "C" int hat_rect_sum(const void* x, int32_t dtype, int32_t ld, int32_t C, int32_t W, int32_t r0, int32_t r1, int32_t c0,
                            int32_t c1, int64_t bstride, int32_t B, float* out, int32_t ldo, float* tmp, uint32_t* counter,
                            void* stream) {
    if (!x || !out || !tmp || !counter || B < 1 || C < 1 || C > 256 || ld < C || ld % 4 || W < 1 || r0 < 0 || r1 <= r0 || c0 < 0 ||
        c1 <= c0 || c1 > W || ldo < 1)
        return HAT_EINVAL;
    const long npx = (long)(r1 - r0) * (c1 - c0);
    const int nwg = (int)(npx >= 64 * 1024 ? 64 : (npx + 1023) / 1024);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == HAT_BF16)
        HAT_LAUNCH(rect_sum_kernel<bf16_t>, dim3(nwg, B), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(x), ld, C, W, r0, c0, c1 - c0, npx,
                   (long)bstride, out, ldo, tmp, counter);
    else if (dtype == HAT_F32)
        HAT_LAUNCH(rect_sum_kernel<float>, dim3(nwg, B), dim3(256), 0, s, reinterpret_cast<const float*>(x), ld, C, W, r0, c0, c1 - c0, npx,
                   (long)bstride, out, ldo, tmp, counter);
    else
        return HAT_EINVAL;
    return hat_check_launch();
}

extern "C" int hat_add_f32(const float* a, const float* c, float* out, int32_t B, int64_t n, int64_t c_bstride, void* stream) {
    if (!a || !c || !out || B < 1 || n < 4 || n % 4 || c_bstride < 0 || c_bstride % 4) return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    HAT_LAUNCH(add_f32_kernel, dim3(blocks, B), dim3(256), 0, s, a, c, out, n4, (long)n, (long)c_bstride);
    return hat_check_launch();
}

extern "C" int hat_sgfn_gate(const void* u, const float* wdw, const float* bdw, void* out, int32_t B, int32_t H, int32_t W,
                             int32_t half, int32_t ldu, int32_t ldo, int32_t dtype, void* stream) {
    if (!u || !wdw || !bdw || !out || u == out || B < 1 || H < 1 || W < 1 || half < 4 || half % 4 || ldu < 2 * half || ldu % 4 ||
        ldo < 2 * half || ldo % 4)
        return HAT_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)B * H * W * (half / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == HAT_BF16)
        HAT_LAUNCH(sgfn_gate_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(u), wdw, bdw,
                   reinterpret_cast<bf16_t*>(out), B, H, W, half, ldu, ldo);
    else if (dtype == HAT_F32)
        HAT_LAUNCH(sgfn_gate_kernel<float>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float*>(u), wdw, bdw,
                   reinterpret_cast<float*>(out), B, H, W, half, ldu, ldo);
    else
        return HAT_EINVAL;
    return hat_check_launch();
}

extern "C" int hat_abi_version(void) { return HAT_ABI_VERSION; }
extern "C" const char* hat_target_arch(void) { return "gfx950"; }
