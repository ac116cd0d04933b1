/*
 * hat_mi355x.h — C ABI of libhat_mi355x.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * forward pass of the HAT super-resolution network of imjaegyun/super_resolution.
 *
 * The reference implements this path in pure PyTorch (no FFI exists upstream); each entry point
 * below replaces a group of ATen calls inside `HAT.forward`.  File:line references are relative
 * to /root/reference/HAT.  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - every function returns 0 on success, a negative HAT_E* code on a bad argument or the
 *     (positive) hipError_t of a failed launch; the Python host turns non-zero into RuntimeError;
 *   - all pointers are DEVICE pointers owned by the caller; nothing is allocated, freed or
 *     synchronised inside; kernels are enqueued on `stream` (a hipStream_t passed as void*);
 *   - activations are channel-last ("tokens": (B, H, W, C) == the reference's (B, N, C) layout,
 *     hat_arch.py:571-575); `dtype` selects the storage/MFMA operand type of T-typed buffers:
 *     HAT_F32 (exact fp32 MFMA, parity path) or HAT_BF16 (bf16 MFMA operands, fp32 accumulate);
 *   - the residual stream, LayerNorm statistics, softmax and all accumulations are fp32.
 */
#ifndef HAT_MI355X_H
#define HAT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever a descriptor layout, a packed-weight layout or the entry-point table changes (2: round 3 — HatConvDesc /
 * HatFfnDesc grew in round 2 without a bump; hat_hab_tail3 and its packing; plan files carry the version) */
#define HAT_ABI_VERSION 2

enum { HAT_F32 = 0, HAT_BF16 = 1 };
enum { HAT_EINVAL = -1, HAT_ELDS = -2, HAT_EUNSUPPORTED = -3 };

/* activation codes */
enum { HAT_ACT_NONE = 0, HAT_ACT_GELU = 1 /* exact erf, nn.GELU() */, HAT_ACT_LRELU = 2 /* slope 0.01 */ };
/* conv input modes */
enum { HAT_X_NHWC_T = 0, HAT_X_NHWC_F32 = 1, HAT_X_NCHW_F32_MEAN = 2 };
/* conv output modes */
enum { HAT_O_NHWC_T = 0, HAT_O_NHWC_F32 = 1, HAT_O_PIXSHUF_T = 2, HAT_O_NCHW_F32 = 3 };

/*
 * Implicit-GEMM convolution / pointwise linear on channel-last data with a fused epilogue:
 *     out[p, n] = epi( sum_{tap, ci} X[p + tap, ci] * Wp[n, tap * Cin_p + ci] + bias[n] )
 * zero padding ksize/2, stride 1.  Replaces (ksize == 1) nn.Linear / 1x1 nn.Conv2d:
 * hat_arch.py:111,117 (FFN fc1/fc2), :347,350,391 (OCAB q/kv/proj), :309-313 (OCAB MLP),
 * esc_arch.py:144 (ESC aggr); (ksize == 3) hat_arch.py:84,86 (CAB), :544 (RHAG conv), :673,747,
 * 754,757 (head/tail), :598,601 (Upsample convs, with nn.PixelShuffle :599,602 folded into the
 * store); (ksize == 13) esc_arch.py:122-123 (large-kernel conv + dynamic depthwise conv, the
 * latter folded into per-sample weights by hat_esc_weights).
 *
 * Wp is packed [n_slices * nt * 16][Kpad] in T with K index = tap * Cin_p + ci,
 * Cin_p = round_up(Cin, 8), Kpad = round_up(ksize^2 * Cin_p, KC), KC = 64 (bf16) / 32 (f32), x3 when nt == 1;
 * zero padded.  epi: v = acc + bias[n]; v = act(v); v += r1[p, n] (fp32, optional);
 * v += r2scale[n] * r2[p, n] (T, optional); store by out_mode; optional per-tile column sums
 * of v (for the ECA global average pool, hat_arch.py:73).
 */
typedef struct HatConvDesc {
    const void* x;        /* main input */
    const void* x0;       /* optional: channels [0, c_split) are read from x0 (NHWC T, ldx0) */
    const void* w;        /* packed weights (T) */
    const float* bias;    /* [n_slices*nt*16] fp32, zero padded */
    void* out;
    const float* r1;      /* optional fp32 residual, NHWC stride ldr1 */
    const void* r2;       /* optional T residual, NHWC stride ldr2, scaled per channel by r2scale */
    const float* r2scale; /* [B][r2scale_bstride] fp32 */
    float* colsum;        /* optional [B][tiles][n_slices*nt*16] per-tile column sums */
    int32_t B, H, W;
    int32_t Cin, ldx, x_mode;
    int32_t c_split, ldx0;
    int32_t ksize, Kpad;
    int32_t nt, n_slices;
    int64_t w_bstride;    /* elements between per-sample weight sets (0 = shared) */
    int32_t n_store;      /* channels actually stored (multiple of 4 unless out_mode NCHW) */
    int32_t ldo, out_mode, act;
    int32_t ldr1, ldr2, r2scale_bstride;
    int32_t ps_r;         /* pixel-shuffle factor for HAT_O_PIXSHUF_T */
    float in_scale, out_scale;
    float mean[4];        /* HAT_X_NCHW_F32_MEAN: x = (x - mean[c]) * in_scale;  HAT_O_NCHW_F32: v*out_scale + mean[n] */
    int32_t dtype;
    /* hat_linear only (n_slices == 1, n_store == channels): also emit LayerNorm(v) (eps 1e-5) of the finished pixel,
     * i.e. the nn.LayerNorm that consumes this layer's output (hat_arch.py:214 after :236; :306 after :391), as T
     * rows of ld_ln elements.  ln_ones != 0: element [n_store] of every row is 1.0 and the rest up to ld_ln is 0 —
     * the image hat_ffn's m_in expects.  ln_out == NULL: off. */
    int32_t ld_ln, ln_ones;
    const float* ln_g;
    const float* ln_b;
    void* ln_out;
    /* hat_conv with ln_out (n_slices == 1, n_store == 16 nt, NHWC output, ln_ones == 0): the same fused LayerNorm in
     * the conv's epilogue — the norm1 of the next residual group's first block, or HAT.norm, after the group conv
     * (hat_arch.py:556 then :214 / :844) — with what the ESC path of that block needs from it:
     * gap_out (B, hat_conv_tiles, 16) per-tile sums of the first gap_c (<= 16, % 4) LayerNorm channels over the tile's
     * pixels (the partial sums hat_esc_weights reduces), n16_out (B,H,W,16) T a compact copy of LayerNorm channels
     * [0,16).  Both optional (NULL). */
    float* gap_out;
    void* n16_out;
    int32_t gap_c;
    int32_t reserved0;
} HatConvDesc;

/* Debug query (not thread safe): how many workgroups of the kernel hat_conv would launch for `d` fit one CU. */
int hat_conv_occupancy(const HatConvDesc* d, int32_t* wgs_per_cu);
/* Number of spatial tiles hat_conv uses for (H, W, Cin, ksize, nt, dtype): the leading dimension of `colsum`. */
int hat_conv_tiles(const HatConvDesc* d, int32_t* tiles_out);
/* The launch plan hat_conv picks for `d`: waves per workgroup, pixel rows per wave, spatial tiles and dynamic
 * LDS bytes.  The kernel instantiation is conv_kernel<T, waves, rows_per_wave, nt> (used to label profiles). */
int hat_conv_plan(const HatConvDesc* d, int32_t* waves, int32_t* rows_per_wave, int32_t* tiles, int64_t* lds_bytes);
int hat_conv(const HatConvDesc* d, void* stream);

/*
 * Pointwise linear layer (ksize == 1, channel-last T input, HAT_O_NHWC_T / HAT_O_NHWC_F32 output, same epilogue
 * as hat_conv without column sums) as a weight-stationary, barrier-free streaming GEMM: the layers it serves
 * (hat_arch.py:309-313,347,350,391; esc_arch.py:144) are HBM-bound.  Same descriptor as hat_conv, but `w` is
 * FRAGMENT packed: [n_slices][nt][ceil(Cin/32)][64 lanes][8] with element (lane l, j) =
 * W[slice*nt*16 + t*16 + (l & 15)][32*ks + 8*(l >> 4) + j] (zero beyond Cin / n); Kpad is ignored.
 * Returns HAT_EUNSUPPORTED for (nt, Cin) pairs that are not instantiated: call hat_conv instead.
 */
int hat_linear(const HatConvDesc* d, void* stream);

/*
 * 3x3 convolution (zero padding 1) whose whole weight slice fits in LDS — the two CAB convolutions
 * (hat_arch.py:84,86) — in the same free-running structure: neighbour pixels are gathered straight from global
 * memory.  Same descriptor as hat_conv (ksize 3, n_slices 1, channel-last T input, no residuals), weights
 * FRAGMENT packed like hat_linear with K index = tap * Cin_p + ci.  Optional colsum is [B][groups][nt*16]
 * (one row per workgroup), groups from hat_conv3x3_small_groups().  HAT_EUNSUPPORTED: use hat_conv.
 */
int hat_conv3x3_small_groups(const HatConvDesc* d, int32_t* groups_out);
int hat_conv3x3_small(const HatConvDesc* d, void* stream);

/*
 * CAB expand conv + ECA folded into the ESC aggregation (hat_arch.py:84-90, 66-78, 233-236 with esc_arch.py:123) for
 * models whose CAB squeeze width is <= 8 channels (HAT-S: 6).  The reference computes
 *     c2 = conv3x3(c1) + b2;  e = sigmoid(conv1d_k(mean_pixels(c2)));  x = t + aggr(y) + conv_scale * e * c2.
 * mean_pixels(c2) is linear in c1, so it follows from the column sums of c1 and its border rows / columns
 * (zero padding: tap (dy,dx) misses one border row and/or column), BEFORE c2 exists:
 *   hat_cab_fold:  scale = conv_scale * e,  wf = fragment-packed (scale * W2) with K = tap*8 + ci,
 *                  bias_out = bias_in + scale * b2            (tiny per-sample kernel)
 *   hat_aggr_cab:  out = r1 + W_aggr . x + wf . im2col3x3(c1) + bias_out   — c2 never exists in memory.
 */
typedef struct HatCabFoldDesc {
    const void* c1;          /* (B,H,W,ld1) T: GELU(conv3x3(n)), `mid` channels, pad channels zero, ld1 == 8 */
    const float* c1_colsum;  /* [B][tiles][ldcs]: per-tile column sums of c1 as hat_conv's colsum emits them */
    const float* w2;         /* [C][mid][3][3] fp32 */
    const float* b2;         /* [C] */
    const float* wk;         /* ECA conv1d weights [k] */
    const float* bias_in;    /* [C]: the aggregation bias */
    float* scale;            /* out [B][ld_scale] */
    void* wf;                /* out [B][nt][3][64][8] T, nt = ceil(C/16) */
    float* bias_out;         /* out [B][nt*16] */
    float* tmp;              /* scratch [B][32][ldcs] */
    int32_t B, H, W, C, mid, ld1, tiles, ldcs, k, ld_scale, dtype;
    float conv_scale;
    /* optional [B][72] fp32: the sums this kernel otherwise takes from c1 / c1_colsum, supplied by the caller — a frame that is
     * sharded into row bands adds up its bands' hat_rect_sum results: [0,8) sums of c1's channels over the whole
     * frame, [8,40) over its first row, last row, first column, last column, [40,72) its four corner pixels (top-left,
     * top-right, bottom-left, bottom-right).  H, W are then the FULL frame's; c1, c1_colsum, tmp are not read. */
    const float* stats;
    /* optional: W2 once more, in wf's own order, fp32 [nt][3][64][8] with element (t, ks, lane, j) = W2[16 t + (lane & 15)][ci = j]
     * [tap = 4 ks + (lane >> 4)] (zero where co >= C, tap >= 9 or ci >= mid).  The kernel then reads the weights it scales with
     * unit stride (it is ONE workgroup on the critical chain of a block: the gather out of the [C][mid][3][3] layout was half
     * of its time).  NULL: gather from w2. */
    const float* w2f;
} HatCabFoldDesc;
int hat_cab_fold(const HatCabFoldDesc* d, void* stream);

typedef struct HatAggrCabDesc {
    HatConvDesc lin;         /* the aggregation as for hat_linear: x (+x0 / c_split), w (fragment packed, nt = 9, Cin = 144),
                                out (fp32, HAT_O_NHWC_F32), r1 (fp32); bias, r2*, ln_* are ignored */
    const void* c1;          /* (B,H,W,8) T */
    const void* wf;          /* hat_cab_fold's wf */
    const float* bias_b;     /* hat_cab_fold's bias_out, [B][nt*16] */
} HatAggrCabDesc;
int hat_aggr_cab(const HatAggrCabDesc* d, void* stream);

/*
 * LayerNorm over the channel dimension (eps 1e-5, affine), fp32 in -> T or fp32 out
 * (nn.LayerNorm at hat_arch.py:209,214,291,306,743 and PatchEmbed.norm :573-574).
 * Optionally emits per-block partial sums of the first `gap_c` (<= 32, % 4) output channels
 * (the AdaptiveAvgPool2d(1) feeding the ESC dynamic kernel, esc_arch.py:96,121):
 * gap_partial[b][blk][gs], blk < hat_layernorm_blocks(), gs = 16 floats per block, 32 when gap_c > 16.
 */
int hat_layernorm_blocks(void);
int hat_layernorm(const float* x, void* y, const float* gamma, const float* beta, float* gap_partial,
                  int32_t B, int64_t npix, int32_t C, int32_t ldy, int32_t out_f32, int32_t gap_c,
                  int32_t dtype, void* stream);

/*
 * out[b][i] = a[b][i] + c[b * c_bstride + i], i < n (fp32, n % 4 == 0; out may alias a).  The plain adds of the
 * reference that no producing kernel can absorb: the absolute position embedding (hat_arch.py:837-838,
 * c_bstride = 0: broadcast over the batch) and the residuals around nn.Identity when
 * resi_connection == 'identity' (:545-546 with :556, :748 with :854).
 */
int hat_add_f32(const float* a, const float* c, float* out, int32_t B, int64_t n, int64_t c_bstride, void* stream);

/*
 * ESC per-sample conv weights (esc_arch.py:95-100,121-123): p = mean(gap partials);
 * dk = W2 * gelu(W1 * p + b1) + b2 (pdim*9 values); Wp[b][co][tap*Cin_p + ci] =
 * T( plk_packed[co][tap*Cin_p+ci] + (co == ci && tap in central 3x3 ? dk[co*9 + ..] : 0) ).
 * pdim <= 32 (% 4): the GAP partial blocks are 16 floats (32 for pdim > 16, hat_layernorm's layout for gap_c = pdim) and
 * plk_packed / w_out hold 16 rows per sample (32 for pdim > 16: two 16-row slices for hat_conv with nt = 1).
 */
int hat_esc_weights(const float* gap_partial, int32_t nblk, int64_t npix, const float* w1, const float* b1,
                    const float* w2, const float* b2, const float* plk_packed, void* w_out, int32_t B,
                    int32_t pdim, int32_t ksize, int32_t Kpad, int32_t dtype, void* stream);

/*
 * The ESC large-kernel conv as a dedicated kernel (esc_arch.py:121-123), bf16, pdim 16, 13x13: y16 = conv2d(x[..., :16], Wp[b])
 * with zero padding 6, no bias.  x: (B,H,W,ldx) T (the first 16 channels are used); wp: hat_esc_weights' output
 * [B][16][Kpad] (K = tap * 16 + ci, tap = ty * 13 + tx); y16: (B,H,W,16) T.  All weights and the haloed tile stay in LDS.
 * HAT_EUNSUPPORTED for other dtypes: use hat_conv (ksize 13).
 */
int hat_esc_conv13(const void* x, int32_t ldx, const void* wp, int32_t Kpad, void* y16, int32_t B, int32_t H, int32_t W,
                   int32_t dtype, void* stream);

/*
 * ECA channel attention scale (hat_arch.py:73-77) times conv_scale (:236):
 * scale[b][c] = conv_scale * sigmoid( conv1d_k(mean_pixels(c2))[c] ), from hat_conv's colsum.
 * `tmp` is fp32 scratch of B*32*ldc floats.
 */
int hat_eca_scale(const float* colsum, int32_t tiles, int32_t ldc, int64_t npix, const float* wk, int32_t k,
                  float conv_scale, float* tmp, float* scale, int32_t B, int32_t C, void* stream);

/*
 * Depthwise 3x3 (+bias, zero pad) on 2*hid channels, chunk(2), a * SiLU(g)   (hat_arch.py:112-116).
 * u: (B,H,W,ldu) T with 2*hid channels; wdw packed [9][2*hid] fp32; out (B,H,W,ldo) T with hid channels.
 */
int hat_dwconv_gate(const void* u, const float* wdw, const float* bdw, void* out, int32_t B, int32_t H,
                    int32_t W, int32_t hid, int32_t ldu, int32_t ldo, int32_t dtype, void* stream);

/*
 * The OCAB's MLP with its residual in one launch (hat_arch.py:309-313, :391): out = r1 + fc2(GELU(fc1(x))), exact-erf GELU
 * (the bf16 path's approximation of hat_linear), for embed_dim 144, hidden 288, bf16 (HAT_EUNSUPPORTED otherwise: run the two
 * hat_linear launches).  The hidden tensor stays in registers.
 *   x    (B,H,W,ldx) T: LayerNorm2 output;  r1 (B,H,W,ldr1) fp32: the residual stream;
 *   out  (B,H,W,ldo): fp32 when out_f32 (may alias r1), else T rows (16-byte aligned, ldo % 8 == 0);
 *   w1f  fc1 as MFMA A fragments [18 n-tiles][4 k-steps][64 lanes][8] bf16 (rows = hidden unit 16 nt + (lane & 15), k = 32 ks +
 *        8 (lane >> 4) + j), followed by the 16-deep tail [18][64 lanes][4] (k = 128 + 4 (lane >> 4) + j);  b1 [288] fp32;
 *   w2f  fc2 as A fragments [9 n-tiles][9 k-steps][64 lanes][8] bf16, rows = output channel, k-slot (g = lane >> 4, j) of k-step kk =
 *        hidden unit 32 kk + 4 g + j (j < 4) or 32 kk + 16 + 4 g + j - 4 (j >= 4);  b2 [144] fp32.
 */
typedef struct HatMlpDesc {
    const void* x;
    const void* w1f;
    const float* b1;
    const void* w2f;
    const float* b2;
    const float* r1;
    void* out;
    int32_t B, H, W, C, hidden;
    int32_t ldx, ldr1, ldo;
    int32_t out_f32, dtype;
} HatMlpDesc;
int hat_ocab_mlp(const HatMlpDesc* d, void* stream);
/*
 * The OCAB's q and kv projections (hat_arch.py:347, :350) in one launch for embed_dim 144, bf16: out rows of 432 channels
 * [q | k | v] (T, 16-byte aligned, ldo % 8 == 0) = W x + b.  Uses HatMlpDesc: x, ldx as above; w1f = the stacked weight
 * [q_proj * head_dim^-0.5 ; kv_proj] (432 x 144) in hat_ocab_mlp's fc1 fragment layout ([27][4][64][8] + [27][64][4]); b1 [432]
 * (q part scaled likewise); hidden = 432; w2f, b2, r1, ldr1, out_f32 unused.  hat_ocab_attention then takes q = out,
 * kv = out + 144 elements, ldq = ldkv = ldo.
 */
int hat_ocab_qkv(const HatMlpDesc* d, void* stream);

/*
 * HATX's OCAB options (hatx_arch.py:421-449), for the key windows the generic attention kernel is built for (wse = 24, 12 and
 * the odd 25, 13):
 *   hat_ocab_keybias      kb[b][window][key] (fp32, rows of round_up(wse*wse, 16) floats per window: whole key tiles) = tanh(sal at the key's pixel) for a kept key — 0
 *                         without a focus head (sal == NULL: the score is then ||k||_2 over the C key channels of kv) —
 *                         and -inf for a pruned one; kept = the k_keep keys of a window with the largest score, ties by
 *                         the lower key index (the reference leaves tie order to torch.topk); zero-padded keys outside
 *                         the image score tanh(0) = 0 / norm 0.  sal: (B,H,W,ldsal) T, channel 0 = the saliency map.
 *   hat_ocab_attention_kb hat_ocab_attention with `kb` applied before the relative-position bias: logit + kb, or -1e4 in
 *                         place of the logit where kb = -inf; `pad` = ceil((wse - ws) / 2), HATX's unfold padding.
 * ldsal < 0 (dtype HAT_BF16 only): `sal` is an FP32 map of row stride -ldsal — the saliency head's last conv writes its fp32
 * accumulators (HAT_O_NHWC_F32) so that the keys are ranked on unrounded scores; kv stays bf16.
 */
int hat_ocab_keybias(const void* sal, int32_t ldsal, const void* kv, int32_t ldkv, float* kb, int32_t B, int32_t H, int32_t W,
                     int32_t C, int32_t ws, int32_t wse, int32_t pad, int32_t k_keep, int32_t dtype, void* stream);
int hat_ocab_attention_kb(const void* q, const void* kv, const float* bias_rot, const float* kb, void* out, int32_t B, int32_t H,
                          int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t wse, int32_t pad, int32_t ldq, int32_t ldkv,
                          int32_t ldo, int32_t dtype, void* stream);

/*
 * Spatial-gate step of HATX's SGFN (hatx_arch.py:165-177) between its fc1 and fc2: depthwise 3x3 (+bias, zero pad) on the
 * FIRST `half` channels of u, gated by SiLU of the second half, which is also passed on:
 *     out[..., :half] = dw3x3(u[..., :half]) * SiLU(u[..., half:]),   out[..., half:] = u[..., half:].
 * u: (B,H,W,ldu) T with 2*half channels; wdw packed [9][half] fp32, bdw [half]; out (B,H,W,ldo) T, must not alias u.
 */
int hat_sgfn_gate(const void* u, const float* wdw, const float* bdw, void* out, int32_t B, int32_t H, int32_t W,
                  int32_t half, int32_t ldu, int32_t ldo, int32_t dtype, void* stream);

/*
 * Overlapping cross-attention core (hat_arch.py:353-388): per 'ws x ws' query window and head,
 * softmax(q k^T + RPB) v over the 'wse x wse' key window (stride ws, zero padded, NOT masked).
 * q: (B,H,W,ldq) T (already multiplied by head_dim^-0.5), kv: (B,H,W,ldkv) T with k at channel 0
 * and v at channel C; bias_rot: [heads][(ws+wse-1)^2] fp32, the relative-position-bias table
 * rotated so that index (kh-qh+ws-1)*(ws+wse-1) + (kw-qw+ws-1) is non-negative (the reference's
 * negative-index wraparound, hat_arch.py:378, is applied when the table is packed).
 * out: (B,H,W,ldo) T in window_reverse order (:387-388).
 * Key windows: wse = 24 and 12 (window 16 / 8, overlap 0.5; padding (wse - ws) / 2 on every side) and the odd 25 and 13
 * (HATX: overlap 0.6 / 0.7, padding ceil((wse - ws) / 2), hatx_arch.py:303-305); HAT_EUNSUPPORTED for others, HAT_ELDS when
 * K and V of one key window do not fit the LDS (fp32, wse 25, head_dim 30).
 */
int hat_ocab_attention(const void* q, const void* kv, const float* bias_rot, void* out, int32_t B, int32_t H,
                       int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t wse, int32_t ldq,
                       int32_t ldkv, int32_t ldo, int32_t dtype, void* stream);
/*
 * hat_ocab_attention for the tuned kernel of the embed_dim-144 models (bf16, 16 x 16 windows, 24 x 24 key windows, head_dim 24;
 * HAT_EUNSUPPORTED otherwise), with q ALREADY multiplied by head_dim^-1/2 * log2(e) — the caller folds the factor into the q
 * projection's weights before they are rounded, so the scores are in log2 units at no extra rounding.  The kernel then carries
 * the softmax's offset in a spare k-slot of the QK^T MFMA (K rows hold 1.0, the query fragment -offset): p = exp2(score) with no
 * per-score FMA and no rescale of O.  The offset is the row maximum of the first 96-key chunk; the rest of the key window runs
 * without range checks, and a query tile whose softmax denominator comes out non-finite or above 1e30 (a later score more than
 * ~2^100 above the first chunk's maximum) is computed again with a check and a re-centring step per chunk.  Same result as
 * hat_ocab_attention up to rounding (hat_arch.py:375-384).
 */
int hat_ocab_attention_log2(const void* q, const void* kv, const float* bias_rot, void* out, int32_t B, int32_t H, int32_t W,
                            int32_t C, int32_t heads, int32_t ws, int32_t wse, int32_t ldq, int32_t ldkv, int32_t ldo,
                            int32_t dtype, void* stream);

/*
 * CAB squeeze conv: GELU_erf(conv3x3(x, C -> mid <= 8 channels, zero pad) + bias)   (hat_arch.py:84-85, cab.0 + GELU)
 * for bf16 rows without LDS operand traffic: a wave sweeps a strip of 14 output columns (16 loaded ones: a halo column on each
 * side) top to bottom, every input row's activations come straight from global memory ONCE (the fragments of the two
 * horizontally shifted taps are lane shifts of the loaded one), the weights stay in registers, and the three taps of a
 * kernel column are routed to three rolling output-row accumulators by the choice of MFMA C operand (csrc/hat_cabsq.hip).
 * x: (B,H,W,ldx) bf16, 128 < C <= 160, C % 8 == 0; W % 16 == 0.
 * wpk: 6 tiles x 5 k-steps of MFMA A fragments [tile][kstep][64 lanes][8] bf16, tile = 2*kx + j:
 *      j = 0: rows 0-7 = w[ch][.][ky=0][kx], rows 8-15 = w[ch][.][ky=1][kx];  j = 1: rows 0-7 = w[ch][.][ky=2][kx], rest 0
 *      (fragment element [lane][e] = row lane%16, input channel 32*kstep + 8*(lane/16) + e; channels >= C are zero).
 * bias: 8 floats (zeros past mid).  out: (B,H,W,8) bf16, channels >= mid are exact zeros.
 * colsum (optional): [B][units][16] fp32 per-wave-unit channel sums of the stored values (what hat_cab_fold consumes as
 * c1_colsum with tiles = units, ldcs = 16); units from hat_cab_squeeze_units.  dtype must be HAT_BF16 (the fp32 parity
 * path uses hat_conv).
 */
int hat_cab_squeeze_units(int32_t H, int32_t W, int32_t* rows_per_band, int32_t* units);
int hat_cab_squeeze(const void* x, const void* wpk, const float* bias, void* out, float* colsum, int32_t B, int32_t H,
                    int32_t W, int32_t C, int32_t ldx, int32_t dtype, void* stream);

/*
 * conv_last (hat_arch.py:757, 856-858): out = (conv3x3(x, 64 -> n_out <= 8) + bias) * out_scale + mean[ch], written as
 * (B, n_out, H, W) fp32 planes — the same row-sweep kernel as hat_cab_squeeze with two k-steps (num_feat = 64,
 * hat_arch.py:656).  x: (B,H,W,ldx) bf16; wpk: 6 tiles x 2 k-steps of A fragments in hat_cab_squeeze's tile order;
 * bias: 8 floats; mean4: 4 floats (the RGB mean, or zeros); W % 16 == 0; dtype must be HAT_BF16.
 */
int hat_conv3x3_to_planes(const void* x, const void* wpk, const float* bias, float* out, int32_t B, int32_t H, int32_t W,
                          int32_t C, int32_t ldx, int32_t n_out, float out_scale, const float* mean4, int32_t dtype,
                          void* stream);

/*
 * (Shifted-)window self-attention, (S)W-MSA — SURVEY §8 row f2.  Replaces, for one attention branch of a Swin / upstream-HAT
 * block, ESC/basicsr/archs/swinir_arch.py:291-317 (torch.roll by -shift, window_partition, WindowAttention core :147-168
 * with the relative-position bias :153-156 and the shift mask of calculate_mask :262-280, window_reverse, torch.roll by
 * +shift); the same buffers this fork's HAT still registers (hat_arch.py:770-781, 805-818).  The two Linear layers
 * (qkv :146, proj :169) are hat_linear launches on either side.
 * q: (B,H,W,ldq) T, already multiplied by head_dim^-0.5; kv: (B,H,W,ldkv) T with k at channel 0 and v at channel C (for
 * a packed qkv map of row stride 3C: q = base, kv = base + C); head h owns channels [h*d, (h+1)*d), d = C/heads even, <= 32.
 * bias_flip: [heads][(2ws-1)^2] fp32 with bias_flip[h][i] = relative_position_bias_table[(2ws-1)^2 - 1 - i][h]
 * (the kernel indexes by key - query offsets, the reference's relative_position_index by query - key).
 * shift: 0 (W-MSA) or a multiple of 4 below ws (SW-MSA; the reference uses ws/2): windows are taken on the cyclically
 * shifted frame and pairs in different mask bands get -100 added to the logit (not -inf), as in the reference.
 * out: (B,H,W,ldo) T at the un-shifted pixel positions.  ws in {8, 16}; H, W multiples of ws.
 */
int hat_window_attention(const void* q, const void* kv, const float* bias_flip, void* out, int32_t B, int32_t H,
                         int32_t W, int32_t C, int32_t heads, int32_t ws, int32_t shift, int32_t ldq, int32_t ldkv,
                         int32_t ldo, int32_t dtype, void* stream);

/*
 * Fused HAB feed-forward half (hat_arch.py:237 with :107-119):
 *     t_out = t_in + fc2( a * SiLU(g) ),  [a | g] = dwconv3x3( fc1( LayerNorm2(t_in) ) )
 * in ONE kernel: the 4C-wide intermediate never leaves the CU (LDS), HBM traffic is one read and one
 * write of the fp32 residual stream.  Optionally also emits the NEXT block's LayerNorm of t_out
 * (n_out, T) and its ESC global-average-pool partials (gap_out[b][tile][16]), saving that pass.
 * t_out must not alias t_in (3x3 halo).  Weights are "fragment packed" by the host:
 *   w1f [chunk][4][KS][64 lanes][8]  : fc1 rows {a: 32c..32c+31, g: hid_p+32c..} of chunk c, MFMA A-fragment order,
 *                                      K = round_up(C+1, 32) with the fc1 BIAS stored as column k = C
 *   w2f [chunk][nt][64 lanes][8]     : fc2 columns 32c..32c+31, k order (g, j<4) -> 4g+j, (g, j>=4) -> 16+4g+j-4
 *   dww [chunk][64 lanes][4 groups x 5 tap pairs] (fp32, or a bf16 duplicated in both halves of a dword): the
 *        depthwise weight of channel (lane & 15) of each 16-channel group {a0, a1, g0, g1} for tap
 *        2*pair + (lane >> 5) — "tap 9" is the depthwise BIAS — zero in lanes with
 *        ((lane & 15) >> 3) != ((lane >> 4) & 1); the kernel expands it to a diagonal MFMA operand
 *   b2 : [nt*16] fp32; hid_p = 32*chunks >= hidden, zero padded.  (b1, dwb: [2*hid_p] fp32 copies of the biases
 *   that are folded into w1f / dww; kept for reference, not read by the kernel.)
 * Spatial tile = (2*waves) rows x 16 columns; hat_ffn_tiles() returns the tile count (leading dim of gap_out).
 */
typedef struct HatFfnDesc {
    const float* t_in;
    float* t_out;
    const float* ln_g;
    const float* ln_b;
    const void* w1f;
    const float* b1;
    const void* dww;
    const float* dwb;
    const void* w2f;
    const float* b2;
    const float* ln1_g;  /* optional fused next LayerNorm (NULL: off) */
    const float* ln1_b;
    void* n_out;         /* (B,H,W,ldn) T */
    float* gap_out;      /* optional [B][tiles][16] */
    int32_t B, H, W, C;
    int32_t chunks;      /* hid_p / 32 */
    int32_t ldn, gap_c;
    int32_t dtype;
    /* optional: LayerNorm2(t_in) already computed by the producer (hat_linear's ln_out with ln_ones): T rows of
     * ldm_in >= 32*ceil((C+1)/32) elements = [LN (C) | 1.0 | zeros].  The kernel then copies instead of normalising
     * (ln_g / ln_b are ignored). */
    int32_t ldm_in;
    const void* m_in;
    /* optional (hat_ffn2 / hat_hab_tail, with ln1_g): also write channels [0, 16) of n_out as a compact (B,H,W,16) T plane —
     * what the next block's ESC 13x13 conv reads (32 contiguous bytes per pixel instead of 32 out of every ldn * 2) */
    void* n16_out;
} HatFfnDesc;

int hat_ffn_tiles(const HatFfnDesc* d, int32_t* tiles_out);
int hat_ffn(const HatFfnDesc* d, void* stream);

/*
 * Second-generation fused feed-forward half for embed_dim 144 / bf16 (same HatFfnDesc, same tile geometry and outputs as
 * hat_ffn; m_in must be NULL): the hidden tensor lives on chip as FP16 (fc1 accumulators converted round-toward-zero:
 * saturating), the depthwise 3x3 and the gate run as packed-fp16 VALU, fc2 is an fp16 MFMA.  Different packing:
 *   w1f [chunk][4][5][64 lanes][8] bf16 : fc1 rows {a: 32c..32c+31 | gate: hid+32c..} of chunk c, A-fragment order,
 *                                         K = 144 zero padded to 160 (NO bias column)
 *   b1  [chunk][64] fp32                : fc1 bias of the chunk's rows in the same order (the MFMA C operand)
 *   dww [chunk][4 groups][10][16] fp16  : depthwise weights of hidden units 32c+8g..+7 per tap 0..8 and the depthwise
 *                                         BIAS as "tap 9": eight a-unit values then eight gate-unit values
 *   w2f [chunk][9][64 lanes][8] fp16    : fc2 columns 32c..32c+31 in natural k order (k = 8*(lane>>4) + j)
 *   b2  [144] fp32.  dwb is not read.  HAT_EUNSUPPORTED unless C == 144 and dtype == HAT_BF16.
 */
int hat_ffn2(const HatFfnDesc* d, void* stream);

/*
 * The whole second half of a HAB (hat_arch.py:233-237 with esc_arch.py:123) in ONE kernel, embed_dim 144 / bf16:
 *     tB    = t + W_aggr . [y16 | n[16:]] + wf . im2col3x3(c1) + bias_b        (= hat_aggr_cab, kept in registers / LDS)
 *     t_out = tB + fc2( a * SiLU(g) ),  [a | g] = dwconv3x3( fc1( LayerNorm2(tB) ) )   (= hat_ffn2)
 * `ffn` as for hat_ffn2 except that ffn.t_in is the residual stream BEFORE the aggregation (t); the fp32 tB that
 * hat_aggr_cab would write and hat_ffn2 read back (with its halo) never exists in memory: 1 152 of the pair's 2 896
 * algorithmic bytes per pixel.  The aggregation is evaluated on the haloed tile (10 x 18 pixels for 8 x 16 outputs).
 * n: (B,H,W,ldn_in) T = LayerNorm1(t) whose first 16 channels are replaced by y16 (B,H,W,16) T; c1 (B,H,W,8) T;
 * w_aggr: the aggregation weights fragment packed as for hat_linear (nt = 9, K = 160); wf, bias_b: hat_cab_fold's outputs.
 */
typedef struct HatHabTailDesc {
    HatFfnDesc ffn;
    const void* n;
    const void* y16;
    const void* c1;
    const void* w_aggr;
    const void* wf;
    const float* bias_b;
    int32_t ldn_in;
    /* hat_hab_tail3 at embed_dim 180 only (c1 / wf unused there): the CAB expand conv's output map and its per-sample scale */
    int32_t ldr2;               /* row stride of r2 (elements) */
    const void* r2;             /* (B,H,W,ldr2) T: c2 = conv3x3(c1) + b2 */
    const float* r2scale;       /* [B][r2scale_bstride] fp32: conv_scale * ECA(c2); 1 KiB must be readable from every sample's row */
    int32_t r2scale_bstride;
    int32_t reserved1;          /* hat_hab_tail3 at embed_dim 144: bit 0 = ffn.t_in, bit 1 = ffn.t_out are FP16 rows (B,H,W,C) instead of
                                 * fp32 ones (a 16-bit residual stream between the blocks of a group; values are clamped to the finite
                                 * FP16 range when written); 0 everywhere else */
} HatHabTailDesc;
int hat_hab_tail(const HatHabTailDesc* d, void* stream);

/*
 * Third generation of the same launch (same HatHabTailDesc, same arithmetic, tiles, outputs and rounding points as
 * hat_hab_tail; replaces the same reference lines).  LayerNorm2(tB) of the haloed tile stays in the REGISTERS of the wave
 * that computed it (fc1 B fragments), and the weights of the current 32 (+32 gate) hidden units are copied into LDS once
 * per workgroup by LDS-DMA and shared by its four waves, instead of every wave streaming its own fragments through L1/L2
 * (344 KB of weights per 8 x 16 tile instead of 1 460 KB).  Different packing of the three per-chunk records:
 *   w1f [chunk][4][5][64 lanes][8] bf16 : as for hat_ffn2, but K = 144 + the fc1 BIAS as column k = 144 (the kernel keeps
 *                                         a 1.0 in k-slot 144 of every pixel inside the image and 0 outside: a pixel the
 *                                         depthwise conv must see as zero padding gets U = 0 exactly, hat_arch.py:112-114)
 *   dww [chunk][1024] fp16              : hat_ffn2's 640-element record zero padded to 2 KiB (two whole LDS-DMA pieces)
 *   w2f [chunk][9][64 lanes][8] fp16    : as for hat_ffn2.   b1 and dwb are not read.
 * embed_dim 180 (HAT / HAT-L; ffn.C == 180, ffn.chunks == 12: the hidden width 360 zero padded to 384): 12 channel tiles, K = 180
 * in 6 k-steps with the fc1 bias as column k = 180 — w1f [12][4][6][64][8], w2f [12][12][64][8], dww [12][1024], b2 a 1 KiB
 * record — and no folded CAB (their squeeze is 60 channels wide): tB = t + W_aggr . [y16 | n[16:]] + r2scale * r2 + bias_b with
 * w_aggr fragment packed [12][6][64][8], bias_b a 1 KiB record [256] (zero padded), r2 / r2scale as below; c1 and wf are ignored.
 */
int hat_hab_tail3(const HatHabTailDesc* d, void* stream);

/*
 * Forward plans — the whole network behind three calls, for hosts without Python (SURVEY §8b's hat_forward(handle ...)).
 * A plan file holds ONE input shape's complete forward: every call of this header in launch order with its descriptors
 * and scalars, every device pointer as (buffer, offset), the packed weights and constants with their bytes, and the sizes
 * of the workspace buffers.  `python -m super_resolution_amd.plan -opt x.yml --shape B H W -o net.hatplan` (or
 * `super_resolution_amd.plan.export_plan(net, shape, path)`) writes it by recording what the engine launches.
 *   hat_plan_load     reads the file, allocates (hipMalloc) and uploads; *out owns the device memory.
 *   hat_plan_info     dims8 = {B, Cin, H, W, scale, Cout, dtype, 0}; number of launches; device bytes held.
 *   hat_plan_forward  x: (B,Cin,H,W) fp32 NCHW in [0,1], y: (B,Cout,scale*H,scale*W) fp32, both device memory of that
 *                     shape; issues the recorded launches on `stream` (one stream; no allocation, no sync).  Results
 *                     are bit-identical to the Python engine's for the same weights.  Not re-entrant per plan (the
 *                     workspace is the plan's): one forward at a time.
 *   hat_plan_free     releases everything.
 */
typedef struct hat_plan hat_plan;
int hat_plan_load(const char* path, hat_plan** out);
int hat_plan_info(const hat_plan* plan, int32_t* dims8, int64_t* n_calls, int64_t* device_bytes);
int hat_plan_forward(const hat_plan* plan, const float* x, float* y, void* stream);
void hat_plan_free(hat_plan* plan);

/*
 * Per-channel sums of a channel-last map over the pixel rectangle rows [r0, r1) x columns [c0, c1):
 *     out[b*ldo + ch] = sum x[b][(r*W + c)*ld + ch],  ch < C,  x: T = bf16 / fp32, ld % 4 == 0 with zero pad channels
 * — the global average pools of the path (ECA, hat_arch.py:73; ESC dynamic kernel, esc_arch.py:96,121) when a frame is sharded
 * into row bands and every band contributes the sums of the rows it OWNS; the bands' vectors are then added
 * (hat_add_f32 on one GPU, an RCCL all-reduce across GPUs) and fed to hat_esc_weights / hat_eca_scale as ONE block, to
 * hat_cab_fold as `stats`.  Deterministic: fixed reduction order, cross-workgroup part in fp64.
 * bstride: elements between samples; tmp: [B][64][256] fp32 scratch; counter: [B] uint32, zero before the first call
 * (the kernel leaves it zero).
 */
int hat_rect_sum(const void* x, int32_t dtype, int32_t ld, int32_t C, int32_t W, int32_t r0, int32_t r1, int32_t c0, int32_t c1,
                 int64_t bstride, int32_t B, float* out, int32_t ldo, float* tmp, uint32_t* counter, void* stream);

int hat_abi_version(void);
/* name of the architecture the code objects in this library were compiled for ("gfx950") */
const char* hat_target_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* HAT_MI355X_H */
