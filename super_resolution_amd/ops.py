"""Tensor-level wrappers over the C ABI: torch tensors in (device memory, current stream), raw
pointers out.  PyTorch is plumbing here — allocation and stream ownership — no torch op computes
anything on the hot path.  Every wrapper raises if the library is missing or a call fails.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_LRELU, ACT_NONE, HAT_BF16, HAT_F32, O_NCHW_F32, O_NHWC_F32, O_NHWC_T, O_PIXSHUF_T,
                   X_NCHW_F32_MEAN, X_NHWC_F32, X_NHWC_T, HatAggrCabDesc, HatCabFoldDesc, HatConvDesc, HatFfnDesc, HatHabTailDesc, HatMlpDesc)

TORCH_DTYPE = {HAT_F32: torch.float32, HAT_BF16: torch.bfloat16}
DTYPE_CODE = {"f32": HAT_F32, "fp32": HAT_F32, "float32": HAT_F32, "bf16": HAT_BF16, "bfloat16": HAT_BF16}
KC = {HAT_F32: 32, HAT_BF16: 64}


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("HAT HIP ops need device tensors (no CPU path exists)")
    if not t.is_contiguous():
        raise RuntimeError("HAT HIP ops need contiguous tensors")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- optional per-launch timing (bench.py): HIP events on the stream the kernels are enqueued on ----
_prof = None  # None, or a list of (kernel_name, algorithmic_flops, start_event, end_event)


class profile:
    """`with ops.profile() as records:` brackets every kernel launch with HIP events on the current
    stream and records (kernel name as rocprof prints it, algorithmic FLOPs of the launch)."""

    def __enter__(self):
        global _prof
        _prof = []
        return _prof

    def __exit__(self, *exc):
        global _prof
        _prof = None
        return False


def profiling() -> bool:
    return _prof is not None


def _timed(name, flops, fn, tag="", nbytes=0.0):
    """nbytes: ALGORITHMIC HBM bytes of the launch (each operand read once, each result written once)."""
    if _prof is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    _prof.append((name, float(flops), s, e, tag, float(nbytes)))
    return r


_TNAME = {HAT_F32: "float", HAT_BF16: "__bf16"}


class PackedConv:
    """Packed weights of one conv/linear layer (see HatConvDesc in include/hat_mi355x.h)."""
    __slots__ = ("w", "bias", "ksize", "cin", "kpad", "nt", "n_slices", "nout", "w_bstride", "frag", "ksplit")

    def __init__(self, w, bias, ksize, cin, kpad, nt, n_slices, nout, w_bstride=0, frag=False):
        self.w, self.bias, self.ksize, self.cin, self.kpad = w, bias, ksize, cin, kpad
        self.nt, self.n_slices, self.nout, self.w_bstride = nt, n_slices, nout, w_bstride
        self.frag = frag  # True: MFMA-fragment order for hat_linear; False: [Npad][Kpad] rows for hat_conv
        self.ksplit = None  # second half of a layer whose K is split over two launches (engine._lin)

    @property
    def npad(self):
        return self.nt * 16 * self.n_slices


def choose_nt(nout: int):
    """n-tiles per slice in {12, 9, 8, 4, 1}: least padded work, weighted by LDS fragment reads per MFMA."""
    best = None
    for nt in (12, 9, 8, 4, 1):
        npad = -(-nout // (16 * nt)) * 16 * nt
        cost = npad * (nt + 2) / nt
        if best is None or cost < best[0]:
            best = (cost, nt, npad // (16 * nt))
    return best[1], best[2]


def pack_conv_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], dtype: int, device, out_perm=None,
                     scale: float = 1.0, nt: Optional[int] = None) -> PackedConv:
    """weight (O, I, k, k) [or (O, I) for nn.Linear] -> [Npad][Kpad] with K = tap * Cin_p + ci."""
    w = weight.detach().to(torch.float32).cpu()
    if w.dim() == 2:
        w = w[:, :, None, None]
    o, i, kh, kw = w.shape
    assert kh == kw
    b = torch.zeros(o) if bias is None else bias.detach().to(torch.float32).cpu()
    if scale != 1.0:
        w, b = w * scale, b * scale
    if out_perm is not None:
        w, b = w[out_perm], b[out_perm]
    cin_p = (i + 7) // 8 * 8
    k = kh * kw * cin_p
    if nt is None:
        nt, n_slices = choose_nt(o)
    else:
        n_slices = -(-o // (16 * nt))
    kc = KC[dtype] * (3 if nt == 1 else (2 if nt <= 4 else 1))  # weight chunk length of hat_conv.hip: longer for few n-tiles
    kpad = -(-k // kc) * kc
    npad = nt * 16 * n_slices
    wp = torch.zeros(npad, kpad, dtype=torch.float32)
    wt = torch.zeros(o, kh * kw, cin_p)
    wt[:, :, :i] = w.permute(0, 2, 3, 1).reshape(o, kh * kw, i)
    wp[:o, :k] = wt.reshape(o, k)
    bp = torch.zeros(npad, dtype=torch.float32)
    bp[:o] = b
    return PackedConv(wp.to(TORCH_DTYPE[dtype]).to(device).contiguous(), bp.to(device), kh, i, kpad, nt, n_slices, o)


def conv(pw: PackedConv, x: torch.Tensor, out: torch.Tensor, *, B: int, H: int, W: int, dtype: int, ldx: int, ldo: int,
         x_mode: int = X_NHWC_T, out_mode: int = O_NHWC_T, act: int = ACT_NONE, n_store: Optional[int] = None,
         x0: Optional[torch.Tensor] = None, c_split: int = 0, ldx0: int = 0,
         r1: Optional[torch.Tensor] = None, ldr1: int = 0, r2: Optional[torch.Tensor] = None, ldr2: int = 0,
         r2scale: Optional[torch.Tensor] = None, r2scale_bstride: int = 0, colsum: Optional[torch.Tensor] = None,
         ps_r: int = 0, in_scale: float = 1.0, out_scale: float = 1.0, mean=(0.0, 0.0, 0.0, 0.0), cin: Optional[int] = None,
         ln=None, ln_out: Optional[torch.Tensor] = None, ld_ln: int = 0, gap_out: Optional[torch.Tensor] = None, gap_c: int = 0,
         n16_out: Optional[torch.Tensor] = None):
    """ln=(gamma, beta), ln_out, ld_ln: also emit LayerNorm(result) as T rows from the conv's epilogue (one slice that stores
    every channel), with the per-tile sums of its first gap_c channels (gap_out, conv_tiles blocks) and a compact copy of
    its first 16 channels (n16_out) for the ESC path of the block that consumes it."""
    lib = _lib.load()
    d = HatConvDesc()
    if ln is not None:
        d.ln_g, d.ln_b, d.ln_out, d.ld_ln = _ptr(ln[0]), _ptr(ln[1]), _ptr(ln_out), ld_ln
    if gap_out is not None and gap_c:
        d.gap_out, d.gap_c = _ptr(gap_out), gap_c
    d.n16_out = _ptr(n16_out)
    d.x, d.x0, d.w, d.bias, d.out = _ptr(x), _ptr(x0), _ptr(pw.w), _ptr(pw.bias), _ptr(out)
    d.r1, d.r2, d.r2scale, d.colsum = _ptr(r1), _ptr(r2), _ptr(r2scale), _ptr(colsum)
    d.B, d.H, d.W = B, H, W
    d.Cin, d.ldx, d.x_mode = (pw.cin if cin is None else cin), ldx, x_mode
    d.c_split, d.ldx0 = c_split, ldx0
    d.ksize, d.Kpad, d.nt, d.n_slices, d.w_bstride = pw.ksize, pw.kpad, pw.nt, pw.n_slices, pw.w_bstride
    d.n_store = pw.nout if n_store is None else n_store
    d.ldo, d.out_mode, d.act = ldo, out_mode, act
    d.ldr1, d.ldr2, d.r2scale_bstride, d.ps_r = ldr1, ldr2, r2scale_bstride, ps_r
    d.in_scale, d.out_scale = in_scale, out_scale
    for i in range(4):
        d.mean[i] = float(mean[i]) if i < len(mean) else 0.0
    d.dtype = dtype
    name, flops = "conv_kernel", 0.0
    if _prof is not None:
        wv, pt, tl, lds = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int64(0)
        lib.hat_conv_plan(C.byref(d), C.byref(wv), C.byref(pt), C.byref(tl), C.byref(lds))
        name = f"conv_kernel<{_TNAME[dtype]}, {wv.value}, {pt.value}, {pw.nt}>"
        # algorithmic FLOPs (2*MAC, unpadded).  The 13x13 ESC conv also carries the
        # dynamic depthwise 3x3 that is folded into its weights (2*9*pdim per pixel).
        flops = 2.0 * B * H * W * (pw.ksize ** 2) * d.Cin * pw.nout + (2.0 * 9 * pw.nout * B * H * W if pw.w_bstride else 0.0)
    _timed(name, flops, lambda: _lib.check(lib.hat_conv(C.byref(d), _stream()),
                                           f"hat_conv(k={pw.ksize}, Cin={d.Cin}, N={pw.nout})"),
           tag=f"k{pw.ksize} {d.Cin}->{pw.nout} {H}x{W} x{x_mode} o{out_mode}{' r1' if r1 is not None else ''}{' r2' if r2 is not None else ''}")


def conv_tiles(pw: PackedConv, H: int, W: int, dtype: int) -> int:
    lib = _lib.load()
    d = HatConvDesc()
    d.B, d.H, d.W, d.Cin, d.ksize, d.nt, d.n_slices, d.dtype = 1, H, W, pw.cin, pw.ksize, pw.nt, pw.n_slices, dtype
    n = C.c_int32(0)
    _lib.check(lib.hat_conv_tiles(C.byref(d), C.byref(n)), "hat_conv_tiles")
    return n.value


def layernorm_blocks() -> int:
    return _lib.load().hat_layernorm_blocks()


def layernorm(x: torch.Tensor, y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, B: int, npix: int, C_: int,
              ldy: int, out_f32: bool, dtype: int, gap: Optional[torch.Tensor] = None, gap_c: int = 0):
    lib = _lib.load()
    _timed("ln_kernel", 0.0, lambda: _lib.check(
        lib.hat_layernorm(_ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(gap), B, npix, C_, ldy, int(out_f32), gap_c, dtype,
                          _stream()), "hat_layernorm"))


def add_f32(a: torch.Tensor, c: torch.Tensor, out: torch.Tensor, *, B: int, n: int, c_bstride: Optional[int] = None):
    """out[b] = a[b] + c[b] (c_bstride = n) or + c (c_bstride = 0: broadcast over the batch); fp32."""
    lib = _lib.load()
    cb = n if c_bstride is None else c_bstride
    _timed("add_f32_kernel", 0.0, lambda: _lib.check(lib.hat_add_f32(_ptr(a), _ptr(c), _ptr(out), B, n, cb, _stream()), "hat_add_f32"))


def rect_sum(x: torch.Tensor, out: torch.Tensor, tmp: torch.Tensor, counter: torch.Tensor, *, B: int, W: int, ld: int, C_: int, r0: int,
             r1: int, c0: int = 0, c1: Optional[int] = None, out_off: int = 0, dtype: Optional[int] = None):
    """out[b][out_off : out_off + C_] = per-channel sums of x (B, rows*W, ld) over rows [r0, r1) x columns [c0, c1) (hat_rect_sum):
    the pooled sums a row band of a sharded frame contributes."""
    lib = _lib.load()
    dt = dtype if dtype is not None else (HAT_BF16 if x.dtype == torch.bfloat16 else HAT_F32)
    o = out.view(B, -1)
    bstride = x.numel() // B
    _timed("rect_sum_kernel", 0.0, lambda: _lib.check(
        lib.hat_rect_sum(_ptr(x), dt, ld, C_, W, r0, r1, c0, (W if c1 is None else c1), bstride, B, o.data_ptr() + 4 * out_off, o.shape[1],
                         _ptr(tmp), _ptr(counter), _stream()), "hat_rect_sum"))


def esc_weights(gap: torch.Tensor, nblk: int, npix: int, w1, b1, w2, b2, plk_packed, w_out, *, B: int, pdim: int,
                ksize: int, kpad: int, dtype: int):
    lib = _lib.load()
    _timed("esc_weights_kernel", 0.0, lambda: _lib.check(
        lib.hat_esc_weights(_ptr(gap), nblk, npix, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(plk_packed), _ptr(w_out), B,
                            pdim, ksize, kpad, dtype, _stream()), "hat_esc_weights"))


def esc_conv13_supported(pdim: int, ksize: int, dtype: int) -> bool:
    return pdim == 16 and ksize == 13 and dtype == HAT_BF16


def esc_conv13(x, wp, y16, *, B: int, H: int, W: int, ldx: int, kpad: int, dtype: int):
    """ESC 13x13 conv on the dedicated kernel (hat_esc_conv13): weights and haloed tile resident in LDS."""
    lib = _lib.load()
    _timed("esc13_kernel", 2.0 * B * H * W * 169 * 256 + 2.0 * 9 * 16 * B * H * W, lambda: _lib.check(
        lib.hat_esc_conv13(_ptr(x), ldx, _ptr(wp), kpad, _ptr(y16), B, H, W, dtype, _stream()), "hat_esc_conv13"),
        tag=f"k13 16->16 {H}x{W} resident")


def eca_scale(colsum, tiles: int, ldc: int, npix: int, wk, k: int, conv_scale: float, tmp, scale, *, B: int, C_: int):
    lib = _lib.load()
    _timed("eca_reduce+scale", 0.0, lambda: _lib.check(
        lib.hat_eca_scale(_ptr(colsum), tiles, ldc, npix, _ptr(wk), k, conv_scale, _ptr(tmp), _ptr(scale), B, C_, _stream()),
        "hat_eca_scale"))


def dwconv_gate(u, wdw, bdw, out, *, B: int, H: int, W: int, hid: int, ldu: int, ldo: int, dtype: int):
    lib = _lib.load()
    _timed(f"dwgate_kernel<{_TNAME[dtype]}>", 2.0 * 9 * 2 * hid * B * H * W, lambda: _lib.check(
        lib.hat_dwconv_gate(_ptr(u), _ptr(wdw), _ptr(bdw), _ptr(out), B, H, W, hid, ldu, ldo, dtype, _stream()),
        "hat_dwconv_gate"))


def ocab_keybias(sal, kv, kb, *, B: int, H: int, W: int, C_: int, ws: int, wse: int, pad: int, k_keep: int, ldsal: int, ldkv: int,
                 dtype: int):
    """HATX focus bias / top-k prune mask per (window, key) (hat_ocab_keybias)."""
    lib = _lib.load()
    _timed("keybias_kernel", 0.0, lambda: _lib.check(
        lib.hat_ocab_keybias(_ptr(sal), ldsal, _ptr(kv), ldkv, _ptr(kb), B, H, W, C_, ws, wse, pad, k_keep, dtype, _stream()), "hat_ocab_keybias"))


def ocab_attention_kb(q, kv, bias_rot, kb, out, *, B: int, H: int, W: int, C_: int, heads: int, ws: int, wse: int, pad: int, ldq: int,
                      ldkv: int, ldo: int, dtype: int):
    lib = _lib.load()
    _timed(f"ocab_attn_kernel<{_TNAME[dtype]}, kb>", 2.0 * 2 * wse * wse * C_ * B * H * W, lambda: _lib.check(
        lib.hat_ocab_attention_kb(_ptr(q), _ptr(kv), _ptr(bias_rot), _ptr(kb), _ptr(out), B, H, W, C_, heads, ws, wse, pad, ldq, ldkv, ldo,
                                  dtype, _stream()), "hat_ocab_attention_kb"))


class PackedMlp:
    """fc1 / fc2 of the OCAB's MLP in hat_ocab_mlp's fragment layouts (include/hat_mi355x.h)."""
    __slots__ = ("w1f", "b1", "w2f", "b2", "C", "hidden")


def ocab_mlp_supported(C_: int, hidden: int, dtype: int) -> bool:
    return C_ == 144 and hidden == 288 and dtype == HAT_BF16


def pack_ocab_mlp(fc1_w, fc1_b, fc2_w, fc2_b, device) -> PackedMlp:
    f = lambda t: t.detach().to(torch.float32).cpu()
    W1, b1, W2, b2 = f(fc1_w), f(fc1_b), f(fc2_w), f(fc2_b)
    hid, C_ = W1.shape
    assert (C_, hid) == (144, 288) and W2.shape == (C_, hid)
    lane = torch.arange(64)
    n16, g4 = lane & 15, lane >> 4
    # fc1: full fragments [nt][ks][lane][j] = W1[16 nt + n16][32 ks + 8 g + j], then the 16-deep tail [nt][lane][j] = W1[..][128 + 4 g + j]
    r = (torch.arange(18)[:, None, None, None] * 16 + n16[None, None, :, None]).expand(18, 4, 64, 8)
    c = (torch.arange(4)[None, :, None, None] * 32 + 8 * g4[None, None, :, None] + torch.arange(8)[None, None, None, :]).expand(18, 4, 64, 8)
    full = W1[r, c]
    rh = (torch.arange(18)[:, None, None] * 16 + n16[None, :, None]).expand(18, 64, 4)
    ch = (128 + 4 * g4[None, :, None] + torch.arange(4)[None, None, :]).expand(18, 64, 4)
    half = W1[rh, ch]
    w1f = torch.cat([full.reshape(-1), half.reshape(-1)])
    # fc2: [nt2][kk][lane][j] = W2[16 nt2 + n16][unit], unit = 32 kk + 4 g + j (j < 4) | 32 kk + 16 + 4 g + j - 4
    j8 = torch.arange(8)
    unit = (torch.arange(9)[:, None, None] * 32 + torch.where(j8[None, None, :] < 4, 4 * g4[None, :, None] + j8[None, None, :],
                                                               16 + 4 * g4[None, :, None] + j8[None, None, :] - 4))   # (kk, lane, j)
    r2 = (torch.arange(9)[:, None, None, None] * 16 + n16[None, None, :, None]).expand(9, 9, 64, 8)
    w2f = W2[r2, unit[None].expand(9, 9, 64, 8)]
    p = PackedMlp()
    p.w1f = w1f.to(torch.bfloat16).contiguous().to(device)
    p.w2f = w2f.to(torch.bfloat16).contiguous().to(device)
    p.b1, p.b2, p.C, p.hidden = b1.contiguous().to(device), b2.contiguous().to(device), C_, hid
    return p


def _pack_fc1_frags(W1: torch.Tensor) -> torch.Tensor:
    """(16 nt, 144) fp32 -> hat_ocab_mlp's fc1 layout: [nt][4][64 lanes][8] full k-steps then [nt][64][4] the 16-deep tail."""
    nt_ = W1.shape[0] // 16
    lane = torch.arange(64)
    n16, g4 = lane & 15, lane >> 4
    r = (torch.arange(nt_)[:, None, None, None] * 16 + n16[None, None, :, None]).expand(nt_, 4, 64, 8)
    c = (torch.arange(4)[None, :, None, None] * 32 + 8 * g4[None, None, :, None] + torch.arange(8)[None, None, None, :]).expand(nt_, 4, 64, 8)
    rh = (torch.arange(nt_)[:, None, None] * 16 + n16[None, :, None]).expand(nt_, 64, 4)
    ch = (128 + 4 * g4[None, :, None] + torch.arange(4)[None, None, :]).expand(nt_, 64, 4)
    return torch.cat([W1[r, c].reshape(-1), W1[rh, ch].reshape(-1)])


def pack_ocab_qkv(q_w, q_b, kv_w, kv_b, qscale: float, device) -> PackedMlp:
    """Stacked [q_proj * qscale ; kv_proj] (432 x 144) for hat_ocab_qkv."""
    f = lambda t: t.detach().to(torch.float32).cpu()
    Wq, Wkv = f(q_w) * qscale, f(kv_w)
    bq = (torch.zeros(Wq.shape[0]) if q_b is None else f(q_b)) * qscale
    bkv = torch.zeros(Wkv.shape[0]) if kv_b is None else f(kv_b)
    W = torch.cat([Wq, Wkv], 0)
    assert W.shape == (432, 144)
    p = PackedMlp()
    p.w1f = _pack_fc1_frags(W).to(torch.bfloat16).contiguous().to(device)
    p.b1 = torch.cat([bq, bkv]).contiguous().to(device)
    p.w2f = p.b2 = None
    p.C, p.hidden = 144, 432
    return p


def ocab_qkv(pm: PackedMlp, x, out, *, B: int, H: int, W: int, ldx: int, ldo: int, dtype: int):
    """out rows [q | k | v] (432 channels) = both OCAB projections of x in one launch (hat_ocab_qkv)."""
    lib = _lib.load()
    d = HatMlpDesc()
    d.x, d.w1f, d.b1, d.out = _ptr(x), _ptr(pm.w1f), _ptr(pm.b1), _ptr(out)
    d.B, d.H, d.W, d.C, d.hidden, d.ldx, d.ldo, d.dtype = B, H, W, pm.C, pm.hidden, ldx, ldo, dtype
    _timed("ocab_qkv_kernel", 2.0 * pm.C * pm.hidden * B * H * W, lambda: _lib.check(lib.hat_ocab_qkv(C.byref(d), _stream()), "hat_ocab_qkv"),
           tag=f"qkv {pm.C}->{pm.hidden} {H}x{W}", nbytes=float(B * H * W) * 2 * (ldx + pm.hidden))


def ocab_mlp(pm: PackedMlp, x, r1, out, *, B: int, H: int, W: int, ldx: int, ldr1: int, ldo: int, out_f32: bool, dtype: int):
    """out = r1 + fc2(GELU(fc1(x))) in one launch (hat_ocab_mlp): the 288-wide hidden tensor never reaches HBM."""
    lib = _lib.load()
    d = HatMlpDesc()
    d.x, d.w1f, d.b1, d.w2f, d.b2, d.r1, d.out = _ptr(x), _ptr(pm.w1f), _ptr(pm.b1), _ptr(pm.w2f), _ptr(pm.b2), _ptr(r1), _ptr(out)
    d.B, d.H, d.W, d.C, d.hidden, d.ldx, d.ldr1, d.ldo, d.out_f32, d.dtype = B, H, W, pm.C, pm.hidden, ldx, ldr1, ldo, int(out_f32), dtype
    _timed(f"ocab_mlp_kernel<{'true' if out_f32 else 'false'}>", 2.0 * 2 * pm.C * pm.hidden * B * H * W,
           lambda: _lib.check(lib.hat_ocab_mlp(C.byref(d), _stream()), "hat_ocab_mlp"),
           tag=f"mlp {pm.C}->{pm.hidden}->{pm.C} {H}x{W}", nbytes=float(B * H * W) * (2 * ldx + 4 * pm.C + (4 if out_f32 else 2) * pm.C))


def sgfn_gate(u, wdw, bdw, out, *, B: int, H: int, W: int, half: int, ldu: int, ldo: int, dtype: int):
    """HATX SGFN: out = [dw3x3(u[:half]) * silu(u[half:]) | u[half:]] (hat_sgfn_gate)."""
    lib = _lib.load()
    _timed(f"sgfn_gate_kernel<{_TNAME[dtype]}>", 2.0 * 9 * half * B * H * W, lambda: _lib.check(
        lib.hat_sgfn_gate(_ptr(u), _ptr(wdw), _ptr(bdw), _ptr(out), B, H, W, half, ldu, ldo, dtype, _stream()), "hat_sgfn_gate"))


LOG2E = 1.4426950408889634


def ocab_attention_log2_supported(C_: int, heads: int, ws: int, wse: int, dtype: int) -> bool:
    """Shapes hat_ocab_attention_log2 (softmax offset in a spare k-slot; q pre-multiplied by log2 e) is built for."""
    return dtype == HAT_BF16 and C_ % heads == 0 and C_ // heads == 24 and ws == 16 and wse == 24 and C_ % 8 == 0


def ocab_attention(q, kv, bias_rot, out, *, B: int, H: int, W: int, C_: int, heads: int, ws: int, wse: int, ldq: int,
                   ldkv: int, ldo: int, dtype: int, q_log2: bool = False):
    """q_log2: q was projected with head_dim^-1/2 * log2(e) folded into its weights (hat_ocab_attention_log2)."""
    lib = _lib.load()
    fn, nm = (lib.hat_ocab_attention_log2, "hat_ocab_attention_log2") if q_log2 else (lib.hat_ocab_attention, "hat_ocab_attention")
    _timed(f"ocab_attn_kernel<{_TNAME[dtype]}>", 2.0 * 2 * wse * wse * C_ * B * H * W, lambda: _lib.check(
        fn(_ptr(q), _ptr(kv), _ptr(bias_rot), _ptr(out), B, H, W, C_, heads, ws, wse, ldq, ldkv, ldo, dtype, _stream()), nm))


# ------------------------------------------------------------------------------------------------
# CAB squeeze conv on the row-sweep kernel (hat_cab_squeeze)
# ------------------------------------------------------------------------------------------------
def cab_squeeze_supported(C_: int, mid: int, W: int, dtype: int) -> bool:
    return dtype == HAT_BF16 and 128 < C_ <= 160 and C_ % 8 == 0 and mid <= 8 and W % 16 == 0


def pack_cab_squeeze(weight: torch.Tensor, bias: torch.Tensor, device):
    """3x3 weight (mid <= 8, C, 3, 3) -> the 6 x ceil(C/32) MFMA A fragments the row-sweep kernel keeps in registers
    (hat_cab_squeeze / hat_conv3x3_to_planes), + 8 bias floats."""
    w = weight.detach().to(torch.float32).cpu()
    mid, cin = w.shape[0], w.shape[1]
    ks = -(-cin // 32)
    A = torch.zeros(6, 16, 32 * ks)                  # [tile][row][k]
    for kx in range(3):
        A[2 * kx, 0:mid, :cin] = w[:, :, 0, kx]      # ky = 0 -> output row r + 1
        A[2 * kx, 8:8 + mid, :cin] = w[:, :, 1, kx]  # ky = 1 -> output row r
        A[2 * kx + 1, 0:mid, :cin] = w[:, :, 2, kx]  # ky = 2 -> output row r - 1
    lane = torch.arange(64)
    shape = (6, ks, 64, 8)
    row = (lane & 15)[None, None, :, None].expand(shape)
    col = (torch.arange(ks)[None, :, None, None] * 32 + 8 * (lane >> 4)[None, None, :, None] + torch.arange(8)[None, None, None, :]).expand(shape)
    tile = torch.arange(6)[:, None, None, None].expand(shape)
    wpk = A[tile, row, col].to(torch.bfloat16).contiguous().to(device)
    b8 = torch.zeros(8)
    b8[:mid] = bias.detach().to(torch.float32).cpu()
    return wpk, b8.to(device)


def conv3x3_to_planes_supported(nout: int, cin: int, W: int, dtype: int) -> bool:
    return dtype == HAT_BF16 and cin == 64 and nout <= 8 and W % 16 == 0


def conv3x3_to_planes(x, wpk, bias8, out, *, B: int, H: int, W: int, C_: int, ldx: int, n_out: int, out_scale: float, mean,
                      dtype: int):
    """conv_last on the row-sweep kernel: (conv3x3 + bias) * out_scale + mean -> (B, n_out, H, W) fp32."""
    lib = _lib.load()
    m4 = (C.c_float * 4)(*[float(mean[i]) if i < len(mean) else 0.0 for i in range(4)])
    _timed("cab_squeeze_kernel<2, planes>", 2.0 * B * H * W * 9 * C_ * n_out, lambda: _lib.check(
        lib.hat_conv3x3_to_planes(_ptr(x), _ptr(wpk), _ptr(bias8), _ptr(out), B, H, W, C_, ldx, n_out, out_scale, m4, dtype,
                                  _stream()), "hat_conv3x3_to_planes"), tag=f"k3 {C_}->{n_out} {H}x{W} row sweep planes")


def cab_squeeze_units(H: int, W: int) -> int:
    lib = _lib.load()
    rows, units = C.c_int32(0), C.c_int32(0)
    _lib.check(lib.hat_cab_squeeze_units(H, W, C.byref(rows), C.byref(units)), "hat_cab_squeeze_units")
    return units.value


def cab_squeeze(x, wpk, bias8, out, colsum, *, B: int, H: int, W: int, C_: int, ldx: int, dtype: int):
    lib = _lib.load()
    _timed("cab_squeeze_kernel", 2.0 * B * H * W * 9 * C_ * 6, lambda: _lib.check(
        lib.hat_cab_squeeze(_ptr(x), _ptr(wpk), _ptr(bias8), _ptr(out), _ptr(colsum), B, H, W, C_, ldx, dtype, _stream()),
        "hat_cab_squeeze"), tag=f"k3 {C_}->8 {H}x{W} row sweep")


def window_attention(q, kv, bias_flip, out, *, B: int, H: int, W: int, C_: int, heads: int, ws: int, shift: int, ldq: int,
                     ldkv: int, ldo: int, dtype: int):
    """(S)W-MSA core (hat_window_attention; swinir_arch.py:147-168 + roll / partition / mask / reverse :291-317)."""
    lib = _lib.load()
    _timed(f"ocab_attn_kernel<{_TNAME[dtype]}, self>", 2.0 * 2 * ws * ws * C_ * B * H * W, lambda: _lib.check(
        lib.hat_window_attention(_ptr(q), _ptr(kv), _ptr(bias_flip), _ptr(out), B, H, W, C_, heads, ws, shift, ldq, ldkv, ldo,
                                 dtype, _stream()), "hat_window_attention"))


# ------------------------------------------------------------------------------------------------
# fused feed-forward half of the HAB (hat_ffn)
# ------------------------------------------------------------------------------------------------
class PackedFFN:
    __slots__ = ("w1f", "b1", "dww", "dwb", "w2f", "b2", "chunks", "C", "hid", "nt", "ks", "khalf")


def ffn_supported(C_: int) -> bool:
    """Shapes hat_ffn is instantiated for (anything else uses the unfused kernels)."""
    return C_ in (144, 180) or (C_ <= 32 and C_ % 32 != 16 and C_ >= 8)


def pack_ffn(fc1_w, fc1_b, dw_w, dw_b, fc2_w, fc2_b, dtype: int, device) -> PackedFFN:
    """Fragment-pack GatedDconvFFN weights (hat_arch.py:99-104) for hat_ffn; layouts in include/hat_mi355x.h."""
    f = lambda t: t.detach().to(torch.float32).cpu()
    W1, b1, Wd, bd, W2, b2 = f(fc1_w), f(fc1_b), f(dw_w).reshape(-1, 9), f(dw_b), f(fc2_w), f(fc2_b)
    C_, hid = W2.shape
    assert W1.shape == (2 * hid, C_) and Wd.shape[0] == 2 * hid
    chunks = -(-hid // 32)
    hid_p = 32 * chunks
    khalf = False
    ks = -(-(C_ + 1) // 32)          # K padded to a multiple of 32 with room for the bias column at k = C
    nt = 9 if C_ == 144 else (12 if C_ == 180 else 2)
    tdt = TORCH_DTYPE[dtype]
    lane = torch.arange(64)
    n16, g4 = lane & 15, lane >> 4
    j8 = torch.arange(8)
    # ---- fc1: w1f[chunk][nt4][ks][lane][8]; column C of the padded weight matrix is the fc1 bias (the kernel
    # keeps a constant 1 in column C of the LayerNorm'ed activations)
    W1p = torch.zeros(2 * hid_p, ks * 32)
    W1p[:hid, :C_] = W1[:hid]
    W1p[hid_p:hid_p + hid, :C_] = W1[hid:]
    W1p[:hid, C_] = b1[:hid]
    W1p[hid_p:hid_p + hid, C_] = b1[hid:]
    c_i = torch.arange(chunks)[:, None, None, None, None]
    nt_i = torch.arange(4)[None, :, None, None, None]
    ks_i = torch.arange(ks)[None, None, :, None, None]
    nl = nt_i * 16 + n16[None, None, None, :, None]                       # chunk-local channel 0..63
    row = torch.where(nl < 32, c_i * 32 + nl, hid_p + c_i * 32 + (nl - 32)).expand(chunks, 4, ks, 64, 8)
    col = (ks_i * 32 + 8 * g4[None, None, None, :, None] + j8[None, None, None, None, :]).expand(chunks, 4, ks, 64, 8)
    w1f = W1p[row, col]
    # ---- fc2: w2f[chunk][nt][lane][8]; the k order inside the 32-deep chunk is the accumulator order of the
    # depthwise stage: element (g, j<4) <-> channel 4g+j of a-group 0, (g, j>=4) <-> channel 16+4g+(j-4)
    W2p = torch.zeros(nt * 16, hid_p)
    W2p[:C_, :hid] = W2
    n_i = (torch.arange(nt)[:, None] * 16 + n16[None, :])                  # (nt, 64)
    kloc = torch.where(j8[None, :] < 4, 4 * g4[:, None] + j8[None, :], 16 + 4 * g4[:, None] + (j8[None, :] - 4))  # (64, 8)
    k_i = torch.arange(chunks)[:, None, None] * 32 + kloc[None]            # (chunks, 64, 8)
    w2f = W2p[n_i[None, :, :, None].expand(chunks, nt, 64, 8), k_i[:, None].expand(chunks, nt, 64, 8)]
    # ---- biases
    b1p = torch.zeros(2 * hid_p)
    b1p[:hid], b1p[hid_p:hid_p + hid] = b1[:hid], b1[hid:]
    dwb = torch.zeros(2 * hid_p)
    dwb[:hid], dwb[hid_p:hid_p + hid] = bd[:hid], bd[hid:]
    b2p = torch.zeros(nt * 16)
    b2p[:C_] = b2
    # ---- depthwise weights, one value per (chunk, lane, group of 16 channels, tap pair): lane (n = l&15, g = l>>4)
    # owns channel n of the group and tap 2*pair + (g>>1) (tap 9 = the depthwise BIAS, multiplied by a constant 1 in
    # the kernel); non-zero only in the lanes whose 8-wide k group holds channel n (n>>3 == g&1), so the kernel
    # builds its diagonal A fragment from this single value.  bf16: stored duplicated in both halves of a dword.
    Wd_a, Wd_g = torch.zeros(hid_p, 10), torch.zeros(hid_p, 10)
    Wd_a[:hid, :9], Wd_g[:hid, :9] = Wd[:hid], Wd[hid:]
    Wd_a[:hid, 9], Wd_g[:hid, 9] = bd[:hid], bd[hid:]
    Wdp = torch.stack([Wd_a.reshape(chunks, 2, 16, 10)[:, 0], Wd_a.reshape(chunks, 2, 16, 10)[:, 1],
                       Wd_g.reshape(chunks, 2, 16, 10)[:, 0], Wd_g.reshape(chunks, 2, 16, 10)[:, 1]])  # [group][chunk][ch][tap]
    tap = 2 * torch.arange(5)[None, :] + (g4[:, None] >> 1)                  # (64, 5)
    active = ((n16 >> 3) == (g4 & 1)).to(torch.float32)                      # (64,)
    dww = torch.zeros(chunks, 64, 4, 5)
    for gi in range(4):
        dww[:, :, gi, :] = Wdp[gi][:, n16[:, None].expand(64, 5), tap] * active[None, :, None]
    dww = dww.reshape(chunks, 64, 20)
    if dtype == HAT_BF16:
        bits = dww.to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
        dww = (bits | (bits << 16)).to(torch.int32)
    p = PackedFFN()
    p.w1f, p.w2f = w1f.to(tdt).contiguous().to(device), w2f.to(tdt).contiguous().to(device)
    p.b1, p.dwb, p.b2 = b1p.to(device), dwb.to(device), b2p.to(device)
    p.dww = dww.contiguous().to(device)
    p.chunks, p.C, p.hid, p.nt, p.ks, p.khalf = chunks, C_, hid, nt, ks, khalf
    return p


def ffn2_supported(C_: int, hid: int, dtype: int) -> bool:
    """Shapes hat_ffn2 (fp16 hidden tensor, VALU depthwise conv) is built for."""
    return C_ == 144 and hid % 32 == 0 and dtype == HAT_BF16


FP16_SAFE = 6.0e4   # below _Float16's largest finite value, 65504


def ffn_fp16_range_bound(fc1_w, fc1_b, dw_w, dw_b, ln_g, ln_b) -> float:
    """Worst-case magnitude of anything hat_ffn2 / hat_hab_tail3 hold in FP16 — the hidden tensor u = fc1(LayerNorm2(x)), the
    depthwise conv's outputs a and g, and the gated product a * g * sigmoid(g) — for ANY input: LayerNorm's normalised row
    has Euclidean norm <= sqrt(C), so |u_j| <= sqrt(C) * ||W1[j] * gamma||_2 + |W1[j] . beta + b1[j]| =: U_j (Cauchy-Schwarz),
    |a_j| <= sum_taps |wd[j, tap]| * U_j + |bd_j|, likewise g, and |a * g * sigmoid(g)| <= |a| * |g|.
    The kernels convert to FP16 with round-toward-zero (a value past the range saturates at 65504 instead of becoming an
    infinity) but the packed-FP16 products behind that conversion can still overflow, so the engine uses these kernels only
    while this bound stays below FP16_SAFE and otherwise keeps hat_ffn (hidden tensor in bf16, fp32 range).  The bound is loose
    by design (a trained HAT-S sits orders of magnitude below it: unit-variance rows, weights of norm ~1 give U ~ 12, a * g ~ 10^3)."""
    f = lambda t: t.detach().to(torch.float64).cpu()
    W1, b1, Wd, bd, g_, b_ = f(fc1_w), f(fc1_b), f(dw_w).reshape(-1, 9), f(dw_b), f(ln_g), f(ln_b)
    C_ = W1.shape[1]
    hid = W1.shape[0] // 2
    U = (C_ ** 0.5) * (W1 * g_[None, :]).norm(dim=1) + (W1 @ b_ + b1).abs()
    A = Wd.abs().sum(1) * U + bd.abs()
    return float(max(U.max(), A.max(), (A[:hid] * A[hid:]).max()))


def pack_ffn2(fc1_w, fc1_b, dw_w, dw_b, fc2_w, fc2_b, device) -> PackedFFN:
    """GatedDconvFFN weights (hat_arch.py:99-104) in hat_ffn2's layouts (include/hat_mi355x.h)."""
    f = lambda t: t.detach().to(torch.float32).cpu()
    W1, b1, Wd, bd, W2, b2 = f(fc1_w), f(fc1_b), f(dw_w).reshape(-1, 9), f(dw_b), f(fc2_w), f(fc2_b)
    C_, hid = W2.shape
    assert W1.shape == (2 * hid, C_) and Wd.shape[0] == 2 * hid and hid % 32 == 0 and C_ == 144
    chunks, ks, nt = hid // 32, 5, 9
    lane = torch.arange(64)
    n16, g4, j8 = lane & 15, lane >> 4, torch.arange(8)
    # fc1 output row nl = tile * 16 + n16 of a chunk (tiles 0, 1: a half; 2, 3: gate half) computes hidden unit
    # q = 8 g + 4 ii + r of its half (ii = tile & 1, n16 = 4 g + r): a lane's 4 + 4 results of the two tiles are then 16
    # contiguous bytes of the U row, unit order natural (one conflict-free-enough ds_write_b128, hat_ffn2.hip `ust`)
    nl = torch.arange(64)
    q = 8 * ((nl & 15) >> 2) + 4 * ((nl >> 4) & 1) + (nl & 3)
    rows = (nl[None, :] >> 5) * hid + torch.arange(chunks)[:, None] * 32 + q[None, :]            # (chunks, 64)
    W1p = torch.zeros(2 * hid, ks * 32)
    W1p[:, :C_] = W1
    r = rows.reshape(chunks, 4, 16)[:, :, None, n16, None].expand(chunks, 4, ks, 64, 8)
    col = (torch.arange(ks)[None, None, :, None, None] * 32 + 8 * g4[None, None, None, :, None] + j8).expand(chunks, 4, ks, 64, 8)
    w1f = W1p[r, col]
    b1c = b1[rows]                                                                               # (chunks, 64)
    # depthwise: [chunk][g][tap 0..8, bias][a-units 8g..8g+7 | gate-units 8g..8g+7]
    Wd10 = torch.cat([Wd, bd[:, None]], dim=1)                                                    # (2*hid, 10)
    unit = (torch.arange(chunks)[:, None, None] * 32 + 8 * torch.arange(4)[None, :, None] + j8[None, None, :])   # (chunks, 4, 8)
    dww = torch.cat([Wd10[unit].permute(0, 1, 3, 2), Wd10[hid + unit].permute(0, 1, 3, 2)], dim=-1)  # (chunks, 4, 10, 16)
    W2p = torch.zeros(nt * 16, hid)
    W2p[:C_] = W2
    n_i = torch.arange(nt)[:, None] * 16 + n16[None, :]                                           # (nt, 64)
    k_i = torch.arange(chunks)[:, None, None] * 32 + (8 * g4[:, None] + j8[None, :])[None]        # (chunks, 64, 8)
    w2f = W2p[n_i[None, :, :, None].expand(chunks, nt, 64, 8), k_i[:, None].expand(chunks, nt, 64, 8)]
    b2p = torch.zeros(nt * 16)
    b2p[:C_] = b2
    p = PackedFFN()
    p.w1f = w1f.to(torch.bfloat16).contiguous().to(device)
    p.w2f = w2f.to(torch.float16).contiguous().to(device)
    p.dww = dww.to(torch.float16).contiguous().to(device)
    p.b1, p.dwb, p.b2 = b1c.contiguous().to(device), bd.to(device), b2p.to(device)
    p.chunks, p.C, p.hid, p.nt, p.ks, p.khalf = chunks, C_, hid, nt, ks, "v2"
    return p


def tail3_supported(C_: int, hid: int, dtype: int) -> bool:
    """Shapes hat_hab_tail3 is built for: embed_dim 144 (hidden 288, folded CAB) and 180 (hidden 360 padded to 384, c2 as a map)."""
    return dtype == HAT_BF16 and ((C_ == 144 and hid % 32 == 0) or (C_ == 180 and hid == 360))


def pack_ffn3(fc1_w, fc1_b, dw_w, dw_b, fc2_w, fc2_b, ln_g, ln_b, device) -> PackedFFN:
    """GatedDconvFFN weights (hat_arch.py:99-104) + the affine part of the LayerNorm in front of them (norm2, hat_arch.py:237)
    in hat_hab_tail3's layouts (include/hat_mi355x.h), embed_dim 144 or 180:
      * LayerNorm2's gamma folded into the fc1 columns and W1 . beta into the fc1 bias (fc1(xhat * gamma + beta) =
        (W1 diag(gamma)) xhat + (W1 beta + b1), exact in real arithmetic): the kernel normalises without an affine step;
      * the fc1 bias as column k = C of the fc1 fragments (K = C + 1 padded to a multiple of 32);
      * the hidden width padded to a multiple of 32 with zero units (embed_dim 180: 360 -> 384);
      * fc1 output row (tile, n16) of a chunk computes hidden unit 8 (n16 >> 2) + 4 (tile & 1) + (n16 & 3) of its half, so that a
        lane's 4 + 4 results of the two tiles are 16 contiguous bytes of the U row (as pack_ffn2);
      * the depthwise record of a chunk zero padded to 2 KiB and the fc2 bias to 1 KiB (every record the kernel copies is then
        a whole number of 1 KiB LDS-DMA pieces)."""
    f = lambda t: t.detach().to(torch.float32).cpu()
    W1, b1, gam, bet = f(fc1_w), f(fc1_b), f(ln_g), f(ln_b)
    Wd, bd, W2, b2 = f(dw_w).reshape(-1, 9), f(dw_b), f(fc2_w), f(fc2_b)
    C_, hid = W2.shape
    assert W1.shape == (2 * hid, C_) and C_ in (144, 180)
    b1 = b1 + W1 @ bet
    W1 = W1 * gam[None, :]
    ks, nt = (C_ + 1 + 31) // 32, (C_ + 15) // 16
    chunks = -(-hid // 32)
    hid_p = 32 * chunks
    z2 = lambda m, rows: torch.cat([m, torch.zeros(rows - m.shape[0], *m.shape[1:])]) if rows > m.shape[0] else m
    W1a, W1g = z2(W1[:hid], hid_p), z2(W1[hid:], hid_p)             # (hid_p, C) each
    b1a, b1g = z2(b1[:hid], hid_p), z2(b1[hid:], hid_p)
    Wd10 = torch.cat([Wd, bd[:, None]], dim=1)                      # (2*hid, 10): taps 0..8 + the depthwise bias as "tap 9"
    Wda, Wdg = z2(Wd10[:hid], hid_p), z2(Wd10[hid:], hid_p)
    lane = torch.arange(64)
    n16, g4, j8 = lane & 15, lane >> 4, torch.arange(8)
    nl = torch.arange(64)                                           # fc1 output row of a chunk: tiles 0, 1 = a half; 2, 3 = gate half
    q = 8 * ((nl & 15) >> 2) + 4 * ((nl >> 4) & 1) + (nl & 3)       # hidden unit of its half, chunk-local
    unit = torch.arange(chunks)[:, None] * 32 + q[None, :]          # (chunks, 64)
    half = (nl >> 5)[None, :].expand(chunks, 64)
    Wp = torch.zeros(2, hid_p, ks * 32)
    Wp[0, :, :C_], Wp[1, :, :C_] = W1a, W1g
    Wp[0, :, C_], Wp[1, :, C_] = b1a, b1g                           # the bias column (the kernel keeps 1.0 in k-slot C)
    rows = Wp[half, unit]                                           # (chunks, 64, ks*32)
    r = rows.reshape(chunks, 4, 16, ks * 32)[:, :, n16]             # (chunks, 4, 64 lanes, K)
    col = (torch.arange(ks)[:, None, None] * 32 + 8 * g4[None, :, None] + j8[None, None, :])   # (ks, 64, 8)
    w1f = torch.stack([r[:, :, lane[:, None], col[k]] for k in range(ks)], dim=2)                   # (chunks, 4, ks, 64, 8)
    # depthwise: [chunk][g][tap 0..8, bias][a-units 8g..8g+7 | gate-units 8g..8g+7], padded to 1024 elements per chunk
    u8 = (torch.arange(chunks)[:, None, None] * 32 + 8 * torch.arange(4)[None, :, None] + j8[None, None, :])   # (chunks, 4, 8)
    dww = torch.cat([Wda[u8].permute(0, 1, 3, 2), Wdg[u8].permute(0, 1, 3, 2)], dim=-1)         # (chunks, 4, 10, 16)
    dw = torch.zeros(chunks, 1024)
    dw[:, :640] = dww.reshape(chunks, 640)
    W2p = torch.zeros(nt * 16, hid_p)
    W2p[:C_, :hid] = W2
    n_i = torch.arange(nt)[:, None] * 16 + n16[None, :]
    k_i = torch.arange(chunks)[:, None, None] * 32 + (8 * g4[:, None] + j8[None, :])[None]
    w2f = W2p[n_i[None, :, :, None].expand(chunks, nt, 64, 8), k_i[:, None].expand(chunks, nt, 64, 8)]
    b2p = torch.zeros(256)
    b2p[:C_] = b2
    p = PackedFFN()
    p.w1f = w1f.to(torch.bfloat16).contiguous().to(device)
    p.w2f = w2f.to(torch.float16).contiguous().to(device)
    p.dww = dw.to(torch.float16).contiguous().to(device)
    p.b1, p.dwb, p.b2 = torch.zeros(4, device=device), torch.zeros(4, device=device), b2p.to(device)
    p.chunks, p.C, p.hid, p.nt, p.ks, p.khalf = chunks, C_, hid, nt, ks, "v3"
    return p


def _ffn_desc(pf: PackedFFN, B, H, W, dtype):
    d = HatFfnDesc()
    d.B, d.H, d.W, d.C, d.chunks, d.dtype = B, H, W, pf.C, pf.chunks, dtype
    return d


def ffn_tiles(pf: PackedFFN, H: int, W: int, dtype: int) -> int:
    lib = _lib.load()
    d = _ffn_desc(pf, 1, H, W, dtype)
    n = C.c_int32(0)
    _lib.check(lib.hat_ffn_tiles(C.byref(d), C.byref(n)), "hat_ffn_tiles")
    return n.value


def ffn_m_ld(C_: int) -> int:
    """Row length (elements) of the pre-normalised input image hat_ffn's m_in expects: [LN (C) | 1.0 | zeros]."""
    return (C_ + 1 + 31) // 32 * 32


def _fill_ffn(d, pf: PackedFFN, t_in, t_out, ln_g, ln_b, ln1, n_out, ldn, gap_out, gap_c, m_in=None, ldm_in=0, n16_out=None):
    d.t_in, d.t_out, d.ln_g, d.ln_b = _ptr(t_in), _ptr(t_out), _ptr(ln_g), _ptr(ln_b)
    if m_in is not None:
        d.m_in, d.ldm_in = _ptr(m_in), ldm_in
    d.w1f, d.b1, d.dww, d.dwb, d.w2f, d.b2 = _ptr(pf.w1f), _ptr(pf.b1), _ptr(pf.dww), _ptr(pf.dwb), _ptr(pf.w2f), _ptr(pf.b2)
    if ln1 is not None:
        d.ln1_g, d.ln1_b, d.n_out, d.ldn = _ptr(ln1[0]), _ptr(ln1[1]), _ptr(n_out), ldn
        d.gap_out, d.gap_c = (_ptr(gap_out) if gap_c else None), gap_c
        d.n16_out = _ptr(n16_out)


def ffn(pf: PackedFFN, t_in, t_out, ln_g, ln_b, *, B: int, H: int, W: int, dtype: int, ln1=None, n_out=None, ldn: int = 0,
        gap_out=None, gap_c: int = 0, m_in=None, ldm_in: int = 0):
    lib = _lib.load()
    d = _ffn_desc(pf, B, H, W, dtype)
    _fill_ffn(d, pf, t_in, t_out, ln_g, ln_b, ln1, n_out, ldn, gap_out, gap_c, m_in, ldm_in)
    flops = B * H * W * (2.0 * pf.C * 2 * pf.hid + 2.0 * 9 * 2 * pf.hid + 2.0 * pf.hid * pf.C)
    # algorithmic HBM bytes per pixel: t_in read once (fp32), t_out written (fp32), the next block's LayerNorm output
    # written (T); weights and the on-chip hidden tensor do not count
    es = 2 if dtype == HAT_BF16 else 4
    # (with m_in, LayerNorm2(t_in) arrives pre-computed as T rows: it replaces the haloed fp32 read; t_in is still read
    # once for the residual)
    nbytes = B * H * W * (4.0 * pf.C + 4.0 * pf.C + (es * ldn if ln1 is not None else 0) + (es * ldm_in if m_in is not None else 0))
    if pf.khalf == "v3":
        raise RuntimeError("a hat_hab_tail3-packed FFN has no stand-alone launch: pack with pack_ffn2 for hat_ffn2")
    if pf.khalf == "v2":
        _timed("ffn2_kernel", flops, lambda: _lib.check(lib.hat_ffn2(C.byref(d), _stream()), "hat_ffn2"), nbytes=nbytes)
        return
    _timed(f"ffn_kernel<{_TNAME[dtype]}>", flops, lambda: _lib.check(lib.hat_ffn(C.byref(d), _stream()), "hat_ffn"),
           nbytes=nbytes)


def hab_tail_supported(pf: PackedFFN, aggr: PackedConv, mid: int, dtype: int) -> bool:
    if pf.C == 180:   # hat_hab_tail3 with the CAB's c2 as a map (no fold: mid is 60)
        return pf.khalf == "v3" and aggr.frag and aggr.nt == 12 and aggr.n_slices == 1 and aggr.kpad == 192 and dtype == HAT_BF16
    return pf.khalf in ("v2", "v3") and aggr.frag and aggr.nt == 9 and aggr.n_slices == 1 and aggr.kpad == 160 and mid <= 8 and dtype == HAT_BF16


def hab_tail(pf: PackedFFN, aggr: PackedConv, t_in, t_out, ln_g, ln_b, *, n, ldn_in: int, y16, c1=None, wf=None, bias_b, B: int, H: int,
             W: int, dtype: int, ln1=None, n_out=None, ldn: int = 0, gap_out=None, gap_c: int = 0, n16_out=None,
             r2=None, ldr2: int = 0, r2scale=None, r2scale_bstride: int = 0):
    """hat_hab_tail: aggregation + folded CAB + residuals + the whole gated FFN in one launch (t_in = the residual stream
    BEFORE the aggregation).  hat_hab_tail3 at embed_dim 144 takes t_in / t_out as fp32 or FP16 rows (by the tensors' dtype)."""
    lib = _lib.load()
    h = HatHabTailDesc()
    d = h.ffn
    d.B, d.H, d.W, d.C, d.chunks, d.dtype = B, H, W, pf.C, pf.chunks, dtype
    _fill_ffn(d, pf, t_in, t_out, ln_g, ln_b, ln1, n_out, ldn, gap_out, gap_c, n16_out=n16_out)
    h.n, h.y16, h.c1, h.w_aggr, h.wf, h.bias_b, h.ldn_in = _ptr(n), _ptr(y16), _ptr(c1), _ptr(aggr.w), _ptr(wf), _ptr(bias_b), ldn_in
    h.r2, h.ldr2, h.r2scale, h.r2scale_bstride = _ptr(r2), ldr2, _ptr(r2scale), r2scale_bstride
    in_half, out_half = t_in.dtype == torch.float16, t_out.dtype == torch.float16
    if (in_half or out_half) and not (pf.C == 144 and pf.khalf == "v3"):
        raise RuntimeError("an FP16 residual stream is only instantiated for hat_hab_tail3 at embed_dim 144")
    if (not in_half and t_in.dtype != torch.float32) or (not out_half and t_out.dtype != torch.float32):
        raise RuntimeError("hab_tail: t_in / t_out must be fp32 or FP16 rows")
    h.reserved1 = (1 if in_half else 0) | (2 if out_half else 0)
    if pf.C == 180:   # aggregation + c2 term + FFN; n, y16, c2 (T), t read once, t_out and the next LayerNorm written
        flops = B * H * W * (2.0 * pf.C * pf.C + 2.0 * pf.C * 2 * pf.hid + 2.0 * 9 * 2 * pf.hid + 2.0 * pf.hid * pf.C)
        nbytes = B * H * W * (2.0 * ldn_in + 32 + 2.0 * ldr2 + 4.0 * pf.C + 4.0 * pf.C + (2 * ldn if ln1 is not None else 0))
        _timed("tail3l_kernel", flops, lambda: _lib.check(lib.hat_hab_tail3(C.byref(h), _stream()), "hat_hab_tail3"), nbytes=nbytes)
        return
    flops = B * H * W * (2.0 * pf.C * (pf.C + 72) + 2.0 * pf.C * 2 * pf.hid + 2.0 * 9 * 2 * pf.hid + 2.0 * pf.hid * pf.C)
    # algorithmic HBM bytes per pixel: n (T), y16 (T x 16), c1 (T x 8), t (fp32) read once; t_out (fp32) and the next
    # block's LayerNorm output (T) written
    nbytes = B * H * W * (2.0 * ldn_in + 32 + 16 + (2.0 if in_half else 4.0) * pf.C + (2.0 if out_half else 4.0) * pf.C + (2 * ldn if ln1 is not None else 0))
    if pf.khalf == "v3":
        _timed("tail3_kernel", flops, lambda: _lib.check(lib.hat_hab_tail3(C.byref(h), _stream()), "hat_hab_tail3"), nbytes=nbytes)
        return
    _timed("ffn2_kernel<aggr>", flops, lambda: _lib.check(lib.hat_hab_tail(C.byref(h), _stream()), "hat_hab_tail"), nbytes=nbytes)


# ------------------------------------------------------------------------------------------------
# pointwise linear layers (hat_linear): fragment-packed weights, weight-stationary streaming GEMM
# ------------------------------------------------------------------------------------------------
_PW_SHAPES = {(9, 5), (18, 5), (9, 9), (12, 6), (23, 6), (12, 12), (4, 1), (4, 2)}  # (nt, ceil(Cin/32)) instantiated in hat_pw.hip


def choose_nt_linear(nout: int, cin: int, dtype: int):
    """hat_linear's n-tiling: like choose_nt, except that a 288-wide output over 144 inputs (OCAB kv and MLP fc1 of the
    embed_dim-144 models) is ONE slice of 18 n-tiles in bf16 (90 KB of weights in LDS): every slice re-reads the input,
    and these layers are HBM-bound."""
    if nout == 288 and -(-cin // 32) == 5 and dtype == HAT_BF16:
        return 18, 1
    if nout == 360 and -(-cin // 32) == 6 and dtype == HAT_BF16:   # the same layers of the embed_dim-180 models: 23 n-tiles, 138 KB
        return 23, 1
    return choose_nt(nout)


def linear_supported(nout: int, cin: int, dtype: int) -> bool:
    nt, _ = choose_nt_linear(nout, cin, dtype)
    ks = -(-cin // 32)
    lds = nt * ks * 64 * 8 * (2 if dtype == HAT_BF16 else 4)
    return (nt, ks) in _PW_SHAPES and lds <= 163840 and cin % 4 == 0


def pack_linear_weight(weight: torch.Tensor, bias: Optional[torch.Tensor], dtype: int, device, scale: float = 1.0) -> PackedConv:
    """weight (O, I) -> MFMA A-fragment order [n_slices][nt][ceil(I/32)][64 lanes][8] for hat_linear.
    A conv weight (O, I, k, k) is first flattened to K = tap * Cin_p + ci (hat_conv3x3_small)."""
    w = weight.detach().to(torch.float32).cpu()
    ksize, cin = 1, None
    if w.dim() == 4 and w.shape[-1] > 1:
        o_, i_, ksize, _ = w.shape
        cin, cin_p = i_, (i_ + 7) // 8 * 8
        wt = torch.zeros(o_, ksize * ksize, cin_p)
        wt[:, :, :i_] = w.permute(0, 2, 3, 1).reshape(o_, ksize * ksize, i_)
        w = wt.reshape(o_, -1)
    w = w.reshape(w.shape[0], -1) * scale
    o, i = w.shape
    b = (torch.zeros(o) if bias is None else bias.detach().to(torch.float32).cpu()) * scale
    nt, n_slices = choose_nt_linear(o, i, dtype) if ksize == 1 else choose_nt(o)
    ks = -(-i // 32)
    npad = nt * 16 * n_slices
    wp = torch.zeros(npad, ks * 32)
    wp[:o, :i] = w
    lane = torch.arange(64)
    row = (torch.arange(n_slices)[:, None, None, None, None] * nt * 16 + torch.arange(nt)[None, :, None, None, None] * 16
           + (lane & 15)[None, None, None, :, None]).expand(n_slices, nt, ks, 64, 8)
    col = (torch.arange(ks)[None, None, :, None, None] * 32 + 8 * (lane >> 4)[None, None, None, :, None]
           + torch.arange(8)[None, None, None, None, :]).expand(n_slices, nt, ks, 64, 8)
    wf = wp[row, col].to(TORCH_DTYPE[dtype]).contiguous().to(device)
    bp = torch.zeros(npad)
    bp[:o] = b
    return PackedConv(wf, bp.to(device), ksize, (i if cin is None else cin), ks * 32, nt, n_slices, o, frag=True)


def linear(pw: PackedConv, x: torch.Tensor, out: torch.Tensor, *, B: int, H: int, W: int, dtype: int, ldx: int, ldo: int,
           out_mode: int = O_NHWC_T, act: int = ACT_NONE, n_store: Optional[int] = None, x0: Optional[torch.Tensor] = None,
           c_split: int = 0, ldx0: int = 0, r1: Optional[torch.Tensor] = None, ldr1: int = 0, r2: Optional[torch.Tensor] = None,
           ldr2: int = 0, r2scale: Optional[torch.Tensor] = None, r2scale_bstride: int = 0, ln=None,
           ln_out: Optional[torch.Tensor] = None, ld_ln: int = 0, ln_ones: bool = False):
    """ln=(gamma, beta), ln_out, ld_ln: also emit LayerNorm(result) as T rows (hat_linear's fused LayerNorm; needs a
    residual operand); ln_ones appends the [1.0, 0...] tail that hat_ffn's m_in expects."""
    lib = _lib.load()
    d = HatConvDesc()
    d.x, d.x0, d.w, d.bias, d.out = _ptr(x), _ptr(x0), _ptr(pw.w), _ptr(pw.bias), _ptr(out)
    d.r1, d.r2, d.r2scale = _ptr(r1), _ptr(r2), _ptr(r2scale)
    if ln is not None:
        d.ln_g, d.ln_b, d.ln_out, d.ld_ln, d.ln_ones = _ptr(ln[0]), _ptr(ln[1]), _ptr(ln_out), ld_ln, int(ln_ones)
    d.B, d.H, d.W, d.Cin, d.ldx, d.x_mode = B, H, W, pw.cin, ldx, X_NHWC_T
    d.c_split, d.ldx0, d.ksize, d.Kpad, d.nt, d.n_slices = c_split, ldx0, 1, pw.kpad, pw.nt, pw.n_slices
    d.n_store = pw.nout if n_store is None else n_store
    d.ldo, d.out_mode, d.act, d.ldr1, d.ldr2, d.r2scale_bstride, d.dtype = ldo, out_mode, act, ldr1, ldr2, r2scale_bstride, dtype
    flops = 2.0 * B * H * W * pw.cin * pw.nout
    _timed(f"pw_kernel<{_TNAME[dtype]}, {pw.nt}, {pw.kpad // 32}>", flops,
           lambda: _lib.check(lib.hat_linear(C.byref(d), _stream()), f"hat_linear(Cin={pw.cin}, N={pw.nout})"),
           tag=f"lin {pw.cin}->{pw.nout} {H}x{W} o{out_mode}{' r1' if r1 is not None else ''}{' r2' if r2 is not None else ''}")


_T3_SHAPES = {(1, 144), (9, 8), (1, 24), (4, 8)}  # (nt, Cin_p) instantiated in tap3_kernel


def conv3x3_small_supported(nout: int, cin: int, dtype: int = HAT_BF16) -> bool:
    nt, n_slices = choose_nt(nout)
    cin_p = (cin + 7) // 8 * 8
    lds = nt * (-(-9 * cin_p // 32)) * 64 * 8 * (2 if dtype == HAT_BF16 else 4)
    return (nt, cin_p) in _T3_SHAPES and n_slices == 1 and lds <= 81920


def _c3_desc(pw, B, H, W, dtype, ldx):
    d = HatConvDesc()
    d.B, d.H, d.W, d.Cin, d.ldx, d.x_mode = B, H, W, pw.cin, ldx, X_NHWC_T
    d.ksize, d.Kpad, d.nt, d.n_slices, d.dtype = 3, pw.kpad, pw.nt, 1, dtype
    return d


def conv3x3_small_groups(pw: PackedConv, B: int, H: int, W: int, dtype: int) -> int:
    lib = _lib.load()
    d = _c3_desc(pw, B, H, W, dtype, 8)
    n = C.c_int32(0)
    _lib.check(lib.hat_conv3x3_small_groups(C.byref(d), C.byref(n)), "hat_conv3x3_small_groups")
    return n.value


def conv3x3_small(pw: PackedConv, x, out, *, B: int, H: int, W: int, dtype: int, ldx: int, ldo: int, act: int = ACT_NONE,
                  n_store: Optional[int] = None, out_mode: int = O_NHWC_T, colsum=None):
    lib = _lib.load()
    d = _c3_desc(pw, B, H, W, dtype, ldx)
    d.x, d.w, d.bias, d.out, d.colsum = _ptr(x), _ptr(pw.w), _ptr(pw.bias), _ptr(out), _ptr(colsum)
    d.n_store = pw.nout if n_store is None else n_store
    d.ldo, d.out_mode, d.act = ldo, out_mode, act
    flops = 2.0 * B * H * W * 9 * pw.cin * pw.nout
    _timed(f"tap3_kernel<{_TNAME[dtype]}, {pw.nt}>", flops,
           lambda: _lib.check(lib.hat_conv3x3_small(C.byref(d), _stream()), f"hat_conv3x3_small(Cin={pw.cin}, N={pw.nout})"),
           tag=f"c3s {pw.cin}->{pw.nout} {H}x{W}")


# ------------------------------------------------------------------------------------------------
# CAB expand conv + ECA folded into the ESC aggregation (hat_cab_fold + hat_aggr_cab)
# ------------------------------------------------------------------------------------------------
def aggr_cab_supported(C_: int, mid: int, dtype: int) -> bool:
    return C_ == 144 and mid <= 8 and dtype == HAT_BF16


def pack_cab_w2f(w2: torch.Tensor, device) -> torch.Tensor:
    """(C, mid <= 8, 3, 3) expand-conv weight -> fp32 [nt][3][64][8] in the order of hat_cab_fold's output `wf` (HatCabFoldDesc.w2f)."""
    w = w2.detach().to(torch.float32).cpu()
    C_, mid = w.shape[0], w.shape[1]
    nt = -(-C_ // 16)
    full = torch.zeros(nt * 16, 12, 8)                     # [co][tap 0..11][ci 0..7]
    full[:C_, :9, :mid] = w.reshape(C_, mid, 9).permute(0, 2, 1)
    lane = torch.arange(64)
    t = torch.arange(nt)[:, None, None, None]
    ks = torch.arange(3)[None, :, None, None]
    co = (t * 16 + (lane & 15)[None, None, :, None]).expand(nt, 3, 64, 8)
    tap = (4 * ks + (lane >> 4)[None, None, :, None]).expand(nt, 3, 64, 8)
    ci = torch.arange(8)[None, None, None, :].expand(nt, 3, 64, 8)
    return full[co, tap, ci].contiguous().to(device)


def cab_fold(c1, c1_colsum, tiles: int, ldcs: int, w2, b2, wk, k: int, bias_in, conv_scale: float, scale, wf, bias_out, tmp, *,
             B: int, H: int, W: int, C_: int, mid: int, dtype: int, stats=None, w2f=None):
    """stats: (B, >= 72) fp32 = the frame-wide sums of c1 ([total 8 | first row | last row | first column | last column | four
    corner pixels]); H, W are then the FULL frame's and c1 / c1_colsum are not read (band-sharded frames)."""
    lib = _lib.load()
    d = HatCabFoldDesc()
    d.stats = _ptr(stats)
    d.w2f = _ptr(w2f)
    d.c1, d.c1_colsum, d.w2, d.b2, d.wk, d.bias_in = _ptr(c1), _ptr(c1_colsum), _ptr(w2), _ptr(b2), _ptr(wk), _ptr(bias_in)
    d.scale, d.wf, d.bias_out, d.tmp = _ptr(scale), _ptr(wf), _ptr(bias_out), _ptr(tmp)
    d.B, d.H, d.W, d.C, d.mid, d.ld1, d.tiles, d.ldcs, d.k, d.ld_scale, d.dtype = B, H, W, C_, mid, 8, tiles, ldcs, k, scale.shape[1], dtype
    d.conv_scale = conv_scale
    _timed("cab_fold", 0.0, lambda: _lib.check(lib.hat_cab_fold(C.byref(d), _stream()), "hat_cab_fold"))


def aggr_cab(pw: PackedConv, x, out, c1, wf, bias_b, *, B: int, H: int, W: int, dtype: int, ldx: int, ldo: int, x0=None,
             c_split: int = 0, ldx0: int = 0, r1=None, ldr1: int = 0):
    lib = _lib.load()
    dd = HatAggrCabDesc()
    d = dd.lin
    d.x, d.x0, d.w, d.bias, d.out, d.r1 = _ptr(x), _ptr(x0), _ptr(pw.w), _ptr(pw.bias), _ptr(out), _ptr(r1)
    d.B, d.H, d.W, d.Cin, d.ldx, d.x_mode = B, H, W, pw.cin, ldx, X_NHWC_T
    d.c_split, d.ldx0, d.ksize, d.Kpad, d.nt, d.n_slices, d.n_store = c_split, ldx0, 1, pw.kpad, pw.nt, pw.n_slices, pw.nout
    d.ldo, d.out_mode, d.act, d.ldr1, d.dtype = ldo, O_NHWC_F32, ACT_NONE, ldr1, dtype
    dd.c1, dd.wf, dd.bias_b = _ptr(c1), _ptr(wf), _ptr(bias_b)
    flops = 2.0 * B * H * W * pw.nout * (pw.cin + 9 * 8)
    _timed("aggr_cab_kernel", flops, lambda: _lib.check(lib.hat_aggr_cab(C.byref(dd), _stream()), "hat_aggr_cab"),
           tag=f"aggr+cab {pw.cin}+72->{pw.nout} {H}x{W}")
