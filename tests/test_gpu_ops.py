"""GPU parity tests, per kernel, THROUGH THE C ABI (super_resolution_amd.ops -> libhat_mi355x.so),
against the CPU oracle / plain torch fp64 restatements of the same op on the same seeded inputs.

Tolerances: HAT_F32 path (exact-fp32 MFMA): max-abs <= 2e-5 * scale (accumulation-order noise only);
HAT_BF16 path: relative L2 error <= 1.2e-2 and max-abs <= 6e-2 * scale (bf16 operands, fp32 accumulate).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import hat_oracle as O
from super_resolution_amd import synth

pytestmark = pytest.mark.gpu

DT = ["f32", "bf16"]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _ops():
    from super_resolution_amd import ops
    return ops


def _r8(x):
    return (x + 7) // 8 * 8


def to_dev(x_bhwc: torch.Tensor, ld: int, tdt, dev):
    """(B,H,W,C) float -> device (B, H*W, ld) of dtype tdt with zero pad channels."""
    b, h, w, c = x_bhwc.shape
    out = torch.zeros(b, h * w, ld, dtype=tdt, device=dev)
    out[:, :, :c] = x_bhwc.reshape(b, h * w, c).to(dev).to(tdt)
    return out


def check(got: torch.Tensor, ref: torch.Tensor, dtype: str, what: str, f32_tol=2e-5):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ": non-finite output"
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    rel = float((got - ref).norm() / max(float(ref.norm()), 1e-12))
    if dtype == "f32":
        assert err <= f32_tol * max(scale, 1.0), f"{what}: max-abs {err:.3e} (scale {scale:.3g}, rel {rel:.3e})"
    else:
        assert rel <= 1.2e-2 and err <= 6e-2 * max(scale, 1.0), f"{what}: rel {rel:.3e} max-abs {err:.3e} (scale {scale:.3g})"


def rnd(key, shape, std=1.0):
    return synth.normal(11, key, shape, std=std)


def q(x, dtype):
    """Round inputs the way the kernel's storage type does, so the oracle sees the same operands."""
    return x.to(torch.bfloat16).to(torch.float32) if dtype == "bf16" else x


# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # name, B, H, W, Cin, Cout, k, act
    ("lin144", 1, 16, 32, 144, 144, 1, 0),
    ("fc1_576", 1, 24, 16, 144, 576, 1, 0),
    ("fc2_288", 2, 16, 16, 288, 144, 1, 1),
    ("lin180", 1, 16, 24, 180, 180, 1, 0),
    ("lin360_180", 1, 8, 40, 360, 180, 1, 0),
    ("lin24", 1, 8, 24, 24, 48, 1, 1),
    ("cab0_144_6", 1, 32, 16, 144, 6, 3, 1),
    ("cab2_6_144", 1, 16, 48, 6, 144, 3, 0),
    ("conv144", 1, 32, 32, 144, 144, 3, 0),
    ("conv180_60", 1, 16, 16, 180, 60, 3, 1),
    ("conv180", 2, 16, 16, 180, 180, 3, 0),
    ("conv_c64", 1, 20, 28, 144, 64, 3, 2),
    ("lk13", 1, 32, 32, 16, 16, 13, 0),
    ("lk5", 1, 16, 24, 8, 8, 5, 0),
    ("small_3x5", 1, 3, 5, 24, 24, 3, 0),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_plain(case, dtype):
    name, B, H, W, Cin, Cout, k, act = case
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    x = q(rnd(name + "x", (B, H, W, Cin)), dtype)
    wgt = q(rnd(name + "w", (Cout, Cin, k, k), std=(Cin * k * k) ** -0.5), dtype)
    bias = rnd(name + "b", (Cout,), std=0.1)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wgt.double(), bias.double(), padding=k // 2)
    ref = {0: lambda v: v, 1: F.gelu, 2: lambda v: F.leaky_relu(v, 0.01)}[act](ref).permute(0, 2, 3, 1)
    pw = ops.pack_conv_weight(wgt, bias, dt, dev)
    ldx, ldo = _r8(Cin), _r8(Cout)
    xd = to_dev(x, ldx, tdt, dev)
    out = torch.zeros(B, H * W, ldo, dtype=tdt, device=dev)
    ops.conv(pw, xd, out, B=B, H=H, W=W, dtype=dt, ldx=ldx, ldo=ldo, act=act, n_store=(Cout + 3) // 4 * 4)
    torch.cuda.synchronize()
    got = out[:, :, :Cout].float().reshape(B, H, W, Cout)
    if dtype == "bf16":
        ref = ref.float().to(torch.bfloat16).double()
    check(got, ref, dtype, name)
    assert float(out[:, :, (Cout + 3) // 4 * 4:].float().abs().max() if ldo > (Cout + 3) // 4 * 4 else 0.0) == 0.0


@pytest.mark.parametrize("dtype", DT)
def test_conv_residual_f32_and_r2_and_colsum(dtype):
    """Epilogue: out_f32 = acc + bias + r1 + scale[n] * r2, split input source, per-tile column sums."""
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    B, H, W, C = 2, 24, 40, 144
    x = q(rnd("rx", (B, H, W, C)), dtype)
    x0 = q(rnd("rx0", (B, H, W, 16)), dtype)
    wgt = q(rnd("rw", (C, C, 1, 1), std=C ** -0.5), dtype)
    bias = rnd("rb", (C,), std=0.1)
    r1 = rnd("r1", (B, H, W, C))
    r2 = q(rnd("r2", (B, H, W, C)), dtype)
    sc = rnd("sc", (B, C), std=0.3)
    xin = torch.cat([x0, x[..., 16:]], -1)
    lin = F.conv2d(xin.permute(0, 3, 1, 2).double(), wgt.double(), bias.double()).permute(0, 2, 3, 1)
    ref = lin + r1.double() + sc.double()[:, None, None, :] * r2.double()
    pw = ops.pack_conv_weight(wgt, bias, dt, dev)
    out = torch.zeros(B, H * W, C, dtype=torch.float32, device=dev)
    tiles = ops.conv_tiles(pw, H, W, dt)
    colsum = torch.zeros(B, tiles, pw.npad, dtype=torch.float32, device=dev)
    scd = torch.zeros(B, pw.npad, device=dev)
    scd[:, :C] = sc.to(dev)
    ops.conv(pw, to_dev(x, C, tdt, dev), out, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C, out_mode=ops.O_NHWC_F32,
             x0=to_dev(x0, 16, tdt, dev), c_split=16, ldx0=16, r1=r1.reshape(B, H * W, C).to(dev).contiguous(), ldr1=C,
             r2=to_dev(r2, C, tdt, dev), ldr2=C, r2scale=scd, r2scale_bstride=pw.npad, colsum=colsum)
    torch.cuda.synchronize()
    check(out.reshape(B, H, W, C), ref, dtype, "residual epilogue", f32_tol=3e-5)
    got_cs = colsum.sum(1)[:, :C]
    check(got_cs / (H * W), ref.sum((1, 2)) / (H * W), dtype, "column sums", f32_tol=3e-5)


@pytest.mark.parametrize("dtype", DT)
def test_conv_f32_source_inplace_residual(dtype):
    """RHAG tail (hat_arch.py:556): x read as fp32 tokens, out = conv3x3(x) + r1 written over r1."""
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    B, H, W, C = 1, 32, 16, 144
    x = rnd("fx", (B, H, W, C))
    wgt = q(rnd("fw", (C, C, 3, 3), std=(9 * C) ** -0.5), dtype)
    bias = rnd("fb", (C,), std=0.1)
    r1 = rnd("fr", (B, H, W, C))
    ref = F.conv2d(q(x, dtype).permute(0, 3, 1, 2).double(), wgt.double(), bias.double(), padding=1).permute(0, 2, 3, 1) + r1.double()
    pw = ops.pack_conv_weight(wgt, bias, dt, dev)
    xd = x.reshape(B, H * W, C).to(dev).contiguous()
    rd = r1.reshape(B, H * W, C).to(dev).contiguous()
    ops.conv(pw, xd, rd, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C, x_mode=ops.X_NHWC_F32, out_mode=ops.O_NHWC_F32, r1=rd, ldr1=C)
    torch.cuda.synchronize()
    check(rd.reshape(B, H, W, C), ref, dtype, "f32-source conv + in-place residual", f32_tol=3e-5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geom", [(1, 32, 48, 16), (2, 19, 21, 16), (1, 16, 16, 0)], ids=["32x48", "B2_ragged_19x21", "no_gap"])
def test_conv_group_tail_with_fused_layernorm(geom, dtype):
    """RHAG tail with the consumer's LayerNorm in the epilogue (hat_arch.py:556 then :214): out = conv3x3(x) + r1 in place,
    ln_out = LayerNorm(out), gap_out = per-tile sums of its first gap_c channels, n16_out = its first 16 channels."""
    dev, ops = _dev(), _ops()
    dt, tdt = ops.DTYPE_CODE[dtype], ops.TORCH_DTYPE[ops.DTYPE_CODE[dtype]]
    B, H, W, gap_c = geom
    C = 144
    x = q(rnd("lx", (B, H, W, C)), dtype)
    wgt = q(rnd("lw", (C, C, 3, 3), std=(9 * C) ** -0.5), dtype)
    bias = rnd("lb", (C,), std=0.1)
    r1 = rnd("lr", (B, H, W, C))
    g, bt = 1.0 + rnd("lg", (C,), std=0.2), rnd("lbt", (C,), std=0.2)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wgt.double(), bias.double(), padding=1).permute(0, 2, 3, 1) + r1.double()
    ref_ln = F.layer_norm(ref, (C,), g.double(), bt.double(), 1e-5)
    pw = ops.pack_conv_weight(wgt, bias, dt, dev)
    xd = to_dev(x, C, tdt, dev)
    rd = r1.reshape(B, H * W, C).to(dev).contiguous()
    tiles = ops.conv_tiles(pw, H, W, dt)
    n_out = torch.full((B, H * W, C), float("nan"), dtype=tdt, device=dev)
    n16 = torch.full((B, H * W, 16), float("nan"), dtype=tdt, device=dev)
    gap = torch.full((B, tiles, 16), float("nan"), dtype=torch.float32, device=dev)
    ops.conv(pw, xd, rd, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C, out_mode=ops.O_NHWC_F32, r1=rd, ldr1=C,
             ln=(g.to(dev), bt.to(dev)), ln_out=n_out, ld_ln=C, gap_out=(gap if gap_c else None), gap_c=gap_c,
             n16_out=(n16 if gap_c else None))
    torch.cuda.synchronize()
    check(rd.reshape(B, H, W, C), ref, dtype, "conv + in-place residual", f32_tol=3e-5)
    check(n_out.reshape(B, H, W, C), ref_ln, dtype, "fused LayerNorm of the conv output", f32_tol=6e-5)
    if gap_c:
        assert torch.equal(n16.reshape(B, H, W, 16), n_out.reshape(B, H, W, C)[..., :16]), "compact 16-channel copy"
        got = gap.double().sum(1).cpu() / (H * W)                      # mean over pixels, as hat_esc_weights forms it
        want = n_out.reshape(B, H * W, C)[..., :16].double().mean(1).cpu()
        want[:, gap_c:] = 0
        assert torch.allclose(got, want, atol=2e-3 if dtype == "bf16" else 1e-5), (got - want).abs().max()


@pytest.mark.parametrize("dtype", DT)
def test_conv_first_nchw_mean(dtype):
    """(x - mean) * img_range then conv_first (hat_arch.py:849-853): NCHW fp32 in, fp32 tokens out."""
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    B, H, W, C = 2, 24, 40, 144
    x = synth.uniform(3, "cfx", (B, 3, H, W))
    wgt = q(rnd("cfw", (C, 3, 3, 3), std=27 ** -0.5), dtype)
    bias = rnd("cfb", (C,), std=0.1)
    mean = torch.tensor(O.RGB_MEAN).view(1, 3, 1, 1)
    xin = q((x - mean) * 1.0, dtype)
    ref = F.conv2d(xin.double(), wgt.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    pw = ops.pack_conv_weight(wgt, bias, dt, dev)
    out = torch.zeros(B, H * W, C, dtype=torch.float32, device=dev)
    ops.conv(pw, x.to(dev), out, B=B, H=H, W=W, dtype=dt, ldx=0, ldo=C, x_mode=ops.X_NCHW_F32_MEAN, out_mode=ops.O_NHWC_F32,
             in_scale=1.0, mean=O.RGB_MEAN)
    torch.cuda.synchronize()
    check(out.reshape(B, H, W, C), ref, dtype, "conv_first")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("r", [2, 3])
def test_conv_pixelshuffle_and_last(dtype, r):
    """Upsample conv + PixelShuffle folded into the store, then conv_last to NCHW (hat_arch.py:593-605,856-858)."""
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    B, H, W = 1, 12, 20
    x = q(rnd("px", (B, H, W, 64)), dtype)
    w1 = q(rnd("pw", (64 * r * r, 64, 3, 3), std=576 ** -0.5), dtype)
    b1 = rnd("pb", (64 * r * r,), std=0.1)
    w2 = q(rnd("lw", (3, 64, 3, 3), std=576 ** -0.5), dtype)
    b2 = rnd("lb", (3,), std=0.1)
    up = F.pixel_shuffle(F.conv2d(x.permute(0, 3, 1, 2).double(), w1.double(), b1.double(), padding=1), r)
    n = torch.arange(64 * r * r)
    perm = (n % 64) * (r * r) + n // 64
    pw1 = ops.pack_conv_weight(w1, b1, dt, dev, out_perm=perm)
    mid = torch.zeros(B, H * r * W * r, 64, dtype=tdt, device=dev)
    ops.conv(pw1, to_dev(x, 64, tdt, dev), mid, B=B, H=H, W=W, dtype=dt, ldx=64, ldo=64, out_mode=ops.O_PIXSHUF_T, ps_r=r)
    torch.cuda.synchronize()
    check(mid.float().reshape(B, H * r, W * r, 64), q(up.float(), dtype).double().permute(0, 2, 3, 1), dtype, "pixel shuffle")
    up_q = mid.float().cpu().reshape(B, H * r, W * r, 64).permute(0, 3, 1, 2)
    mean = torch.tensor(O.RGB_MEAN).view(1, 3, 1, 1)
    ref = F.conv2d(up_q.double(), w2.double(), b2.double(), padding=1) / 2.0 + mean.double()
    pw2 = ops.pack_conv_weight(w2, b2, dt, dev)
    y = torch.zeros(B, 3, H * r, W * r, dtype=torch.float32, device=dev)
    ops.conv(pw2, mid, y, B=B, H=H * r, W=W * r, dtype=dt, ldx=64, ldo=0, out_mode=ops.O_NCHW_F32, out_scale=0.5, mean=O.RGB_MEAN)
    torch.cuda.synchronize()
    check(y, ref, dtype, "conv_last")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C", [24, 144, 180])
def test_layernorm(dtype, C):
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    B, N = 2, 1000
    x = rnd("lnx", (B, N, C), std=2.0) + 0.5
    gm, bt = 1 + rnd("lng", (C,), std=0.1), rnd("lnb", (C,), std=0.1)
    ref = F.layer_norm(x.double(), (C,), gm.double(), bt.double(), 1e-5)
    ld = _r8(C)
    y = torch.zeros(B, N, ld, dtype=tdt, device=dev)
    gap_c = {24: 8, 144: 16, 180: 24}[C]   # 24: more than 16 pooled channels -> blocks of 32 floats (wide ESC of HATX)
    gap = torch.zeros(B, ops.layernorm_blocks(), 32 if gap_c > 16 else 16, device=dev)
    ops.layernorm(x.to(dev), y, gm.to(dev), bt.to(dev), B=B, npix=N, C_=C, ldy=ld, out_f32=False, dtype=dt, gap=gap, gap_c=gap_c)
    yf = torch.zeros(B, N, C, device=dev)
    ops.layernorm(x.to(dev), yf, gm.to(dev), bt.to(dev), B=B, npix=N, C_=C, ldy=C, out_f32=True, dtype=dt)
    torch.cuda.synchronize()
    check(y[:, :, :C].float(), ref, dtype, "layernorm -> T")
    check(yf, ref, "f32", "layernorm -> fp32")
    # the pool adds the STORED values (T-rounded: the tensor the ESC conv then reads; a band-sharded frame pools the same rows)
    check(gap.sum(1)[:, :gap_c] / N, y[:, :, :gap_c].double().cpu().mean(1), "f32", "gap partial sums = pool of the stored map", f32_tol=1e-5)
    check(gap.sum(1)[:, :gap_c] / N, ref[:, :, :gap_c].mean(1), "f32", "gap partial sums vs the unrounded LayerNorm", f32_tol=1e-5 if dtype == "f32" else 1e-3)


@pytest.mark.parametrize("geom", [(2, 24, 40, True), (1, 19, 27, False), (1, 3, 5, True)], ids=["B2_24x40_f32out", "ragged_19x27_bf16out", "15px"])
def test_ocab_mlp_fused(geom):
    """hat_ocab_mlp (hat_arch.py:309-313 + the residual of :391): out = r1 + fc2(GELU(fc1(x))) for C = 144, hidden 288, bf16
    storage, against fp64 torch; fp32 output in place over r1, and T rows (the group conv's input)."""
    B, H, W, f32out = geom
    dev, ops = _dev(), _ops()
    C, hid = 144, 288
    x = q(rnd("mx", (B, H, W, C)), "bf16")
    w1, b1 = q(rnd("mw1", (hid, C), std=C ** -0.5), "bf16"), rnd("mb1", (hid,), std=0.2)
    w2, b2 = q(rnd("mw2", (C, hid), std=hid ** -0.5), "bf16"), rnd("mb2", (C,), std=0.2)
    r1 = rnd("mr1", (B, H, W, C))
    hdn = F.gelu(x.double() @ w1.double().t() + b1.double())
    ref = r1.double() + hdn @ w2.double().t() + b2.double()
    pm = ops.pack_ocab_mlp(w1, b1, w2, b2, dev)
    xd = to_dev(x, C, torch.bfloat16, dev)
    rd = r1.reshape(B, H * W, C).to(dev).contiguous()
    if f32out:
        ops.ocab_mlp(pm, xd, rd, rd, B=B, H=H, W=W, ldx=C, ldr1=C, ldo=C, out_f32=True, dtype=ops.HAT_BF16)
        got = rd
    else:
        got = torch.full((B, H * W, C), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.ocab_mlp(pm, xd, rd, got, B=B, H=H, W=W, ldx=C, ldr1=C, ldo=C, out_f32=False, dtype=ops.HAT_BF16)
    torch.cuda.synchronize()
    check(got.float().reshape(B, H, W, C), ref, "bf16", "fused OCAB MLP")


def test_ocab_qkv_fused():
    """hat_ocab_qkv: [q * d^-0.5 | k | v] = both OCAB projections (hat_arch.py:347, :350, :375) of one 144-channel map, bf16."""
    dev, ops = _dev(), _ops()
    B, H, W, C = 2, 19, 27, 144
    x = q(rnd("qx", (B, H, W, C)), "bf16")
    wq, bq = rnd("qw", (C, C), std=C ** -0.5), rnd("qb", (C,), std=0.2)
    wkv, bkv = rnd("kvw", (2 * C, C), std=C ** -0.5), rnd("kvb", (2 * C,), std=0.2)
    sc = 24 ** -0.5
    ref = torch.cat([(x.double() @ q(wq * sc, "bf16").double().t() + bq.double() * sc), x.double() @ q(wkv, "bf16").double().t() + bkv.double()], -1)
    pm = ops.pack_ocab_qkv(wq, bq, wkv, bkv, sc, dev)
    out = torch.full((B, H * W, 432), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.ocab_qkv(pm, to_dev(x, C, torch.bfloat16, dev), out, B=B, H=H, W=W, ldx=C, ldo=432, dtype=ops.HAT_BF16)
    torch.cuda.synchronize()
    check(out.float().reshape(B, H, W, 432), ref, "bf16", "fused q / kv projection")


@pytest.mark.parametrize("dtype", DT)
def test_dwconv_gate(dtype):
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    B, H, W, hid = 2, 19, 23, 288
    u = q(rnd("dwu", (B, H, W, 2 * hid)), dtype)
    wd, bd = rnd("dww", (2 * hid, 1, 3, 3), std=1 / 3), rnd("dwb", (2 * hid,), std=0.1)
    v = F.conv2d(u.permute(0, 3, 1, 2).double(), wd.double(), bd.double(), padding=1, groups=2 * hid).permute(0, 2, 3, 1)
    a, g = v.chunk(2, dim=-1)
    ref = a * F.silu(g)
    out = torch.zeros(B, H * W, hid, dtype=tdt, device=dev)
    ops.dwconv_gate(to_dev(u, 2 * hid, tdt, dev), wd.reshape(2 * hid, 9).t().contiguous().to(dev), bd.to(dev), out, B=B, H=H,
                    W=W, hid=hid, ldu=2 * hid, ldo=hid, dtype=dt)
    torch.cuda.synchronize()
    check(out.float().reshape(B, H, W, hid), ref, dtype, "dwconv+gate")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geom", [(16, 6, 144, 32, 48, 0.5), (16, 6, 180, 32, 16, 0.5), (8, 2, 24, 16, 24, 0.5), (8, 2, 48, 16, 24, 0.7),
                                  (16, 6, 144, 32, 32, 0.6)],
                         ids=["ws16_d24", "ws16_d30", "ws8_d12", "ws8_odd13_d24", "ws16_odd25_d24"])
def test_ocab_attention(dtype, geom):
    """hat_ocab_attention against the oracle's attention core; the last two geometries have ODD key windows (HATX: ceil
    padding, hatx_arch.py:303-305; key tiles padded with dead keys)."""
    ws, heads, C, H, W, ov = geom
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    B = 2
    wse = int(ws * ov) + ws
    d = C // heads
    qv = q(rnd("aq", (B, H, W, C)) * d ** -0.5, dtype)
    kv = q(rnd("akv", (B, H, W, 2 * C)), dtype)
    table = rnd("atab", ((ws + wse - 1) ** 2, heads), std=0.5)
    rpi = O.rpi_oca(ws, ov)
    attn = O.hatx_ocab_attention if wse % 2 else O.ocab_attention     # (identical for even overlaps; HATX's pads odd ones)
    ref = attn(qv.double(), kv[..., :C].double(), kv[..., C:].double(), table.double(), rpi, ws, wse, heads, 1.0)
    M = ws + wse - 1
    shift = (ws - wse + 1 - (ws - 1)) * (M + 1)
    rot = (torch.arange(M * M) + shift) % (M * M)
    bias_rot = table[rot].t().contiguous().to(dev)
    out = torch.zeros(B, H * W, _r8(C), dtype=tdt, device=dev)
    ops.ocab_attention(to_dev(qv, _r8(C), tdt, dev), to_dev(kv, _r8(2 * C), tdt, dev), bias_rot, out, B=B, H=H, W=W, C_=C,
                       heads=heads, ws=ws, wse=wse, ldq=_r8(C), ldkv=_r8(2 * C), ldo=_r8(C), dtype=dt)
    torch.cuda.synchronize()
    check(out[:, :, :C].float().reshape(B, H, W, C), ref, dtype, "ocab attention", f32_tol=5e-5)


def test_ocab_attention_softmax_spike():
    """A large logit in a late key chunk forces the online-softmax rescale branch (guide rule 26)."""
    dev, ops = _dev(), _ops()
    ws, heads, C, H, W, B = 16, 6, 144, 16, 16, 1
    wse, d = 24, 24
    qv = rnd("sq", (B, H, W, C)) * d ** -0.5
    kv = rnd("skv", (B, H, W, 2 * C))
    kv[0, 15, 15, :C] *= 40.0  # bottom-right key: last key chunk of the window
    table = rnd("stab", ((ws + wse - 1) ** 2, heads), std=0.5)
    ref = O.ocab_attention(qv.double(), kv[..., :C].double(), kv[..., C:].double(), table.double(), O.rpi_oca(ws, 0.5), ws, wse, heads, 1.0)
    M = ws + wse - 1
    rot = (torch.arange(M * M) + (ws - wse + 1 - (ws - 1)) * (M + 1)) % (M * M)
    out = torch.zeros(B, H * W, C, device=dev)
    ops.ocab_attention(qv.reshape(B, H * W, C).to(dev), kv.reshape(B, H * W, 2 * C).to(dev), table[rot].t().contiguous().to(dev),
                       out, B=B, H=H, W=W, C_=C, heads=heads, ws=ws, wse=wse, ldq=C, ldkv=2 * C, ldo=C, dtype=ops.HAT_F32)
    torch.cuda.synchronize()
    check(out.reshape(B, H, W, C), ref, "f32", "attention with a logit spike", f32_tol=1e-4)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("pdim,ks", [(16, 13), (8, 5), (24, 15), (32, 17)])
def test_esc_weights_and_eca(dtype, pdim, ks):
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    B, N, nblk = 2, 777, ops.layernorm_blocks()
    npad = 16 if pdim <= 16 else 32   # weight rows per sample and floats per GAP block
    gap = rnd("gp", (B, nblk, npad), std=1.0)
    gap[:, :, pdim:] = 0
    w1, b1 = rnd("w1", (pdim // 2, pdim), std=0.3), rnd("b1", (pdim // 2,), std=0.1)
    w2, b2 = rnd("w2", (pdim * 9, pdim // 2), std=0.3), rnd("b2", (pdim * 9,), std=0.1)
    plk = rnd("plk", (pdim, pdim, ks, ks), std=0.05)
    p = gap.double().sum(1)[:, :pdim] / N
    h = F.gelu(p @ w1.double().t() + b1.double())
    dk = (h @ w2.double().t() + b2.double()).reshape(B, pdim, 3, 3)
    weff = plk.double()[None].repeat(B, 1, 1, 1, 1)
    c = ks // 2
    for i in range(pdim):
        weff[:, i, i, c - 1:c + 2, c - 1:c + 2] += dk[:, i]
    lk = ops.pack_conv_weight(plk, None, ops.HAT_F32, dev, nt=1)
    kc = ops.KC[dt] * 3
    kpad = -(-(ks * ks * _r8(pdim)) // kc) * kc
    plkp = torch.zeros(npad, kpad, device=dev)
    plkp[:lk.w.shape[0], :min(kpad, lk.kpad)] = lk.w[:npad, :min(kpad, lk.kpad)]
    wout = torch.full((B, npad, kpad), 7.0, dtype=tdt, device=dev)
    ops.esc_weights(gap.to(dev), nblk, N, w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), plkp, wout, B=B, pdim=pdim, ksize=ks,
                    kpad=kpad, dtype=dt)
    torch.cuda.synchronize()
    cin_p = _r8(pdim)
    got = wout.float().cpu()[:, :pdim, :ks * ks * cin_p].reshape(B, pdim, ks, ks, cin_p)[..., :pdim].permute(0, 1, 4, 2, 3)
    check(got, weff, dtype, "esc weights", f32_tol=1e-5)
    assert float(wout[:, pdim:].float().abs().max()) == 0.0 if pdim < npad else True
    # ECA
    C, tiles, ldc = 144, 37, 144
    cs = rnd("cs", (B, tiles, ldc), std=3.0)
    wk = rnd("wk", (5,), std=0.5)
    m = cs.double().sum(1) / N
    e = F.conv1d(m[:, None, :], wk.double()[None, None], padding=2)[:, 0]
    ref = 0.01 * torch.sigmoid(e)
    scale = torch.zeros(B, ldc, device=dev)
    ops.eca_scale(cs.to(dev), tiles, ldc, N, wk.to(dev), 5, 0.01, torch.zeros(B, 32, ldc, device=dev), scale, B=B, C_=C)
    torch.cuda.synchronize()
    check(scale[:, :C], ref, "f32", "eca scale", f32_tol=1e-6)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("geom", [(2, 40, 48), (1, 13, 21), (1, 8, 16), (1, 67, 50)], ids=["B2_40x48", "ragged_13x21", "one_tile", "ragged_67x50"])
def test_fused_ffn2(geom):
    """hat_ffn2 (embed_dim 144, bf16 storage; fp16 hidden tensor, packed-fp16 depthwise conv + gate, fp16 fc2 MFMA) against
    the oracle's GatedDconvFFN restatement in fp64 (hat_arch.py:107-119,237): interior tiles, every kind of border tile
    (the depthwise conv zero-pads u AFTER the fc1 bias), ragged sizes, the fused next LayerNorm and its GAP partials."""
    C = 144
    B, H, W = geom
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE["bf16"]
    hid = 2 * C
    t = rnd("ft2", (B, H * W, C), std=1.5) + 0.3
    sd = {
        "n2.weight": 1 + rnd("fg", (C,), std=0.1), "n2.bias": rnd("fbb", (C,), std=0.1),
        "m.fc1.weight": q(rnd("f1w", (2 * hid, C), std=C ** -0.5), "bf16"), "m.fc1.bias": rnd("f1b", (2 * hid,), std=0.1),
        "m.dw.weight": rnd("fdw", (2 * hid, 1, 3, 3), std=1 / 3).half().float(), "m.dw.bias": rnd("fdb", (2 * hid,), std=0.1).half().float(),
        "m.fc2.weight": rnd("f2w", (C, hid), std=hid ** -0.5).half().float(), "m.fc2.bias": rnd("f2b", (C,), std=0.1),
        "n1.weight": 1 + rnd("fg1", (C,), std=0.1), "n1.bias": rnd("fb1", (C,), std=0.1),
    }
    sdd = {k: v.double() for k, v in sd.items()}
    ref = t.double() + O.gated_dconv_ffn(O._ln(t.double(), sdd, "n2"), (H, W), sdd, "m")
    ref_n = O._ln(ref, sdd, "n1")
    assert ops.ffn2_supported(C, hid, dt)
    pf = ops.pack_ffn2(sd["m.fc1.weight"], sd["m.fc1.bias"], sd["m.dw.weight"], sd["m.dw.bias"], sd["m.fc2.weight"],
                       sd["m.fc2.bias"], dev)
    tin = t.to(dev).contiguous()
    tout = torch.full_like(tin, 123.0)
    nout = torch.zeros(B, H * W, C, dtype=torch.bfloat16, device=dev)
    tiles = ops.ffn_tiles(pf, H, W, dt)
    gap = torch.zeros(B, tiles, 16, device=dev)
    dv = lambda k: sd[k].to(dev).contiguous()
    ops.ffn(pf, tin, tout, dv("n2.weight"), dv("n2.bias"), B=B, H=H, W=W, dtype=dt, ln1=(dv("n1.weight"), dv("n1.bias")),
            n_out=nout, ldn=C, gap_out=gap, gap_c=16)
    torch.cuda.synchronize()
    assert torch.isfinite(tout).all()
    upd, upd_ref = (tout.double().cpu() - t.double()), (ref - t.double())
    rel = float((upd - upd_ref).norm() / upd_ref.norm())
    assert rel <= 1.0e-2, f"hat_ffn2 update rel err {rel:.3e}"
    # per-pixel: no pixel (border ones in particular) may be off by more than a few bf16 ulps of the update scale
    worst = float((upd - upd_ref).abs().max() / upd_ref.abs().max())
    assert worst <= 4e-2, f"hat_ffn2 worst element {worst:.3e} of the update's range"
    check(nout.float(), ref_n, "bf16", "hat_ffn2 next-LN")
    check(gap.sum(1) / (H * W), ref_n[:, :, :16].mean(1), "f32", "hat_ffn2 gap partials", f32_tol=2e-3)
    tout2 = torch.zeros_like(tin)
    ops.ffn(pf, tin, tout2, dv("n2.weight"), dv("n2.bias"), B=B, H=H, W=W, dtype=dt)
    torch.cuda.synchronize()
    assert torch.equal(tout, tout2)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geom", [(144, 2, 40, 48), (180, 1, 24, 32), (24, 1, 13, 21)], ids=["C144", "C180", "C24"])
def test_fused_ffn(dtype, geom):
    """hat_ffn: t + fc2(a*silu(g)), [a|g] = dw3x3(fc1(LN2(t))) in one kernel, plus the fused NEXT LayerNorm and
    its GAP partials, against the oracle's GatedDconvFFN restatement (hat_arch.py:107-119,237)."""
    C, B, H, W = geom
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    hid = 2 * C
    t = rnd("ft", (B, H * W, C), std=1.5) + 0.3
    sd = {
        "n2.weight": 1 + rnd("fg", (C,), std=0.1), "n2.bias": rnd("fbb", (C,), std=0.1),
        "m.fc1.weight": q(rnd("f1w", (2 * hid, C), std=C ** -0.5), dtype), "m.fc1.bias": rnd("f1b", (2 * hid,), std=0.1),
        "m.dw.weight": q(rnd("fdw", (2 * hid, 1, 3, 3), std=1 / 3), dtype), "m.dw.bias": rnd("fdb", (2 * hid,), std=0.1),
        "m.fc2.weight": q(rnd("f2w", (C, hid), std=hid ** -0.5), dtype), "m.fc2.bias": rnd("f2b", (C,), std=0.1),
        "n1.weight": 1 + rnd("fg1", (C,), std=0.1), "n1.bias": rnd("fb1", (C,), std=0.1),
    }
    sdd = {k: v.double() for k, v in sd.items()}
    ref = t.double() + O.gated_dconv_ffn(O._ln(t.double(), sdd, "n2"), (H, W), sdd, "m")
    ref_n = O._ln(ref, sdd, "n1")
    pf = ops.pack_ffn(sd["m.fc1.weight"], sd["m.fc1.bias"], sd["m.dw.weight"], sd["m.dw.bias"], sd["m.fc2.weight"],
                      sd["m.fc2.bias"], dt, dev)
    tin = t.to(dev).contiguous()
    tout = torch.full_like(tin, 123.0)
    ld = _r8(C)
    nout = torch.zeros(B, H * W, ld, dtype=tdt, device=dev)
    tiles = ops.ffn_tiles(pf, H, W, dt)
    gap = torch.zeros(B, tiles, 16, device=dev)
    gap_c = 16 if C >= 16 else 8
    dv = lambda k: sd[k].to(dev).contiguous()
    ops.ffn(pf, tin, tout, dv("n2.weight"), dv("n2.bias"), B=B, H=H, W=W, dtype=dt, ln1=(dv("n1.weight"), dv("n1.bias")),
            n_out=nout, ldn=ld, gap_out=gap, gap_c=gap_c)
    torch.cuda.synchronize()
    if dtype == "f32":
        check(tout, ref, dtype, "fused ffn", f32_tol=3e-5)
        check(nout[:, :, :C].float(), ref_n, dtype, "fused next-LN", f32_tol=3e-5)
    else:
        # the hidden tensor is rounded to bf16 twice (u and a*silu(g)); judge the FFN update itself
        upd, upd_ref = (tout.double().cpu() - t.double()), (ref - t.double())
        rel = float((upd - upd_ref).norm() / upd_ref.norm())
        assert rel <= 1.5e-2, f"fused ffn update rel err {rel:.3e}"
        check(nout[:, :, :C].float(), ref_n, dtype, "fused next-LN")
    check(gap.sum(1)[:, :gap_c] / (H * W), ref_n[:, :, :gap_c].mean(1), "f32", "fused gap partials",
          f32_tol=(1e-5 if dtype == "f32" else 2e-3))
    # without the fused LN
    tout2 = torch.zeros_like(tin)
    ops.ffn(pf, tin, tout2, dv("n2.weight"), dv("n2.bias"), B=B, H=H, W=W, dtype=dt)
    torch.cuda.synchronize()
    assert torch.equal(tout, tout2)
    # LayerNorm2 supplied by the producer (m_in): rows [LN | 1.0 | zeros], as hat_linear's ln_ones epilogue writes them
    ldm = ops.ffn_m_ld(C)
    m = torch.zeros(B, H * W, ldm, dtype=tdt, device=dev)
    m[:, :, :C] = O._ln(t.double(), sdd, "n2").to(tdt).to(dev)
    m[:, :, C] = 1.0
    tout3 = torch.zeros_like(tin)
    ops.ffn(pf, tin, tout3, dv("n2.weight"), dv("n2.bias"), B=B, H=H, W=W, dtype=dt, m_in=m, ldm_in=ldm)
    torch.cuda.synchronize()
    if dtype == "f32":
        check(tout3, ref, dtype, "fused ffn (m_in)", f32_tol=3e-5)
    else:
        upd = tout3.double().cpu() - t.double()
        rel = float((upd - upd_ref).norm() / upd_ref.norm())
        assert rel <= 1.5e-2, f"fused ffn (m_in) update rel err {rel:.3e}"


# ------------------------------------------------------------------------------------------------
LIN_CASES = [("aggr144", 144, 144), ("kv288", 144, 288), ("mlp2_288", 288, 144), ("lin180", 180, 180), ("mlp360", 360, 180),
             ("kv360", 180, 360), ("tiny24", 24, 48), ("tiny48", 48, 24)]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", LIN_CASES, ids=[c[0] for c in LIN_CASES])
def test_linear_streaming(case, dtype):
    """hat_linear (weight-stationary streaming GEMM): plain, GELU, fp32 residual in place, split source + scaled T residual."""
    name, Cin, Cout = case
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    if not ops.linear_supported(Cout, Cin, dt):
        pytest.skip("shape not instantiated in hat_linear (the engine uses hat_conv for it)")
    B, H, W = 2, 19, 27  # 1026 pixels: not a multiple of 16, exercises the tail tile
    x = q(rnd(name + "x", (B, H, W, Cin)), dtype)
    wgt = q(rnd(name + "w", (Cout, Cin), std=Cin ** -0.5), dtype)
    bias = rnd(name + "b", (Cout,), std=0.1)
    pw = ops.pack_linear_weight(wgt, bias, dt, dev)
    ldx, ldo = _r8(Cin), _r8(Cout)
    xd = to_dev(x, ldx, tdt, dev)
    lin = F.linear(x.double(), wgt.double(), bias.double())
    out = torch.zeros(B, H * W, ldo, dtype=tdt, device=dev)
    ops.linear(pw, xd, out, B=B, H=H, W=W, dtype=dt, ldx=ldx, ldo=ldo, act=ops.ACT_GELU)
    torch.cuda.synchronize()
    ref = F.gelu(lin)
    check(out[:, :, :Cout].float().reshape(B, H, W, Cout), q(ref.float(), dtype).double() if dtype == "bf16" else ref, dtype, name + " gelu")
    # fp32 residual written in place, plus a scaled T residual, plus a split source for the first 8 channels
    c_split = 8
    x0 = q(rnd(name + "x0", (B, H, W, c_split)), dtype)
    r1 = rnd(name + "r1", (B, H, W, Cout))
    r2 = q(rnd(name + "r2", (B, H, W, Cout)), dtype)
    sc = rnd(name + "sc", (B, Cout), std=0.3)
    xin = torch.cat([x0, x[..., c_split:]], -1)
    ref2 = F.linear(xin.double(), wgt.double(), bias.double()) + r1.double() + sc.double()[:, None, None, :] * r2.double()
    rd = r1.reshape(B, H * W, Cout).to(dev).contiguous()
    scd = torch.zeros(B, pw.npad, device=dev)
    scd[:, :Cout] = sc.to(dev)
    ops.linear(pw, xd, rd, B=B, H=H, W=W, dtype=dt, ldx=ldx, ldo=Cout, out_mode=ops.O_NHWC_F32, x0=to_dev(x0, c_split, tdt, dev),
               c_split=c_split, ldx0=c_split, r1=rd, ldr1=Cout, r2=to_dev(r2, ldo, tdt, dev), ldr2=ldo, r2scale=scd,
               r2scale_bstride=pw.npad)
    torch.cuda.synchronize()
    check(rd.reshape(B, H, W, Cout), ref2, dtype, name + " residual epilogue", f32_tol=3e-5)
    if pw.n_slices == 1:
        # the same launch with the consumer's LayerNorm fused behind it, with and without hat_ffn's [1, 0...] tail
        g_, b_ = 1 + rnd(name + "lg", (Cout,), std=0.1), rnd(name + "lb", (Cout,), std=0.1)
        ref_ln = F.layer_norm(ref2, (Cout,), g_.double(), b_.double(), 1e-5)
        for ones in (False, True):
            ld_ln = ops.ffn_m_ld(Cout) if ones else ldo
            rd2 = r1.reshape(B, H * W, Cout).to(dev).contiguous()
            lno = torch.full((B, H * W, ld_ln), 7.0, dtype=tdt, device=dev)
            ops.linear(pw, xd, rd2, B=B, H=H, W=W, dtype=dt, ldx=ldx, ldo=Cout, out_mode=ops.O_NHWC_F32,
                       x0=to_dev(x0, c_split, tdt, dev), c_split=c_split, ldx0=c_split, r1=rd2, ldr1=Cout,
                       r2=to_dev(r2, ldo, tdt, dev), ldr2=ldo, r2scale=scd, r2scale_bstride=pw.npad,
                       ln=(g_.to(dev), b_.to(dev)), ln_out=lno, ld_ln=ld_ln, ln_ones=ones)
            torch.cuda.synchronize()
            assert torch.equal(rd2, rd), "the fused LayerNorm must not change the main output"
            check(lno[:, :, :Cout].float().reshape(B, H, W, Cout), ref_ln, dtype, name + " fused LayerNorm", f32_tol=5e-5)
            if ones:
                tail = lno[:, :, Cout:].float().cpu()
                assert torch.all(tail[:, :, 0] == 1.0) and torch.all(tail[:, :, 1:] == 0.0), "bias column / zero tail"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [("cab0", 144, 6, 1), ("cab2", 6, 144, 0), ("t_cab0", 24, 8, 1), ("t_cab2", 8, 24, 0)],
                         ids=lambda c: c[0])
def test_conv3x3_small(case, dtype):
    """hat_conv3x3_small (CAB convs, weights resident in LDS, neighbours gathered from global) + column sums."""
    name, Cin, Cout, act = case
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    if not ops.conv3x3_small_supported(Cout, Cin, dt):
        pytest.skip("weight slice does not fit half the LDS in this dtype (the engine uses hat_conv)")
    B, H, W = 2, 24, 40
    x = q(rnd(name + "x", (B, H, W, Cin)), dtype)
    wgt = q(rnd(name + "w", (Cout, Cin, 3, 3), std=(9 * Cin) ** -0.5), dtype)
    bias = rnd(name + "b", (Cout,), std=0.1)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wgt.double(), bias.double(), padding=1)
    ref = (F.gelu(ref) if act else ref).permute(0, 2, 3, 1)
    pw = ops.pack_linear_weight(wgt, bias, dt, dev)
    ldx, ldo = _r8(Cin), _r8(Cout)
    out = torch.zeros(B, H * W, ldo, dtype=tdt, device=dev)
    groups = ops.conv3x3_small_groups(pw, B, H, W, dt)
    colsum = torch.zeros(B, groups, pw.npad, device=dev)
    ops.conv3x3_small(pw, to_dev(x, ldx, tdt, dev), out, B=B, H=H, W=W, dtype=dt, ldx=ldx, ldo=ldo, act=act,
                      n_store=(Cout + 3) // 4 * 4, colsum=colsum)
    torch.cuda.synchronize()
    check(out[:, :, :Cout].float().reshape(B, H, W, Cout), q(ref.float(), dtype).double() if dtype == "bf16" else ref, dtype, name)
    check(colsum.sum(1)[:, :Cout] / (H * W), ref.mean((1, 2)), dtype, name + " column sums", f32_tol=3e-5)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("geom", [(2, 24, 40), (1, 19, 27)], ids=["B2_24x40", "B1_19x27_ragged"])
def test_aggr_with_folded_cab(geom):
    """hat_cab_fold + hat_aggr_cab: x = t + aggr(y) + conv_scale * ECA(c2) * c2 with c2 = conv3x3(c1) + b2 never
    materialised (hat_arch.py:84-90, 66-78, 233-236; esc_arch.py:123): the ECA pooling comes analytically from the
    sums of c1, the scaled expand conv rides on the aggregation GEMM."""
    B, H, W = geom
    C, mid, dtype = 144, 6, "bf16"
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE[dtype]
    tdt = ops.TORCH_DTYPE[dt]
    x = q(rnd("acx", (B, H, W, C)), dtype)
    x0 = q(rnd("acx0", (B, H, W, 16)), dtype)
    c1 = q(F.gelu(rnd("acc1", (B, H, W, mid))), dtype)
    t = rnd("act", (B, H, W, C))
    wa, ba = q(rnd("acwa", (C, C), std=C ** -0.5), dtype), rnd("acba", (C,), std=0.1)
    w2, b2 = rnd("acw2", (C, mid, 3, 3), std=(9 * mid) ** -0.5), rnd("acb2", (C,), std=0.1)
    wk = rnd("acwk", (5,), std=1.0)
    conv_scale = 0.37
    # reference in fp64
    c2 = F.conv2d(c1.permute(0, 3, 1, 2).double(), w2.double(), b2.double(), padding=1)          # (B,C,H,W)
    e = torch.sigmoid(F.conv1d(c2.mean((2, 3))[:, None, :], wk.double()[None, None, :], padding=2))[:, 0]  # (B,C)
    xin = torch.cat([x0, x[..., 16:]], -1)
    ref = t.double() + F.linear(xin.double(), wa.double(), ba.double()) + conv_scale * (e[:, :, None, None] * c2).permute(0, 2, 3, 1)
    # device
    pw = ops.pack_linear_weight(wa, ba, dt, dev)
    c1d = to_dev(c1, 8, tdt, dev)
    colsum = torch.zeros(B, 1, 16, device=dev)
    colsum[:, 0, :8] = c1d.float().sum(1)
    scale = torch.zeros(B, pw.npad, device=dev)
    wf = torch.zeros(B, pw.nt * 3 * 512, dtype=tdt, device=dev)
    bias_b = torch.zeros(B, pw.npad, device=dev)
    tmp = torch.zeros(B, 32, 16, device=dev)
    ops.cab_fold(c1d, colsum, 1, 16, w2.to(dev).contiguous(), b2.to(dev), wk.to(dev), 5, ba.to(dev), conv_scale, scale, wf, bias_b, tmp,
                 B=B, H=H, W=W, C_=C, mid=mid, dtype=dt)
    torch.cuda.synchronize()
    check(scale[:, :C], conv_scale * e, "f32", "folded ECA scale", f32_tol=2e-6)
    out = torch.zeros(B, H * W, C, device=dev)
    ops.aggr_cab(pw, to_dev(x, C, tdt, dev), out, c1d, wf, bias_b, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C,
                 x0=to_dev(x0, 16, tdt, dev), c_split=16, ldx0=16, r1=t.reshape(B, H * W, C).to(dev).contiguous(), ldr1=C)
    torch.cuda.synchronize()
    check(out.reshape(B, H, W, C), ref, dtype, "aggr + folded cab")


@pytest.mark.parametrize("geom", [(1, 64, 96), (2, 45, 70), (1, 8, 8), (1, 33, 129)], ids=["64x96", "B2_ragged_45x70", "8x8", "33x129"])
def test_esc_conv13_resident(geom):
    """hat_esc_conv13 (the ESC large-kernel conv with all weights and the haloed tile resident in LDS, esc_arch.py:121-123)
    against conv2d in fp64 with per-sample weights in hat_esc_weights' [B][16][Kpad] layout, and against hat_conv (ksize 13),
    the path it replaces: several tiles per workgroup, frames smaller than a tile, ragged edges, B = 2."""
    B, H, W = geom
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE["bf16"]
    C, pd, ks = 144, 16, 13
    x = q(rnd("e13x", (B, H, W, C)), "bf16")
    wt = q(rnd("e13w", (B, pd, pd, ks, ks), std=(pd * ks * ks) ** -0.5), "bf16")          # [b][co][ci][ty][tx]
    ref = torch.cat([F.conv2d(x[i:i + 1, ..., :pd].permute(0, 3, 1, 2).double(), wt[i].double(), padding=ks // 2) for i in range(B)], 0)
    ref = ref.permute(0, 2, 3, 1)
    kc = ops.KC[dt] * 3
    kpad = -(-(ks * ks * pd) // kc) * kc
    wp = torch.zeros(B, 16, kpad)
    wp[:, :pd, :ks * ks * pd] = wt.permute(0, 1, 3, 4, 2).reshape(B, pd, ks * ks * pd)     # K = tap * 16 + ci
    wpd = wp.to(torch.bfloat16).to(dev).contiguous()
    xd = to_dev(x, C, torch.bfloat16, dev)
    y = torch.full((B, H * W, 16), 3.0, dtype=torch.bfloat16, device=dev)
    ops.esc_conv13(xd, wpd, y, B=B, H=H, W=W, ldx=C, kpad=kpad, dtype=dt)
    torch.cuda.synchronize()
    check(y.reshape(B, H, W, 16).float(), ref, "bf16", "esc conv13 vs fp64")
    pw = ops.PackedConv(wpd, torch.zeros(16, device=dev), ks, pd, kpad, 1, 1, pd, w_bstride=16 * kpad)
    y2 = torch.zeros_like(y)
    ops.conv(pw, xd, y2, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=16, n_store=16)
    torch.cuda.synchronize()
    assert float((y.float() - y2.float()).abs().max()) <= 0.04 * max(1.0, float(y2.float().abs().max()))   # two bf16 roundings of the same sums


@pytest.mark.parametrize("gen", ["v2", "v3"])
@pytest.mark.parametrize("geom", [(2, 24, 40), (1, 19, 27), (1, 8, 16), (1, 35, 64)], ids=["B2_24x40", "ragged_19x27", "one_tile", "35x64"])
def test_hab_tail_fused(geom, gen):
    """hat_hab_tail (v2) / hat_hab_tail3 (v3: activation-stationary fc1, LayerNorm2's affine folded into fc1 at pack time, bias
    terms as k-slots) = hat_aggr_cab + hat_ffn2 in one launch (hat_arch.py:233-237, esc_arch.py:123): t + aggr([y16 | n[16:]]) +
    folded CAB, LayerNorm2, gated depthwise FFN, residual, next LayerNorm + GAP partials.  Checked against (a) the two-kernel
    sequence it replaces (same arithmetic, tB merely never leaves the chip: agreement to a few flipped bf16 roundings) and (b) an
    fp64 restatement, on interior and border tiles, ragged sizes and B = 2."""
    B, H, W = geom
    C, mid, hid = 144, 6, 288
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE["bf16"]
    tdt = torch.bfloat16
    n = q(rnd("htn", (B, H, W, C)), "bf16")
    y16 = q(rnd("hty", (B, H, W, 16)), "bf16")
    c1 = q(F.gelu(rnd("htc1", (B, H, W, mid))), "bf16")
    t = rnd("htt", (B, H, W, C), std=1.5) + 0.3
    wa, ba = q(rnd("htwa", (C, C), std=C ** -0.5), "bf16"), rnd("htba", (C,), std=0.1)
    w2, b2c = rnd("htw2", (C, mid, 3, 3), std=(9 * mid) ** -0.5), rnd("htb2", (C,), std=0.1)
    wk = rnd("htwk", (5,), std=1.0)
    conv_scale = 0.37
    sd = {
        "n2.weight": 1 + rnd("fg", (C,), std=0.1), "n2.bias": rnd("fbb", (C,), std=0.1),
        "m.fc1.weight": q(rnd("f1w", (2 * hid, C), std=C ** -0.5), "bf16"), "m.fc1.bias": rnd("f1b", (2 * hid,), std=0.1),
        "m.dw.weight": rnd("fdw", (2 * hid, 1, 3, 3), std=1 / 3).half().float(), "m.dw.bias": rnd("fdb", (2 * hid,), std=0.1).half().float(),
        "m.fc2.weight": rnd("f2w", (C, hid), std=hid ** -0.5).half().float(), "m.fc2.bias": rnd("f2b", (C,), std=0.1),
        "n1.weight": 1 + rnd("fg1", (C,), std=0.1), "n1.bias": rnd("fb1", (C,), std=0.1),
    }
    sdd = {k: v.double() for k, v in sd.items()}
    # fp64 restatement
    c2 = F.conv2d(c1.permute(0, 3, 1, 2).double(), w2.double(), b2c.double(), padding=1)
    e = torch.sigmoid(F.conv1d(c2.mean((2, 3))[:, None, :], wk.double()[None, None, :], padding=2))[:, 0]
    xin = torch.cat([y16, n[..., 16:]], -1)
    tb = t.double() + F.linear(xin.double(), wa.double(), ba.double()) + conv_scale * (e[:, :, None, None] * c2).permute(0, 2, 3, 1)
    tb = tb.reshape(B, H * W, C)
    ref = tb + O.gated_dconv_ffn(O._ln(tb, sdd, "n2"), (H, W), sdd, "m")
    ref_n = O._ln(ref, sdd, "n1")
    # device: fold, then the two-kernel sequence and the fused kernel
    pw = ops.pack_linear_weight(wa, ba, dt, dev)
    pf = ops.pack_ffn2(sd["m.fc1.weight"], sd["m.fc1.bias"], sd["m.dw.weight"], sd["m.dw.bias"], sd["m.fc2.weight"], sd["m.fc2.bias"], dev)
    assert ops.hab_tail_supported(pf, pw, mid, dt)
    c1d = to_dev(c1, 8, tdt, dev)
    colsum = torch.zeros(B, 1, 16, device=dev)
    colsum[:, 0, :8] = c1d.float().sum(1)
    scale = torch.zeros(B, pw.npad, device=dev)
    wf = torch.zeros(B, pw.nt * 3 * 512, dtype=tdt, device=dev)
    bias_b = torch.zeros(B, pw.npad, device=dev)
    ops.cab_fold(c1d, colsum, 1, 16, w2.to(dev).contiguous(), b2c.to(dev), wk.to(dev), 5, ba.to(dev), conv_scale, scale, wf, bias_b,
                 torch.zeros(B, 32, 16, device=dev), B=B, H=H, W=W, C_=C, mid=mid, dtype=dt)
    nd, yd, td = to_dev(n, C, tdt, dev), to_dev(y16, 16, tdt, dev), t.reshape(B, H * W, C).to(dev).contiguous()
    dv = lambda k: sd[k].to(dev).contiguous()
    tiles = ops.ffn_tiles(pf, H, W, dt)
    kw = dict(B=B, H=H, W=W, dtype=dt, ln1=(dv("n1.weight"), dv("n1.bias")), ldn=C, gap_c=16)
    tB = torch.zeros(B, H * W, C, device=dev)
    ops.aggr_cab(pw, nd, tB, c1d, wf, bias_b, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C, x0=yd, c_split=16, ldx0=16, r1=td, ldr1=C)
    out2, n2, gap2 = torch.zeros_like(tB), torch.zeros(B, H * W, C, dtype=tdt, device=dev), torch.zeros(B, tiles, 16, device=dev)
    ops.ffn(pf, tB, out2, dv("n2.weight"), dv("n2.bias"), n_out=n2, gap_out=gap2, **kw)
    out1, n1, gap1 = torch.full_like(tB, 7.0), torch.zeros_like(n2), torch.zeros_like(gap2)
    pft = pf if gen == "v2" else ops.pack_ffn3(sd["m.fc1.weight"], sd["m.fc1.bias"], sd["m.dw.weight"], sd["m.dw.bias"], sd["m.fc2.weight"],
                                               sd["m.fc2.bias"], sd["n2.weight"], sd["n2.bias"], dev)
    ops.hab_tail(pft, pw, td, out1, dv("n2.weight"), dv("n2.bias"), n=nd, ldn_in=C, y16=yd, c1=c1d, wf=wf, bias_b=bias_b,
                 n_out=n1, gap_out=gap1, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(out1).all()
    d12 = float((out1 - out2).abs().max())
    # same arithmetic; fp32 round-off in tB may flip single bf16 roundings of LayerNorm2's output (an ulp is 0.4 %)
    # (v3 rounds W1 * gamma to bf16 where the two-kernel sequence rounds W1 and LayerNorm2's output: two bf16 evaluations)
    assert d12 <= (1e-3 if gen == "v2" else 2e-2) * max(1.0, float(out2.abs().max())), f"fused vs two-kernel sequence: max-abs {d12:.3e}"
    assert float((n1.float() - n2.float()).abs().max()) <= (0.04 if gen == "v2" else 0.08) and float((gap1 - gap2).abs().max()) <= 1e-2 * H * W / tiles
    upd, upd_ref = out1.double().cpu() - tb, ref - tb
    rel = float((upd - upd_ref).norm() / upd_ref.norm())
    assert rel <= 1.2e-2, f"hab tail FFN update rel err {rel:.3e}"
    check(out1, ref, "bf16", "hab tail t_out")
    check(n1.float(), ref_n, "bf16", "hab tail next-LN")


@pytest.mark.parametrize("geom", [(2, 24, 40), (1, 19, 27)], ids=["B2_24x40", "ragged_19x27"])
def test_hab_tail3_fp16_residual_rows(geom):
    """hat_hab_tail3 with the residual stream as FP16 rows (HatHabTailDesc.reserved1 bits 0 / 1; the engine uses them between
    the blocks of a residual group): the arithmetic is the fp32-stream kernel's, only the load / store differ — FP16 in is
    exact, FP16 out is round-to-nearest of the same fp32 result.  So on an input that is representable in FP16 all four
    combinations must agree BIT FOR BIT with the fp32 / fp32 launch (t_out after .half(), the next LayerNorm rows, the GAP
    partials and the compact copy as they are); and a value beyond the FP16 range is clamped, not turned into inf."""
    B, H, W = geom
    C, mid, hid = 144, 6, 288
    dev, ops = _dev(), _ops()
    dt, tdt = ops.DTYPE_CODE["bf16"], torch.bfloat16
    n = q(rnd("hhn", (B, H, W, C)), "bf16")
    y16 = q(rnd("hhy", (B, H, W, 16)), "bf16")
    c1 = q(F.gelu(rnd("hhc1", (B, H, W, mid))), "bf16")
    t16 = (rnd("hht", (B, H, W, C), std=1.5) + 0.3).half()
    wa, ba = q(rnd("hhwa", (C, C), std=C ** -0.5), "bf16"), rnd("hhba", (C,), std=0.1)
    sd = {
        "n2.weight": 1 + rnd("hg", (C,), std=0.1), "n2.bias": rnd("hbb", (C,), std=0.1),
        "m.fc1.weight": q(rnd("h1w", (2 * hid, C), std=C ** -0.5), "bf16"), "m.fc1.bias": rnd("h1b", (2 * hid,), std=0.1),
        "m.dw.weight": rnd("hdw", (2 * hid, 1, 3, 3), std=1 / 3).half().float(), "m.dw.bias": rnd("hdb", (2 * hid,), std=0.1).half().float(),
        "m.fc2.weight": rnd("h2w", (C, hid), std=hid ** -0.5).half().float(), "m.fc2.bias": rnd("h2b", (C,), std=0.1),
        "n1.weight": 1 + rnd("hg1", (C,), std=0.1), "n1.bias": rnd("hb1", (C,), std=0.1),
    }
    pw = ops.pack_linear_weight(wa, ba, dt, dev)
    pf = ops.pack_ffn3(sd["m.fc1.weight"], sd["m.fc1.bias"], sd["m.dw.weight"], sd["m.dw.bias"], sd["m.fc2.weight"], sd["m.fc2.bias"],
                       sd["n2.weight"], sd["n2.bias"], dev)
    c1d = to_dev(c1, 8, tdt, dev)
    wf = (torch.randn(B, pw.nt * 3 * 512, generator=torch.Generator().manual_seed(5)) * 0.02).to(tdt).to(dev)
    bias_b = rnd("hhbb", (B, pw.npad), std=0.1).to(dev)
    nd, yd = to_dev(n, C, tdt, dev), to_dev(y16, 16, tdt, dev)
    dv = lambda k: sd[k].to(dev).contiguous()
    tiles = ops.ffn_tiles(pf, H, W, dt)
    res = {}
    for in_half in (False, True):
        for out_half in (False, True):
            td = t16.reshape(B, H * W, C).to(dev)
            td = td.contiguous() if in_half else td.float().contiguous()
            out = torch.full((B, H * W, C), 7.0, dtype=(torch.float16 if out_half else torch.float32), device=dev)
            n1, gap = torch.zeros(B, H * W, C, dtype=tdt, device=dev), torch.zeros(B, tiles, 16, device=dev)
            n16 = torch.zeros(B, H * W, 16, dtype=tdt, device=dev)
            ops.hab_tail(pf, pw, td, out, dv("n2.weight"), dv("n2.bias"), n=nd, ldn_in=C, y16=yd, c1=c1d, wf=wf, bias_b=bias_b, B=B, H=H, W=W,
                         dtype=dt, ln1=(dv("n1.weight"), dv("n1.bias")), ldn=C, gap_c=16, n_out=n1, gap_out=gap, n16_out=n16)
            torch.cuda.synchronize()
            res[(in_half, out_half)] = (out.cpu(), n1.cpu(), gap.cpu(), n16.cpu())
    o0, n0, g0, s0 = res[(False, False)]
    assert torch.isfinite(o0).all() and float(o0.abs().max()) < 6e4
    for key, (o, n1, gap, n16) in res.items():
        want = o0.half() if key[1] else o0
        assert torch.equal(o, want), f"t_out differs for (in_half, out_half) = {key}: max-abs {float((o.float() - want.float()).abs().max()):.3e}"
        assert torch.equal(n1, n0) and torch.equal(gap, g0) and torch.equal(n16, s0), key
    # beyond the FP16 range: clamped to +-65504
    td = torch.full((1, 8 * 16, C), 7.0e4, device=dev)
    out = torch.zeros(1, 8 * 16, C, dtype=torch.float16, device=dev)
    ops.hab_tail(pf, pw, td, out, dv("n2.weight"), dv("n2.bias"), n=nd[:1, :128].contiguous(), ldn_in=C, y16=yd[:1, :128].contiguous(),
                 c1=c1d[:1, :128].contiguous(), wf=wf[:1].contiguous(), bias_b=bias_b[:1].contiguous(), B=1, H=8, W=16, dtype=dt)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(out.float().abs().max()) == 65504.0


# ------------------------------------------------------------------------------------------------
# (S)W-MSA branch: hat_linear -> hat_window_attention -> hat_linear
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("name", ["wmsa_c48_h2_ws16", "wmsa_c48_h6_ws8"])
def test_window_msa_vs_reference_golden(name, dtype):
    """The product WindowAttention (same state-dict keys as swinir_arch.WindowAttention) against the outputs of the
    reference's own module + roll / partition / mask / reverse, for W-MSA (shift 0) and SW-MSA (shift ws/2)."""
    from helpers import golden, wmsa_sd
    from super_resolution_amd.archs.window_msa import WindowAttention
    dev = _dev()
    g = golden(name + ".npz")
    C, heads, ws, H, W = (int(v) for v in g["dims"])
    m = WindowAttention(C, (ws, ws), heads, compute_dtype=dtype).eval()
    m.load_state_dict(wmsa_sd(C, heads, ws), strict=True)
    m = m.to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    for shift in (0, ws // 2):
        y = m.forward_map(x, shift)
        torch.cuda.synchronize()
        check(y, torch.from_numpy(g[f"y_shift{shift}"]), dtype, f"{name} shift {shift}")
    # reference call signature: pre-partitioned windows, no mask
    xw = x.reshape(1, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C).contiguous()
    yw = m(xw)
    ref = torch.from_numpy(g["y_shift0"]).reshape(1, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    check(yw, ref, dtype, f"{name} windows")
    with pytest.raises(NotImplementedError):
        m(xw, mask=torch.zeros(1, ws * ws, ws * ws, device=dev))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geom", [(144, 6, 2, 48, 32), (180, 6, 1, 32, 48), (60, 2, 1, 16, 32)], ids=["C144_d24", "C180_d30", "C60_d30_one_row"])
def test_window_msa_shipped_head_sizes_vs_oracle(geom, dtype):
    """Head sizes of the shipped variants (24 for C = 144, 30 for C = 180: the bf16 fast kernels), B = 2, non-square
    frames, a frame of a single window row, shifts 0 / 8 / 4 — against the CPU oracle."""
    from helpers import wmsa_sd
    from super_resolution_amd.archs.window_msa import WindowAttention
    dev = _dev()
    C, heads, B, H, W = geom
    ws = 16
    sd = wmsa_sd(C, heads, ws)
    m = WindowAttention(C, ws, heads, compute_dtype=dtype).eval()
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    x = rnd(f"wmsa{C}.x", (B, H, W, C))
    for shift in (0, 8, 4):
        ref = O.window_msa(x, {"a." + k: v for k, v in sd.items()}, "a", ws, heads, shift)
        y = m.forward_map(x.to(dev), shift)
        torch.cuda.synchronize()
        check(y, ref, dtype, f"C{C} shift {shift}")


def test_window_attention_rejects_bad_arguments():
    dev = _dev()
    ops = _ops()
    from super_resolution_amd._lib import HAT_F32
    t = torch.zeros(1, 16, 16, 3 * 48, device=dev)
    o = torch.zeros(1, 16, 16, 48, device=dev)
    bias = torch.zeros(2, 31 * 31, device=dev)
    kw = dict(B=1, H=16, W=16, C_=48, heads=2, ws=16, ldq=144, ldkv=144, ldo=48, dtype=HAT_F32)
    for bad in (dict(shift=3), dict(shift=16), dict(H=24), dict(ws=12)):
        with pytest.raises(RuntimeError):
            ops.window_attention(t, t.view(-1)[48:], bias, o, **{**kw, "shift": 0, **bad})


# ------------------------------------------------------------------------------------------------
# CAB squeeze conv on the row-sweep kernel (hat_cab_squeeze)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("geom", [(1, 48, 64, 144, 6), (2, 21, 32, 144, 6), (1, 8, 16, 136, 8), (1, 100, 48, 160, 3), (1, 9, 112, 144, 6),
                                  (1, 12, 272, 144, 6)],
                         ids=["48x64", "B2_21x32_ragged_rows", "one_band_C136_mid8", "C160_mid3", "width_8_full_strips", "width_272"])
def test_cab_squeeze_row_sweep(geom):
    """GELU(conv3x3(x) + b) with <= 8 output channels against F.conv2d in fp64 on the bf16-rounded operands, including the
    per-unit channel sums (their total must equal the sum of the stored fp32 values within bf16 rounding of the outputs).
    A wave's strip is 14 output columns (16 loaded ones, the dx taps are lane shifts): the widths cover a last strip of 2, 4,
    6 and 8 columns, a width of exactly eight strips (the last strip's right halo column is outside the image) and 20 strips."""
    dev = _dev()
    ops = _ops()
    from super_resolution_amd._lib import HAT_BF16
    B, H, W, C, mid = geom
    x = rnd(f"cabsq.x{geom}", (B, H, W, C)).to(torch.bfloat16)
    w = rnd(f"cabsq.w{geom}", (mid, C, 3, 3), std=(9 * C) ** -0.5)
    b = rnd(f"cabsq.b{geom}", (mid,), std=0.1)
    assert ops.cab_squeeze_supported(C, mid, W, HAT_BF16)
    wpk, b8 = ops.pack_cab_squeeze(w, b, dev)
    units = ops.cab_squeeze_units(H, W)
    xd = x.to(dev).contiguous()
    out = torch.full((B, H, W, 8), 7.0, dtype=torch.bfloat16, device=dev)
    cs = torch.full((B, units, 16), 7.0, dtype=torch.float32, device=dev)
    ops.cab_squeeze(xd, wpk, b8, out, cs, B=B, H=H, W=W, C_=C, ldx=C, dtype=HAT_BF16)
    torch.cuda.synchronize()
    ref = F.gelu(F.conv2d(x.double().permute(0, 3, 1, 2), w.to(torch.bfloat16).double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    got = out.float().cpu()
    check(got[..., :mid], ref.float(), "bf16", f"cab squeeze {geom}")
    if mid < 8:
        assert float(got[..., mid:].abs().max()) == 0.0
    tot = cs.sum(1).cpu()
    assert float(tot[:, 8:].abs().max()) == 0.0
    assert torch.allclose(tot[:, :mid].double(), ref.sum((1, 2)), rtol=2e-3, atol=2e-2 * (H * W) ** 0.5)
    # ... and they are the sums of the STORED values (what a band-sharded frame pools with hat_rect_sum): fp32 round-off only
    stored = got[..., :mid].double().sum((1, 2))
    assert float((cs.double().sum(1).cpu()[:, :mid] - stored).abs().max()) <= 2e-6 * float(got[..., :mid].double().abs().sum((1, 2)).max())


@pytest.mark.parametrize("geom", [(1, 40, 64, 3), (2, 19, 32, 3), (1, 64, 48, 1)], ids=["40x64", "B2_19x32", "one_channel"])
def test_conv_last_row_sweep_planes(geom):
    """(conv3x3(x, 64 -> n) + b) * out_scale + mean as (B, n, H, W) fp32 planes against F.conv2d in fp64."""
    dev = _dev()
    ops = _ops()
    from super_resolution_amd._lib import HAT_BF16
    B, H, W, nout = geom
    x = rnd(f"planes.x{geom}", (B, H, W, 64)).to(torch.bfloat16)
    w = rnd(f"planes.w{geom}", (nout, 64, 3, 3), std=(9 * 64) ** -0.5)
    b = rnd(f"planes.b{geom}", (nout,), std=0.1)
    mean, scale = (0.4488, 0.4371, 0.4040), 0.5
    assert ops.conv3x3_to_planes_supported(nout, 64, W, HAT_BF16)
    wpk, b8 = ops.pack_cab_squeeze(w, b, dev)
    out = torch.full((B, nout, H, W), 7.0, dtype=torch.float32, device=dev)
    ops.conv3x3_to_planes(x.to(dev).contiguous(), wpk, b8, out, B=B, H=H, W=W, C_=64, ldx=64, n_out=nout, out_scale=scale,
                          mean=mean, dtype=HAT_BF16)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.to(torch.bfloat16).double(), b.double(), padding=1) * scale
    ref = ref + torch.tensor(mean[:nout], dtype=torch.float64).view(1, -1, 1, 1)
    check(out.cpu(), ref.float(), "f32", f"conv_last planes {geom}", f32_tol=2e-5)


# ------------------------------------------------------------------------------------------------
# pooled sums of a band-sharded frame: hat_rect_sum, hat_cab_fold from supplied statistics
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [(2, 40, 64, 16, 16, "bf16"), (1, 96, 1280, 16, 16, "bf16"), (1, 33, 48, 144, 144, "bf16"), (2, 19, 24, 24, 24, "f32"),
                                  (1, 64, 256, 8, 8, "bf16")], ids=lambda c: f"B{c[0]}_{c[1]}x{c[2]}_C{c[3]}_{c[5]}")
def test_rect_sum(case):
    """hat_rect_sum against fp64 sums over full-width row ranges, single columns, single rows and single pixels — the
    rectangles the band-sharded engine pools (rows a band owns; frame borders and corners for hat_cab_fold).  Many workgroups
    at the larger sizes (cross-workgroup part through the last-block pattern); repeated calls are bit-identical."""
    B, H, W, C, ld, dtype = case
    dev, ops = _dev(), _ops()
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    x = q(rnd("rsx", (B, H, W, ld)), dtype)
    xd = x.reshape(B, H * W, ld).to(tdt).to(dev).contiguous()
    out = torch.zeros(B, 64 + 192, device=dev)
    tmp, cnt = torch.zeros(B, 64, 256, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    rects = [(0, H, 0, W), (3, H - 2, 0, W), (0, H, 0, 1), (2, H, W - 1, W), (0, 1, 0, W), (H - 1, H, 0, W), (0, 1, 0, 1), (H - 1, H, W - 1, W)]
    for r0, r1, c0, c1 in rects:
        for rep in range(2):
            o = torch.zeros_like(out)
            ops.rect_sum(xd, o, tmp, cnt, B=B, W=W, ld=ld, C_=C, r0=r0, r1=r1, c0=c0, c1=c1, out_off=8)
            torch.cuda.synchronize()
            if rep == 0:
                first = o.clone()
        assert torch.equal(first, o) and int(cnt.abs().sum()) == 0
        ref = x[:, r0:r1, c0:c1, :C].double().sum((1, 2))
        got = o[:, 8:8 + C].double().cpu()
        assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(x[:, r0:r1, c0:c1].double().abs().sum((1, 2)).max())), (r0, r1, c0, c1)
        assert float(o[:, :8].abs().max()) == 0 and float(o[:, 8 + C:].abs().max()) == 0


def test_cab_fold_from_supplied_statistics():
    """hat_cab_fold with `stats` (the frame-wide sums a band-sharded frame adds up from its bands' hat_rect_sum results) against
    the same kernel computing them from c1 and its per-tile column sums: scale, folded weights and bias agree to fp32 round-off."""
    B, H, W, C, mid = 2, 24, 40, 144, 6
    dev, ops = _dev(), _ops()
    dt, tdt = ops.DTYPE_CODE["bf16"], torch.bfloat16
    c1 = q(F.gelu(rnd("cfc1", (B, H, W, mid))), "bf16")
    w2, b2c, wk, ba = rnd("cfw2", (C, mid, 3, 3), std=(9 * mid) ** -0.5), rnd("cfb2", (C,), std=0.1), rnd("cfwk", (5,), std=1.0), rnd("cfba", (C,), std=0.1)
    c1d = to_dev(c1, 8, tdt, dev)
    colsum = torch.zeros(B, 1, 16, device=dev)
    colsum[:, 0, :8] = c1d.float().sum(1)
    res = []
    for use_stats in (False, True):
        scale, wf, bias_b = torch.zeros(B, 256, device=dev), torch.zeros(B, 9 * 3 * 512, dtype=tdt, device=dev), torch.zeros(B, 144, device=dev)
        stats = None
        if use_stats:
            stats = torch.zeros(B, 72, device=dev)
            tmp, cnt = torch.zeros(B, 64, 256, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
            for off, (r0, r1, a, b_) in {0: (0, H, 0, W), 8: (0, 1, 0, W), 16: (H - 1, H, 0, W), 24: (0, H, 0, 1), 32: (0, H, W - 1, W), 40: (0, 1, 0, 1),
                                          48: (0, 1, W - 1, W), 56: (H - 1, H, 0, 1), 64: (H - 1, H, W - 1, W)}.items():
                ops.rect_sum(c1d, stats, tmp, cnt, B=B, W=W, ld=8, C_=8, r0=r0, r1=r1, c0=a, c1=b_, out_off=off)
        ops.cab_fold(None if use_stats else c1d, None if use_stats else colsum, 1, 16, w2.to(dev).contiguous(), b2c.to(dev), wk.to(dev), 5, ba.to(dev),
                     0.37, scale, wf, bias_b, None if use_stats else torch.zeros(B, 32, 16, device=dev), B=B, H=H, W=W, C_=C, mid=mid, dtype=dt, stats=stats)
        torch.cuda.synchronize()
        res.append((scale.cpu(), wf.float().cpu(), bias_b.cpu()))
    for a, b_ in zip(*res):
        assert float((a - b_).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_cab_fold_with_fragment_ordered_weights(dtype):
    """HatCabFoldDesc.w2f (the expand weights once more in the order of the kernel's output, ops.pack_cab_w2f — what the engine
    passes) against the gather out of the [C][mid][3][3] layout: same multiplications in the same order, so scale, folded
    weights (including the two bias slots at k = 72, 73) and bias must be bit-identical; C = 144 / mid = 6 and the widest
    supported shape, C = 160 / mid = 8."""
    dev, ops = _dev(), _ops()
    dt, tdt = ops.DTYPE_CODE[dtype], (torch.bfloat16 if dtype == "bf16" else torch.float32)
    for (B, H, W, C, mid) in [(2, 24, 40, 144, 6), (1, 16, 32, 160, 8)]:
        nt = -(-C // 16)
        c1 = q(F.gelu(rnd(f"wfc1{C}", (B, H, W, mid))), dtype)
        w2, b2c, wk, ba = rnd(f"wfw2{C}", (C, mid, 3, 3), std=(9 * mid) ** -0.5), rnd(f"wfb2{C}", (C,), std=0.1), rnd("wfwk", (5,), std=1.0), rnd(f"wfba{C}", (C,), std=0.1)
        c1d = to_dev(c1, 8, tdt, dev)
        colsum = torch.zeros(B, 1, 16, device=dev)
        colsum[:, 0, :8] = c1d.float().sum(1)
        w2f = ops.pack_cab_w2f(w2, dev)
        assert w2f.shape == (nt, 3, 64, 8)
        res = []
        for use in (False, True):
            scale, wf, bias_b = torch.zeros(B, 256, device=dev), torch.full((B, nt * 3 * 512), 7.0, dtype=tdt, device=dev), torch.zeros(B, nt * 16, device=dev)
            ops.cab_fold(c1d, colsum, 1, 16, w2.to(dev).contiguous(), b2c.to(dev), wk.to(dev), 5, ba.to(dev), 0.37, scale, wf, bias_b,
                         torch.zeros(B, 32, 16, device=dev), B=B, H=H, W=W, C_=C, mid=mid, dtype=dt, w2f=(w2f if use else None))
            torch.cuda.synchronize()
            res.append((scale.cpu(), wf.float().cpu(), bias_b.cpu()))
        for a, b_ in zip(*res):
            assert torch.equal(a, b_)
        assert float(res[1][1].abs().max()) > 0


# ------------------------------------------------------------------------------------------------
# FP16 range of the fused FFN kernels (no test drove |u| or a * SiLU(g) near the FP16 range)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("s1", [1.0, 20.0, 45.0], ids=["unit", "x20", "x45"])
def test_fused_ffn2_large_hidden_magnitudes(s1):
    """hat_ffn2 with fc1 scaled by s1 (fc2 by 1 / s1^2 so that the update stays O(1)): at s1 = 45 the hidden tensor reaches
    |u| ~ 200 and the gated product a * SiLU(g) more than 1e4, a sixth of the FP16 range the kernel stores them in (the
    test states the magnitudes it reached).  As long as the
    pack-time worst-case bound (ops.ffn_fp16_range_bound — what the engine checks before it uses the FP16 kernels) holds, the
    result must be finite and as accurate against fp64 as at unit scale: FP16 carries 11 significant bits at every magnitude
    of its normal range."""
    C, B, H, W = 144, 1, 24, 32
    dev, ops = _dev(), _ops()
    dt = ops.DTYPE_CODE["bf16"]
    hid = 2 * C
    t = rnd("fr2", (B, H * W, C), std=1.5) + 0.3
    sd = {
        "n2.weight": 1 + rnd("fg", (C,), std=0.1), "n2.bias": rnd("fbb", (C,), std=0.1),
        "m.fc1.weight": q(s1 * rnd("f1w", (2 * hid, C), std=C ** -0.5), "bf16"), "m.fc1.bias": s1 * rnd("f1b", (2 * hid,), std=0.1),
        "m.dw.weight": rnd("fdw", (2 * hid, 1, 3, 3), std=1 / 3).half().float(), "m.dw.bias": rnd("fdb", (2 * hid,), std=0.1).half().float(),
        "m.fc2.weight": (rnd("f2w", (C, hid), std=hid ** -0.5) / s1 ** 2).half().float(), "m.fc2.bias": rnd("f2b", (C,), std=0.1),
    }
    sdd = {k: v.double() for k, v in sd.items()}
    m = O._ln(t.double(), sdd, "n2")
    ref = t.double() + O.gated_dconv_ffn(m, (H, W), sdd, "m")
    u = F.linear(m, sdd["m.fc1.weight"], sdd["m.fc1.bias"])
    bound = ops.ffn_fp16_range_bound(sd["m.fc1.weight"], sd["m.fc1.bias"], sd["m.dw.weight"], sd["m.dw.bias"], sd["n2.weight"], sd["n2.bias"])
    assert float(u.abs().max()) <= bound                      # the bound is a bound
    ud = F.conv2d(u.reshape(B, H, W, 2 * hid).permute(0, 3, 1, 2), sdd["m.dw.weight"], sdd["m.dw.bias"], padding=1, groups=2 * hid)
    gated = float((ud[:, :hid] * F.silu(ud[:, hid:])).abs().max())
    assert gated < 6.0e4, gated                                # inside the FP16 range: the case the kernel is specified for
    if s1 >= 45.0:
        assert float(u.abs().max()) >= 150.0 and gated >= 1.0e4, (float(u.abs().max()), gated)
    pf = ops.pack_ffn2(sd["m.fc1.weight"], sd["m.fc1.bias"], sd["m.dw.weight"], sd["m.dw.bias"], sd["m.fc2.weight"], sd["m.fc2.bias"], dev)
    tin = t.to(dev).contiguous()
    tout = torch.full_like(tin, 123.0)
    dv = lambda k: sd[k].to(dev).contiguous()
    ops.ffn(pf, tin, tout, dv("n2.weight"), dv("n2.bias"), B=B, H=H, W=W, dtype=dt)
    torch.cuda.synchronize()
    assert torch.isfinite(tout).all()
    upd, upd_ref = (tout.double().cpu() - t.double()), (ref - t.double())
    rel = float((upd - upd_ref).norm() / upd_ref.norm())
    assert rel <= 1.2e-2, f"s1 = {s1}: hat_ffn2 update rel err {rel:.3e} (|u| max {float(u.abs().max()):.0f})"


def test_fp16_range_bound_switches_the_engine_to_the_bf16_hidden_kernel():
    """A HAT-S block whose fc1 is scaled until a * SiLU(g) WOULD leave the FP16 range (s1 = 300: worst-case bound 2e8, actual
    products ~1e7): the engine must not run hat_ffn2 / hat_hab_tail3 for that block — it packs hat_ffn (bf16 hidden tensor, fp32
    range) for it — and the whole forward stays finite and within the bf16 bar of the fp32 oracle; the other blocks keep the FP16
    kernels."""
    from helpers import META, max_abs, oracle_sd
    from super_resolution_amd.registry import build_network
    import super_resolution_amd.archs  # noqa: F401
    dev = _dev()
    cfg, sd = oracle_sd("hats_1g_x4")
    sd = {k: v.clone() for k, v in sd.items()}
    p = "layers.0.residual_group.blocks.1.mlp."
    s1 = 300.0
    sd[p + "fc1.weight"] *= s1
    sd[p + "fc1.bias"] *= s1
    sd[p + "fc2.weight"] /= s1 ** 2
    x = synth.synth_input(7, (1, 3, 32, 48))
    ref = O.hat_forward(x, sd, cfg)
    net = build_network(dict(type="HAT", compute_dtype="bf16", **META["cfgs"]["hats_1g_x4"])).eval()
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    y = net(x.to(dev)).float().cpu()
    torch.cuda.synchronize()
    eng = net.engine()
    assert getattr(eng, "fp16_fallbacks", 0) == 1
    kinds = [hb["ffn"].khalf for hb in eng.layers[0]["habs"]]
    assert kinds[1] not in ("v2", "v3") and all(k == "v2" for i, k in enumerate(kinds) if i != 1), kinds
    assert torch.isfinite(y).all()
    assert O.psnr_float(y, ref) >= 40.0 and max_abs(y, ref) <= 0.08, (O.psnr_float(y, ref), max_abs(y, ref))


@pytest.mark.parametrize("geom", [(2, 24, 40), (1, 19, 27), (1, 8, 16), (1, 35, 64)], ids=["B2_24x40", "ragged_19x27", "one_tile", "35x64"])
def test_hab_tail3_embed_dim_180(geom):
    """hat_hab_tail3 at embed_dim 180 (HAT / HAT-L, BASELINE configs 3-5): t + aggr([y16 | n[16:]]) + scale * c2 + bias, LayerNorm2
    (its affine folded into fc1 at pack time), gated depthwise FFN over 360 hidden units padded to 384, residual, next LayerNorm,
    its GAP partials and compact 16-channel copy — against an fp64 restatement (hat_arch.py:233-237, :107-119; esc_arch.py:123),
    interior and border tiles, ragged sizes, B = 2 with per-sample scales.  Channels 180..191 of the 12th channel tile are dead
    lanes: nothing may leak from them."""
    B, H, W = geom
    C, hid, ldc = 180, 360, 184
    dev, ops = _dev(), _ops()
    dt, tdt = ops.DTYPE_CODE["bf16"], torch.bfloat16
    n = q(rnd("l8n", (B, H, W, C)), "bf16")
    y16 = q(rnd("l8y", (B, H, W, 16)), "bf16")
    c2 = q(rnd("l8c2", (B, H, W, C), std=2.0), "bf16")
    scale = 0.05 + 0.3 * torch.rand(B, C, generator=torch.Generator().manual_seed(5))
    t = rnd("l8t", (B, H, W, C), std=1.5) + 0.3
    wa, ba = q(rnd("l8wa", (C, C), std=C ** -0.5), "bf16"), rnd("l8ba", (C,), std=0.1)
    sd = {
        "n2.weight": 1 + rnd("l8g", (C,), std=0.1), "n2.bias": rnd("l8bb", (C,), std=0.1),
        "m.fc1.weight": rnd("l8f1w", (2 * hid, C), std=C ** -0.5), "m.fc1.bias": rnd("l8f1b", (2 * hid,), std=0.1),
        "m.dw.weight": rnd("l8dw", (2 * hid, 1, 3, 3), std=1 / 3).half().float(), "m.dw.bias": rnd("l8db", (2 * hid,), std=0.1).half().float(),
        "m.fc2.weight": rnd("l8f2w", (C, hid), std=hid ** -0.5).half().float(), "m.fc2.bias": rnd("l8f2b", (C,), std=0.1),
        "n1.weight": 1 + rnd("l8g1", (C,), std=0.1), "n1.bias": rnd("l8b1", (C,), std=0.1),
    }
    sdd = {k: v.double() for k, v in sd.items()}
    xin = torch.cat([y16, n[..., 16:]], -1)
    tb = (t.double() + F.linear(xin.double(), wa.double(), ba.double()) + scale.double()[:, None, None, :] * c2.double()).reshape(B, H * W, C)
    ref = tb + O.gated_dconv_ffn(O._ln(tb, sdd, "n2"), (H, W), sdd, "m")
    ref_n = O._ln(ref, sdd, "n1")
    pw = ops.pack_linear_weight(wa, ba, dt, dev)
    assert (pw.nt, pw.kpad) == (12, 192) and ops.tail3_supported(C, hid, dt)
    pf = ops.pack_ffn3(sd["m.fc1.weight"], sd["m.fc1.bias"], sd["m.dw.weight"], sd["m.dw.bias"], sd["m.fc2.weight"], sd["m.fc2.bias"],
                       sd["n2.weight"], sd["n2.bias"], dev)
    assert ops.hab_tail_supported(pf, pw, 60, dt) and pf.chunks == 12
    nd, yd, cd = to_dev(n, ldc, tdt, dev), to_dev(y16, 16, tdt, dev), to_dev(c2, ldc, tdt, dev)
    td = t.reshape(B, H * W, C).to(dev).contiguous()
    scd = torch.zeros(B, 256, device=dev)
    scd.view(-1)[:B * 192].view(B, 192)[:, :C] = scale.to(dev)      # rows of 192 floats, as hat_eca_scale writes them for C = 180
    b256 = torch.zeros(256, device=dev)
    b256[:C] = ba.to(dev)
    dv = lambda k: sd[k].to(dev).contiguous()
    tiles = -(-H // 8) * -(-W // 16)
    out, n1, gap, n16 = torch.full_like(td, 7.0), torch.zeros(B, H * W, ldc, dtype=tdt, device=dev), torch.zeros(B, tiles, 16, device=dev), \
        torch.zeros(B, H * W, 16, dtype=tdt, device=dev)
    ops.hab_tail(pf, pw, td, out, dv("n2.weight"), dv("n2.bias"), n=nd, ldn_in=ldc, y16=yd, bias_b=b256, B=B, H=H, W=W, dtype=dt,
                 ln1=(dv("n1.weight"), dv("n1.bias")), n_out=n1, ldn=ldc, gap_out=gap, gap_c=16, n16_out=n16,
                 r2=cd, ldr2=ldc, r2scale=scd, r2scale_bstride=192)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    upd, upd_ref = out.double().cpu() - tb, ref - tb
    rel = float((upd - upd_ref).norm() / upd_ref.norm())
    assert rel <= 1.2e-2, f"hab tail (embed_dim 180) FFN update rel err {rel:.3e}"
    check(out, ref, "bf16", "hab tail 180 t_out")
    check(n1[:, :, :C].float(), ref_n, "bf16", "hab tail 180 next-LN")
    assert float(n1[:, :, C:].float().abs().max()) == 0.0                     # pad channels of the rows stay untouched
    assert torch.equal(n16, n1[:, :, :16].contiguous())
    check(gap.sum(1) / (H * W), n1[:, :, :16].float().mean(1), "f32", "hab tail 180 gap = pool of the stored rows", f32_tol=1e-5)


@pytest.mark.parametrize("case", ["late_spike", "late_spike_overflow", "first_chunk_huge", "all_very_negative", "mixed"])
def test_ocab_attention_fast_kernel_offset_range(case):
    """The bf16 OCAB kernel of the embed_dim-144 models carries the softmax offset in a spare k-slot of the QK^T MFMA, centres
    it on the first key chunk's row maximum and then runs the key window WITHOUT range checks; a query tile whose denominator
    comes out non-finite or above 1e30 repeats its pass with a check and a re-centring step per chunk.  Logits far outside the
    comfortable range — a spike in the last key chunk (~2^87 above the offset: still the unchecked pass; ~2^300 above it: exp2
    overflows, the checked pass runs), a first chunk hundreds above the rest, every logit far below zero, and all of it at once
    across windows — must still match the fp64 softmax (hat_arch.py:375-384) at the bf16 bar."""
    dev, ops = _dev(), _ops()
    ws, heads, C, H, W, B = 16, 6, 144, 32, 48, 1
    wse, d = 24, 24
    qv = q(rnd("oq", (B, H, W, C)) * d ** -0.5, "bf16")
    kv = rnd("okv", (B, H, W, 2 * C))
    table = rnd("otab", ((ws + wse - 1) ** 2, heads), std=0.5)
    if case in ("late_spike", "mixed"):
        kv[0, 19, 19, :C] *= 60.0          # bottom-right of window (0, 0)'s key window: its last key chunk
    if case == "late_spike_overflow":
        kv[0, 19, 19, :C] *= 200.0
    if case in ("first_chunk_huge", "mixed"):
        kv[0, 12:14, 16:40, :C] *= 45.0     # first key rows of the windows in window row 1
    if case == "all_very_negative":
        table = table - 150.0               # every logit ~ -150: exp2 of the raw scores underflows without a re-centre
    kv = q(kv, "bf16")
    ref = O.ocab_attention(qv.double(), kv[..., :C].double(), kv[..., C:].double(), table.double(), O.rpi_oca(ws, 0.5), ws, wse, heads, 1.0)
    M = ws + wse - 1
    rot = (torch.arange(M * M) + (ws - wse + 1 - (ws - 1)) * (M + 1)) % (M * M)
    out = torch.zeros(B, H * W, C, dtype=torch.bfloat16, device=dev)
    # the log2 entry takes q * log2(e), rounded ONCE (the engine folds the factor into the projection's weights): the reference
    # is the softmax of exactly those queries
    ql = q(qv * ops.LOG2E, "bf16")
    ref = O.ocab_attention((ql / ops.LOG2E).double(), kv[..., :C].double(), kv[..., C:].double(), table.double(), O.rpi_oca(ws, 0.5), ws, wse, heads, 1.0)
    assert ops.ocab_attention_log2_supported(C, heads, ws, wse, ops.HAT_BF16)
    ops.ocab_attention(to_dev(ql, C, torch.bfloat16, dev), to_dev(kv, 2 * C, torch.bfloat16, dev), table[rot].t().contiguous().to(dev),
                       out, B=B, H=H, W=W, C_=C, heads=heads, ws=ws, wse=wse, ldq=C, ldkv=2 * C, ldo=C, dtype=ops.HAT_BF16, q_log2=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    check(out.float().reshape(B, H, W, C), ref, "bf16", f"fast OCAB kernel (log2 queries), {case}")
