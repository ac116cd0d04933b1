"""Per-phase cycle counts of hat_conv's kernel (staging / K loop / epilogue) on the group conv of the headline frame.

Needs a library built with -DHAT_CONV_STAMPS (tools/conv_phases.sh builds it next to the shipped one and points
HAT_MI355X_LIB at it); the stamps ride in the gap_out field, which the shipped build rejects without ln_out."""
import sys

import torch

sys.path.insert(0, ".")
from super_resolution_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, W, C = 1, 720, 1280, 144
dt = ops.HAT_BF16
for name, cin, cout, h, w_, xf32 in [("group conv 144->144 720p (bf16 in)", 144, 144, H, W, False),
                                     ("group conv 144->144 720p (f32 in)", 144, 144, H, W, True),
                                     ("upsample 64->256 1440p", 64, 256, 2 * H, 2 * W, False)]:
    wgt = torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5
    pw = ops.pack_conv_weight(wgt, torch.zeros(cout), dt, dev)
    x = torch.randn(B, h * w_, cin, device=dev).to(torch.float32 if xf32 else torch.bfloat16)
    tiles = ops.conv_tiles(pw, h, w_, dt)
    stamps = torch.zeros(tiles * 8 * 8, dtype=torch.float32, device=dev)
    if cout == 144:
        r1 = torch.randn(B, h * w_, C, device=dev)
        kw = dict(ldo=C, out_mode=ops.O_NHWC_F32, r1=r1, ldr1=C)
        out = r1
    else:
        out = torch.zeros(B, h * w_ * 4, 64, dtype=torch.bfloat16, device=dev)
        kw = dict(ldo=64, out_mode=ops.O_PIXSHUF_T, ps_r=2)
    def run(g):
        ops.conv(pw, x, out, B=B, H=h, W=w_, dtype=dt, ldx=cin, x_mode=(ops.X_NHWC_F32 if xf32 else ops.X_NHWC_T), gap_out=g, gap_c=4, ln=None, **kw)
    import ctypes as Cc
    from super_resolution_amd import _lib
    dd = ops.HatConvDesc()
    dd.B, dd.H, dd.W, dd.Cin, dd.ksize, dd.nt, dd.n_slices, dd.dtype, dd.Kpad = 1, h, w_, cin, 3, pw.nt, pw.n_slices, dt, pw.kpad
    occ = Cc.c_int32(-1)
    rc = _lib.load().hat_conv_occupancy(Cc.byref(dd), Cc.byref(occ))
    print(f"   occupancy query rc={rc}: {occ.value} workgroups per CU")
    for it in range(3):
        run(None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(10):
        run(None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.3f} ms/launch", flush=True)
    run(stamps)
    torch.cuda.synchronize()
    wv, pt_, tl, lds = Cc.c_int32(0), Cc.c_int32(0), Cc.c_int32(0), Cc.c_int64(0)
    _lib.load().hat_conv_plan(Cc.byref(dd), Cc.byref(wv), Cc.byref(pt_), Cc.byref(tl), Cc.byref(lds))
    nw = wv.value
    s = stamps[:tiles * nw * 8].reshape(tiles, nw, 8).cpu().double()
    # timeline per CU from wave 0 of every workgroup: start (48-bit), HW_ID (cu 11:8, sh 12, se 15:13), XCC_ID
    w0 = s[:, 0]
    start = w0[:, 3] + w0[:, 6] * 2.0 ** 24
    dur = w0[:, :3].sum(1)
    hw = w0[:, 4].long()
    cu = ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7) + 256 * (w0[:, 5].long() & 0xf)
    if float(dur.max()) <= 0:
        continue
    t0 = float(start.min())
    span = float((start + dur).max()) - t0
    ncu = int(cu.unique().numel())
    print(f"   {nw} waves, LDS {lds.value} B; span {span:.0f} ticks ({span / ms / 1e6:.2f} ticks/ns), {ncu} CUs seen, "
          f"mean workgroups in flight per CU {float(dur.sum()) / span / ncu:.2f}")
    print(f"{name}: {ms:.3f} ms/launch, {tiles} tiles; mean ticks per tile: stage {s[:,:,0].mean():.0f}  kloop {s[:,:,1].mean():.0f}  "
          f"epilogue {s[:,:,2].mean():.0f}")
