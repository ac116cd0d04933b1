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
    run(stamps)
    torch.cuda.synchronize()
    s = stamps.reshape(tiles, -1, 8).cpu().double()
    nw = int((s[:, :, 0] > 0).sum(1).max())
    s = s[:, :nw]
    # timeline per CU from wave 0 of every workgroup: start (48-bit), HW_ID (cu 11:8, sh 12, se 15:13), XCC_ID
    w0 = s[:, 0]
    start = w0[:, 3] + w0[:, 6] * 2.0 ** 24
    dur = w0[:, :3].sum(1)
    hw = w0[:, 4].long()
    cu = ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7) + 256 * (w0[:, 5].long() & 0xf)
    t0 = float(start.min())
    span = float((start + dur).max()) - t0
    busy, gaps, ncu = 0.0, [], 0
    for c in cu.unique():
        m = cu == c
        st, du = start[m] - t0, dur[m]
        order = st.argsort()
        st, du = st[order], du[order]
        ncu += 1
        busy += float(du.sum())
        if ncu <= 2:
            print("   cu", int(c), [(int(a), int(b)) for a, b in zip(st[:8].tolist(), du[:8].tolist())])
    print(f"   span {span:.0f} ticks ({span / ms / 1e6:.2f} ticks/ns), {ncu} CUs seen, mean workgroups in flight per CU {busy / span / ncu:.2f}")
    print(f"{name}: {ms:.3f} ms/launch, {tiles} tiles, {nw} waves; mean cycles (100 MHz ticks x ~24): stage {s[:,:,0].mean():.0f}  kloop {s[:,:,1].mean():.0f}  "
          f"epilogue {s[:,:,2].mean():.0f}  (ticks)")
