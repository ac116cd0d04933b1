"""Time hat_linear vs hat_conv (ksize 1) on the 720p 144->144 layer (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from super_resolution_amd import ops
dev = torch.device("cuda:0")
B, H, W, C = 1, 720, 1280, 144
dt = ops.HAT_BF16
x = torch.randn(B, H * W, C, device=dev).to(torch.bfloat16)
w = torch.randn(C, C) * C ** -0.5
b = torch.randn(C) * 0.1
out = torch.zeros(B, H * W, C, device=dev, dtype=torch.bfloat16)
outf = torch.zeros(B, H * W, C, device=dev)
pl, pc = ops.pack_linear_weight(w, b, dt, dev), ops.pack_conv_weight(w, b, dt, dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "lin"):
    print("linear plain  %.3f ms" % t(lambda: ops.linear(pl, x, out, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C)))
    print("linear r1 f32 %.3f ms" % t(lambda: ops.linear(pl, x, outf, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C, out_mode=ops.O_NHWC_F32, r1=outf, ldr1=C)))
if which in ("all", "conv"):
    print("conv   plain  %.3f ms" % t(lambda: ops.conv(pc, x, out, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C)))
    print("conv   r1 f32 %.3f ms" % t(lambda: ops.conv(pc, x, outf, B=B, H=H, W=W, dtype=dt, ldx=C, ldo=C, out_mode=ops.O_NHWC_F32, r1=outf, ldr1=C)))
