"""Time the Upsample convs (64 -> 256 + PixelShuffle(2)) of the headline frame through hat_conv (HAT_NO_CONV64R=1: hat_conv's
general kernel instead of the resident-weight one) and check the two against each other."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_resolution_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = ops.HAT_BF16
torch.manual_seed(0)
for h, w_ in ((720, 1280), (1440, 2560)):
    wgt = torch.randn(256, 64, 3, 3) * (9 * 64) ** -0.5
    pw = ops.pack_conv_weight(wgt, torch.randn(256) * 0.1, dt, dev)
    x = torch.randn(1, h * w_, 64, device=dev).to(torch.bfloat16)
    out = torch.zeros(1, h * w_ * 4, 64, dtype=torch.bfloat16, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(13):
        if i == 3:
            a.record()
        ops.conv(pw, x, out, B=1, H=h, W=w_, dtype=dt, ldx=64, ldo=64, out_mode=ops.O_PIXSHUF_T, ps_r=2)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"64->256 3x3 + PixelShuffle at {h}x{w_}: {ms:.3f} ms = {2 * h * w_ * 64 * 256 * 9 / ms / 1e9:.0f} TFLOP/s; checksum {float(out.float().abs().sum()):.6e}")
