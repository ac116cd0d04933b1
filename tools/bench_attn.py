"""Time hat_ocab_attention alone at the headline geometry (C = 144, six heads, 16 -> 24 windows, one 720 x 1280 map)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_resolution_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
import sys as _s
C = int(_s.argv[1]) if len(_s.argv) > 1 else 144
heads, ws, wse, H, W = 6, 16, 24, 720, 1280
dt = ops.HAT_BF16
LD = (C + 7) // 8 * 8
q = torch.randn(1, H * W, LD, device=dev).to(torch.bfloat16)
kv = torch.randn(1, H * W, 2 * LD, device=dev).to(torch.bfloat16)
M = ws + wse - 1
bias = torch.randn(heads, M * M, device=dev) * 0.1
out = torch.empty(1, H * W, LD, dtype=torch.bfloat16, device=dev)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(23):
    if i == 3:
        a.record()
    ops.ocab_attention(q, kv, bias, out, B=1, H=H, W=W, C_=C, heads=heads, ws=ws, wse=wse, ldq=LD, ldkv=2 * LD, ldo=LD, dtype=dt)
b.record()
torch.cuda.synchronize()
print(f"ocab attention 720p C={C}: {a.elapsed_time(b) / 20:.3f} ms per launch")
