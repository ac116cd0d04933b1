"""Time the (S)W-MSA branch (hat_linear -> hat_window_attention -> hat_linear) at the headline geometry:
C = 144, six heads, 16 x 16 windows, one 720 x 1280 map.  python tools/bench_wmsa.py [bf16|f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from super_resolution_amd import ops, synth  # noqa: E402
from super_resolution_amd.archs.window_msa import WindowAttention  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0")
C, heads, ws, H, W = 144, 6, 16, 720, 1280
m = WindowAttention(C, ws, heads, compute_dtype=dtype).eval()
m.load_state_dict(synth.synth_state_dict(m.state_dict(), 1234))
m = m.to(dev)
x = synth.normal(7, "x", (1, H, W, C)).to(dev).to(ops.TORCH_DTYPE[ops.DTYPE_CODE[dtype]])
for shift in (0, 8):
    for _ in range(3):
        m.forward_map(x, shift)
    rec = ops.start_profile() if hasattr(ops, "start_profile") else None
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        m.forward_map(x, shift)
    b.record()
    torch.cuda.synchronize()
    print(f"{dtype} shift {shift}: {a.elapsed_time(b) / 10:.3f} ms per branch (qkv + attention + proj)")
qkv_w, proj_w, bias_flip, dt = m._pack(dev)
tdt = ops.TORCH_DTYPE[dt]
qkv = torch.randn(1, H, W, 3 * C, device=dev).to(tdt)
att = torch.empty(1, H, W, C, dtype=tdt, device=dev)
for shift in (0, 8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(13):
        if i == 3:
            a.record()
        ops.window_attention(qkv, qkv.view(-1)[C:], bias_flip, att, B=1, H=H, W=W, C_=C, heads=heads, ws=ws, shift=shift,
                             ldq=3 * C, ldkv=3 * C, ldo=C, dtype=dt)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{dtype} shift {shift}: attention kernel {ms:.3f} ms = {2 * 2 * 256 * C * H * W / ms / 1e9:.1f} TFLOP/s")
