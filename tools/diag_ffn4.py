"""Debug helper (GPU box): with the HAT_FFN_DEBUG_DUMP build of hat_ffn, compare chunk 0's U and depthwise
accumulators against expectations computed with torch."""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["HAT_MI355X_LIB"] = os.path.join(root, "super_resolution_amd/variants/lib_DUMP.so")
sys.path.insert(0, root)
import torch, torch.nn.functional as F
from super_resolution_amd import ops, synth
dev = torch.device("cuda:0")
C, B, H, W = 144, 1, 8, 16
hid = 2 * C
rnd = lambda k, s, std=1.0: synth.normal(5, k, s, std=std)
t = rnd("t", (B, H * W, C), 1.5) + 0.3
g_, b_ = 1 + rnd("g", (C,), 0.1), rnd("b", (C,), 0.1)
W1, b1 = rnd("w1", (2 * hid, C), C ** -0.5), rnd("b1", (2 * hid,), 0.5)
Wd, bd = rnd("wd", (2 * hid, 1, 3, 3), 1 / 3), rnd("bd", (2 * hid,), 0.5)
W2, b2 = rnd("w2", (C, hid), hid ** -0.5), rnd("b2", (C,), 0.5)
for dtype in ("f32", "bf16"):
    dt = ops.DTYPE_CODE[dtype]
    pf = ops.pack_ffn(W1, b1, Wd, bd, W2, b2, dt, dev)
    tout = torch.zeros(B, H * W, C, device=dev)
    ops.ffn(pf, t.to(dev), tout, g_.to(dev), b_.to(dev), B=B, H=H, W=W, dtype=dt)
    torch.cuda.synchronize()
    got = tout.cpu().reshape(H, W, C)
    m = F.layer_norm(t.double(), (C,), g_.double(), b_.double(), 1e-5)
    u = F.linear(m, W1.double(), b1.double()).reshape(H, W, 2 * hid)
    u0 = torch.cat([u[..., :32], u[..., hid:hid + 32]], -1)            # chunk 0: a 0..31, gate 0..31
    e_u = (got[..., :64].double() - u0).abs()
    print(dtype, "U err max", float(e_u.max()), " per 16-ch group", [round(float(e_u[..., i:i + 16].max()), 4) for i in range(0, 64, 16)],
          " |U|max", float(u0.abs().max()))
    ud = F.conv2d(u.permute(2, 0, 1)[None], Wd.double(), bd.double(), padding=1, groups=2 * hid)[0].permute(1, 2, 0)
    d0 = torch.cat([ud[..., :32], ud[..., hid:hid + 32]], -1)
    e_d = (got[..., 64:128].double() - d0).abs()
    print(dtype, "dw err max", float(e_d.max()), " per group [a0,a1,g0,g1]", [round(float(e_d[..., i:i + 16].max()), 4) for i in range(0, 64, 16)],
          " |dw|max", float(d0.abs().max()))
    print("   sample got dw a0[0:4] at (3,5):", got[3, 5, 64:68].tolist(), "exp", d0[3, 5, 0:4].tolist())
    print("   sample got dw g0[0:4] at (3,5):", got[3, 5, 96:100].tolist(), "exp", d0[3, 5, 32:36].tolist())
    print("   sample got U[0:4] at (3,5):", got[3, 5, 0:4].tolist(), "exp", u0[3, 5, 0:4].tolist())
