"""Localise a fused-FFN bug: run hat_ffn on structured weights and compare with the oracle (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import hat_oracle as O
from super_resolution_amd import ops, synth

dev = torch.device("cuda:0")
C, B, H, W = 144, 1, 24, 32
hid = 2 * C
rnd = lambda k, s, std=1.0: synth.normal(5, k, s, std=std)


def run(name, dtype, W1, b1, Wd, bd, W2, b2):
    dt = ops.DTYPE_CODE[dtype]
    t = rnd("t", (B, H * W, C), 1.5) + 0.3
    sd = {"n2.weight": 1 + rnd("g", (C,), 0.1), "n2.bias": rnd("b", (C,), 0.1), "m.fc1.weight": W1, "m.fc1.bias": b1,
          "m.dw.weight": Wd, "m.dw.bias": bd, "m.fc2.weight": W2, "m.fc2.bias": b2}
    sdd = {k: v.double() for k, v in sd.items()}
    ref = t.double() + O.gated_dconv_ffn(O._ln(t.double(), sdd, "n2"), (H, W), sdd, "m")
    pf = ops.pack_ffn(W1, b1, Wd, bd, W2, b2, dt, dev)
    tout = torch.zeros(B, H * W, C, device=dev)
    ops.ffn(pf, t.to(dev), tout, sd["n2.weight"].to(dev), sd["n2.bias"].to(dev), B=B, H=H, W=W, dtype=dt)
    torch.cuda.synchronize()
    upd, upd_ref = tout.double().cpu() - t.double(), ref - t.double()
    err = (upd - upd_ref).abs()
    e_img = err.reshape(H, W, C).amax(-1)
    print(f"{name:28s} {dtype}: max|upd-ref| {float(err.max()):.3e}  |ref| {float(upd_ref.abs().max()):.3e}  "
          f"interior err {float(e_img[2:-2, 2:-2].max()):.2e}  border err {float(torch.cat([e_img[0], e_img[-1], e_img[:, 0], e_img[:, -1]]).max()):.2e}"
          f"  per-channel-group err {[round(float(err.reshape(-1, C)[:, i:i+16].max()), 4) for i in range(0, C, 48)]}")


Z = torch.zeros
W1r, b1r = rnd("w1", (2 * hid, C), C ** -0.5), rnd("b1", (2 * hid,), 0.5)
Wdr, bdr = rnd("wd", (2 * hid, 1, 3, 3), 1 / 3), rnd("bd", (2 * hid,), 0.5)
W2r, b2r = rnd("w2", (C, hid), hid ** -0.5), rnd("b2", (C,), 0.5)
ctr = Z(2 * hid, 1, 3, 3); ctr[:, 0, 1, 1] = 1.0
avg = torch.full((2 * hid, 1, 3, 3), 1 / 9)
for dtype in ("f32", "bf16"):
    run("A only b2", dtype, Z(2 * hid, C), Z(2 * hid), Z(2 * hid, 1, 3, 3), Z(2 * hid), Z(C, hid), b2r)
    run("B b1 const, center tap, W2", dtype, Z(2 * hid, C), b1r, ctr, Z(2 * hid), W2r, Z(C))
    run("C b1 const, avg taps, W2", dtype, Z(2 * hid, C), b1r, avg, Z(2 * hid), W2r, Z(C))
    run("D W1 rand, center tap", dtype, W1r, b1r, ctr, Z(2 * hid), W2r, Z(C))
    run("E dw bias only", dtype, Z(2 * hid, C), Z(2 * hid), Z(2 * hid, 1, 3, 3), bdr, W2r, Z(C))
    run("F dw rand taps, b1 const", dtype, Z(2 * hid, C), b1r, Wdr, Z(2 * hid), W2r, Z(C))
    run("G everything", dtype, W1r, b1r, Wdr, bdr, W2r, b2r)

print("---- single taps (b1 const, W1 = 0): which taps contribute? f32")
for tp in range(9):
    wt = Z(2 * hid, 1, 3, 3); wt[:, 0, tp // 3, tp % 3] = 1.0
    run(f"tap {tp}", "f32", Z(2 * hid, C), b1r, wt, Z(2 * hid), W2r, Z(C))
