"""What would a 16-bit residual stream cost in parity?  Runs the 720p headline frame three times in separate processes' worth of
engines (HAT_EMU_T16 unset / fp16 / bf16: the engine then rounds the fp32 residual stream after every HAB tail and group conv)
and reports PSNR and max-abs against the REFERENCE's crops (tests/golden/big_headline_HAT-S_x4_720p.npz).
    PYTHONPATH=. python tools/residual16_psnr.py <mode>      (mode: none | fp16 | bf16; one mode per process: the flag is read at import)"""
import os, sys
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    os.environ["HAT_EMU_T16"] = mode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import hat_oracle as O
from super_resolution_amd import synth
from test_gpu_model import build_net
from helpers import X_SEED, golden
dev = torch.device("cuda:0")
g = golden("big_headline_HAT-S_x4_720p.npz")
x = synth.synth_input(X_SEED, (1, 3, 720, 1280)).to(dev)
y = build_net("HAT-S_x4", "bf16", dev)(x).float().cpu()
ps, mx = [], 0.0
for k in ("tl", "tr", "bl", "br", "ce", "q1", "q3"):
    a, b, c = (int(v) for v in g["pos_" + k])
    ref = torch.as_tensor(g["crop_" + k]).float()
    got = y[..., a:a + c, b:b + c]
    ps.append(O.psnr_float(got, ref)); mx = max(mx, float((got - ref).abs().max()))
print(f"residual stream {mode:5s}: PSNR vs reference crops min {min(ps):.2f} mean {sum(ps)/len(ps):.2f} dB, max-abs {mx:.4f}")
