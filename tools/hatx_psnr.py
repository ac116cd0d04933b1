"""bf16 PSNR of the three HATX whole-model cases against the oracle (lowest-index ties) — what the 35 / 40 dB bars are about."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import hat_oracle as O
from super_resolution_amd import synth
from super_resolution_amd.registry import build_network
import super_resolution_amd.archs  # noqa
from helpers import META, W_SEED, X_SEED
dev = torch.device("cuda:0")
cases = [("hatx_tiny_focus_x2", dict(META["cfgs"]["hatx_tiny_focus_x2"]), (1, 3, 16, 24)),
         ("hatx_live_x2", dict(META["cfgs"]["hatx_live_x2"]), (1, 3, 16, 24)),
         ("hatx_train_yml[1,1]", dict(META["cfgs"]["hatx_train_yml"], depths=[1, 1], num_heads=[6, 6]), (1, 3, 32, 48)),
         ("hatx_train_yml[1,1] 64x96", dict(META["cfgs"]["hatx_train_yml"], depths=[1, 1], num_heads=[6, 6]), (1, 3, 64, 96))]
for name, kw, shape in cases:
    cfg = O.make_hatx_cfg(**kw)
    sd = synth.synth_state_dict(O.hatx_blank_state_dict(cfg), W_SEED)
    x = synth.synth_input(X_SEED, shape)
    ref = O.hatx_forward(x, sd, cfg, tie="lowest_index")
    for dtype in ("bf16", "f32"):
        try:
            net = build_network(dict(type="HATX", compute_dtype=dtype, **kw)).eval()
            net.load_state_dict(sd, strict=True)
            y = net.to(dev)(x.to(dev)).float().cpu()
            print(name, shape, dtype, "psnr %.2f dB" % O.psnr_float(y, ref), "max-abs %.3e" % float((y - ref).abs().max()), flush=True)
        except Exception as e:
            print(name, shape, dtype, "EXC", repr(e)[:200], flush=True)
