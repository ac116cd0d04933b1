"""Error of the band-sharded forward against the unsharded one over a matrix of cases (debug aid for §8 f4)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from super_resolution_amd import synth
from test_gpu_model import build_net
from helpers import X_SEED
dev = torch.device("cuda:0")
cases = [("tiny_x2", "f32", (1, 3, 48, 24), 2), ("tiny_x2", "f32", (2, 3, 48, 24), 2), ("tiny_x2", "f32", (1, 3, 48, 24), 3), ("tiny_x2", "f32", (1, 3, 64, 24), 4),
         ("tiny_x2", "f32", (2, 3, 64, 24), 2), ("tiny_x4", "bf16", (1, 3, 64, 32), 4), ("hats_1g_x4", "f32", (1, 3, 96, 48), 2), ("hats_1g_x4", "bf16", (1, 3, 96, 64), 3),
         ("hats_1g_x4", "bf16", (2, 3, 128, 48), 4), ("hat_1g_x2", "bf16", (1, 3, 64, 48), 2)]
for name, dtype, shape, n in cases:
    try:
        net = build_net(name, dtype, dev)
        x = synth.synth_input(X_SEED, shape).to(dev)
        y0 = net(x).float().cpu()
        y1 = net.forward_bands(x, n).float().cpu()
        d = (y1 - y0).abs()
        rows = d.amax(dim=(0, 1, 3))
        bad = (rows > 1e-4).nonzero().flatten().tolist()
        from oracle import hat_oracle as O
        print(name, dtype, shape, n, "psnr %.1f" % O.psnr_float(y1, y0), "max", float(d.max()), "per-sample", [float(d[b].max()) for b in range(shape[0])], "bad HR rows", (bad[:4], bad[-4:], len(bad)), flush=True)
    except Exception as e:
        print(name, dtype, shape, n, "EXC", repr(e)[:300], flush=True)
