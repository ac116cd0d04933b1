import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, bench
from super_resolution_amd import synth, tile_parallel as tp
from super_resolution_amd.registry import build_network
import super_resolution_amd.archs
dev = torch.device("cuda:0")
net = build_network(dict(type="HAT", upscale=4, compute_dtype="bf16", **bench.MODELS["HAT-S"])).eval()
net.load_state_dict(synth.synth_state_dict(net.state_dict(), bench.W_SEED), strict=True)
net = net.to(dev)
x = synth.synth_input(bench.X_SEED, (1, 3, 720, 1280)).to(dev)
for n in (2, 4, 8):
    tiles = tp.balanced_tiles(720, 1280, n, 16, 32)
    print(n, [(t.py1 - t.py0, t.px1 - t.px0) for t in tiles])
    ts = []
    for t in tiles:
        tp.run_tile(x, net, t, 4); torch.cuda.synchronize()
        t0 = time.perf_counter(); o = tp.run_tile(x, net, t, 4); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        assert torch.isfinite(o).all()
    print("  per-tile ms:", [round(v, 1) for v in ts], " max", round(max(ts), 1), " -> ideal speedup vs 70 ms:", round(70.0 / max(ts), 2))
y = tp.tile_forward(x, net, 4, tp.balanced_tiles(720, 1280, 8, 16, 32))
print("tile_forward ok", tuple(y.shape), bool(torch.isfinite(y).all()))
