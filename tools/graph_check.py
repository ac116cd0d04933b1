"""Graph replay vs eager: equality and time per forward (GPU box).  python tools/graph_check.py [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from super_resolution_amd import synth
from super_resolution_amd.registry import build_network
import super_resolution_amd.archs  # noqa: F401

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (720, 1280)
dev = torch.device("cuda:0")
net = build_network(dict(type="HAT", upscale=4, compute_dtype="bf16", **bench.MODELS["HAT-S"])).eval()
net.load_state_dict(synth.synth_state_dict(net.state_dict(), bench.W_SEED), strict=True)
net = net.to(dev)
x = synth.synth_input(bench.X_SEED, (1, 3, H, W)).to(dev)
x2 = synth.synth_input(bench.X_SEED + 1, (1, 3, H, W)).to(dev)


def timeit(n=10):
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


net.use_graph = False
ye, ye2 = net(x).clone(), net(x2).clone()
te = timeit()
net.use_graph = True
yg, yg2 = net(x), net(x2)
tg = timeit()
torch.cuda.synchronize()
print(f"{H}x{W}: eager {te:.3f} ms, graph {tg:.3f} ms; equal: {torch.equal(ye, yg)} {torch.equal(ye2, yg2)}; "
      f"max diff {float((ye - yg).abs().max()):.3e}")
