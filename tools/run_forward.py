"""Two full-frame forwards of the bench model (GPU box) -- the target of `rocprofv3 --pmc ... -- python3 tools/run_forward.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from super_resolution_amd import synth
from super_resolution_amd.registry import build_network
import super_resolution_amd.archs  # noqa: F401

model = sys.argv[1] if len(sys.argv) > 1 else "HAT-S"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (720, 1280)
dev = torch.device("cuda:0")
net = build_network(dict(type="HAT", upscale=4, compute_dtype="bf16", **bench.MODELS[model])).eval()
net.load_state_dict(synth.synth_state_dict(net.state_dict(), bench.W_SEED), strict=True)
net = net.to(dev)
x = synth.synth_input(bench.X_SEED, (1, 3, H, W)).to(dev)
for _ in range(2):
    y = net(x)
torch.cuda.synchronize()
print("ok", tuple(y.shape))
