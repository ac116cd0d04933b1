"""Time HAT.forward on the BASELINE.json configurations that fit one GPU (bf16, synthetic weights and inputs, inputs
resident in HBM): ms per forward and output MP/s.   python tools/bench_configs.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from super_resolution_amd import synth
from super_resolution_amd.registry import build_network
import super_resolution_amd.archs  # noqa: F401

dev = torch.device("cuda:0")
CASES = [("cfg1 HAT-S x2 1x64x64", "HAT-S", 2, (1, 3, 64, 64)), ("cfg2 HAT-S x4 1x256x256", "HAT-S", 4, (1, 3, 256, 256)),
         ("headline HAT-S x4 1x720x1280", "HAT-S", 4, (1, 3, 720, 1280)), ("cfg3 HAT-L x4 1x512x512", "HAT-L", 4, (1, 3, 512, 512)),
         ("cfg4 (one GPU, whole frame) HAT-L x4 1x720x1280", "HAT-L", 4, (1, 3, 720, 1280)),
         ("cfg5 HAT-L x4 32x256x256", "HAT-L", 4, (32, 3, 256, 256))]
nets = {}
for name, model, s, shape in CASES:
    if (model, s) not in nets:
        net = build_network(dict(type="HAT", upscale=s, compute_dtype="bf16", **bench.MODELS[model])).eval()
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), bench.W_SEED), strict=True)
        nets[(model, s)] = net.to(dev)
    net = nets[(model, s)]
    x = synth.synth_input(bench.X_SEED, shape).to(dev)
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    n = 10 if shape[2] * shape[3] * shape[0] < 500000 else 5
    t0 = time.perf_counter()
    for _ in range(n):
        net(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    mp = shape[0] * shape[2] * s * shape[3] * s / 1e6
    fl = bench.algorithmic_flops_per_lr_pixel(dict(bench.MODELS[model]), s) * shape[0] * shape[2] * shape[3]
    print(f"{name:50s} {ms:9.3f} ms  {mp / ms * 1e3:8.1f} MP/s  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
