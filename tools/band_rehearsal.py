"""SURVEY §8 f4, stage 2 rehearsal on a ONE-GPU box: world_size ranks share cuda:0 and talk over gloo (payloads staged through
host memory) — exercises band_parallel.forward_band_distributed end to end: halo rows by batched send / recv between neighbouring
ranks, pool sums by all-reduce, output rows by all-gather.  On an 8-GPU node the same call runs one rank per GPU over RCCL.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/band_rehearsal.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from oracle import hat_oracle as O
    from super_resolution_amd import synth
    from test_gpu_model import build_net
    from helpers import X_SEED
    res = []
    for name, dtype, shape in (("tiny_x2", "f32", (2, 3, 64, 24)), ("hats_1g_x4", "bf16", (1, 3, 96, 64)), ("HAT-S_x4", "bf16", (1, 3, 720, 1280))):
        net = build_net(name, dtype, dev)
        x = synth.synth_input(X_SEED, shape).to(dev)
        y0 = net(x).float()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.time()
        y1 = net.forward_band_parallel(x).float()
        torch.cuda.synchronize()
        dt = time.time() - t0
        err = float((y1 - y0).abs().max())
        res.append((name, dtype, shape, world, "max-abs %.3e" % err, "psnr %.1f dB" % O.psnr_float(y1.cpu(), y0.cpu()), "%.2f s" % dt))
        ok = err <= 1e-5 if dtype == "f32" else O.psnr_float(y1.cpu(), y0.cpu()) >= (48.0 if shape[2] < 720 else 43.0)   # (bars of tests/test_gpu_bands.py)
        assert ok, res[-1]
    dist.barrier()
    if rank == 0:
        for r in res:
            print("band rehearsal:", *r, flush=True)
        print("band rehearsal OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
