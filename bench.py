#!/usr/bin/env python3
"""bench.py — headline benchmark: SR output megapixels/s, HAT-S x4, 1280x720 LR in (BASELINE.json).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one forward of the whole HAT network on one synthetic 3x720x1280 frame already
resident in HBM (weights resident, H2D/D2H excluded).
  N == 1 : the full frame on one GPU (the configuration the metric is quoted on).
  N  > 1 : the SAME frame cut into N balanced window-aligned tiles with the reference's tile
           semantics (tile_pad 32, hat_model.py:40-108), one tile per rank, one RCCL all-gather of
           the output cores per step  ->  "strong" scaling.  (`--mode frames` instead gives every
           rank its own full frame, no collective: "weak" scaling.)
Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (largest share of the
step), timed live with HIP events on the launch stream; `cpu_baseline` is the CPU oracle (a
restatement of the reference forward, pinned to reference-generated goldens) timed on the host
cores on a bounded crop of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODELS = {
    "HAT-S": dict(in_chans=3, img_size=64, window_size=16, compress_ratio=24, squeeze_factor=24, conv_scale=0.01,
                  overlap_ratio=0.5, img_range=1.0, depths=[6] * 6, embed_dim=144, num_heads=[6] * 6, mlp_ratio=2,
                  upsampler="pixelshuffle", resi_connection="1conv"),
    "HAT": dict(in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30, conv_scale=0.01,
                overlap_ratio=0.5, img_range=1.0, depths=[6] * 6, embed_dim=180, num_heads=[6] * 6, mlp_ratio=2,
                upsampler="pixelshuffle", resi_connection="1conv"),
    "HAT-L": dict(in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30, conv_scale=0.01,
                  overlap_ratio=0.5, img_range=1.0, depths=[6] * 12, embed_dim=180, num_heads=[6] * 12, mlp_ratio=2,
                  upsampler="pixelshuffle", resi_connection="1conv"),
}
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # HBM3E spec peak (same guide; ~6300 GB/s is what streaming kernels reach)
W_SEED, X_SEED = 1234, 7


def algorithmic_flops_per_lr_pixel(cfg, scale):
    """SURVEY App. B (2*MAC over convs, linears, QK^T, AV)."""
    C = cfg["embed_dim"]
    hid, pd, K, ow = int(C * cfg["mlp_ratio"]), 16, 13, 24
    mid = C // cfg["compress_ratio"]
    hab = 2 * (2 * 9 * C * mid) + 2 * K * K * pd * pd + 2 * 9 * pd + 2 * C * C + 2 * C * (2 * hid) + 2 * 9 * (2 * hid) + 2 * hid * C
    ocab = 2 * C * C + 4 * C * C + 2 * C * C + 2 * (2 * C * hid) + 2 * (2 * ow * ow * C)
    head = 2 * 27 * C + 18 * C * C + 2 * 9 * C * 64
    if scale == 4:
        tail = 2 * 9 * 64 * 256 * (1 + 4) + 2 * 9 * 64 * 3 * 16
    elif scale == 2:
        tail = 2 * 9 * 64 * 256 + 2 * 9 * 64 * 3 * 4
    else:
        tail = 2 * 9 * 64 * 576 + 2 * 9 * 64 * 3 * 9
    total = head + tail
    for d in cfg["depths"]:
        total += d * hab + ocab + 18 * C * C
    return float(total)


def cpu_baseline(model, scale, crop, threads):
    """The CPU oracle (restatement of the reference forward) on `threads` host cores, one forward of a crop."""
    from oracle import hat_oracle as O
    from super_resolution_amd import synth
    torch.set_num_threads(threads)
    cfg = O.make_cfg(upscale=scale, **MODELS[model])
    sd = synth.synth_state_dict(O.blank_state_dict(cfg), W_SEED)
    x = synth.synth_input(X_SEED, (1, 3, crop, crop))
    with torch.no_grad():
        t0 = time.perf_counter()
        y = O.hat_forward(x, sd, cfg)
        dt = time.perf_counter() - t0
    return {"value": round(y.shape[-1] * y.shape[-2] / 1e6 / dt, 5), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": f"{model} x{scale} fp32, one forward of a 3x{crop}x{crop} crop of the synthetic frame in {dt:.1f} s "
                      f"(cost is linear in pixels: fixed-size windows)"}


def f32_path(args, cfg, x, dev, fl_frame):
    """The same frame on the exact-fp32 kernel path (fp32 storage, v_mfma_f32_16x16x4_f32 = an fp32 fmaf chain): the
    precision the reference itself computes in.  Timed OUTSIDE the bf16 region, after it; reported beside the headline."""
    from super_resolution_amd import synth
    from super_resolution_amd.registry import build_network
    net = build_network(dict(type="HAT", upscale=args.scale, compute_dtype="f32", **cfg)).eval()
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), W_SEED), strict=True)
    net = net.to(dev)
    net(x)
    torch.cuda.synchronize()
    n = 2
    t0 = time.perf_counter()
    for _ in range(n):
        net(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    tf = fl_frame / (ms * 1e-3) / 1e12
    return {"dtype": "f32", "ms_per_frame": round(ms, 2), "frames_timed": n, "achieved_tflops": round(tf, 2),
            "frac_of_f32_mfma_peak": round(tf / MFMA_PEAK_TFLOPS["f32"], 4), "peak_tflops": MFMA_PEAK_TFLOPS["f32"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="HAT-S", choices=list(MODELS))
    ap.add_argument("--scale", type=int, default=4)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--mode", default="tiles", choices=["tiles", "frames", "shard"],
                    help="N>1: one frame cut in independent tiles (reference tile semantics), one frame per rank, or one frame "
                         "sharded EXACTLY into row bands that exchange halo rows and pool sums (SURVEY §8 f4)")
    ap.add_argument("--tile-pad", type=int, default=32)
    ap.add_argument("--cpu-crop", type=int, default=256, help="side of the crop timed on the CPU (0 disables the baseline)")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-f32-path", action="store_true", help="skip the exact-fp32 path timing reported beside the bf16 headline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as children, BEFORE anything in
        # this process touches the GPU (a process that has initialised HIP must never exec another program on this
        # pool), relay their output (rank 0 prints the one JSON line) and exit with their code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only path")
    # HAT_BENCH_REHEARSE_ON_ONE_GPU=1: all ranks share cuda:0 and talk over gloo — only to exercise the N > 1 code path on
    # a one-GPU box (the numbers mean nothing); the real run is one rank per GPU over RCCL
    one_gpu = os.environ.get("HAT_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    dev = torch.device("cuda", 0 if one_gpu else local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL

    from super_resolution_amd import ops, synth, tile_parallel as tp
    from super_resolution_amd.registry import build_network
    import super_resolution_amd.archs  # noqa: F401

    cfg = dict(MODELS[args.model])
    net = build_network(dict(type="HAT", upscale=args.scale, compute_dtype=args.dtype, **cfg)).eval()
    net.load_state_dict(synth.synth_state_dict(net.state_dict(), W_SEED), strict=True)
    net = net.to(dev)
    H, W, s = args.height, args.width, args.scale
    x = synth.synth_input(X_SEED, (1, 3, H, W)).to(dev)
    ws = cfg["window_size"]
    if H % ws or W % ws:  # HATModel.pre_process (hat_model.py:16-26); 720x1280 needs none
        x = torch.nn.functional.pad(x, (0, (ws - W % ws) % ws, 0, (ws - H % ws) % ws), "reflect")
    Hp, Wp = x.shape[-2:]

    frames_per_step = 1
    if world == 1 or args.mode == "frames":
        workload = f"{args.model} x{s}, one 3x{H}x{W} LR frame per GPU, full-frame forward"
        scaling = "weak"
        frames_per_step = world

        def step():
            return net(x)
    elif args.mode == "shard":
        workload = (f"{args.model} x{s}, one 3x{H}x{W} LR frame sharded exactly into {world} row bands (halo rows by send/recv "
                    f"between neighbours, pool sums by all-reduce), one band per GPU + all-gather of output rows")
        scaling = "strong"

        def step():
            return net.forward_band_parallel(x)
    else:
        tiles = tp.balanced_tiles(Hp, Wp, world, ws, args.tile_pad)
        workload = (f"{args.model} x{s}, one 3x{H}x{W} LR frame cut into {world} window-aligned tiles "
                    f"(reference tile semantics, tile_pad {args.tile_pad}), one tile per GPU + all-gather of output cores")
        scaling = "strong"
        out_buf = torch.zeros(1, 3, Hp * s, Wp * s, device=dev)

        def step():
            return tp.tile_parallel_forward(x, net, s, tiles, out=out_buf)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    out_mp = frames_per_step * (H * s) * (W * s) / 1e6
    value = out_mp / (ms_per_step / 1e3)

    # ---- per-kernel timing with HIP events on the launch stream (rank 0) -------------------------
    roofline, kernels = None, None
    if rank == 0 and not args.no_kernel_profile:
        agg, layers = {}, {}
        reps = 2
        for _ in range(reps):
            with ops.profile() as rec:
                if world == 1 or args.mode in ("frames", "shard"):   # (shard: the kernels of a full frame; a band's are a slice of them)
                    net(x)
                else:
                    tp.run_tile(x, net, tiles[tp.assign(tiles, world)[0][0]], s)
                torch.cuda.synchronize()
                for name, fl, a, b, tag, nb in rec:
                    ms = a.elapsed_time(b)
                    e = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
                    e[0] += 1
                    e[1] += ms
                    e[2] += fl
                    e[3] += nb
                    if tag:
                        e2 = layers.setdefault(name + " | " + tag, [0, 0.0, 0.0])
                        e2[0] += 1
                        e2[1] += ms
                        e2[2] += fl
        total_ms = sum(e[1] for e in agg.values())
        kernels = {k: {"launches_per_step": e[0] // reps, "avg_ms": round(e[1] / e[0], 4), "share": round(e[1] / total_ms, 4),
                       "tflops": round(e[2] / (e[1] * 1e-3) / 1e12, 2) if e[2] else None}
                   for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1])}
        if os.environ.get("HAT_BENCH_LAYERS"):
            for k, e2 in sorted(layers.items(), key=lambda kv: -kv[1][1]):
                print(f"# {e2[1] / reps:8.3f} ms/step  {e2[0] // reps:3d}x {e2[1] / e2[0]:7.4f} ms  "
                      f"{e2[2] / (e2[1] * 1e-3) / 1e12 if e2[2] else 0:7.1f} TF  {k}", file=sys.stderr)
        dom, e = max(agg.items(), key=lambda kv: kv[1][1])
        avg_s = e[1] / e[0] * 1e-3
        tflops = e[2] / e[0] / avg_s / 1e12
        peak_tf = MFMA_PEAK_TFLOPS[args.dtype]
        if e[3] > 0 and (e[2] / e[3]) < peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9):
            # arithmetic intensity below the machine balance: the HBM roof is the lower one for this kernel
            gbs = e[3] / e[0] / avg_s / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                        "bytes_per_launch": round(e[3] / e[0]), "bytes_unit": "algorithmic HBM bytes per launch (DESIGN.md §5: operands read once, results written once)",
                        "mfma_tflops": round(tflops, 2), "mfma_frac": round(tflops / peak_tf, 4),
                        "intensity_flop_per_byte": round(e[2] / e[3], 1)}
        else:
            roofline = {"bound": "mfma", "kernel": dom, "achieved": round(tflops, 2), "peak": peak_tf, "unit": "TFLOP/s",
                        "frac": round(tflops / peak_tf, 4), "traffic": None}
        roofline.update({"launches_per_step": e[0] // reps, "avg_launch_ms": round(e[1] / e[0], 4),
                         "flop_per_launch": round(e[2] / e[0] / 1e9, 3), "flop_unit": "GFLOP (algorithmic, 2*MAC)"})
        if dom == "tail3_kernel" and roofline["bound"] == "hbm":
            # continuity with rounds 1-3a: until the residual stream between two tails became FP16 rows the same launch needed
            # 1776 B per pixel; `frac` above is on the bytes it needs NOW (1200 / 1488 B per pixel by launch)
            ref_b = frames_per_step / world * H * W * 1776.0
            roofline["frac_on_fp32_stream_bytes"] = round(ref_b / avg_s / 1e9 / HBM_PEAK_GBS, 4)
        # HBM bytes per launch from the PMC counters: measured in separate rocprofv3 --pmc passes (profiles/README.md)
        # and recorded in profiles/; quoted only when the recording is of this exact workload
        try:
            pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            tfile = sorted(f for f in os.listdir(pdir) if f.endswith("_hbm_traffic.json"))[-1]   # the latest round's recording
            with open(os.path.join(pdir, tfile)) as f:
                tr = json.load(f)
            full_frame = world == 1 or args.mode == "frames"   # the recording is of full-frame launches, not of a tile's
            if full_frame and tr.get("workload") == [args.model, s, H, W, args.dtype] and dom in tr.get("kernels", {}):
                roofline["traffic"] = tr["kernels"][dom]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = f"profiles/{tfile} (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
        except (OSError, ValueError):
            pass
    if world > 1:
        dist.barrier()

    if rank == 0:
        fl_frame = algorithmic_flops_per_lr_pixel(cfg, s) * H * W
        path_tflops = fl_frame * frames_per_step / (ms_per_step * 1e-3) / 1e12
        res = {
            "metric": "SR output megapixels/sec, HAT-S x4, 1280x720 LR in" if (args.model, s, H, W) == ("HAT-S", 4, 720, 1280)
            else f"SR output megapixels/sec, {args.model} x{s}, {W}x{H} LR in",
            "value": round(value, 3), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic (uniform[0,1) LR frame, seeded random weights; no checkpoints exist for this fork)",
            "config": {"workload": workload, "lr_size": [H, W], "upscale": s, "parallelism": (f"band{world}" if args.mode == "shard" else f"tile{world}") if scaling == "strong" else f"dp{world}"},
            "path": {"algorithmic_tflop_per_frame": round(fl_frame / 1e12, 3), "achieved_tflops": round(path_tflops, 2),
                     "frac_of_mfma_peak": round(path_tflops / (MFMA_PEAK_TFLOPS[args.dtype] * world), 4)},
            "roofline": roofline,
            "kernels": kernels,
        }
        if world == 1 and args.dtype == "bf16" and not args.no_f32_path:
            res["path_f32"] = f32_path(args, cfg, x, dev, fl_frame)
        if world == 1 and args.cpu_crop > 0:
            # the GPU box gives one GPU a share of 16 host cores; more threads than that only oversubscribe
            cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
            res["cpu_baseline"] = cpu_baseline(args.model, s, args.cpu_crop, cores)
            res["cpu_baseline"]["gpu_over_cpu"] = round(value / res["cpu_baseline"]["value"], 1)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
