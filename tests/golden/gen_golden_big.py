#!/usr/bin/env python3
"""Reference outputs AT THE BENCHMARKED SIZES (BASELINE.json headline, cfg3, cfg4, cfg5).

Same recipe as gen_golden.py: the reference `HAT` itself is imported in the build container (loader
shim of SURVEY.md §8c) and run on CPU; only DATA is stored — crops, a strided sample of the whole
output and global checksums (the whole tensors are 50-180 MB each, too large to commit).  Inputs and
weights are regenerated from seeds by `super_resolution_amd.synth`.  Takes ~25 minutes of CPU time.

    python tests/golden/gen_golden_big.py [--only headline,cfg3,cfg5,cfg4] [--threads 8]

  headline  HAT-S x4, 1x3x720x1280                      -> big_headline_HAT-S_x4_720p.npz
  cfg3      HAT-L x4, 1x3x512x512                       -> big_cfg3_HAT-L_x4_512.npz
  cfg5      HAT-L x4, samples 0 / 13 / 31 of the 32x3x256x256 batch, run alone (the reference's eval
            branch is B = 1 only)            -> big_cfg5_HAT-L_x4_256.npz
  cfg4      HAT-L x4 on two of the 8 balanced tiles of the 720x1280 frame (the padded crops the
            tile loop of hat_model.py:40-108 hands to the net; core of the result kept)
                                                        -> big_cfg4_HAT-L_x4_tiles.npz
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gen_golden import CFGS, W_SEED, X_SEED, build, import_reference  # noqa: E402
from super_resolution_amd import synth  # noqa: E402
from super_resolution_amd import tile_parallel as tp  # noqa: E402

STRIDE = (37, 41)   # co-prime with the 16-pixel window, the x4 scale and each other: hits every residue class


def big_summary(y: torch.Tensor, crop: int = 64):
    """Crops at 7 positions, a strided sample of the whole tensor and float64 checksums."""
    h, w = y.shape[-2:]
    c = min(crop, h, w)
    pos = {"tl": (0, 0), "tr": (0, w - c), "bl": (h - c, 0), "br": (h - c, w - c), "ce": ((h - c) // 2, (w - c) // 2),
           "q1": (h // 4 + 3, w // 4 + 5), "q3": (3 * h // 4 - c - 7, 3 * w // 4 - c - 11)}
    yd = y.double()
    out = {"mean": np.float64(yd.mean()), "std": np.float64(yd.std()), "abs_sum": np.float64(yd.abs().sum()),
           "strided": y[..., ::STRIDE[0], ::STRIDE[1]].numpy().copy(), "stride": np.array(STRIDE),
           "row_sums": yd.sum(dim=-1).float().numpy().copy(), "col_sums": yd.sum(dim=-2).float().numpy().copy()}
    for k, (a, b) in pos.items():
        a, b = max(0, min(a, h - c)), max(0, min(b, w - c))
        out["crop_" + k] = y[..., a:a + c, b:b + c].numpy().copy()
        out["pos_" + k] = np.array([a, b, c])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="headline,cfg3,cfg5,cfg4")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    todo = set(args.only.split(","))
    torch.set_num_threads(args.threads)
    HAT = import_reference()
    t0 = time.time()
    with torch.no_grad():
        if "headline" in todo:
            net, _ = build(HAT, CFGS["HAT-S_x4"])
            shape = (1, 3, 720, 1280)
            y = net(synth.synth_input(X_SEED, shape))
            np.savez_compressed(f"{HERE}/big_headline_HAT-S_x4_720p.npz", x_shape=np.array(shape), **big_summary(y))
            print(f"headline done {time.time() - t0:.0f}s", flush=True)
            del net, y
        if todo & {"cfg3", "cfg5", "cfg4"}:
            net, _ = build(HAT, CFGS["HAT-L_x4"])
        if "cfg3" in todo:
            shape = (1, 3, 512, 512)
            y = net(synth.synth_input(X_SEED, shape))
            np.savez_compressed(f"{HERE}/big_cfg3_HAT-L_x4_512.npz", x_shape=np.array(shape), **big_summary(y))
            print(f"cfg3 done {time.time() - t0:.0f}s", flush=True)
            del y
        if "cfg5" in todo:
            shape = (32, 3, 256, 256)
            x = synth.synth_input(X_SEED, shape)
            out = {"x_shape": np.array(shape), "samples": np.array([0, 13, 31])}
            for i in (0, 13, 31):
                y = net(x[i:i + 1].contiguous())
                for k, v in big_summary(y, 48).items():
                    out[f"s{i}_{k}"] = v
                print(f"cfg5 sample {i} done {time.time() - t0:.0f}s", flush=True)
            np.savez_compressed(f"{HERE}/big_cfg5_HAT-L_x4_256.npz", **out)
        if "cfg4" in todo:
            shape = (1, 3, 720, 1280)
            x = synth.synth_input(X_SEED, shape)
            tiles = tp.balanced_tiles(720, 1280, 8, 16, 32)
            out = {"x_shape": np.array(shape), "tiles": np.array([list(t) for t in tiles]), "picked": np.array([0, 5])}
            for i in (0, 5):    # a corner tile (two frame borders) and an interior-column tile of the second row
                core = tp.run_tile(x, net, tiles[i], 4)
                for k, v in big_summary(core).items():
                    out[f"t{i}_{k}"] = v
                print(f"cfg4 tile {i} {tuple(tiles[i])} done {time.time() - t0:.0f}s", flush=True)
            np.savez_compressed(f"{HERE}/big_cfg4_HAT-L_x4_tiles.npz", **out)
    print("big goldens written", flush=True)


if __name__ == "__main__":
    main()
