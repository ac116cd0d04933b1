#!/usr/bin/env python3
"""Round-2 additions to the small golden set, made like gen_golden.py by running THE REFERENCE ITSELF
(build container only): the two constructor options the first round refused.

  tiny_identity_ape_x2   resi_connection='identity' (hat_arch.py:545-546, :748) + ape=True (:699-702, :837-838):
                         whole-model in/out on the one input size ape allows (img_size^2 patches) and the
                         state-dict surface (absolute_pos_embed comes first).

    python tests/golden/gen_golden_r2.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gen_golden import W_SEED, X_SEED, build, import_reference  # noqa: E402
from super_resolution_amd import synth  # noqa: E402

CFGS_R2 = {
    "tiny_identity_ape_x2": dict(upscale=2, in_chans=3, img_size=16, window_size=8, compress_ratio=3, squeeze_factor=30,
                                 conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[2, 1], embed_dim=24,
                                 num_heads=[2, 2], mlp_ratio=2, upsampler="pixelshuffle", resi_connection="identity",
                                 ape=True, esc_pdim=8, esc_kernel=5),
}


def main():
    torch.set_num_threads(4)
    HAT = import_reference()
    with open(f"{HERE}/meta.json") as f:
        meta = json.load(f)
    with open(f"{HERE}/state_dict_surface.json") as f:
        surface = json.load(f)
    with torch.no_grad():
        for name, cfg in CFGS_R2.items():
            net, _ = build(HAT, cfg)
            surface[name] = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
            surface[name + ":nparams"] = sum(p.numel() for p in net.parameters())
            shape = (1, 3, cfg["img_size"], cfg["img_size"])
            np.savez(f"{HERE}/whole_{name}.npz", y=net(synth.synth_input(X_SEED, shape)).numpy(), x_shape=np.array(shape))
            meta["cfgs"][name] = cfg
    with open(f"{HERE}/state_dict_surface.json", "w") as f:
        json.dump(surface, f)
    with open(f"{HERE}/meta.json", "w") as f:
        json.dump(meta, f, indent=1)
    print("round-2 goldens written")


if __name__ == "__main__":
    main()
