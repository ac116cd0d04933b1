#!/usr/bin/env python3
"""Golden vectors for the HATX variant, made like gen_golden.py by running THE REFERENCE ITSELF
(`hat.archs.hatx_arch.HATX`, imported with the same loader shim; build container only).  Data only.

  hatx_tiny_plain_x2    HATX defaults (no focus bias, no pruning): SGFN + ceil-padded OCAB, whole model
  hatx_tiny_focus_x2    use_focus_bias + kv_topk_ratio 0.6 + overlap_ratio 0.6 (odd key window) + OCAB-ESC, whole model
  hatx_blocks           per-block outputs of the second config's first group: SGFN, HAB, OCAB, plus the OCAB with the
                        pruning switched to ||k||_2 scores (use_focus_bias off) — the other branch of hatx_arch.py:437-441
  + the state-dict surface of both configs and of the one live training config (embed_dim 180 ...: surface only)

    python tests/golden/gen_golden_hatx.py
"""
from __future__ import annotations

import importlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gen_golden import W_SEED, X_SEED, import_reference  # noqa: E402
from super_resolution_amd import synth  # noqa: E402

TINY = dict(upscale=2, in_chans=3, img_size=16, window_size=8, compress_ratio=3, squeeze_factor=30, conv_scale=0.01,
            img_range=1.0, depths=[2, 1], embed_dim=24, num_heads=[2, 2], mlp_ratio=2, upsampler="pixelshuffle",
            resi_connection="1conv", esc_pdim=8, esc_kernel=5)
CFGS_X = {
    "hatx_tiny_plain_x2": dict(TINY, overlap_ratio=0.5),
    "hatx_tiny_focus_x2": dict(TINY, overlap_ratio=0.6, kv_topk_ratio=0.6, use_focus_bias=True, ocab_esc_enable=True,
                               ocab_esc_pdim=8, ocab_esc_kernel=5, hab_ffn_ratio=3.0),
    # options/train/train_HAT_SRx2_ESC_OCAB_from_scratch.yml:48-81 (surface only: 21 M parameters)
    "hatx_train_yml": dict(upscale=2, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30,
                           conv_scale=0.03, overlap_ratio=0.6, img_range=1.0, depths=[6] * 6, embed_dim=180,
                           num_heads=[6] * 6, hab_ffn_ratio=3.0, mlp_ratio=4.0, upsampler="pixelshuffle",
                           resi_connection="1conv", esc_pdim=24, esc_kernel=15, ocab_esc_enable=True, ocab_esc_pdim=32,
                           ocab_esc_kernel=17, kv_topk_ratio=0.6, use_focus_bias=True),
}


def main():
    torch.set_num_threads(4)
    import_reference()
    importlib.import_module("hat.archs.hatx_arch")
    HATX = sys.modules["basicsr.utils.registry"].ARCH_REGISTRY.get("HATX")
    with open(f"{HERE}/meta.json") as f:
        meta = json.load(f)
    with open(f"{HERE}/state_dict_surface.json") as f:
        surface = json.load(f)
    with torch.no_grad():
        for name, cfg in CFGS_X.items():
            net = HATX(**cfg).eval()
            surface[name] = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
            surface[name + ":nparams"] = sum(p.numel() for p in net.parameters())
            meta["cfgs"][name] = cfg
            if name == "hatx_train_yml":
                continue
            net.load_state_dict(synth.synth_state_dict(net.state_dict(), W_SEED), strict=True)
            shape = (1, 3, 16, 24)
            np.savez(f"{HERE}/whole_{name}.npz", y=net(synth.synth_input(X_SEED, shape)).numpy(), x_shape=np.array(shape))
            if name == "hatx_tiny_focus_x2":
                hw = (16, 24)
                t = synth.normal(X_SEED, "tokens", (1, hw[0] * hw[1], cfg["embed_dim"]))
                grp = net.layers[0].residual_group
                rpi = net.relative_position_index_OCA
                out = {"hw": np.array(hw), "sgfn0": grp.blocks[0].mlp(t, hw).numpy(), "hab0": grp.blocks[0](t, hw).numpy(),
                       "ocab": grp.overlap_attn(t, hw, rpi).numpy()}
                grp.overlap_attn.use_focus_bias = False        # prune by ||k||_2 instead (hatx_arch.py:437-439)
                out["ocab_knorm"] = grp.overlap_attn(t, hw, rpi).numpy()
                grp.overlap_attn.use_focus_bias = True
                np.savez(f"{HERE}/blocks_hatx_tiny_focus_x2.npz", **out)
                # a 48x48 map: its 4x4 interior windows see no padded key, so their top-k has no ties and the reference's
                # output there is well defined (what the GPU kernels are held to)
                hw2 = (48, 48)
                t2 = synth.normal(X_SEED, "tokens48", (1, hw2[0] * hw2[1], cfg["embed_dim"]))
                out2 = {"hw": np.array(hw2), "ocab": grp.overlap_attn(t2, hw2, rpi).numpy()}
                grp.overlap_attn.use_focus_bias = False
                out2["ocab_knorm"] = grp.overlap_attn(t2, hw2, rpi).numpy()
                grp.overlap_attn.use_focus_bias = True
                np.savez_compressed(f"{HERE}/blocks_hatx_tiny_focus_48.npz", **out2)
    with open(f"{HERE}/state_dict_surface.json", "w") as f:
        json.dump(surface, f)
    with open(f"{HERE}/meta.json", "w") as f:
        json.dump(meta, f, indent=1)
    print("HATX goldens written")


if __name__ == "__main__":
    main()
