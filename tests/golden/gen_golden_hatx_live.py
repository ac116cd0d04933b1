#!/usr/bin/env python3
"""Golden vectors for HATX at the SHAPES of the fork's live training config (options/train/
train_HAT_SRx2_ESC_OCAB_from_scratch.yml:48-81) on a small model, made like gen_golden_hatx.py by running THE REFERENCE
ITSELF (build container only).  What the live config has that the other HATX fixtures do not: an ODD key window
(window 16, overlap 0.6 -> 25; here window 8, overlap 0.7 -> 13: ceil padding 3), ESC on 24 channels with a 15 x 15
kernel in the HABs, OCAB-ESC on 32 channels with a 17 x 17 kernel, hab_ffn_ratio 3 / mlp_ratio 4, conv_scale 0.03.

  whole_hatx_live_x2.npz     whole model on a 16 x 24 frame
  blocks_hatx_live_48.npz    first group on a 48 x 48 token map: HAB 0, OCAB (focus bias + top-k) and OCAB with ||k||_2
                             pruning; the interior windows see no padded key (no top-k ties)

    python tests/golden/gen_golden_hatx_live.py
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

LIVE = dict(upscale=2, in_chans=3, img_size=16, window_size=8, compress_ratio=3, squeeze_factor=30, conv_scale=0.03,
            overlap_ratio=0.7, img_range=1.0, depths=[2, 1], embed_dim=48, num_heads=[2, 2], hab_ffn_ratio=3.0,
            mlp_ratio=4.0, upsampler="pixelshuffle", resi_connection="1conv", esc_pdim=24, esc_kernel=15,
            ocab_esc_enable=True, ocab_esc_pdim=32, ocab_esc_kernel=17, kv_topk_ratio=0.6, use_focus_bias=True)


def main():
    torch.set_num_threads(4)
    import_reference()
    importlib.import_module("hat.archs.hatx_arch")
    HATX = sys.modules["basicsr.utils.registry"].ARCH_REGISTRY.get("HATX")
    with open(f"{HERE}/meta.json") as f:
        meta = json.load(f)
    with open(f"{HERE}/state_dict_surface.json") as f:
        surface = json.load(f)
    name = "hatx_live_x2"
    with torch.no_grad():
        net = HATX(**LIVE).eval()
        surface[name] = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
        surface[name + ":nparams"] = sum(p.numel() for p in net.parameters())
        meta["cfgs"][name] = LIVE
        net.load_state_dict(synth.synth_state_dict(net.state_dict(), W_SEED), strict=True)
        shape = (1, 3, 16, 24)
        np.savez(f"{HERE}/whole_{name}.npz", y=net(synth.synth_input(X_SEED, shape)).numpy(), x_shape=np.array(shape))
        hw = (48, 48)
        t = synth.normal(X_SEED, "tokens48", (1, hw[0] * hw[1], LIVE["embed_dim"]))
        grp = net.layers[0].residual_group
        rpi = net.relative_position_index_OCA
        out = {"hw": np.array(hw), "hab0": grp.blocks[0](t, hw).numpy(), "ocab": grp.overlap_attn(t, hw, rpi).numpy()}
        grp.overlap_attn.use_focus_bias = False        # prune by ||k||_2 instead (hatx_arch.py:437-439)
        out["ocab_knorm"] = grp.overlap_attn(t, hw, rpi).numpy()
        grp.overlap_attn.use_focus_bias = True
        np.savez_compressed(f"{HERE}/blocks_hatx_live_48.npz", **out)
    with open(f"{HERE}/state_dict_surface.json", "w") as f:
        json.dump(surface, f)
    with open(f"{HERE}/meta.json", "w") as f:
        json.dump(meta, f, indent=1)
    print("HATX live-shape goldens written")


if __name__ == "__main__":
    main()
