#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; the GPU box has no reference).
The reference network `hat.archs.hat_arch.HAT` is imported with the loader shim described in
SURVEY.md §8(c): the vendored `basicsr` package cannot be imported whole (needs cv2 /
torchvision), so only `basicsr/utils/registry.py` is loaded by path and the two helpers the arch
file imports from `basicsr.archs.arch_util` (`to_2tuple`, `trunc_normal_`) are provided by
equivalents; they affect construction/initialisation only, never the forward arithmetic, and all
parameters are overwritten with the portable synthetic weights of `super_resolution_amd.synth`.

What is stored is DATA ONLY (inputs are regenerated from seeds; outputs are stored): nothing of
the reference's source text is copied.

    python tests/golden/gen_golden.py [--big]     # --big adds cfg2/cfg3 (minutes of CPU time)
"""
from __future__ import annotations

import argparse
import collections.abc
import importlib
import importlib.util
import json
import os
import sys
import types
from itertools import repeat

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/HAT"
sys.path.insert(0, ROOT)

from super_resolution_amd import synth  # noqa: E402
from oracle import hat_oracle as O  # noqa: E402  (the tile loop / pad harness restatement)

W_SEED, X_SEED = 1234, 7


def import_reference():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("basicsr.utils.registry", f"{REF}/ESC/basicsr/utils/registry.py")
    reg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(reg)
    for name in ("basicsr", "basicsr.utils", "basicsr.archs"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["basicsr.utils.registry"] = reg
    au = types.ModuleType("basicsr.archs.arch_util")

    def to_2tuple(x):
        return x if isinstance(x, collections.abc.Iterable) else tuple(repeat(x, 2))

    au.to_2tuple = to_2tuple
    au.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules["basicsr.archs.arch_util"] = au
    for name, path in (("hat", REF + "/hat"), ("hat.archs", REF + "/hat/archs")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    importlib.import_module("hat.archs.hat_arch")
    return reg.ARCH_REGISTRY.get("HAT")


CFGS = {
    # tiny whole-model configs (verified to run on the reference)
    "tiny_x2": dict(upscale=2, in_chans=3, img_size=16, window_size=8, compress_ratio=3, squeeze_factor=30,
                    conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[2, 2], embed_dim=24,
                    num_heads=[2, 2], mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv",
                    esc_pdim=8, esc_kernel=5),
    "tiny_x4": dict(upscale=4, in_chans=3, img_size=16, window_size=8, compress_ratio=3, squeeze_factor=30,
                    conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[2, 2], embed_dim=24,
                    num_heads=[2, 2], mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv",
                    esc_pdim=8, esc_kernel=5),
    "tiny_x3": dict(upscale=3, in_chans=3, img_size=16, window_size=8, compress_ratio=3, squeeze_factor=30,
                    conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[1], embed_dim=24,
                    num_heads=[2], mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv",
                    esc_pdim=8, esc_kernel=5),
    "tiny_ocabesc_x2": dict(upscale=2, in_chans=3, img_size=16, window_size=8, compress_ratio=3, squeeze_factor=30,
                            conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[1, 1], embed_dim=24,
                            num_heads=[2, 2], mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv",
                            esc_pdim=8, esc_kernel=5, ocab_esc_enable=True, ocab_esc_pdim=8, ocab_esc_kernel=5),
    # one-RHAG slices of the shipped variants (options/test/*.yml:49-65) for per-block goldens
    "hats_1g_x4": dict(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=24, squeeze_factor=24,
                       conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[2], embed_dim=144, num_heads=[6],
                       mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv"),
    "hat_1g_x2": dict(upscale=2, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30,
                      conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[1], embed_dim=180, num_heads=[6],
                      mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv"),
    # full shipped variants
    "HAT-S_x2": dict(upscale=2, in_chans=3, img_size=64, window_size=16, compress_ratio=24, squeeze_factor=24,
                     conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[6] * 6, embed_dim=144,
                     num_heads=[6] * 6, mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv"),
    "HAT-S_x4": dict(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=24, squeeze_factor=24,
                     conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[6] * 6, embed_dim=144,
                     num_heads=[6] * 6, mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv"),
    "HAT_x4": dict(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30,
                   conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[6] * 6, embed_dim=180,
                   num_heads=[6] * 6, mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv"),
    "HAT-L_x4": dict(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30,
                     conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[6] * 12, embed_dim=180,
                     num_heads=[6] * 12, mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv"),
}


def build(HAT, cfg):
    net = HAT(**cfg).eval()
    sd = synth.synth_state_dict(net.state_dict(), W_SEED)
    net.load_state_dict(sd, strict=True)
    return net, sd


def summary(y: torch.Tensor, crop=64):
    """Checksums + crops of a big output (cfg2/cfg3 are too large to commit whole)."""
    h, w = y.shape[-2:]
    c = min(crop, h, w)
    pos = {"tl": (0, 0), "br": (h - c, w - c), "ce": ((h - c) // 2, (w - c) // 2)}
    out = {"mean": np.float64(y.double().mean()), "std": np.float64(y.double().std()),
           "abs_sum": np.float64(y.double().abs().sum())}
    for k, (a, b) in pos.items():
        out["crop_" + k] = y[..., a:a + c, b:b + c].numpy().copy()
        out["pos_" + k] = np.array([a, b, c])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also cfg2 (HAT-S x4 256^2) and cfg3 (HAT-L x4 512^2 crop stats)")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    HAT = import_reference()
    meta = {"w_seed": W_SEED, "x_seed": X_SEED, "torch": torch.__version__, "cfgs": CFGS}

    # --- state-dict surface of the shipped variants (keys, shapes, dtypes) + int buffers -----
    surface = {}
    for name in ("tiny_x2", "tiny_ocabesc_x2", "HAT-S_x2", "HAT-S_x4", "HAT_x4", "HAT-L_x4"):
        net = HAT(**CFGS[name])
        surface[name] = [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()]
        surface[name + ":nparams"] = sum(p.numel() for p in net.parameters())
        if name == "HAT-S_x4":
            np.savez_compressed(f"{HERE}/rpi_ws16.npz",
                                sa=net.relative_position_index_SA.numpy(), oca=net.relative_position_index_OCA.numpy())
        if name == "tiny_x2":
            np.savez_compressed(f"{HERE}/rpi_ws8.npz",
                                sa=net.relative_position_index_SA.numpy(), oca=net.relative_position_index_OCA.numpy())
    with open(f"{HERE}/state_dict_surface.json", "w") as f:
        json.dump(surface, f)

    with torch.no_grad():
        # --- (i) tiny whole-model in/out -----------------------------------------------------
        for name, shape in (("tiny_x2", (1, 3, 16, 24)), ("tiny_x4", (1, 3, 24, 16)), ("tiny_x3", (1, 3, 16, 16)),
                            ("tiny_ocabesc_x2", (1, 3, 16, 24))):
            net, _ = build(HAT, CFGS[name])
            x = synth.synth_input(X_SEED, shape)
            np.savez(f"{HERE}/whole_{name}.npz", y=net(x).numpy(), x_shape=np.array(shape))

        # --- (ii) per-block in/out at C=144 / C=180 (HAB, OCAB, RHAG conv) ------------------
        for name, hw in (("hats_1g_x4", (16, 32)), ("hat_1g_x2", (32, 16))):
            net, _ = build(HAT, CFGS[name])
            C = CFGS[name]["embed_dim"]
            t = synth.normal(X_SEED, "tokens", (1, hw[0] * hw[1], C))
            grp = net.layers[0].residual_group
            rpi = net.relative_position_index_OCA
            out = {"hw": np.array(hw)}
            blk = grp.blocks[0]
            out["hab0"] = blk(t, hw).numpy()
            n = blk.norm1(t)
            n_img = n.view(1, hw[0], hw[1], C).permute(0, 3, 1, 2)
            out["cab0"] = blk.conv_block(n_img).numpy()
            out["esc0"] = blk.esc_attn(n, hw).numpy()
            out["ffn0"] = blk.mlp(t, hw).numpy()
            out["ocab"] = grp.overlap_attn(t, hw, rpi).numpy()
            out["rhag"] = net.layers[0](t, hw, {"rpi_oca": rpi}).numpy()
            np.savez(f"{HERE}/blocks_{name}.npz", **out)
            # whole 1-group net on a small frame
            shape = (1, 3, hw[0], hw[1])
            np.savez(f"{HERE}/whole_{name}.npz", y=net(synth.synth_input(X_SEED, shape)).numpy(), x_shape=np.array(shape))

        # --- (iv) tile loop (hat_model.py:40-108 restated) driven through the reference net --
        for name in ("tiny_x2", "tiny_x4"):
            net, _ = build(HAT, CFGS[name])
            s = CFGS[name]["upscale"]
            x = synth.synth_input(X_SEED, (1, 3, 90, 77))  # not a window multiple: exercises reflect pad + crop
            img, ph, pw = O.pre_process(x, CFGS[name]["window_size"])
            y = O.post_process(O.tile_process(img, net, s, 32, 16), ph, pw, s)
            np.savez(f"{HERE}/tiled_{name}.npz", y=y.numpy(), x_shape=np.array(x.shape), tile=np.array([32, 16]))

        # --- (iii) BASELINE cfg1: HAT-S x2, 3x64x64, whole output ---------------------------
        net, _ = build(HAT, CFGS["HAT-S_x2"])
        shape = (1, 3, 64, 64)
        np.savez(f"{HERE}/whole_HAT-S_x2_64.npz", y=net(synth.synth_input(X_SEED, shape)).numpy(), x_shape=np.array(shape))

        if args.big:
            net, _ = build(HAT, CFGS["HAT-S_x4"])
            shape = (1, 3, 256, 256)
            y = net(synth.synth_input(X_SEED, shape))
            np.savez(f"{HERE}/summary_HAT-S_x4_256.npz", x_shape=np.array(shape), **summary(y))
            shape = (1, 3, 64, 64)
            y = net(synth.synth_input(X_SEED, shape))
            np.savez(f"{HERE}/summary_HAT-S_x4_64.npz", x_shape=np.array(shape), **summary(y, 128))
            net, _ = build(HAT, CFGS["HAT-L_x4"])
            shape = (1, 3, 128, 128)
            y = net(synth.synth_input(X_SEED, shape))
            np.savez(f"{HERE}/summary_HAT-L_x4_128.npz", x_shape=np.array(shape), **summary(y, 96))

    with open(f"{HERE}/meta.json", "w") as f:
        json.dump(meta, f, indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
