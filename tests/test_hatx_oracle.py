"""SURVEY §8 f3 — the HATX variant: pin the CPU restatement (oracle/hat_oracle.py: sgfn, hatx_hab, hatx_ocab, hatx_forward,
hatx_state_dict_spec) against vectors produced by the reference's own `hat.archs.hatx_arch.HATX`
(tests/golden/gen_golden_hatx.py, run in the build container with /root/reference imported)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hat_oracle as O
from super_resolution_amd import synth
from helpers import GOLDEN, META, W_SEED, X_SEED, golden, max_abs

TOL = 1e-5


def _cfg_sd(name):
    cfg = O.make_hatx_cfg(**META["cfgs"][name])
    return cfg, synth.synth_state_dict(O.hatx_blank_state_dict(cfg), W_SEED)


@pytest.mark.parametrize("name", ["hatx_tiny_plain_x2", "hatx_tiny_focus_x2", "hatx_train_yml", "hatx_live_x2"])
def test_hatx_state_dict_surface(name):
    """keys, ORDER, shapes, dtypes, parameter count — including the one live training config of the fork
    (options/train/train_HAT_SRx2_ESC_OCAB_from_scratch.yml:48-81: embed_dim 180, 25x25 key windows, focus head)."""
    with open(os.path.join(GOLDEN, "state_dict_surface.json")) as f:
        surf = json.load(f)
    spec = O.hatx_state_dict_spec(O.make_hatx_cfg(**META["cfgs"][name]))
    assert [[k, list(s), str(d)] for k, (s, d) in spec.items()] == surf[name]
    assert sum(int(np.prod(s)) for k, (s, d) in spec.items() if d == torch.float32) == surf[name + ":nparams"]


@pytest.mark.parametrize("name", ["hatx_tiny_plain_x2", "hatx_tiny_focus_x2", "hatx_live_x2"])
def test_hatx_whole_model(name):
    g = golden(f"whole_{name}.npz")
    cfg, sd = _cfg_sd(name)
    y = O.hatx_forward(synth.synth_input(X_SEED, tuple(g["x_shape"])), sd, cfg)
    assert y.shape == g["y"].shape
    assert max_abs(y, g["y"]) <= TOL


def test_hatx_blocks():
    """SGFN, HAB, OCAB with focus bias + top-k by the focus score (ties among the zero-padded keys resolved by the same
    torch.topk the reference runs), and OCAB pruned by ||k||_2."""
    name = "hatx_tiny_focus_x2"
    g = golden("blocks_hatx_tiny_focus_x2.npz")
    cfg, sd = _cfg_sd(name)
    hw = tuple(int(v) for v in g["hw"])
    t = synth.normal(X_SEED, "tokens", (1, hw[0] * hw[1], cfg["embed_dim"]))
    p = "layers.0.residual_group"
    assert max_abs(O.sgfn(t, hw, sd, p + ".blocks.0.mlp"), g["sgfn0"]) <= TOL
    assert max_abs(O.hatx_hab(t, hw, sd, p + ".blocks.0", cfg), g["hab0"]) <= TOL
    rpi = sd["relative_position_index_OCA"]
    assert max_abs(O.hatx_ocab(t, hw, sd, p + ".overlap_attn", rpi, cfg, 2), g["ocab"]) <= TOL
    assert max_abs(O.hatx_ocab(t, hw, sd, p + ".overlap_attn", rpi, dict(cfg, use_focus_bias=False), 2), g["ocab_knorm"]) <= TOL


def test_hatx_defaults_reduce_to_plain_attention():
    """kv_topk_ratio = 1, no focus bias and an even window overlap: the HATX attention core IS the HAT one."""
    q, k, v = (synth.normal(3, n, (1, 16, 24, 24)) for n in "qkv")
    table, rpi = synth.normal(3, "tab", (23 * 23, 2), std=0.5), O.rpi_oca(8, 0.5)
    a = O.ocab_attention(q, k, v, table, rpi, 8, 12, 2, 0.3)
    b = O.hatx_ocab_attention(q, k, v, table, rpi, 8, 12, 2, 0.3)
    assert max_abs(a, b) == 0.0


def test_hatx_interior_windows_do_not_depend_on_the_tie_rule():
    """On the 48x48 block golden the 4x4 interior windows see no padded key: there the reference's output is the same under
    torch.topk's tie order and under "lowest index first" (the GPU kernel's rule); border windows differ."""
    g = golden("blocks_hatx_tiny_focus_48.npz")
    cfg, sd = _cfg_sd("hatx_tiny_focus_x2")
    hw = tuple(int(v) for v in g["hw"])
    t = synth.normal(X_SEED, "tokens48", (1, hw[0] * hw[1], cfg["embed_dim"]))
    p, rpi = "layers.0.residual_group.overlap_attn", sd["relative_position_index_OCA"]
    for key, c in (("ocab", cfg), ("ocab_knorm", dict(cfg, use_focus_bias=False))):
        ref = torch.from_numpy(g[key]).reshape(48, 48, -1)
        assert max_abs(O.hatx_ocab(t, hw, sd, p, rpi, c, 2), g[key]) <= TOL
        low = O.hatx_ocab(t, hw, sd, p, rpi, c, 2, tie="lowest_index").reshape(48, 48, -1)
        assert max_abs(low[8:40, 8:40], ref[8:40, 8:40]) <= TOL


def test_hatx_live_shapes_blocks():
    """The shapes of the fork's live training config on a small model (tests/golden/gen_golden_hatx_live.py): ODD key window
    (13 = 8 + int(0.7 * 8), ceil padding 3), ESC on 24 channels with a 15 x 15 kernel, OCAB-ESC on 32 channels with 17 x 17."""
    g = golden("blocks_hatx_live_48.npz")
    cfg, sd = _cfg_sd("hatx_live_x2")
    hw = tuple(int(v) for v in g["hw"])
    t = synth.normal(X_SEED, "tokens48", (1, hw[0] * hw[1], cfg["embed_dim"]))
    p, rpi = "layers.0.residual_group", sd["relative_position_index_OCA"]
    assert max_abs(O.hatx_hab(t, hw, sd, p + ".blocks.0", cfg), g["hab0"]) <= TOL
    for key, c in (("ocab", cfg), ("ocab_knorm", dict(cfg, use_focus_bias=False))):
        ref = torch.from_numpy(g[key]).reshape(48, 48, -1)
        assert max_abs(O.hatx_ocab(t, hw, sd, p + ".overlap_attn", rpi, c, 2), g[key]) <= TOL
        low = O.hatx_ocab(t, hw, sd, p + ".overlap_attn", rpi, c, 2, tie="lowest_index").reshape(48, 48, -1)
        assert max_abs(low[8:40, 8:40], ref[8:40, 8:40]) <= TOL
