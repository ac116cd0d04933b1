"""CPU-side checks of the PRODUCT package (no kernels run): the `HAT` module's state-dict surface against the one recorded
from the reference itself (tests/golden/state_dict_surface.json), the weights-version bookkeeping that decides when the
packed engine is rebuilt, the dataset GT crop, and bench.py's self-launch of its ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, META, golden
from super_resolution_amd import data as D
from super_resolution_amd.registry import build_network
import super_resolution_amd.archs  # noqa: F401  (registers 'HAT')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = ["tiny_x2", "tiny_ocabesc_x2", "tiny_identity_ape_x2", "HAT-S_x2", "HAT-S_x4", "HAT_x4", "HAT-L_x4"]


@pytest.mark.parametrize("name", VARIANTS)
def test_product_state_dict_equals_reference_surface(name):
    """keys, ORDER, shapes, dtypes and parameter count of `HAT(**cfg).state_dict()` == the reference's (hat_arch.py:610-759);
    the two int64 index buffers equal the reference's values, negative entries included."""
    with open(os.path.join(GOLDEN, "state_dict_surface.json")) as f:
        surf = json.load(f)
    net = build_network(dict(type="HAT", **META["cfgs"][name]))
    sd = net.state_dict()
    got = [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]
    assert got == surf[name]
    assert sum(p.numel() for p in net.parameters()) == surf[name + ":nparams"]
    ws = META["cfgs"][name]["window_size"]
    g = golden(f"rpi_ws{ws}.npz")
    assert torch.equal(sd["relative_position_index_SA"], torch.from_numpy(g["sa"]))
    assert torch.equal(sd["relative_position_index_OCA"], torch.from_numpy(g["oca"]))
    assert int(sd["relative_position_index_OCA"].min()) < 0


@pytest.mark.parametrize("name", ["hatx_tiny_plain_x2", "hatx_tiny_focus_x2", "hatx_train_yml"])
def test_product_hatx_state_dict_equals_reference_surface(name):
    """The HATX drop-in (registered as 'HATX', hatx_arch.py:707): keys, order, shapes, dtypes, parameter count of the
    reference's HATX for two tiny configs and for the fork's one live training config (SGFN, focus head, ESC in the OCAB)."""
    with open(os.path.join(GOLDEN, "state_dict_surface.json")) as f:
        surf = json.load(f)
    net = build_network(dict(type="HATX", **META["cfgs"][name]))
    assert [[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()] == surf[name]
    assert sum(p.numel() for p in net.parameters()) == surf[name + ":nparams"]


def test_unknown_resi_connection_raises_at_build_time():
    with pytest.raises(ValueError):
        build_network(dict(type="HAT", **dict(META["cfgs"]["tiny_x2"], resi_connection="3conv")))


def test_weights_version_bumps_on_bulk_parameter_changes():
    net = build_network(dict(type="HAT", **META["cfgs"]["tiny_x2"])).eval()
    k0 = net._weights_key("cuda:0")
    assert net._weights_key("cuda:0") == k0                 # O(1): no parameter walk
    net.load_state_dict(net.state_dict())
    k1 = net._weights_key("cuda:0")
    assert k1 != k0
    net.double().float()                                    # any _apply (to / cuda / float / ...)
    k2 = net._weights_key("cuda:0")
    assert k2 != k1
    net.mark_weights_changed()
    assert net._weights_key("cuda:0") != k2
    net.set_compute_dtype("f32")
    assert net._weights_key("cuda:0")[1] == "f32"


def test_weights_version_sees_wrapper_loads_and_in_place_edits():
    """The two cases the round-2 key missed: load_state_dict called on a PARENT (torch recurses through
    _load_from_state_dict and never calls the child's load_state_dict: nn.DataParallel, nn.Sequential, user containers),
    and in-place parameter edits that autograd's version counter sees (optimizer steps, p.copy_ / p.mul_ under no_grad)."""
    net = build_network(dict(type="HAT", **META["cfgs"]["tiny_x2"])).eval()
    k = net._weights_key("cuda:0")
    for wrap in (torch.nn.Sequential(net), torch.nn.DataParallel(net), torch.nn.ModuleDict({"g": net})):
        wrap.load_state_dict(wrap.state_dict())
        k, prev = net._weights_key("cuda:0"), k
        assert k != prev, type(wrap).__name__
    with torch.no_grad():
        net.layers[0].residual_group.blocks[0].mlp.fc1.weight.mul_(1.5)
    k, prev = net._weights_key("cuda:0"), k
    assert k != prev
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    net.conv_first.weight.grad = torch.ones_like(net.conv_first.weight)
    opt.step()
    assert net._weights_key("cuda:0") != k
    assert net._weights_key("cuda:0") == net._weights_key("cuda:0")


def test_paired_dataset_crops_gt_to_lq_times_scale(tmp_path):
    """paired_image_dataset.py:92-95: in the test phase GT is cut to lq.shape * scale (benchmark GTs are often a few
    pixels larger), instead of failing the shape check of the metrics."""
    rng = np.random.default_rng(1)
    D.write_image(rng.integers(0, 256, (10, 12, 3), dtype=np.uint8), str(tmp_path / "lq" / "a.png"))
    gt = rng.integers(0, 256, (31, 38, 3), dtype=np.uint8)
    D.write_image(gt, str(tmp_path / "gt" / "a.png"))
    opt = {"name": "toy", "type": "PairedImageDataset", "dataroot_lq": str(tmp_path / "lq"), "dataroot_gt": str(tmp_path / "gt"),
           "scale": 3, "phase": "test"}
    item = D.FolderDataset(opt)[0]
    assert item["gt"].shape == (1, 3, 30, 36)
    assert np.array_equal((item["gt"][0].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8), gt[:30, :36])


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment must start two ranks itself (before touching the
    GPU) and return their exit code.  Without a GPU each rank stops at bench.py's own "needs an MI355X" check — which is
    exactly what shows that two children were started with WORLD_SIZE=2."""
    if torch.cuda.is_available():
        pytest.skip("CPU-side test of the launcher (on a GPU box the children would run the real benchmark)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-2000:]
    assert "WORLD_SIZE=1" not in r.stderr


def test_pack_cab_w2f_is_the_expand_weight_in_fragment_order():
    """ops.pack_cab_w2f (HatCabFoldDesc.w2f, include/hat_mi355x.h): element (t, ks, lane, j) of the fp32 image is
    W2[16 t + lane % 16][ci = j][tap = 4 ks + lane // 16] (hat_arch.py:86's 3x3 expand conv) and zero where the output
    channel, the tap or the input channel does not exist — the order of hat_cab_fold's output, so that the kernel reads it
    with unit stride.  Host-side packing only: runs without a GPU."""
    from super_resolution_amd import ops
    g = torch.Generator().manual_seed(7)
    for C_, mid in ((144, 6), (160, 8), (136, 3)):
        w2 = torch.randn(C_, mid, 3, 3, generator=g)
        f = ops.pack_cab_w2f(w2, "cpu")
        nt = -(-C_ // 16)
        assert f.shape == (nt, 3, 64, 8) and f.dtype == torch.float32
        for (t, ks, lane, j) in [(0, 0, 0, 0), (nt - 1, 2, 63, 7), (3, 1, 17, 2), (nt - 1, 0, 15, mid - 1), (1, 2, 16, 0), (2, 2, 5, 1)]:
            co, tap = 16 * t + lane % 16, 4 * ks + lane // 16
            want = float(w2[co, j, tap // 3, tap % 3]) if (co < C_ and tap < 9 and j < mid) else 0.0
            assert float(f[t, ks, lane, j]) == want, (C_, mid, t, ks, lane, j)
        assert float(f.abs().sum()) == pytest.approx(float(w2.abs().sum()), rel=1e-6)
