"""SURVEY §8 f1 ("next" row): the HATModel test harness — YAML options, PNG I/O, pad / tile / crop, PSNR-Y / SSIM-Y —
on the CPU: metric arithmetic against the oracle's independent restatement and hand-computed answers (the reference
holds no fixtures for its metrics and cannot be imported without OpenCV: "parity unpinned"), the folder
dataset round trip, and the pad/tile/crop logic driven with the oracle as the network."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import hat_oracle as O
from super_resolution_amd import data as D, metrics as M, synth
from super_resolution_amd.models import HATModel

TINY = dict(type="HAT", upscale=2, in_chans=3, img_size=32, window_size=16, compress_ratio=4, squeeze_factor=4,
            conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[1], embed_dim=24, num_heads=[2], mlp_ratio=2,
            upsampler="pixelshuffle", resi_connection="1conv")


def test_metric_known_answers():
    a = np.full((32, 40, 3), 100, np.uint8)
    assert M.calculate_psnr(a, a, 4, True) == float("inf") and M.calculate_ssim(a, a, 4, True) == pytest.approx(1.0, abs=1e-12)
    b = a.copy()
    b[10, 12] = (110, 110, 110)  # one grey pixel off by 10: Y differs by 10 * (65.481 + 128.553 + 24.966) / 255
    dy = 10 * (65.481 + 128.553 + 24.966) / 255.0
    mse = dy * dy / (24 * 32)
    assert M.calculate_psnr(a, b, 4, True) == pytest.approx(10 * math.log10(255 * 255 / mse), rel=1e-5)
    mse_rgb = 3 * 100.0 / (24 * 32 * 3)
    assert M.calculate_psnr(a, b, 4, False) == pytest.approx(10 * math.log10(255 * 255 / mse_rgb), rel=1e-12)
    t = torch.tensor([[[-0.2, 0.0], [0.5, 1.3]]]).repeat(3, 1, 1)
    assert M.tensor2img(t)[..., 0].tolist() == [[0, 0], [128, 255]]  # clamp, round-half-even of 127.5 -> 128


def test_metrics_match_oracle_restatement():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (48, 57, 3), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-12, 13, a.shape), 0, 255).astype(np.uint8)
    for cb in (0, 2, 4):
        assert M.calculate_psnr(a, b, cb, True) == pytest.approx(O.psnr_y(a, b, cb), rel=1e-12)
        assert M.calculate_ssim(a, b, cb, True) == pytest.approx(O.ssim_y(a, b, cb), rel=1e-9)
    t = torch.rand(1, 3, 9, 7) * 1.5 - 0.25
    assert np.array_equal(M.tensor2img(t), O.tensor2img_rgb(t))
    assert M.calculate_metric({"img": a, "img2": b}, {"type": "calculate_ssim", "crop_border": 2, "test_y_channel": True}) \
        == pytest.approx(O.ssim_y(a, b, 2), rel=1e-9)


def test_folder_dataset_roundtrip(tmp_path):
    rng = np.random.default_rng(5)
    for sub, shape in (("lq", (20, 24, 3)), ("gt", (40, 48, 3))):
        for name in ("b", "a"):
            D.write_image(rng.integers(0, 256, shape, dtype=np.uint8), str(tmp_path / sub / f"{name}.png"))
    ds = D.FolderDataset({"name": "toy", "type": "PairedImageDataset", "dataroot_lq": str(tmp_path / "lq"), "dataroot_gt": str(tmp_path / "gt")})
    assert len(ds) == 2
    item = ds[0]
    assert os.path.basename(item["lq_path"][0]) == "a.png" and os.path.basename(item["gt_path"][0]) == "a.png"
    assert item["lq"].shape == (1, 3, 20, 24) and item["gt"].shape == (1, 3, 40, 48) and item["lq"].dtype == torch.float32
    assert np.array_equal(M.tensor2img(item["lq"]), np.asarray(__import__("PIL.Image").Image.open(item["lq_path"][0])))
    (tmp_path / "gt" / "a.png").unlink()
    with pytest.raises(AssertionError):
        D.FolderDataset({"name": "toy", "type": "PairedImageDataset", "dataroot_lq": str(tmp_path / "lq"), "dataroot_gt": str(tmp_path / "gt")})


@pytest.mark.parametrize("tile", [None, {"tile_size": 32, "tile_pad": 16}])
def test_hatmodel_pad_tile_crop_with_oracle_net(tmp_path, tile):
    """HATModel's pre_process / (tile_)process / post_process + metrics + PNG output, with the oracle standing in for
    the network (the product network needs an MI355X), against the oracle's own harness restatement."""
    opt = {"name": "toy", "scale": 2, "network_g": dict(TINY), "path": {"visualization": str(tmp_path / "vis")},
           "val": {"save_img": True, "suffix": None, "metrics": {
               "psnr": {"type": "calculate_psnr", "crop_border": 2, "test_y_channel": True},
               "ssim": {"type": "calculate_ssim", "crop_border": 2, "test_y_channel": True}}}}
    if tile:
        opt["tile"] = tile
    model = HATModel(opt, device="cpu")
    cfg = O.make_cfg(**{k: v for k, v in TINY.items() if k != "type"})
    sd = synth.synth_state_dict(O.blank_state_dict(cfg), 11)
    net = lambda x: O.hat_forward(x, sd, cfg)
    model.net_g = net
    lq = synth.synth_input(3, (1, 3, 45, 38))
    gt = synth.synth_input(4, (1, 3, 90, 76))
    D.write_image(M.tensor2img(lq), str(tmp_path / "lq" / "im.png"))
    D.write_image(M.tensor2img(gt), str(tmp_path / "gt" / "im.png"))
    ds = D.FolderDataset({"name": "toy", "type": "PairedImageDataset", "dataroot_lq": str(tmp_path / "lq"), "dataroot_gt": str(tmp_path / "gt")})
    mean, rows = model.nondist_validation(ds, save_img=True)
    lq8 = D.read_image(str(tmp_path / "lq" / "im.png")).unsqueeze(0)
    img, ph, pw = O.pre_process(lq8, 16)
    ref = O.tile_process(img, net, 2, tile["tile_size"], tile["tile_pad"]) if tile else net(img)
    ref = O.post_process(ref, ph, pw, 2)
    ref8 = O.tensor2img_rgb(ref)
    saved = np.asarray(__import__("PIL.Image").Image.open(str(tmp_path / "vis" / "toy" / "im_toy.png")).convert("RGB"))
    assert saved.shape == (90, 76, 3) and np.array_equal(saved, ref8)
    gt8 = M.tensor2img(D.read_image(str(tmp_path / "gt" / "im.png")))
    assert mean["psnr"] == pytest.approx(O.psnr_y(ref8, gt8, 2), rel=1e-12) and rows[0]["name"] == "im"
    assert mean["ssim"] == pytest.approx(O.ssim_y(ref8, gt8, 2), rel=1e-9)


def test_cli_option_parsing(tmp_path):
    from super_resolution_amd import test as T
    yml = tmp_path / "o.yml"
    yml.write_text("name: t\nscale: 2\ndatasets:\n  test_1:\n    name: A\n    type: SingleImageDataset\n    dataroot_lq: x\nnetwork_g:\n  type: HAT\n  window_size: 16\nval:\n  save_img: false\n")
    opt = T.parse_options(str(yml))
    assert opt["is_train"] is False and opt["datasets"]["test_1"]["phase"] == "test" and opt["datasets"]["test_1"]["scale"] == 2


def test_window_attention_module_surface_and_no_cpu_path():
    """Row f2 host side: same parameter / buffer names and shapes as swinir_arch.WindowAttention (the golden generator
    loads these very keys into the reference module with strict=True), bit-exact index buffer, and no CPU fallback."""
    import numpy as np
    from helpers import golden, wmsa_sd
    from super_resolution_amd.archs.window_msa import WindowAttention, relative_position_index
    m = WindowAttention(48, (16, 16), 2).eval()
    sd = wmsa_sd(48, 2, 16)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    m.load_state_dict(sd, strict=True)
    assert torch.equal(relative_position_index(16), torch.from_numpy(np.asarray(golden("rpi_ws16.npz")["sa"])))
    with pytest.raises(RuntimeError):
        m.forward_map(torch.zeros(1, 16, 16, 48))
