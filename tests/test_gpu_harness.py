"""SURVEY §8 f1 on the GPU: `python -m super_resolution_amd.test -opt x.yml` (YAML options, PNG folders, checkpoint
file, tile mode, PSNR/SSIM) through the MI355X path, against the oracle driven through its own harness restatement."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import hat_oracle as O
from super_resolution_amd import data as D, metrics as M, synth

pytestmark = pytest.mark.gpu

NET = dict(type="HAT", upscale=2, in_chans=3, img_size=32, window_size=16, compress_ratio=4, squeeze_factor=4,
           conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[2], embed_dim=24, num_heads=[2], mlp_ratio=2,
           upsampler="pixelshuffle", resi_connection="1conv")


@pytest.mark.parametrize("tile", [None, {"tile_size": 32, "tile_pad": 16}], ids=["whole", "tiled"])
def test_cli_end_to_end(tmp_path, tile):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from super_resolution_amd import test as T
    cfg = O.make_cfg(**{k: v for k, v in NET.items() if k != "type"})
    sd = synth.synth_state_dict(O.blank_state_dict(cfg), 21)
    ckpt = tmp_path / "net.pth"
    torch.save({"params_ema": {"module." + k: v for k, v in sd.items()}}, ckpt)  # EMA key + DataParallel prefix (base_model.py:302-315)
    for i, (h, w) in enumerate([(45, 38), (32, 64)]):
        D.write_image(M.tensor2img(synth.synth_input(30 + i, (1, 3, h, w))), str(tmp_path / "lq" / f"im{i}.png"))
        D.write_image(M.tensor2img(synth.synth_input(40 + i, (1, 3, 2 * h, 2 * w))), str(tmp_path / "gt" / f"im{i}.png"))
    opt = {"name": "toy", "model_type": "HATModel", "scale": 2, "num_gpu": 1,
           "datasets": {"test_1": {"name": "Toy", "type": "PairedImageDataset", "dataroot_gt": str(tmp_path / "gt"),
                                   "dataroot_lq": str(tmp_path / "lq"), "io_backend": {"type": "disk"}}},
           "network_g": dict(NET, compute_dtype="f32"),
           "path": {"pretrain_network_g": str(ckpt), "strict_load_g": True, "param_key_g": "params_ema",
                    "visualization": str(tmp_path / "vis")},
           "val": {"save_img": True, "suffix": None, "metrics": {
               "psnr": {"type": "calculate_psnr", "crop_border": 2, "test_y_channel": True},
               "ssim": {"type": "calculate_ssim", "crop_border": 2, "test_y_channel": True}}}}
    if tile:
        opt["tile"] = tile
    yml = tmp_path / "opt.yml"
    yml.write_text(yaml.safe_dump(opt))
    res = T.main(["-opt", str(yml)])
    net = lambda x: O.hat_forward(x, sd, cfg)
    psnrs = []
    for i in range(2):
        lq8 = D.read_image(str(tmp_path / "lq" / f"im{i}.png")).unsqueeze(0)
        img, ph, pw = O.pre_process(lq8, 16)
        ref = O.tile_process(img, net, 2, tile["tile_size"], tile["tile_pad"]) if tile else net(img)
        ref8 = O.tensor2img_rgb(O.post_process(ref, ph, pw, 2))
        from PIL import Image
        saved = np.asarray(Image.open(str(tmp_path / "vis" / "Toy" / f"im{i}_toy.png")).convert("RGB"))
        assert saved.shape == ref8.shape
        diff = np.abs(saved.astype(int) - ref8.astype(int))
        assert diff.max() <= 1 and (diff != 0).mean() < 1e-3, "fp32 path: at most isolated +-1 rounding flips"
        gt8 = M.tensor2img(D.read_image(str(tmp_path / "gt" / f"im{i}.png")))
        psnrs.append(O.psnr_y(ref8, gt8, 2))
        assert res["Toy"]["images"][i]["psnr"] == pytest.approx(psnrs[-1], abs=1e-3)  # north_star: within 1e-3 dB
    assert res["Toy"]["mean"]["psnr"] == pytest.approx(float(np.mean(psnrs)), abs=1e-3)


def test_tiled_720p_allocates_one_workspace_per_distinct_tile_shape():
    """VERDICT r1 (host side): `HATModel.tile_process` with the reference grid on a 720x1280 frame at tile_size 256 /
    tile_pad 32 — 15 tiles of 6 distinct padded shapes (interior, edges, corners, the 208-row last band) — must allocate the
    engine workspace once per distinct shape (LRU), not once per tile, and a second frame must allocate nothing."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from super_resolution_amd.models import HATModel
    from super_resolution_amd import tile_parallel as tp
    net_opt = dict(type="HAT", upscale=2, in_chans=3, img_size=64, window_size=16, compress_ratio=24, squeeze_factor=24,
                   conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[1], embed_dim=144, num_heads=[6], mlp_ratio=2,
                   upsampler="pixelshuffle", resi_connection="1conv", compute_dtype="bf16")
    model = HATModel({"name": "t", "scale": 2, "network_g": net_opt, "path": {}, "tile": {"tile_size": 256, "tile_pad": 32}}, device="cuda:0")
    shapes = {(t.py1 - t.py0, t.px1 - t.px0) for t in tp.reference_tiles(720, 1280, 256, 32)}
    for k in range(2):
        model.feed_data({"lq": synth.synth_input(50 + k, (1, 3, 720, 1280))})
        model.test()
        torch.cuda.synchronize()
        assert model.output.shape == (1, 3, 1440, 2560) and torch.isfinite(model.output).all()
        assert model.net_g.engine().ws_allocations == len(shapes), (model.net_g.engine().ws_allocations, len(shapes))


def test_dataparallel_wrapper_on_one_gpu():
    """base_model.py:99-100 wraps the network in nn.DataParallel when num_gpu > 1; with one visible device that is a plain
    call-through and must give the same result as the bare module."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from super_resolution_amd.registry import build_network
    net = build_network(dict(NET, compute_dtype="f32")).eval().to("cuda:0")
    x = synth.synth_input(3, (1, 3, 32, 48)).to("cuda:0")
    y = net(x)
    yd = torch.nn.DataParallel(net, device_ids=[0])(x)
    torch.cuda.synchronize()
    assert torch.equal(y, yd)


def test_hatmodel_num_gpu_2_runs_through_dataparallel(tmp_path):
    """`num_gpu: 2` in the options: HATModel wraps the network in nn.DataParallel exactly like basicsr's model_to_device
    (base_model.py:91-104) and the whole test() — reflect pad, forward THROUGH the wrapper, crop — gives the bare module's
    result; loading a state dict through the wrapper afterwards re-packs the engine (the weights key sees loads made on a
    parent module)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from super_resolution_amd.models import HATModel
    cfg = O.make_cfg(**{k: v for k, v in NET.items() if k != "type"})
    sd = synth.synth_state_dict(O.blank_state_dict(cfg), 21)
    ckpt = tmp_path / "net.pth"
    torch.save({"params": sd}, ckpt)
    base = {"name": "t", "scale": 2, "network_g": dict(NET, compute_dtype="f32"), "path": {"pretrain_network_g": str(ckpt)}}
    m1 = HATModel(dict(base, num_gpu=1), device="cuda:0")
    m2 = HATModel(dict(base, num_gpu=2), device="cuda:0")
    assert isinstance(m2.net_g, torch.nn.DataParallel) and not isinstance(m1.net_g, torch.nn.DataParallel)
    lq = synth.synth_input(31, (1, 3, 45, 38))
    outs = []
    for m in (m1, m2):
        m.feed_data({"lq": lq})
        m.test()
        torch.cuda.synchronize()
        outs.append(m.output.clone())
    assert outs[0].shape == (1, 3, 90, 76) and torch.equal(outs[0], outs[1])
    # new weights loaded THROUGH the wrapper (keys prefixed "module."): the packed engine must follow
    sd2 = synth.synth_state_dict(O.blank_state_dict(cfg), 22)
    m2.net_g.load_state_dict({"module." + k: v for k, v in sd2.items()})
    m2.feed_data({"lq": lq})
    m2.test()
    torch.cuda.synchronize()
    ref = O.hat_forward(O.pre_process(lq, 16)[0], sd2, cfg)[:, :, :90, :76]
    assert float((m2.output.cpu() - ref).abs().max()) <= 1e-4
