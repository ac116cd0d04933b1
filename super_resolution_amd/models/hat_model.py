"""`HATModel` — the reference's test-time caller of the hot path (HAT/hat/models/hat_model.py on top of
basicsr SRModel/BaseModel): pad to a window multiple, run the network whole or tile by tile, crop, convert to uint8,
save, score.  Same option dictionary as the reference's test YAMLs (`network_g`, `path.pretrain_network_g`,
`path.param_key_g`, `path.strict_load_g`, `tile.{tile_size,tile_pad}`, `val.{save_img,suffix,metrics}`).

Deviations (documented in INTEGRATION.md): a tile that fails raises instead of being printed and skipped
(hat_model.py:89-90); images are read/written with PIL; `num_gpu: 0` (CPU) is not supported by the MI355X path.
"""
from __future__ import annotations

import os
from os import path as osp

import torch
from torch.nn import functional as F

from .. import archs  # noqa: F401  (registers 'HAT' in the arch registry, like hat/archs/__init__.py:8-11)
from .. import tile_parallel as tp
from ..data import write_image
from ..metrics import calculate_metric, tensor2img
from ..registry import build_network


class HATModel:
    def __init__(self, opt: dict, device=None):
        self.opt = opt
        self.device = torch.device(device if device is not None else "cuda")
        self.net_g = build_network(dict(opt["network_g"])).eval()
        load_path = (opt.get("path") or {}).get("pretrain_network_g")
        if load_path:
            self.load_network(self.net_g, load_path, (opt["path"].get("strict_load_g", True)),
                              opt["path"].get("param_key_g", "params"))
        self.net_g = self.model_to_device(self.net_g)
        self.scale = opt.get("scale", 1)
        self.metric_results = {}

    def model_to_device(self, net):
        """basicsr BaseModel.model_to_device (base_model.py:91-104): `num_gpu > 1` wraps the network in nn.DataParallel, as the
        reference does (one device: DataParallel forwards straight to the module; several: torch replicates the module per
        call and every replica packs its own engine — it works, but tile_parallel / band_parallel are the multi-GPU paths of
        this build).  `dist: true` (DistributedDataParallel) is a training-time wrapper and not needed for the test pipeline."""
        net = net.to(self.device)
        if self.opt.get("dist"):
            raise NotImplementedError("dist: true wraps the network for training (DistributedDataParallel); the test pipeline "
                                      "of this build runs with dist: false (multi-GPU inference: tile_parallel / band_parallel)")
        if int(self.opt.get("num_gpu", 1) or 1) > 1:
            net = torch.nn.DataParallel(net)
        return net

    def get_bare_model(self, net):  # base_model.py:106-112
        return net.module if isinstance(net, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)) else net

    # basicsr BaseModel.load_network (base_model.py:289-315)
    @staticmethod
    def load_network(net, load_path, strict=True, param_key="params"):
        sd = torch.load(load_path, map_location="cpu")
        if param_key is not None:
            if param_key not in sd and "params" in sd:
                param_key = "params"
            sd = sd[param_key] if param_key in sd else sd
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        net.load_state_dict(sd, strict=strict)

    def feed_data(self, data: dict):
        self.lq = data["lq"].to(self.device)
        if "gt" in data:
            self.gt = data["gt"].to(self.device)

    def pre_process(self):  # hat_model.py:16-26
        window_size = self.opt["network_g"]["window_size"]
        self.scale = self.opt.get("scale", 1)
        _, _, h, w = self.lq.size()
        self.mod_pad_h = (window_size - h % window_size) % window_size
        self.mod_pad_w = (window_size - w % window_size) % window_size
        self.img = F.pad(self.lq, (0, self.mod_pad_w, 0, self.mod_pad_h), "reflect")

    def process(self):  # hat_model.py:28-38
        with torch.no_grad():
            self.output = self.net_g(self.img)

    def tile_process(self):  # hat_model.py:40-108
        _, _, h, w = self.img.shape
        tiles = tp.reference_tiles(h, w, self.opt["tile"]["tile_size"], self.opt["tile"]["tile_pad"])
        with torch.no_grad():
            self.output = tp.tile_forward(self.img, self.net_g, self.scale, tiles)

    def post_process(self):  # hat_model.py:110-112
        _, _, h, w = self.output.size()
        self.output = self.output[:, :, 0:h - self.mod_pad_h * self.scale, 0:w - self.mod_pad_w * self.scale]

    def test(self):
        self.pre_process()
        if "tile" in self.opt:
            self.tile_process()
        else:
            self.process()
        self.post_process()

    def get_current_visuals(self):
        out = {"lq": self.lq.detach().cpu(), "result": self.output.detach().cpu()}
        if hasattr(self, "gt"):
            out["gt"] = self.gt.detach().cpu()
        return out

    def nondist_validation(self, dataset, save_img: bool = True):  # hat_model.py:114-185
        dataset_name = dataset.opt["name"]
        val = self.opt.get("val") or {}
        metrics = val.get("metrics")
        self.metric_results = {m: 0.0 for m in (metrics or {})}
        per_image = []
        n = 0
        for val_data in dataset:
            img_name = osp.splitext(osp.basename(val_data["lq_path"][0]))[0]
            self.feed_data(val_data)
            self.test()
            visuals = self.get_current_visuals()
            sr_img = tensor2img(visuals["result"])
            data = {"img": sr_img}
            if "gt" in visuals:
                data["img2"] = tensor2img(visuals["gt"])
                del self.gt
            del self.lq, self.output
            if save_img:
                suffix = val.get("suffix") or self.opt["name"]
                root = (self.opt.get("path") or {}).get("visualization") or osp.join("results", self.opt["name"], "visualization")
                write_image(sr_img, osp.join(root, dataset_name, f"{img_name}_{suffix}.png"))
            row = {"name": img_name}
            if metrics and "img2" in data:
                for name, mopt in metrics.items():
                    v = calculate_metric(data, mopt)
                    self.metric_results[name] += v
                    row[name] = v
            per_image.append(row)
            n += 1
        for m in self.metric_results:
            self.metric_results[m] /= max(n, 1)
        return dict(self.metric_results), per_image
