"""Folder datasets of the reference's test YAMLs (`PairedImageDataset`, `SingleImageDataset`:
HAT/ESC/basicsr/data/paired_image_dataset.py, single_image_dataset.py) read with PIL instead of OpenCV:
images become float32 RGB CHW tensors in [0, 1] (img_util.py imfrombytes float32 + img2tensor bgr2rgb)."""
from __future__ import annotations

import os

import numpy as np
import torch

_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")


def read_image(path: str) -> torch.Tensor:
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.float32) / np.float32(255.0)
    return torch.from_numpy(a).permute(2, 0, 1).contiguous()


def write_image(img_u8_rgb: np.ndarray, path: str) -> None:
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(img_u8_rgb, "RGB").save(path)


def _scan(folder: str):
    return sorted(os.path.join(folder, f) for f in os.listdir(folder) if f.lower().endswith(_EXT))


class FolderDataset:
    """opt: {name, type: PairedImageDataset|SingleImageDataset, dataroot_lq, [dataroot_gt], [filename_tmpl]}.
    Paired: every LQ file must have a GT file of the same basename (paired_paths_from_folder semantics)."""

    def __init__(self, opt: dict):
        self.opt = opt
        self.lq = _scan(opt["dataroot_lq"])
        self.gt = None
        if opt.get("type", "SingleImageDataset") == "PairedImageDataset" or opt.get("dataroot_gt"):
            tmpl = opt.get("filename_tmpl", "{}")
            gt = {os.path.splitext(os.path.basename(p))[0]: p for p in _scan(opt["dataroot_gt"])}
            if len(gt) != len(self.lq):
                raise AssertionError(f"{opt['name']}: lq and gt folders have different numbers of images: {len(self.lq)}, {len(gt)}.")
            self.gt = []
            for name, p in sorted(gt.items()):
                want = tmpl.format(name)
                match = [q for q in self.lq if os.path.splitext(os.path.basename(q))[0] == want]
                if not match:
                    raise AssertionError(f"{want} is not in the lq folder of {opt['name']}.")
                self.gt.append((match[0], p))
            self.lq = [m for m, _ in self.gt]
            self.gt = [g for _, g in self.gt]

    def __len__(self):
        return len(self.lq)

    def __getitem__(self, i):
        d = {"lq": read_image(self.lq[i]).unsqueeze(0), "lq_path": [self.lq[i]]}
        if self.gt is not None:
            gt = read_image(self.gt[i]).unsqueeze(0)
            # test phase: GT cropped to lq size x scale (paired_image_dataset.py:92-95) so that a GT whose size is not an
            # exact multiple of the LQ size is still scored, as the reference does
            s = int(self.opt.get("scale", 0) or 0)
            if s > 0 and self.opt.get("phase", "test") != "train":
                gt = gt[..., :d["lq"].shape[-2] * s, :d["lq"].shape[-1] * s]
            d["gt"] = gt
            d["gt_path"] = [self.gt[i]]
        return d

    def __iter__(self):
        return (self[i] for i in range(len(self)))
