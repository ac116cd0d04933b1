"""Image conversion and PSNR / SSIM as the reference's validation loop computes them, without OpenCV.

Follows (paths under /root/reference/HAT/ESC/basicsr): `utils/img_util.py:66-67,87-91` (tensor2img),
`metrics/psnr_ssim.py:11-48` (calculate_psnr), `:86-125,170-198` (calculate_ssim / _ssim),
`metrics/metric_util.py:32-45` + `utils/color_util.py:38-68,110-114` (BT.601 Y channel).

The reference keeps images in OpenCV's BGR order; here arrays stay RGB, so the Y coefficients are applied in RGB
order — the numbers are the same.  SSIM's `cv2.filter2D(...)[5:-5, 5:-5]` is a VALID 11x11 Gaussian correlation
(sigma 1.5, `cv2.getGaussianKernel(11, 1.5)`), done here separably in float64.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def tensor2img(t: torch.Tensor, min_max=(0.0, 1.0)) -> np.ndarray:
    """(1,3,H,W) or (3,H,W) float tensor -> uint8 HWC RGB: clamp, normalise to min_max, x255, round."""
    a = t.detach().float().cpu()
    if a.dim() == 4:
        a = a.squeeze(0)
    a = a.clamp(min_max[0], min_max[1])
    a = (a - min_max[0]) / (min_max[1] - min_max[0])
    return np.round(a.permute(1, 2, 0).numpy() * 255.0).astype(np.uint8)


def to_y_channel(img: np.ndarray) -> np.ndarray:
    """uint8/float [0,255] HWC RGB -> float [0,255] Y (H,W,1), without rounding (metric_util.py:32-45)."""
    a = img.astype(np.float32) / np.float32(255.0)
    if a.ndim == 3 and a.shape[2] == 3:
        y = np.dot(a, [65.481, 128.553, 24.966]) + 16.0
        a = (y / 255.0).astype(np.float32)[..., None]
    return a * np.float32(255.0)


def _prep(img, img2, crop_border, test_y_channel):
    if img.shape != img2.shape:
        raise AssertionError(f"Image shapes are different: {img.shape}, {img2.shape}.")
    if img.ndim == 2:
        img, img2 = img[..., None], img2[..., None]
    if crop_border != 0:
        img = img[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        img, img2 = to_y_channel(img), to_y_channel(img2)
    return img.astype(np.float64), img2.astype(np.float64)


def calculate_psnr(img, img2, crop_border, test_y_channel=False, **_):
    a, b = _prep(img, img2, crop_border, test_y_channel)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else float(10.0 * np.log10(255.0 * 255.0 / mse))


def _gauss11():
    x = np.arange(11, dtype=np.float64) - 5.0
    g = np.exp(-(x * x) / (2.0 * 1.5 * 1.5))
    return g / g.sum()


def _valid_blur(a: np.ndarray) -> np.ndarray:
    g = _gauss11()
    h, w = a.shape
    rows = sum(g[i] * a[:, i:w - 10 + i] for i in range(11))
    return sum(g[i] * rows[i:h - 10 + i, :] for i in range(11))


def _ssim(a: np.ndarray, b: np.ndarray) -> float:
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = _valid_blur(a), _valid_blur(b)
    s1 = _valid_blur(a * a) - mu1 * mu1
    s2 = _valid_blur(b * b) - mu2 * mu2
    s12 = _valid_blur(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))
    return float(m.mean())


def calculate_ssim(img, img2, crop_border, test_y_channel=False, **_):
    a, b = _prep(img, img2, crop_border, test_y_channel)
    return float(np.mean([_ssim(a[..., i], b[..., i]) for i in range(a.shape[2])]))


METRICS = {"calculate_psnr": calculate_psnr, "calculate_ssim": calculate_ssim}


def calculate_metric(data: dict, opt: dict) -> float:
    """basicsr.metrics.calculate_metric: opt = {type: calculate_psnr|calculate_ssim, crop_border, test_y_channel}."""
    opt = dict(opt)
    fn = METRICS[opt.pop("type")]
    return fn(data["img"], data["img2"], **opt)
