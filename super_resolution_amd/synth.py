"""Portable, version-independent synthetic weights and inputs.

There are no loadable checkpoints for this fork's HAT (the shipped YAMLs point at
upstream `.pth` files whose keys do not match), so parity and the benchmark use seeded synthetic
parameters.  The generator is counter based (splitmix64 of (seed, key-hash, element index) ->
Box-Muller in float64 -> float32) so that the *same* tensors are produced in the build container
(where the golden vectors are made with the reference imported) and on the GPU box, without
relying on `torch.manual_seed` streams.

Distributions follow SURVEY.md §8(d): matrices/convs N(0, 1/fan_in), biases N(0, 0.02^2),
LayerNorm weight 1 + N(0, 0.1^2), relative-position-bias table N(0, 0.5^2).  *Every* parameter
is randomised, including the ones the reference zero-initialises (`dwc_proj.3.*`,
esc_arch.py:101-102), otherwise the dynamic-kernel path would never be exercised.
"""
from __future__ import annotations

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def _uniform01(seed: int, key: str, n: int, stream: int) -> np.ndarray:
    """n doubles in (0,1), a pure function of (seed, key, stream, index)."""
    base = (_fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15) ^ (stream * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(_splitmix64(idx + np.uint64(base)) ^ np.uint64(base))
    # 53 random bits -> (0,1)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, key: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = _uniform01(seed, key, n, 1)
    u2 = _uniform01(seed, key, n, 2)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    out = (mean + std * z).astype(np.float32).reshape(tuple(shape))
    return torch.from_numpy(out)


def uniform(seed: int, key: str, shape) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    out = _uniform01(seed, key, n, 3).astype(np.float32).reshape(tuple(shape))
    return torch.from_numpy(out)


def synth_input(seed: int, shape) -> torch.Tensor:
    """LR input `uniform[0,1)` (the reference's own timing input, test_direct_metrics.py:60)."""
    return uniform(seed, "input", shape)


def _fan_in(shape) -> int:
    if len(shape) <= 1:
        return 1
    f = 1
    for s in shape[1:]:
        f *= int(s)
    return f


def synth_state_dict(reference_sd: dict, seed: int = 1234) -> dict:
    """Return a fully randomised fp32 state dict with the keys/shapes/dtypes of `reference_sd`.

    Integer buffers (`relative_position_index_*`) are copied unchanged: they are part of the
    state-dict contract and are deterministic functions of the config.
    """
    out = {}
    for k, v in reference_sd.items():
        shape = tuple(v.shape)
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        leaf = k.rsplit(".", 1)[-1]
        if "relative_position_bias_table" in k:
            t = normal(seed, k, shape, std=0.5)
        elif (".norm" in k or k.startswith("norm.")) and leaf == "weight":
            t = normal(seed, k, shape, std=0.1, mean=1.0)
        elif leaf == "bias":
            t = normal(seed, k, shape, std=0.02)
        else:
            t = normal(seed, k, shape, std=float(_fan_in(shape)) ** -0.5)
        out[k] = t.to(v.dtype)
    return out
