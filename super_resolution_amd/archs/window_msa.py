"""(Shifted-)window multi-head self-attention, (S)W-MSA — SURVEY §8 row f2.

Host-side mirror of the reference's `WindowAttention` (ESC/basicsr/archs/swinir_arch.py:95-175: same constructor
arguments, same parameters / buffers and state-dict keys, so an upstream Swin / HAT attention checkpoint loads with
`strict=True`) in front of `hat_window_attention` (include/hat_mi355x.h).  This fork's `HAB` dropped the window attention
 but still registers its index / mask buffers (hat_arch.py:770-781, 805-818); this module is the optional
HAB attention that upstream-HAT checkpoints need.

Where the reference block rolls the map, partitions it into windows, runs `WindowAttention.forward(windows, mask)`,
reverses the partition and rolls back (swinir_arch.py:291-317), `forward_map(x, shift)` takes the normalised map
(B, H, W, C) and does all of that inside one kernel launch between two `hat_linear` launches: the rolls, the partition
and the -100 shift mask (:262-280) are addressing and a band comparison in the kernel, nothing is materialised.
There is no CPU path: the HIP library is the only implementation.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .._lib import O_NHWC_T


def relative_position_index(ws: int) -> torch.Tensor:
    """swinir_arch.py:120-131 (== hat_arch.py:770-781): (qh-kh+ws-1)*(2ws-1) + (qw-kw+ws-1), int64 (ws^2, ws^2)."""
    i = torch.arange(ws * ws)
    h, w = i // ws, i % ws
    return ((h[:, None] - h[None, :] + ws - 1) * (2 * ws - 1) + (w[:, None] - w[None, :] + ws - 1)).to(torch.int64)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.,
                 compute_dtype: str = "bf16"):
        super().__init__()
        ws = window_size if isinstance(window_size, (tuple, list)) else (window_size, window_size)
        if ws[0] != ws[1]:
            raise ValueError("square windows only (the reference only ever builds square ones)")
        self.dim, self.window_size, self.num_heads = dim, tuple(ws), num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws[0] - 1) * (2 * ws[1] - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(ws[0]))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.compute_dtype = compute_dtype
        self._packed, self._key = None, None
        # attn_drop / proj_drop are identities in eval; training is out of scope

    def extra_repr(self) -> str:
        return f"dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}"

    def _pack(self, dev):
        key = (str(dev), self.compute_dtype, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if self._packed is not None and self._key == key:
            return self._packed
        dt = ops.DTYPE_CODE[self.compute_dtype]
        C = self.dim

        def lin(w, b):
            if ops.linear_supported(w.shape[0], w.shape[1], dt):
                pw = ops.pack_linear_weight(w, b, dt, dev)
                pw.frag = True
            else:
                pw = ops.pack_conv_weight(w, b, dt, dev)
                pw.frag = False
            return pw

        w = self.qkv.weight.detach().float().clone()
        b = (self.qkv.bias.detach().float().clone() if self.qkv.bias is not None else torch.zeros(3 * C, device=w.device))
        w[:C] *= self.scale   # q * scale (swinir_arch.py:149) folded into the projection
        b[:C] *= self.scale
        table = self.relative_position_bias_table.detach().float()
        # kernel index = (kh-qh+ws-1)*(2ws-1) + (kw-qw+ws-1) = (2ws-1)^2 - 1 - relative_position_index
        bias_flip = table.flip(0).t().contiguous().to(dev)
        self._packed = (lin(w, b), lin(self.proj.weight.detach().float(), self.proj.bias.detach().float()), bias_flip, dt)
        self._key = key
        return self._packed

    def forward_map(self, x: torch.Tensor, shift: int = 0) -> torch.Tensor:
        """x: normalised map (B, H, W, C) on the GPU -> attention branch (B, H, W, C) at the un-shifted positions
        (what swinir_arch.py:291-317 computes between `norm1` and the residual add)."""
        if self.training:
            raise RuntimeError("inference only: call .eval() first (training is out of scope)")
        if not x.is_cuda:
            raise RuntimeError("WindowAttention needs a GPU tensor: the MI355X HIP path is the only path (no CPU fallback)")
        B, H, W, C = x.shape
        ws = self.window_size[0]
        if C != self.dim or H % ws or W % ws:
            raise RuntimeError(f"map {tuple(x.shape)} is not (B, k*{ws}, l*{ws}, {self.dim})")
        qkv_w, proj_w, bias_flip, dt = self._pack(x.device)
        tdt = ops.TORCH_DTYPE[dt]
        ldc, ld3 = (C + 7) // 8 * 8, (3 * C + 7) // 8 * 8   # T-typed rows: channel stride rounded up to 8, zero pad channels
        with torch.no_grad():
            buf = torch.empty if ldc == C else torch.zeros   # pad channels must read as zeros
            xt = buf(B, H, W, ldc, dtype=tdt, device=x.device)
            xt[..., :C] = x.detach()
            qkv = buf(B, H, W, ld3, dtype=tdt, device=x.device)
            att = buf(B, H, W, ldc, dtype=tdt, device=x.device)
            out = buf(B, H, W, ldc, dtype=tdt, device=x.device)
            run = lambda pw, src, dst, ldx, ldo: (ops.linear if pw.frag else ops.conv)(
                pw, src, dst, B=B, H=H, W=W, dtype=dt, ldx=ldx, ldo=ldo, out_mode=O_NHWC_T)
            run(qkv_w, xt, qkv, ldc, ld3)
            ops.window_attention(qkv, qkv.view(-1)[C:], bias_flip, att, B=B, H=H, W=W, C_=C, heads=self.num_heads, ws=ws,
                                 shift=shift, ldq=ld3, ldkv=ld3, ldo=ldc, dtype=dt)
            run(proj_w, att, out, ldc, ldc)
            out = out[..., :C]
        return out.to(x.dtype)

    def forward(self, x: torch.Tensor, mask=None) -> torch.Tensor:
        """Reference signature (swinir_arch.py:140): x = windows (num_windows*b, ws*ws, C).  Without a mask every window is
        an independent ws x ws map.  With the shift mask use forward_map(map, shift): the kernel derives the mask from the
        window position, it does not read a mask tensor."""
        if mask is not None:
            raise NotImplementedError("pass the un-partitioned map to forward_map(x, shift) for SW-MSA")
        b_, n, c = x.shape
        ws = self.window_size[0]
        if n != ws * ws:
            raise RuntimeError(f"expected windows of {ws * ws} tokens, got {n}")
        return self.forward_map(x.reshape(b_, ws, ws, c), 0).reshape(b_, n, c)
