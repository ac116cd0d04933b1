"""`HAT` — drop-in for the reference's `hat.archs.hat_arch.HAT` (hat/archs/hat_arch.py:607-859).

Same constructor kwargs (unknown keys swallowed, :644), same `state_dict()` key/shape/dtype
surface (checked in tests against the reference's own surface), same
registration name `'HAT'`, same call contract: `net(x)` with `x: (B,3,H,W)` float in [0,1],
H and W multiples of `window_size`, returns `(B,3,sH,sW)` float32 on the same device.

The module tree below only HOLDS parameters (nn.Linear / nn.Conv2d / nn.LayerNorm are used as
typed containers so initialisation matches the reference, :761-768).  `forward` never runs a
torch op on them: it hands the parameters to `HATEngine`, which packs them for the MI355X
kernels and enqueues hand-written HIP code through the C ABI (include/hat_mi355x.h).  There is
no CPU path: a CPU tensor or a missing libhat_mi355x.so raises.
"""
from __future__ import annotations

import collections
import math
import os

import torch
import torch.nn as nn

from ..registry import ARCH_REGISTRY


def _index_table_sa(ws: int) -> torch.Tensor:
    """`relative_position_index_SA` (hat_arch.py:770-781): (qh_i-qh_j+ws-1)*(2ws-1) + (qw_i-qw_j+ws-1).
    Dead on this path (no block consumes it) but part of the state-dict contract."""
    i = torch.arange(ws * ws)
    h, w = i // ws, i % ws
    return ((h[:, None] - h[None, :] + ws - 1) * (2 * ws - 1) + (w[:, None] - w[None, :] + ws - 1)).to(torch.int64)


def _index_table_oca(ws: int, overlap_ratio: float) -> torch.Tensor:
    """`relative_position_index_OCA` (hat_arch.py:783-803); keeps the reference's NEGATIVE values
     for strict state-dict compatibility — the kernels use a rotated table instead."""
    wse = ws + int(overlap_ratio * ws)
    qi, ki = torch.arange(ws * ws), torch.arange(wse * wse)
    off = ws - wse + 1
    dh = (ki // wse)[None, :] - (qi // ws)[:, None] + off
    dw = (ki % wse)[None, :] - (qi % ws)[:, None] + off
    return (dh * (ws + wse - 1) + dw).to(torch.int64)


class _Holder(nn.Module):
    """A module that only owns parameters/sub-modules; calling it is an error by design."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the forward pass runs in HIP kernels via HAT.forward")


class _ECA(_Holder):
    def __init__(self, k_size=5):
        super().__init__()
        self.conv = nn.Conv1d(1, 1, kernel_size=k_size, padding=k_size // 2, bias=False)


class _CAB(_Holder):
    def __init__(self, dim, compress_ratio):
        super().__init__()
        mid = dim // compress_ratio
        self.cab = nn.Sequential(nn.Conv2d(dim, mid, 3, 1, 1), nn.GELU(), nn.Conv2d(mid, dim, 3, 1, 1), _ECA(5))


class _ConvAttn(_Holder):  # esc_arch.py:89-102
    def __init__(self, pdim):
        super().__init__()
        self.dwc_proj = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(pdim, pdim // 2, 1, 1, 0), nn.GELU(),
                                      nn.Conv2d(pdim // 2, pdim * 9, 1, 1, 0))
        nn.init.zeros_(self.dwc_proj[-1].weight)
        nn.init.zeros_(self.dwc_proj[-1].bias)


class _ConvAttnWrapper(_Holder):  # esc_arch.py:136-140
    def __init__(self, dim, pdim):
        super().__init__()
        self.plk = _ConvAttn(pdim)
        self.aggr = nn.Conv2d(dim, dim, 1, 1, 0)


class _ESCAttn(_Holder):  # hat_arch.py:139-149
    def __init__(self, dim, pdim, ksize):
        super().__init__()
        self.core = _ConvAttnWrapper(dim, pdim)
        self.plk_filter = nn.Parameter(torch.randn(pdim, pdim, ksize, ksize))
        nn.init.orthogonal_(self.plk_filter)


class _FFN(_Holder):  # hat_arch.py:95-105
    def __init__(self, dim, mlp_ratio):
        super().__init__()
        hidden = int(dim * mlp_ratio)
        self.fc1 = nn.Linear(dim, 2 * hidden)
        self.dw = nn.Conv2d(2 * hidden, 2 * hidden, 3, 1, 1, groups=2 * hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _SGFN(_Holder):  # hatx_arch.py:144-158 (SpatialGateDConvFFN): the depthwise conv covers the first half only
    def __init__(self, dim, mlp_ratio):
        super().__init__()
        hidden = int(dim * mlp_ratio)
        assert hidden % 2 == 0, f"Hidden({hidden}) must be even for spatial gate split."
        self.fc1 = nn.Linear(dim, hidden)
        self.dw = nn.Conv2d(hidden // 2, hidden // 2, 3, 1, 1, groups=hidden // 2)
        self.fc2 = nn.Linear(hidden, dim)


class _HAB(_Holder):  # hat_arch.py:172-215 (sgfn: hatx_arch.py:185-230)
    def __init__(self, dim, compress_ratio, mlp_ratio, esc_pdim, esc_kernel, sgfn=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.esc_attn = _ESCAttn(dim, esc_pdim, esc_kernel)
        self.conv_block = _CAB(dim, compress_ratio)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _SGFN(dim, mlp_ratio) if sgfn else _FFN(dim, mlp_ratio)


class _OCAB(_Holder):  # hat_arch.py:267-324
    def __init__(self, dim, window_size, overlap_ratio, num_heads, qkv_bias, mlp_ratio, esc_enable, esc_pdim, esc_kernel,
                 focus=False):
        super().__init__()
        wse = int(window_size * overlap_ratio) + window_size
        self.norm1 = nn.LayerNorm(dim)
        self.q_proj = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv_proj = nn.Linear(dim, 2 * dim, bias=qkv_bias)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((window_size + wse - 1) ** 2, num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        hid = int(dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(dim, hid), nn.GELU(), nn.Linear(hid, dim))
        if esc_enable:
            self.esc_core = _ConvAttnWrapper(dim, esc_pdim)
            self.esc_plk = nn.Parameter(torch.randn(esc_pdim, esc_pdim, esc_kernel, esc_kernel))
            nn.init.orthogonal_(self.esc_plk)
        if focus:  # hatx_arch.py:357-361: saliency head of the focus bias
            self.focus_head = nn.Sequential(nn.Conv2d(dim, dim // 4, 1, 1, 0), nn.GELU(), nn.Conv2d(dim // 4, 1, 1, 1, 0))


class _AttenBlocks(_Holder):  # hat_arch.py:395-464
    def __init__(self, dim, depth, num_heads, window_size, compress_ratio, overlap_ratio, mlp_ratio, qkv_bias, esc_pdim,
                 esc_kernel, ocab_esc_enable, ocab_esc_pdim, ocab_esc_kernel, sgfn=False, focus=False):
        super().__init__()
        self.blocks = nn.ModuleList([_HAB(dim, compress_ratio, mlp_ratio, esc_pdim, esc_kernel, sgfn) for _ in range(depth)])
        self.overlap_attn = _OCAB(dim, window_size, overlap_ratio, num_heads, qkv_bias, mlp_ratio, ocab_esc_enable,
                                  ocab_esc_pdim, ocab_esc_kernel, focus)


class _RHAG(_Holder):  # hat_arch.py:484-553
    def __init__(self, dim, resi_connection, **kw):
        super().__init__()
        self.residual_group = _AttenBlocks(dim, **kw)
        if resi_connection == '1conv':
            self.conv = nn.Conv2d(dim, dim, 3, 1, 1)
        elif resi_connection == 'identity':
            self.conv = nn.Identity()
        else:
            raise ValueError(f"Unknown resi_connection: {resi_connection}")


class _PatchEmbed(_Holder):  # hat_arch.py:558-570
    def __init__(self, embed_dim, norm_layer):
        super().__init__()
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None


class _Upsample(nn.Sequential):  # hat_arch.py:593-605
    def __init__(self, scale, num_feat):
        m = []
        if (scale & (scale - 1)) == 0:
            for _ in range(int(math.log(scale, 2))):
                m.append(nn.Conv2d(num_feat, 4 * num_feat, 3, 1, 1))
                m.append(nn.PixelShuffle(2))
        elif scale == 3:
            m.append(nn.Conv2d(num_feat, 9 * num_feat, 3, 1, 1))
            m.append(nn.PixelShuffle(3))
        else:
            raise ValueError(f'scale {scale} is not supported. Supported scales: 2^n and 3.')
        super().__init__(*m)


@ARCH_REGISTRY.register()
class HAT(nn.Module):
    """Hybrid Attention Transformer (this fork: ESC conv-attention HABs + OCAB), MI355X-native forward.

    Extra (non-reference) constructor keyword: `compute_dtype` in {'bf16', 'f32'} selects the
    kernel storage/MFMA type (default 'bf16'; 'f32' is the exact-fp32 parity path).  It can also
    be changed later with `set_compute_dtype`.
    """
    _VARIANT = "hat"

    def __init__(self, img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=(6, 6, 6, 6), num_heads=(6, 6, 6, 6),
                 window_size=7, compress_ratio=3, squeeze_factor=30, conv_scale=0.01, overlap_ratio=0.5, mlp_ratio=4.,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 norm_layer=nn.LayerNorm, ape=False, patch_norm=True, use_checkpoint=False, upscale=2, img_range=1.,
                 upsampler='', resi_connection='1conv', esc_pdim: int = 16, esc_kernel: int = 13,
                 esc_use_dynamic: bool = True, ocab_esc_enable: bool = False, ocab_esc_pdim: int = 16,
                 ocab_esc_kernel: int = 13, compute_dtype: str = 'bf16', **kwargs):
        super().__init__()
        self.cfg = dict(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                        depths=tuple(depths), num_heads=tuple(num_heads), window_size=window_size,
                        compress_ratio=compress_ratio, squeeze_factor=squeeze_factor, conv_scale=conv_scale,
                        overlap_ratio=overlap_ratio, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, ape=ape,
                        patch_norm=patch_norm, upscale=upscale, img_range=img_range, upsampler=upsampler,
                        resi_connection=resi_connection, esc_pdim=esc_pdim, esc_kernel=esc_kernel,
                        ocab_esc_enable=ocab_esc_enable, ocab_esc_pdim=ocab_esc_pdim, ocab_esc_kernel=ocab_esc_kernel,
                        variant=self._VARIANT, kv_topk_ratio=float(kwargs.get("_kv_topk_ratio", 1.0)),
                        use_focus_bias=bool(kwargs.get("_use_focus_bias", False)))
        self.window_size = window_size
        self.shift_size = window_size // 2
        self.overlap_ratio = overlap_ratio
        self.img_range = img_range
        self.upscale = upscale
        self.upsampler = upsampler
        self.embed_dim = embed_dim
        self.num_layers = len(depths)
        self.compute_dtype = compute_dtype
        num_feat = 64  # hard-coded in the reference (:656)
        if in_chans == 3:
            self.mean = torch.Tensor((0.4488, 0.4371, 0.4040)).view(1, 3, 1, 1)
        else:
            self.mean = torch.zeros(1, 1, 1, 1)

        self.register_buffer('relative_position_index_SA', _index_table_sa(window_size))
        self.register_buffer('relative_position_index_OCA', _index_table_oca(window_size, overlap_ratio))

        self.conv_first = nn.Conv2d(in_chans, embed_dim, 3, 1, 1)
        self.patch_embed = _PatchEmbed(embed_dim, norm_layer if patch_norm else None)
        if ape:
            n = (img_size // patch_size) ** 2
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, n, embed_dim))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.layers = nn.ModuleList([
            _RHAG(embed_dim, resi_connection, depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                  compress_ratio=compress_ratio, overlap_ratio=overlap_ratio, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                  esc_pdim=esc_pdim, esc_kernel=esc_kernel, ocab_esc_enable=ocab_esc_enable, ocab_esc_pdim=ocab_esc_pdim,
                  ocab_esc_kernel=ocab_esc_kernel, sgfn=self._VARIANT == "hatx",
                  focus=bool(kwargs.get("_use_focus_bias", False))) for i in range(self.num_layers)])
        self.norm = norm_layer(embed_dim)
        if resi_connection == '1conv':
            self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1)
        elif resi_connection == 'identity':
            self.conv_after_body = nn.Identity()
        else:
            raise ValueError(f"Unknown resi_connection: {resi_connection}")
        if self.upsampler == 'pixelshuffle':
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.upsample = _Upsample(upscale, num_feat)
            self.conv_last = nn.Conv2d(num_feat, in_chans, 3, 1, 1)
        self.apply(self._init_weights)
        self._engine = None
        self._engine_key = None
        self._wver = 0   # bumped whenever the parameters may have changed (load_state_dict, .to()/.cuda()/..., explicit)
        self._plist = None
        self.register_load_state_dict_post_hook(HAT._post_load_hook)
        # extra (non-reference) switch: replay the forward as a HIP graph (also HAT_GRAPH=1 in the environment)
        self.use_graph = bool(kwargs.get("use_graph", False)) or os.environ.get("HAT_GRAPH") == "1"
        self._graphs, self._graph_engine = collections.OrderedDict(), None
        self._graph_max = int(os.environ.get("HAT_GRAPH_CACHE", "4"))

    def _init_weights(self, m):  # hat_arch.py:761-768
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    # ------------------------------------------------------------------------------------------
    def set_compute_dtype(self, dtype: str):
        self.compute_dtype = dtype
        self._engine = None
        return self

    # The engine holds PACKED copies of the parameters.  It is rebuilt when the weights key changes.  The key is cheap
    # (a walk over data_ptr of ~900 parameters on every forward was ~1.5 ms, a third of a 64x64 forward) and has two parts:
    #   * `_wver`, bumped by every bulk path that rewrites parameters: a load_state_dict POST HOOK (fires for this module
    #     also when the call was made on a parent / wrapper — nn.DataParallel(net), nn.Sequential(net), a user container —
    #     where torch recurses through _load_from_state_dict and never calls the child's load_state_dict), `_apply`
    #     (.to / .cuda / .float / ...) and `mark_weights_changed()`;
    #   * the sum of `p._version` over a cached parameter list (~0.1 ms for HAT-S's 910 parameters): catches every in-place edit autograd sees —
    #     optimizer steps, `p.copy_()`, `p.mul_()` under no_grad.
    # What this cannot see: an edit made through `p.data` (`p.data.mul_(d)`: `.data` has its own version counter by design,
    # e.g. BasicSR's EMA update) and a parameter OBJECT replaced on a submodule (`net.conv_first.weight = nn.Parameter(..)`):
    # such code must call `mark_weights_changed()` (the walk that would find them costs 1.5 ms).  HAT_STRICT_WEIGHTS=1 restores the
    # full per-forward walk over (data_ptr, _version) for debugging.
    def mark_weights_changed(self):
        self._wver += 1
        return self

    @staticmethod
    def _post_load_hook(module, incompatible_keys):
        module._wver = getattr(module, "_wver", 0) + 1

    def _apply(self, fn, *args, **kwargs):
        r = super()._apply(fn, *args, **kwargs)
        self._wver = getattr(self, "_wver", 0) + 1
        self._plist = None
        return r

    def _weights_key(self, device):
        pl = getattr(self, "_plist", None)
        if pl is None or self._plist_wver != self._wver:
            pl = self._plist = list(self.parameters())
            self._plist_wver = self._wver
        key = (str(device), self.compute_dtype, self._wver, sum(p._version for p in pl))
        if os.environ.get("HAT_STRICT_WEIGHTS") == "1":
            key += tuple((p.data_ptr(), p._version) for p in self.parameters())
        return key

    def engine(self, device=None):
        """The packed-weight engine for the current parameters (re-packed when they change)."""
        from ..engine import HATEngine
        device = torch.device(device) if device is not None else self.conv_first.weight.device
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = self._weights_key(device)
        if self._engine is None or self._engine_key != key:
            if self.conv_first.weight.device != device:
                raise RuntimeError(f"input is on {device} but the parameters are on {self.conv_first.weight.device}: "
                                   f"move the module first (net.to('{device}'))")
            self._engine = HATEngine(self.cfg, self.state_dict(), device, self.compute_dtype)
            self._engine_key = key
        return self._engine

    def forward(self, x):
        if self.training:
            raise RuntimeError("this HAT implements the inference forward pass only: call .eval() first "
                               "(training is out of scope, SURVEY §2)")
        if not x.is_cuda:
            raise RuntimeError("HAT.forward needs a GPU tensor: the MI355X HIP path is the only path (no CPU fallback)")
        with torch.no_grad():
            from .. import ops
            if self.use_graph and not ops.profiling():
                return self._forward_graph(x)
            return self.engine(x.device).forward(x).to(x.dtype)

    # ---- exact full-frame sharding into row bands (no counterpart in the reference, whose tile loop
    # hat_model.py:40-108 gives a DIFFERENT result than the full frame: SURVEY F6) ----
    def _check_band_input(self, x):
        if self.training:
            raise RuntimeError("this HAT implements the inference forward pass only: call .eval() first")
        if not x.is_cuda:
            raise RuntimeError("HAT.forward needs a GPU tensor: the MI355X HIP path is the only path (no CPU fallback)")

    def forward_bands(self, x, n_bands: int):
        """`forward(x)` computed as n_bands row bands on this GPU that exchange halo rows and pool sums: equal to the
        unsharded forward up to the summation order of the two global pools (band_parallel.forward_bands_local)."""
        from .. import band_parallel
        self._check_band_input(x)
        with torch.no_grad():
            return band_parallel.forward_bands_local(self.engine(x.device), x, n_bands).to(x.dtype)

    def forward_band_parallel(self, x, group=None):
        """`forward(x)` with one row band per rank of `group` (torch.distributed; RCCL send / recv of halo rows between
        neighbours, one small all-reduce per pool, one all-gather of the output rows): every rank passes the same x and
        receives the full output frame (band_parallel.forward_band_distributed)."""
        from .. import band_parallel
        self._check_band_input(x)
        with torch.no_grad():
            return band_parallel.forward_band_distributed(self.engine(x.device), x, group).to(x.dtype)

    def _forward_graph(self, x):
        """Replay the whole forward (about 290 launches on two streams) as one HIP graph: captured once per input
        shape after two eager warm-up passes, input copied into the graph's static buffer, output copied out of it.
        Removes the host launch path and most inter-kernel gaps; matters most for small frames / tiles."""
        eng = self.engine(x.device)
        if self._graph_engine is not eng:
            self._graphs, self._graph_engine = collections.OrderedDict(), eng
        key = (tuple(x.shape), x.dtype)
        with torch.cuda.device(x.device):
            ent = self._graphs.get(key)
            if ent is None:
                sx = x.detach().clone()
                cur = torch.cuda.current_stream(x.device)
                side = torch.cuda.Stream(device=x.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(2):
                        eng.forward(sx)
                cur.wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    sy = eng.forward(sx)
                # the graph bakes in raw pointers into the engine's workspace of this shape: the entry keeps that
                # workspace alive even after the engine's own LRU has dropped it
                ent = self._graphs[key] = (g, sx, sy, eng._workspace(*x.shape[:1], *x.shape[2:]))
                while len(self._graphs) > self._graph_max:
                    self._graphs.popitem(last=False)
            else:
                self._graphs.move_to_end(key)
            g, sx, sy, _ = ent
            sx.copy_(x)
            g.replay()
            return sy.to(x.dtype, copy=True)


@ARCH_REGISTRY.register()
class HATX(HAT):
    """Drop-in for the reference's `hat.archs.hatx_arch.HATX` (hatx_arch.py:707-974): `HAT` with the Spatial-Gate DConv FFN
    (SGFN) in every HAB and an OCAB that pads its key windows with ceil((wse - ws) / 2), may add a focus bias to the logits
    and may prune keys (top-k).  Same constructor keywords (`hab_ffn_ratio` is accepted and, exactly like the reference,
    never used: its AttenBlocks hands `mlp_ratio` to the HABs, hatx_arch.py:513), same `state_dict()` surface.

    The MI355X forward covers the SGFN, the focus bias and the top-k pruning for key windows of 12 / 24 (overlap 0.5) and the
    odd 13 / 25 (ceil padding, hatx_arch.py:303-305: overlap 0.7 at window 8, 0.6 at window 16 — the live training config, in
    bf16), ESC on up to 32 channels with kernels up to 17 x 17.  Among keys of EQUAL score the pruning keeps the lower window index, where the reference
    leaves the order to torch.topk: border windows, whose zero-padded keys all score tanh(0) = 0, can therefore differ from
    the reference; windows without padded keys cannot.
    """
    _VARIANT = "hatx"

    def __init__(self, *args, hab_ffn_ratio: float = 2.0, kv_topk_ratio: float = 1.0, use_focus_bias: bool = False, **kwargs):
        self.hab_ffn_ratio = hab_ffn_ratio
        super().__init__(*args, _kv_topk_ratio=kv_topk_ratio, _use_focus_bias=use_focus_bias, **kwargs)
        self.kv_topk_ratio, self.use_focus_bias = float(kv_topk_ratio), bool(use_focus_bias)
