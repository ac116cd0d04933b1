"""ORACLE — test infrastructure only. NOT a product path.

CPU restatement (plain PyTorch fp32/fp64 ops, functional over a state dict) of the reference's
HAT forward pass, the one hot path named by BASELINE.json.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this module; the
product package `super_resolution_amd` never does (it fails loudly when the HIP library is
missing instead of falling back to this code).

Pinning: `tests/golden/gen_golden.py` imports the *reference itself* from /root/reference in
the build container and stores input/output vectors under `tests/golden/`;
`tests/test_oracle_golden.py` checks this restatement against them (<= 1e-5 max-abs, fp32).
The reference has no tests or fixtures of its own for this path.

Every function cites the reference lines it follows (paths relative to /root/reference/HAT).
Tensors are NCHW at the network boundary and (B, N, C) tokens inside, as in the reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # hat/archs/hat_arch.py:659

DEFAULT_CFG = dict(  # constructor defaults, hat/archs/hat_arch.py:610-644
    img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=(6, 6, 6, 6), num_heads=(6, 6, 6, 6),
    window_size=7, compress_ratio=3, squeeze_factor=30, conv_scale=0.01, overlap_ratio=0.5, mlp_ratio=4.0,
    qkv_bias=True, qk_scale=None, ape=False, patch_norm=True, upscale=2, img_range=1.0, upsampler="",
    resi_connection="1conv", esc_pdim=16, esc_kernel=13, ocab_esc_enable=False, ocab_esc_pdim=16,
    ocab_esc_kernel=13,
)


def make_cfg(**kw) -> dict:
    cfg = dict(DEFAULT_CFG)
    for k, v in kw.items():
        if k in cfg:
            cfg[k] = v  # unknown keys are swallowed like the reference's **kwargs (hat_arch.py:644)
    return cfg


# --------------------------------------------------------------------------------------------
# integer index tables (hat_arch.py:770-803) — registered buffers, part of the state dict
# --------------------------------------------------------------------------------------------
def rpi_sa(window_size: int) -> Tensor:
    """hat_arch.py:770-781.  rpi[i,j] = (qh_i-qh_j+ws-1)*(2ws-1) + (qw_i-qw_j+ws-1)."""
    ws = window_size
    idx = torch.arange(ws * ws)
    h, w = idx // ws, idx % ws
    dh = h[:, None] - h[None, :] + ws - 1
    dw = w[:, None] - w[None, :] + ws - 1
    return (dh * (2 * ws - 1) + dw).to(torch.int64)


def rpi_oca(window_size: int, overlap_ratio: float) -> Tensor:
    """hat_arch.py:783-803.  rpi[i,j] = (kh-qh+ws-wse+1)*(ws+wse-1) + (kw-qw+ws-wse+1).

    Contains NEGATIVE values; consumers rely on negative-index wraparound.
    """
    ws = window_size
    wse = ws + int(overlap_ratio * ws)
    qi = torch.arange(ws * ws)
    ki = torch.arange(wse * wse)
    qh, qw = qi // ws, qi % ws
    kh, kw = ki // wse, ki % wse
    off = ws - wse + 1
    dh = kh[None, :] - qh[:, None] + off
    dw = kw[None, :] - qw[:, None] + off
    return (dh * (ws + wse - 1) + dw).to(torch.int64)


# --------------------------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------------------------
def _ln(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _tok2img(t: Tensor, hw) -> Tensor:
    b, n, c = t.shape
    return t.transpose(1, 2).reshape(b, c, hw[0], hw[1])


def _img2tok(x: Tensor) -> Tensor:
    return x.flatten(2).transpose(1, 2)


def cab(n_img: Tensor, sd: SD, p: str) -> Tensor:
    """CAB + ECA, hat_arch.py:80-90 and :66-78.  n_img: (B,C,H,W)."""
    c1 = F.gelu(F.conv2d(n_img, sd[p + ".cab.0.weight"], sd[p + ".cab.0.bias"], padding=1))
    c2 = F.conv2d(c1, sd[p + ".cab.2.weight"], sd[p + ".cab.2.bias"], padding=1)
    g = c2.mean(dim=(2, 3))  # (B,C)                                   :73
    e = F.conv1d(g[:, None, :], sd[p + ".cab.3.conv.weight"], padding=sd[p + ".cab.3.conv.weight"].shape[-1] // 2)
    e = torch.sigmoid(e)[:, 0, :]  # (B,C)                              :74-77
    return c2 * e[:, :, None, None]


def esc_dynamic_kernel(n_img: Tensor, sd: SD, p: str, pdim: int) -> Tensor:
    """esc_arch.py:95-100,121.  Returns per-sample (B, pdim, 3, 3) depthwise kernels."""
    g = n_img[:, :pdim].mean(dim=(2, 3), keepdim=True)  # (B,pdim,1,1)
    h = F.gelu(F.conv2d(g, sd[p + ".dwc_proj.1.weight"], sd[p + ".dwc_proj.1.bias"]))
    dk = F.conv2d(h, sd[p + ".dwc_proj.3.weight"], sd[p + ".dwc_proj.3.bias"])  # (B,pdim*9,1,1)
    return dk.reshape(n_img.shape[0], pdim, 3, 3)


def esc_conv_attn(n_img: Tensor, plk: Tensor, sd: SD, core: str, pdim: int) -> Tensor:
    """ConvAttnWrapper(ConvolutionalAttention) eval branch, esc_arch.py:119-123,142-145.

    The reference eval branch only accepts B==1; B>1 is defined here as the
    per-sample dynamic kernel of its training branch (esc_arch.py:105-118).
    """
    b = n_img.shape[0]
    dk = esc_dynamic_kernel(n_img, sd, core + ".plk", pdim)
    x1 = n_img[:, :pdim]
    lk = F.conv2d(x1, plk, padding=plk.shape[-1] // 2)
    dyn = torch.cat([F.conv2d(x1[i:i + 1], dk[i][:, None], padding=1, groups=pdim) for i in range(b)], 0)
    y = torch.cat([lk + dyn, n_img[:, pdim:]], dim=1)
    return F.conv2d(y, sd[core + ".aggr.weight"], sd[core + ".aggr.bias"])


def gated_dconv_ffn(m: Tensor, hw, sd: SD, p: str) -> Tensor:
    """GatedDconvFFN, hat_arch.py:107-119.  m: (B,N,C) -> (B,N,C)."""
    u = F.linear(m, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])
    u = _tok2img(u, hw)
    u = F.conv2d(u, sd[p + ".dw.weight"], sd[p + ".dw.bias"], padding=1, groups=u.shape[1])
    u = _img2tok(u)
    a, g = u.chunk(2, dim=-1)
    return F.linear(a * F.silu(g), sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


def hab(t: Tensor, hw, sd: SD, p: str, cfg: dict) -> Tensor:
    """HAB.forward, hat_arch.py:217-238 (DropPath is identity in eval, :46-48)."""
    n = _ln(t, sd, p + ".norm1")
    n_img = _tok2img(n, hw)
    conv_x = _img2tok(cab(n_img, sd, p + ".conv_block"))
    attn_x = _img2tok(esc_conv_attn(n_img, sd[p + ".esc_attn.plk_filter"], sd, p + ".esc_attn.core", cfg["esc_pdim"]))
    t = t + attn_x + conv_x * cfg["conv_scale"]
    return t + gated_dconv_ffn(_ln(t, sd, p + ".norm2"), hw, sd, p + ".mlp")


def ocab_attention(q: Tensor, k: Tensor, v: Tensor, table: Tensor, rpi: Tensor, ws: int, wse: int,
                   heads: int, scale: float) -> Tensor:
    """Window cross-attention core of OCAB, hat_arch.py:353-388.

    q,k,v: (B,H,W,C) projected maps.  K/V windows are `wse x wse`, stride `ws`, ZERO padded
    *after* the biased projection (:296-297,359-360): out-of-image keys are exactly 0, are not
    masked, and still receive exp(RPB) softmax weight.  Returns (B,H,W,C).
    """
    b, h, w, c = q.shape
    d = c // heads
    nh, nw = h // ws, w // ws
    pad = (wse - ws) // 2

    def win_q(x):  # (B,H,W,C) -> (B*nW, heads, ws*ws, d)                  window_partition :124-128
        x = x.reshape(b, nh, ws, nw, ws, heads, d).permute(0, 1, 3, 5, 2, 4, 6)
        return x.reshape(b * nh * nw, heads, ws * ws, d)

    def win_kv(x):  # (B,H,W,C) -> (B*nW, heads, wse*wse, d)               nn.Unfold :296-297
        xp = F.pad(x, (0, 0, pad, pad, pad, pad))
        xs = xp.unfold(1, wse, ws).unfold(2, wse, ws)  # (B,nh,nw,C,wse,wse)
        xs = xs.reshape(b, nh, nw, heads, d, wse * wse).permute(0, 1, 2, 3, 5, 4)
        return xs.reshape(b * nh * nw, heads, wse * wse, d)

    qh, kh, vh = win_q(q) * scale, win_kv(k), win_kv(v)
    attn = qh @ kh.transpose(-2, -1)  # (B*nW, heads, ws^2, wse^2)          :375-376
    bias = table[rpi.reshape(-1)].reshape(ws * ws, wse * wse, heads).permute(2, 0, 1)  # neg. idx wrap :378-382
    attn = torch.softmax(attn + bias[None], dim=-1)
    o = attn @ vh  # (B*nW, heads, ws^2, d)
    o = o.reshape(b, nh, nw, heads, ws, ws, d).permute(0, 1, 4, 2, 5, 3, 6)  # window_reverse :130-134
    return o.reshape(b, h, w, c)


def sw_msa_mask(h: int, w: int, ws: int, shift: int) -> Tensor:
    """Shift mask of SW-MSA: ESC/basicsr/archs/swinir_arch.py:262-280 (same construction as the dead hat_arch.py:805-818).  Region ids are
    assigned on the SHIFTED frame in 3 x 3 bands ([0, h-ws), [h-ws, h-shift), [h-shift, h)); pairs of window positions in
    different regions get -100 (not -inf), same region 0.  Returns (nW, ws*ws, ws*ws)."""
    def band(n):
        r = torch.zeros(n, dtype=torch.int64)
        r[n - ws:n - shift] = 1
        r[n - shift:] = 2
        return r
    rid = (3 * band(h)[:, None] + band(w)[None, :]).reshape(h // ws, ws, w // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = rid[:, None, :] != rid[:, :, None]
    return torch.where(diff, torch.tensor(-100.0), torch.tensor(0.0))


def window_msa(x: Tensor, sd: SD, p: str, ws: int, heads: int, shift: int) -> Tensor:
    """(S)W-MSA branch of a Swin/HAT block on an already-normalised map x (B,H,W,C): cyclic shift, window partition,
    WindowAttention, window reverse, reverse shift — ESC/basicsr/archs/swinir_arch.py:291-317 with WindowAttention.forward :140-172
    (upstream HAT's HAB attention; this fork's HAB dropped it).  Parameters under prefix `p`:
    qkv.{weight,bias} (3C,C), proj.{weight,bias}, relative_position_bias_table ((2ws-1)^2, heads); the index is rpi_sa."""
    b, h, w, c = x.shape
    d = c // heads
    nh, nw = h // ws, w // ws
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))                                   # :297-298
    xw = x.reshape(b, nh, ws, nw, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b * nh * nw, ws * ws, c)  # :303-304
    qkv = F.linear(xw, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"]).reshape(-1, ws * ws, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]                                                   # :147-151
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + ".relative_position_bias_table"][rpi_sa(ws).reshape(-1)].reshape(ws * ws, ws * ws, heads)
    attn = attn + bias.permute(2, 0, 1)[None]                                                      # :153-156
    if shift:
        attn = attn.reshape(b, nh * nw, heads, ws * ws, ws * ws) + sw_msa_mask(h, w, ws, shift)[None, :, None]
        attn = attn.reshape(-1, heads, ws * ws, ws * ws)                                           # :158-161
    o = (torch.softmax(attn, dim=-1) @ v).transpose(1, 2).reshape(-1, ws * ws, c)                  # :162-168
    o = F.linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])                                  # :169
    o = o.reshape(b, nh, nw, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, c)              # :312-313
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))                                      # :316-317
    return o


def ocab(t: Tensor, hw, sd: SD, p: str, rpi: Tensor, cfg: dict, heads: int) -> Tensor:
    """OCAB.forward, hat_arch.py:326-393."""
    b, _, c = t.shape
    ws = cfg["window_size"]
    wse = int(ws * cfg["overlap_ratio"]) + ws
    scale = cfg["qk_scale"] or (c // heads) ** -0.5
    n = _ln(t, sd, p + ".norm1")
    x_img = n.reshape(b, hw[0], hw[1], c)
    y_img = x_img
    if cfg["ocab_esc_enable"]:  # :336-344
        y = esc_conv_attn(_tok2img(n, hw), sd[p + ".esc_plk"], sd, p + ".esc_core", cfg["ocab_esc_pdim"])
        y_img = y.permute(0, 2, 3, 1)
    q = F.linear(x_img, sd[p + ".q_proj.weight"], sd.get(p + ".q_proj.bias"))
    kv = F.linear(y_img, sd[p + ".kv_proj.weight"], sd.get(p + ".kv_proj.bias"))
    k, v = kv.split(c, dim=-1)
    o = ocab_attention(q, k, v, sd[p + ".relative_position_bias_table"], rpi, ws, wse, heads, scale)
    t = F.linear(o.reshape(b, -1, c), sd[p + ".proj.weight"], sd[p + ".proj.bias"]) + t  # :391
    m = _ln(t, sd, p + ".norm2")
    m = F.linear(F.gelu(F.linear(m, sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"])),
                 sd[p + ".mlp.2.weight"], sd[p + ".mlp.2.bias"])
    return t + m  # :392


def rhag(t: Tensor, hw, sd: SD, p: str, rpi: Tensor, cfg: dict, depth: int, heads: int) -> Tensor:
    """RHAG.forward :555-556 / AttenBlocks.forward :466-482."""
    t_in = t
    for i in range(depth):
        t = hab(t, hw, sd, f"{p}.residual_group.blocks.{i}", cfg)
    t = ocab(t, hw, sd, f"{p}.residual_group.overlap_attn", rpi, cfg, heads)
    if cfg["resi_connection"] == "1conv":
        t = _img2tok(F.conv2d(_tok2img(t, hw), sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1))
    return t + t_in


def upsample(f: Tensor, sd: SD, scale: int) -> Tensor:
    """Upsample, hat_arch.py:593-605."""
    if scale & (scale - 1) == 0:
        for i in range(int(math.log2(scale))):
            f = F.pixel_shuffle(F.conv2d(f, sd[f"upsample.{2 * i}.weight"], sd[f"upsample.{2 * i}.bias"], padding=1), 2)
    elif scale == 3:
        f = F.pixel_shuffle(F.conv2d(f, sd["upsample.0.weight"], sd["upsample.0.bias"], padding=1), 3)
    else:
        raise ValueError(f"scale {scale} is not supported. Supported scales: 2^n and 3.")
    return f


def forward_features(f0: Tensor, sd: SD, cfg: dict) -> Tensor:
    """HAT.forward_features, hat_arch.py:828-846 (the SW-MSA mask/rpi_sa are dead)."""
    hw = (f0.shape[2], f0.shape[3])
    ws = cfg["window_size"]
    if hw[0] % ws or hw[1] % ws:  # the reference raises from calculate_mask's view (:815), SURVEY F4
        raise RuntimeError(f"input size {hw} is not a multiple of window_size {ws}")
    t = _img2tok(f0)
    if cfg["patch_norm"]:
        t = _ln(t, sd, "patch_embed.norm")
    if "absolute_pos_embed" in sd:
        t = t + sd["absolute_pos_embed"]
    rpi = sd["relative_position_index_OCA"]
    for g, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        t = rhag(t, hw, sd, f"layers.{g}", rpi, cfg, depth, heads)
    t = _ln(t, sd, "norm")
    return _tok2img(t, hw)


def hat_forward(x: Tensor, sd: SD, cfg: dict) -> Tensor:
    """HAT.forward, hat_arch.py:848-859 (upsampler == 'pixelshuffle')."""
    if cfg["upsampler"] != "pixelshuffle":
        raise NotImplementedError("only the 'pixelshuffle' upsampler is on the hot path (all test YAMLs)")
    if cfg["in_chans"] == 3:
        mean = torch.tensor(RGB_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    else:
        mean = torch.zeros(1, 1, 1, 1, dtype=x.dtype)
    r = cfg["img_range"]
    x = (x - mean) * r
    f0 = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    f = forward_features(f0, sd, cfg)
    if cfg["resi_connection"] == "1conv":
        f = F.conv2d(f, sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1)
    f = f + f0
    f = F.leaky_relu(F.conv2d(f, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"], padding=1), 0.01)
    f = upsample(f, sd, cfg["upscale"])
    y = F.conv2d(f, sd["conv_last.weight"], sd["conv_last.bias"], padding=1)
    return y / r + mean


# --------------------------------------------------------------------------------------------
# state-dict surface — shapes for a config, used to build synthetic weights
# --------------------------------------------------------------------------------------------
def state_dict_spec(cfg: dict) -> Dict[str, tuple]:
    """Key -> (shape, dtype) of `HAT(**cfg).state_dict()` (hat_arch.py:610-759), in reference order."""
    C = cfg["embed_dim"]
    ws = cfg["window_size"]
    wse = ws + int(cfg["overlap_ratio"] * ws)
    mid = C // cfg["compress_ratio"]
    hid = int(C * cfg["mlp_ratio"])
    pd, ks = cfg["esc_pdim"], cfg["esc_kernel"]
    f32, i64 = torch.float32, torch.int64
    spec: Dict[str, tuple] = {}

    def add(k, shape, dt=f32):
        spec[k] = (tuple(shape), dt)

    def lin(p, o, i, bias=True):
        add(p + ".weight", (o, i))
        if bias:
            add(p + ".bias", (o,))

    def conv(p, o, i, k):
        add(p + ".weight", (o, i, k, k))
        add(p + ".bias", (o,))

    def norm(p):
        add(p + ".weight", (C,))
        add(p + ".bias", (C,))

    def esc_core(p, pdim):
        conv(p + ".plk.dwc_proj.1", pdim // 2, pdim, 1)
        conv(p + ".plk.dwc_proj.3", pdim * 9, pdim // 2, 1)
        conv(p + ".aggr", C, C, 1)

    if cfg["ape"]:  # own parameters come before own buffers in Module.state_dict (hat_arch.py:699-702)
        add("absolute_pos_embed", (1, (cfg["img_size"] // cfg.get("patch_size", 1)) ** 2, C))
    add("relative_position_index_SA", (ws * ws, ws * ws), i64)
    add("relative_position_index_OCA", (ws * ws, wse * wse), i64)
    conv("conv_first", C, cfg["in_chans"], 3)
    if cfg["patch_norm"]:
        norm("patch_embed.norm")
    for g, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        for b in range(depth):
            p = f"layers.{g}.residual_group.blocks.{b}"
            norm(p + ".norm1")
            add(p + ".esc_attn.plk_filter", (pd, pd, ks, ks))
            esc_core(p + ".esc_attn.core", pd)
            conv(p + ".conv_block.cab.0", mid, C, 3)
            conv(p + ".conv_block.cab.2", C, mid, 3)
            add(p + ".conv_block.cab.3.conv.weight", (1, 1, 5))
            norm(p + ".norm2")
            lin(p + ".mlp.fc1", 2 * hid, C)
            add(p + ".mlp.dw.weight", (2 * hid, 1, 3, 3))
            add(p + ".mlp.dw.bias", (2 * hid,))
            lin(p + ".mlp.fc2", C, hid)
        p = f"layers.{g}.residual_group.overlap_attn"
        add(p + ".relative_position_bias_table", ((ws + wse - 1) ** 2, heads))
        if cfg["ocab_esc_enable"]:
            add(p + ".esc_plk", (cfg["ocab_esc_pdim"],) * 2 + (cfg["ocab_esc_kernel"],) * 2)
        norm(p + ".norm1")
        lin(p + ".q_proj", C, C, cfg["qkv_bias"])
        lin(p + ".kv_proj", 2 * C, C, cfg["qkv_bias"])
        lin(p + ".proj", C, C)
        norm(p + ".norm2")
        lin(p + ".mlp.0", hid, C)
        lin(p + ".mlp.2", C, hid)
        if cfg["ocab_esc_enable"]:
            esc_core(p + ".esc_core", cfg["ocab_esc_pdim"])
        if cfg["resi_connection"] == "1conv":
            conv(f"layers.{g}.conv", C, C, 3)
    norm("norm")
    if cfg["resi_connection"] == "1conv":
        conv("conv_after_body", C, C, 3)
    conv("conv_before_upsample.0", 64, C, 3)
    s = cfg["upscale"]
    if s & (s - 1) == 0:
        for i in range(int(math.log2(s))):
            conv(f"upsample.{2 * i}", 256, 64, 3)
    elif s == 3:
        conv("upsample.0", 576, 64, 3)
    conv("conv_last", cfg["in_chans"], 64, 3)
    return spec


def blank_state_dict(cfg: dict) -> SD:
    """Zero-valued fp32 tensors + the exact integer index buffers, with the reference's keys."""
    sd: SD = {}
    for k, (shape, dt) in state_dict_spec(cfg).items():
        sd[k] = torch.zeros(shape, dtype=dt)
    sd["relative_position_index_SA"] = rpi_sa(cfg["window_size"])
    sd["relative_position_index_OCA"] = rpi_oca(cfg["window_size"], cfg["overlap_ratio"])
    return sd


# --------------------------------------------------------------------------------------------
# HATX variant: hat/archs/hatx_arch.py — SGFN feed-forward in the HAB, OCAB with ceil padding,
# focus bias and top-k key pruning.  Everything else (CAB, ESC, RHAG, head / tail) is the HAT path above.
# --------------------------------------------------------------------------------------------
HATX_DEFAULT_CFG = dict(DEFAULT_CFG, hab_ffn_ratio=2.0, kv_topk_ratio=1.0, use_focus_bias=False)  # hatx_arch.py:714-751


def make_hatx_cfg(**kw) -> dict:
    cfg = dict(HATX_DEFAULT_CFG)
    for k, v in kw.items():
        if k in cfg:
            cfg[k] = v
    return cfg


def sgfn(m: Tensor, hw, sd: SD, p: str) -> Tensor:
    """SpatialGateDConvFFN.forward, hatx_arch.py:160-180: fc1, depthwise 3x3 on the FIRST half only, the second half gates
    (SiLU) and is also passed on: fc2(cat[dw(a) * silu(b), b])."""
    u = F.linear(m, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])
    c2 = u.shape[-1] // 2
    xa = F.conv2d(_tok2img(u[..., :c2], hw), sd[p + ".dw.weight"], sd[p + ".dw.bias"], padding=1, groups=c2)
    xb = u[..., c2:]
    return F.linear(torch.cat([_img2tok(xa) * F.silu(xb), xb], dim=-1), sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])


def hatx_hab(t: Tensor, hw, sd: SD, p: str, cfg: dict) -> Tensor:
    """HAB.forward of hatx_arch.py:232-257 (= hat_arch's with the SGFN)."""
    n = _ln(t, sd, p + ".norm1")
    n_img = _tok2img(n, hw)
    conv_x = _img2tok(cab(n_img, sd, p + ".conv_block"))
    attn_x = _img2tok(esc_conv_attn(n_img, sd[p + ".esc_attn.plk_filter"], sd, p + ".esc_attn.core", cfg["esc_pdim"]))
    t = t + attn_x + conv_x * cfg["conv_scale"]
    return t + sgfn(_ln(t, sd, p + ".norm2"), hw, sd, p + ".mlp")


def hatx_ocab_attention(q: Tensor, k: Tensor, v: Tensor, table: Tensor, rpi: Tensor, ws: int, wse: int, heads: int, scale: float,
                        sal: Optional[Tensor] = None, topk_ratio: float = 1.0, tie: str = "torch") -> Tensor:
    """Attention core of hatx_arch.py:386-465.  Differences from hat_arch's: the unfold pads ceil((wse - ws) / 2) (:303-305);
    `sal` (B,H,W) = the saliency map: tanh of its zero-padded key window is added to every logit of that key (:421-430);
    top-k pruning (:434-449) keeps the k_keep = max(1, int(ratio * Nk)) keys of a window with the largest score (the focus
    value, else ||k||_2 over all channels) and REPLACES the other keys' logits by -1e4 before the relative-position bias is
    added.  `torch.topk(sorted=False)` decides ties exactly as the reference does on this build of torch."""
    b, h, w, c = q.shape
    d = c // heads
    nh, nw = h // ws, w // ws
    pad = (wse - ws + 1) // 2

    def win_q(x):
        x = x.reshape(b, nh, ws, nw, ws, heads, d).permute(0, 1, 3, 5, 2, 4, 6)
        return x.reshape(b * nh * nw, heads, ws * ws, d)

    def unfold(x):  # (B,H,W,Cx) -> (B*nW, Cx, wse*wse), nn.Unfold(kernel wse, stride ws, padding pad)
        xp = F.pad(x, (0, 0, pad, pad, pad, pad))
        xs = xp.unfold(1, wse, ws).unfold(2, wse, ws)[:, :nh, :nw]  # (B,nh,nw,Cx,wse,wse)
        return xs.reshape(b * nh * nw, x.shape[-1], wse * wse)

    kw_, vw_ = unfold(k), unfold(v)                                                       # (B', C, Nk)
    qh = win_q(q) * scale
    kh = kw_.reshape(-1, heads, d, wse * wse).permute(0, 1, 3, 2)
    vh = vw_.reshape(-1, heads, d, wse * wse).permute(0, 1, 3, 2)
    attn = qh @ kh.transpose(-2, -1)
    b_, nk = attn.shape[0], wse * wse
    focus_k = None
    if sal is not None:
        focus_k = torch.tanh(unfold(sal[..., None])[:, 0])                                # (B', Nk)
        attn = attn + focus_k.view(b_, 1, 1, nk)
    if topk_ratio < 1.0:
        k_keep = max(1, int(topk_ratio * nk))
        score = focus_k if focus_k is not None else torch.linalg.vector_norm(kw_.transpose(1, 2), ord=2, dim=-1)
        if tie == "torch":       # the reference: which of several equal scores survive is whatever torch.topk returns
            idx = torch.topk(score, k_keep, dim=1, sorted=False).indices
        else:                    # "lowest_index": the MI355X kernel's documented rule (hat_ocab_keybias): among equal scores the
            idx = torch.argsort(-score, dim=1, stable=True)[:, :k_keep]   # key with the lower window index is kept
        keep = torch.zeros(b_, nk, dtype=torch.bool).scatter_(1, idx, True)
        attn = attn.masked_fill(~keep.view(b_, 1, 1, nk), -1e4)
    bias = table[rpi.reshape(-1)].reshape(ws * ws, wse * wse, heads).permute(2, 0, 1)
    attn = torch.softmax(attn + bias[None], dim=-1)
    o = attn @ vh
    o = o.reshape(b, nh, nw, heads, ws, ws, d).permute(0, 1, 4, 2, 5, 3, 6)
    return o.reshape(b, h, w, c)


def hatx_ocab(t: Tensor, hw, sd: SD, p: str, rpi: Tensor, cfg: dict, heads: int, tie: str = "torch") -> Tensor:
    """OCAB.forward, hatx_arch.py:367-465."""
    b, _, c = t.shape
    ws = cfg["window_size"]
    wse = int(ws * cfg["overlap_ratio"]) + ws
    scale = cfg["qk_scale"] or (c // heads) ** -0.5
    n = _ln(t, sd, p + ".norm1")
    x_img = n.reshape(b, hw[0], hw[1], c)
    y_chw = _tok2img(n, hw)
    if cfg["ocab_esc_enable"]:
        y_chw = esc_conv_attn(y_chw, sd[p + ".esc_plk"], sd, p + ".esc_core", cfg["ocab_esc_pdim"])
    y_img = y_chw.permute(0, 2, 3, 1)
    q = F.linear(x_img, sd[p + ".q_proj.weight"], sd.get(p + ".q_proj.bias"))
    kv = F.linear(y_img, sd[p + ".kv_proj.weight"], sd.get(p + ".kv_proj.bias"))
    k, v = kv.split(c, dim=-1)
    sal = None
    if cfg["use_focus_bias"]:  # focus_head: 1x1 C -> C/4, GELU, 1x1 -> 1                      :357-361, :423
        hcw = F.gelu(F.conv2d(y_chw, sd[p + ".focus_head.0.weight"], sd[p + ".focus_head.0.bias"]))
        sal = F.conv2d(hcw, sd[p + ".focus_head.2.weight"], sd[p + ".focus_head.2.bias"])[:, 0]
    o = hatx_ocab_attention(q, k, v, sd[p + ".relative_position_bias_table"], rpi, ws, wse, heads, scale, sal, cfg["kv_topk_ratio"], tie)
    t = F.linear(o.reshape(b, -1, c), sd[p + ".proj.weight"], sd[p + ".proj.bias"]) + t
    m = _ln(t, sd, p + ".norm2")
    m = F.linear(F.gelu(F.linear(m, sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"])), sd[p + ".mlp.2.weight"], sd[p + ".mlp.2.bias"])
    return t + m


def hatx_forward(x: Tensor, sd: SD, cfg: dict, tie: str = "torch") -> Tensor:
    """HATX.forward, hatx_arch.py:944-974: hat_forward with the HATX blocks."""
    if cfg["upsampler"] != "pixelshuffle":
        raise NotImplementedError("only the 'pixelshuffle' upsampler is on the hot path")
    mean = torch.tensor(RGB_MEAN, dtype=x.dtype).view(1, 3, 1, 1) if cfg["in_chans"] == 3 else torch.zeros(1, 1, 1, 1, dtype=x.dtype)
    r = cfg["img_range"]
    x = (x - mean) * r
    f0 = F.conv2d(x, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    hw = (f0.shape[2], f0.shape[3])
    ws = cfg["window_size"]
    if hw[0] % ws or hw[1] % ws:
        raise RuntimeError(f"input size {hw} is not a multiple of window_size {ws}")
    t = _img2tok(f0)
    if cfg["patch_norm"]:
        t = _ln(t, sd, "patch_embed.norm")
    if "absolute_pos_embed" in sd:
        t = t + sd["absolute_pos_embed"]
    rpi = sd["relative_position_index_OCA"]
    for g, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        t_in, p = t, f"layers.{g}"
        for i in range(depth):
            t = hatx_hab(t, hw, sd, f"{p}.residual_group.blocks.{i}", cfg)
        t = hatx_ocab(t, hw, sd, f"{p}.residual_group.overlap_attn", rpi, cfg, heads, tie)
        if cfg["resi_connection"] == "1conv":
            t = _img2tok(F.conv2d(_tok2img(t, hw), sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1))
        t = t + t_in
    f = _tok2img(_ln(t, sd, "norm"), hw)
    if cfg["resi_connection"] == "1conv":
        f = F.conv2d(f, sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1)
    f = f + f0
    f = F.leaky_relu(F.conv2d(f, sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"], padding=1), 0.01)
    f = upsample(f, sd, cfg["upscale"])
    return F.conv2d(f, sd["conv_last.weight"], sd["conv_last.bias"], padding=1) / r + mean


def hatx_state_dict_spec(cfg: dict) -> Dict[str, tuple]:
    """Key -> (shape, dtype) of `HATX(**cfg).state_dict()` (hatx_arch.py:714-874), in reference order.  The SGFN is built
    with `mlp_ratio` (AttenBlocks hands the OCAB ratio to the HABs, hatx_arch.py:513; `hab_ffn_ratio` is stored, not used)."""
    base = state_dict_spec(cfg)
    C = cfg["embed_dim"]
    hid = int(C * cfg["mlp_ratio"])
    spec: Dict[str, tuple] = {}
    for k, v in base.items():
        if ".mlp.fc1.weight" in k:
            v = ((hid, C), v[1])
        elif ".mlp.fc1.bias" in k:
            v = ((hid,), v[1])
        elif ".mlp.dw.weight" in k:
            v = ((hid // 2, 1, 3, 3), v[1])
        elif ".mlp.dw.bias" in k:
            v = ((hid // 2,), v[1])
        elif ".mlp.fc2.weight" in k:
            v = ((C, hid), v[1])
        spec[k] = v
        # focus_head sits behind esc_core (or behind mlp.2 when the OCAB has no ESC), hatx_arch.py:342-361
        if cfg["use_focus_bias"] and ".overlap_attn." in k:
            last = ".esc_core.aggr.bias" if cfg["ocab_esc_enable"] else ".mlp.2.bias"
            if k.endswith(last):
                p = k[: -len(last)]
                spec[p + ".focus_head.0.weight"] = ((C // 4, C, 1, 1), torch.float32)
                spec[p + ".focus_head.0.bias"] = ((C // 4,), torch.float32)
                spec[p + ".focus_head.2.weight"] = ((1, C // 4, 1, 1), torch.float32)
                spec[p + ".focus_head.2.bias"] = ((1,), torch.float32)
    return spec


def hatx_blank_state_dict(cfg: dict) -> SD:
    sd: SD = {k: torch.zeros(shape, dtype=dt) for k, (shape, dt) in hatx_state_dict_spec(cfg).items()}
    sd["relative_position_index_SA"] = rpi_sa(cfg["window_size"])
    sd["relative_position_index_OCA"] = rpi_oca(cfg["window_size"], cfg["overlap_ratio"])
    return sd


# --------------------------------------------------------------------------------------------
# caller harness restatement (hat/models/hat_model.py) — cannot be imported (needs basicsr.models)
# --------------------------------------------------------------------------------------------
def pre_process(lq: Tensor, window_size: int):
    """HATModel.pre_process, hat_model.py:16-26: reflect-pad bottom/right to a window multiple."""
    _, _, h, w = lq.shape
    ph = (window_size - h % window_size) % window_size
    pw = (window_size - w % window_size) % window_size
    return F.pad(lq, (0, pw, 0, ph), "reflect"), ph, pw


def tile_process(img: Tensor, net, scale: int, tile_size: int, tile_pad: int) -> Tensor:
    """HATModel.tile_process, hat_model.py:40-108 (a RuntimeError in a tile is raised, not swallowed)."""
    b, c, h, w = img.shape
    out = img.new_zeros((b, c, h * scale, w * scale))
    for y in range(math.ceil(h / tile_size)):
        for x in range(math.ceil(w / tile_size)):
            x0, y0 = x * tile_size, y * tile_size
            x1, y1 = min(x0 + tile_size, w), min(y0 + tile_size, h)
            x0p, x1p = max(x0 - tile_pad, 0), min(x1 + tile_pad, w)
            y0p, y1p = max(y0 - tile_pad, 0), min(y1 + tile_pad, h)
            o = net(img[:, :, y0p:y1p, x0p:x1p])
            ox, oy = (x0 - x0p) * scale, (y0 - y0p) * scale
            out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = \
                o[:, :, oy:oy + (y1 - y0) * scale, ox:ox + (x1 - x0) * scale]
    return out


def post_process(out: Tensor, ph: int, pw: int, scale: int) -> Tensor:
    """HATModel.post_process, hat_model.py:110-112."""
    _, _, h, w = out.shape
    return out[:, :, 0:h - ph * scale, 0:w - pw * scale]


# --------------------------------------------------------------------------------------------
# metric restatement (basicsr needs cv2 and cannot be imported; SURVEY §8c "parity unpinned")
# --------------------------------------------------------------------------------------------
def tensor2img_rgb(t: Tensor):
    """img_util.py:66-67,87-91 without the RGB->BGR flip: clamp [0,1], x255, round, uint8, HWC."""
    import numpy as np
    a = t.detach().float().cpu().clamp(0, 1).squeeze(0).permute(1, 2, 0).numpy()
    return np.round(a * 255.0).astype(np.uint8)


def psnr_y(img_rgb_u8, img2_rgb_u8, crop_border: int) -> float:
    """calculate_psnr(test_y_channel=True): metrics/psnr_ssim.py:11-48, metric_util.py:32-45,
    color_util.py:38-68 (BT.601 Y = (65.481 R + 128.553 G + 24.966 B) + 16 on [0,1] input)."""
    import numpy as np

    def to_y(a):  # to_y_channel: f32/255 -> f64 dot + 16 -> /255 -> f32 -> *255 (color_util.py:110-114,174-178)
        a = a.astype(np.float32) / np.float32(255.0)
        y = np.dot(a, [65.481, 128.553, 24.966]) + 16.0
        y = (y / 255.0).astype(np.float32) * np.float32(255.0)
        return y.astype(np.float64)

    a, b2 = to_y(img_rgb_u8), to_y(img2_rgb_u8)
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b2 = b2[crop_border:-crop_border, crop_border:-crop_border]
    mse = float(np.mean((a - b2) ** 2))
    return float("inf") if mse == 0 else 10.0 * math.log10(255.0 * 255.0 / mse)


def ssim_y(img_rgb_u8, img2_rgb_u8, crop_border: int) -> float:
    """calculate_ssim(test_y_channel=True): metrics/psnr_ssim.py:86-125 with _ssim :170-198 — an 11x11 Gaussian window
    (sigma 1.5, cv2.getGaussianKernel) correlated in VALID mode (filter2D(...)[5:-5, 5:-5]), c1 = (0.01*255)^2,
    c2 = (0.03*255)^2, mean of the SSIM map.  Independent of the product code: full 2-D window via scipy."""
    import numpy as np
    from scipy.signal import correlate2d

    def to_y(a):
        a = a.astype(np.float32) / np.float32(255.0)
        y = np.dot(a, [65.481, 128.553, 24.966]) + 16.0
        return ((y / 255.0).astype(np.float32) * np.float32(255.0)).astype(np.float64)

    a, b2 = to_y(img_rgb_u8), to_y(img2_rgb_u8)
    if crop_border:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b2 = b2[crop_border:-crop_border, crop_border:-crop_border]
    x = np.arange(11) - 5.0
    k = np.exp(-(x ** 2) / (2 * 1.5 ** 2))
    k /= k.sum()
    win = np.outer(k, k)
    f = lambda z: correlate2d(z, win, mode="valid")
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = f(a), f(b2)
    s1, s2, s12 = f(a * a) - mu1 ** 2, f(b2 * b2) - mu2 ** 2, f(a * b2) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))
    return float(m.mean())


def psnr_float(a: Tensor, b: Tensor, peak: float = 1.0) -> float:
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(peak * peak / mse)
