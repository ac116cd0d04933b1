"""HATEngine — host-side orchestration of the HIP kernels for one HAT network.

Holds the packed weights (per storage dtype) and a per-shape workspace in HBM, and enqueues the
kernel sequence of `HAT.forward` (reference: hat/archs/hat_arch.py:848-859 and the blocks it
calls; op-by-op map in SURVEY.md App. A) on the current stream.  Nothing here computes on the
CPU or through torch ops: torch only allocates device buffers and packs weights at load time.

Data layout in HBM (B = batch, N = H*W pixels, C = embed_dim):
  residual stream      tA, tB : fp32 (B, N, C)  two buffers: tA = RHAG input/output, tB = working copy
  shallow feature      f0     : fp32 (B, N, C)
  MFMA operands        n, c2, q, ao : T (B, N, ld(C));  u : T (B, N, ld(4C));  g, kv : T (B, N, ld(2C))
                       c1 : T (B, N, ld(C/cr));  y16 : T (B, N, pdim)       ld(x) = round_up(x, 8)
  upsampler            fb : T (B, N, 64);  up_i : T (B, r^2i N, 64);  output y : fp32 (B, 3, sH, sW) NCHW
T = bf16 (performance path) or fp32 (exact-fp32 parity path).
"""
from __future__ import annotations

import collections
import math
import os
import threading
from typing import Dict, Optional

import torch

from . import ops
from .ops import (ACT_GELU, ACT_LRELU, ACT_NONE, O_NCHW_F32, O_NHWC_F32, O_NHWC_T, O_PIXSHUF_T, X_NCHW_F32_MEAN,
                  X_NHWC_F32, X_NHWC_T)

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # hat_arch.py:659


def _r8(x: int) -> int:
    return (x + 7) // 8 * 8


def _r4(x: int) -> int:
    return (x + 3) // 4 * 4


# HAT_EMU_T16=fp16|bf16: round the fp32 residual stream to that type after every HAB tail and group conv (a torch copy:
# slower, NOT a product path) — the parity cost of a 16-bit residual stream, measured before anyone builds it.
_EMU_T16 = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(os.environ.get("HAT_EMU_T16", ""))


class _ESC:
    """Packed parameters of one ConvAttnWrapper (esc_arch.py:136-145) + its large-kernel filter."""

    def __init__(self, sd, core: str, plk_key: str, pdim: int, ksize: int, C: int, dtype: int, dev):
        f32 = dict(dtype=torch.float32, device=dev)
        self.pdim, self.ksize = pdim, ksize
        self.w1 = sd[core + ".plk.dwc_proj.1.weight"].detach().reshape(pdim // 2, pdim).to(**f32).contiguous()
        self.b1 = sd[core + ".plk.dwc_proj.1.bias"].detach().to(**f32).contiguous()
        self.w2 = sd[core + ".plk.dwc_proj.3.weight"].detach().reshape(pdim * 9, pdim // 2).to(**f32).contiguous()
        self.b2 = sd[core + ".plk.dwc_proj.3.bias"].detach().to(**f32).contiguous()
        # static large-kernel filter packed in fp32 [16 or 32][Kpad]; hat_esc_weights adds the dynamic
        # depthwise 3x3 on the diagonal of the central taps and casts to T per forward
        lk = ops.pack_conv_weight(sd[plk_key], None, ops.HAT_F32, dev, nt=1)
        kc = ops.KC[dtype] * 3  # nt == 1 chunk length (hat_conv.hip)
        self.kpad = -(-(ksize * ksize * _r8(pdim)) // kc) * kc
        self.npad = 16 if pdim <= 16 else 32   # weight rows / output channels of the conv (one or two 16-row slices)
        plk = torch.zeros(self.npad, self.kpad, **f32)
        plk[:lk.w.shape[0], :min(self.kpad, lk.kpad)] = lk.w[:self.npad, :min(self.kpad, lk.kpad)]
        self.plk = plk.contiguous()
        self.zero_bias = torch.zeros(self.npad, **f32)
        self.aggr = None  # packed by the engine (hat_linear when the shape is instantiated)
        self.aggr_keys = (core + ".aggr.weight", core + ".aggr.bias")


class HATEngine:
    def __init__(self, cfg: dict, state_dict: Dict[str, torch.Tensor], device, dtype: str = "bf16"):
        if cfg.get("upsampler") != "pixelshuffle":
            raise NotImplementedError("only upsampler='pixelshuffle' is on the hot path (all shipped test YAMLs)")
        if cfg.get("resi_connection", "1conv") not in ("1conv", "identity"):
            raise ValueError(f"Unknown resi_connection: {cfg.get('resi_connection')}")   # hat_arch.py:547-548, :750-751
        self.cfg = cfg
        self.identity = cfg.get("resi_connection", "1conv") == "identity"
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("HATEngine needs a GPU device: there is no CPU path")
        self.dtype = ops.DTYPE_CODE[dtype]
        self.tdt = ops.TORCH_DTYPE[self.dtype]
        self.C = cfg["embed_dim"]
        self.ws = cfg["window_size"]
        self.wse = self.ws + int(cfg["overlap_ratio"] * self.ws)
        self.scale = cfg["upscale"]
        # per-shape workspaces, least recently used first; tiled inference alternates between a few shapes (interior /
        # edge / corner tiles), so a small LRU allocates and zero-fills each of them once
        self._ws_cache = collections.OrderedDict()
        self._ws_max = int(os.environ.get("HAT_WS_CACHE", "12"))
        self._ws_max_bytes = int(float(os.environ.get("HAT_WS_CACHE_GIB", "96")) * 2 ** 30)
        self.use_n16 = os.environ.get("HAT_NO_N16") != "1"
        # FP16 residual rows between the fused HAB tails of a residual group (bf16 path, embed_dim 144; HAT_NO_T16=1: fp32 everywhere)
        self.t16 = os.environ.get("HAT_NO_T16") != "1" and _EMU_T16 is None and self.dtype == ops.HAT_BF16 and self.C == 144
        self._lock = threading.Lock()   # one forward at a time per engine: the workspace and side stream are shared state
        ops._lib.load()
        # fused FFN kernel (hat_ffn) for the shapes it is instantiated for; HAT_NO_FUSED_FFN=1 forces the
        # unfused kernel sequence (fc1 -> dw+gate -> fc2), kept for A/B validation of the fusion
        self.hatx = cfg.get("variant", "hat") == "hatx"
        if self.hatx:
            # HATX (hatx_arch.py): the SGFN runs as fc1 -> hat_sgfn_gate -> fc2; odd window overlaps are padded with
            # ceil((wse - ws) / 2) (hatx_arch.py:303-305) and run on the generic attention kernel (key windows 25 and 13)
            pass
        self.topk = float(cfg.get("kv_topk_ratio", 1.0)) if self.hatx else 1.0
        self.focus = bool(cfg.get("use_focus_bias", False)) if self.hatx else False
        self.fuse_ffn = not self.hatx and ops.ffn_supported(self.C) and os.environ.get("HAT_NO_FUSED_FFN", "0") != "1"
        self._pack(state_dict)

    # ------------------------------------------------------------------------------------------
    def _lin(self, sd, wkey, bkey, scale=1.0):
        """Pack a pointwise layer for hat_linear when its shape is instantiated, else for hat_conv (ksize 1)."""
        w = sd[wkey]
        b = sd.get(bkey) if bkey else None
        o, i = w.shape[0], w.reshape(w.shape[0], -1).shape[1]

        def pack(wm, bias):
            if ops.linear_supported(wm.shape[0], wm.shape[1], self.dtype):
                pw = ops.pack_linear_weight(wm, bias, self.dtype, self.dev, scale=scale)
                pw.frag = True
                return pw
            pw = ops.pack_conv_weight(wm, bias, self.dtype, self.dev, scale=scale)
            pw.frag = False
            return pw
        w2 = w.reshape(o, -1)
        if w2.shape[1] == i and i > 512 and i % 16 == 0 and not ops.linear_supported(o, i, self.dtype):
            # K too wide for any tiling (HATX's SGFN fc2 at embed_dim 180: 720 -> 180; its fp32 rows do not fit a hat_conv tile):
            # two launches over the two halves of K, the second adding onto the first's fp32 result
            h1, h2 = pack(w2[:, :i // 2].contiguous(), b), pack(w2[:, i // 2:].contiguous(), None)
            h1.ksplit = h2
            return h1
        return pack(w, b)

    def _c3(self, sd, wkey, bkey):
        """Pack a CAB 3x3 conv for hat_conv3x3_small (weights resident in LDS) when instantiated, else for hat_conv."""
        w = sd[wkey]
        # squeeze conv (C -> C/cr, one n-tile): hat_conv's LDS-staged haloed tile beats gathering 9 neighbours per
        # k-step through L1 (0.18 vs 0.23 ms at 720p); expand conv (C/cr -> C, K = 72): the gathers are cheap, tap3 wins
        if w.shape[0] > 16 and ops.conv3x3_small_supported(w.shape[0], w.shape[1], self.dtype):
            return ops.pack_linear_weight(w, sd[bkey], self.dtype, self.dev)
        return ops.pack_conv_weight(w, sd[bkey], self.dtype, self.dev)

    def _run_lin(self, pw, x, out, **kw):
        h2 = getattr(pw, "ksplit", None)
        if h2 is not None:   # out = W[:, :K/2] x[:, :K/2] + b (+ r1), then out += W[:, K/2:] x[:, K/2:]   (fp32 output only)
            if kw.get("out_mode") != O_NHWC_F32 or kw.get("act", ACT_NONE) != ACT_NONE:
                raise NotImplementedError("a K-split linear accumulates through its fp32 output")
            (ops.linear if pw.frag else ops.conv)(pw, x, out, **kw)
            (ops.linear if h2.frag else ops.conv)(h2, x.reshape(-1)[pw.cin:], out, **dict(kw, r1=out, ldr1=kw["ldo"]))
            return
        (ops.linear if pw.frag else ops.conv)(pw, x, out, **kw)

    def _pack(self, sd):
        cfg, dt, dev, C = self.cfg, self.dtype, self.dev, self.C
        f32 = dict(dtype=torch.float32, device=dev)
        P = lambda w, b=None, **kw: ops.pack_conv_weight(sd[w], None if b is None else sd[b], dt, dev, **kw)
        vec = lambda k: sd[k].detach().to(**f32).contiguous()
        self.conv_first = P("conv_first.weight", "conv_first.bias")
        self.pe_norm = (vec("patch_embed.norm.weight"), vec("patch_embed.norm.bias")) if cfg.get("patch_norm", True) else None
        self.ape = vec("absolute_pos_embed").reshape(-1) if cfg.get("ape", False) else None   # (num_patches * C,)  :699-702
        self.layers = []
        ws, wse = self.ws, self.wse
        M = ws + wse - 1
        shift = (ws - wse + 1 - (ws - 1)) * (M + 1)  # rotated index i' -> reference index rpi = i' + shift (may be < 0)
        rot = (torch.arange(M * M) + shift) % (M * M)  # negative-index wraparound of hat_arch.py:378
        for g, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
            L = {"heads": heads, "habs": []}
            for i in range(depth):
                p = f"layers.{g}.residual_group.blocks.{i}"
                hb = {
                    "n1": (vec(p + ".norm1.weight"), vec(p + ".norm1.bias")),
                    "n2": (vec(p + ".norm2.weight"), vec(p + ".norm2.bias")),
                    "esc": _ESC(sd, p + ".esc_attn.core", p + ".esc_attn.plk_filter", cfg["esc_pdim"], cfg["esc_kernel"], C, dt, dev),
                    "cab0": self._c3(sd, p + ".conv_block.cab.0.weight", p + ".conv_block.cab.0.bias"),
                    "cab2": self._c3(sd, p + ".conv_block.cab.2.weight", p + ".conv_block.cab.2.bias"),
                    "eca_w": vec(p + ".conv_block.cab.3.conv.weight").reshape(-1),
                }
                hb["esc"].aggr = self._lin(sd, *hb["esc"].aggr_keys)
                w2raw = sd[p + ".conv_block.cab.2.weight"]
                # CAB expand conv + ECA folded into the aggregation (hat_cab_fold / hat_aggr_cab): squeeze width <= 8 only
                if (hb["esc"].aggr.frag and not hb["cab0"].frag and ops.aggr_cab_supported(C, w2raw.shape[1], dt)
                        and not os.environ.get("HAT_NO_CAB_FOLD")):
                    hb["fold"] = {"w2": w2raw.detach().to(**f32).contiguous(), "b2": vec(p + ".conv_block.cab.2.bias"),
                                  "ba": vec(hb["esc"].aggr_keys[1]), "w2f": ops.pack_cab_w2f(w2raw, dev)}
                    # squeeze conv on the row-sweep kernel (no LDS operand traffic) where it is instantiated
                    if ops.cab_squeeze_supported(C, w2raw.shape[1], 16, dt) and not os.environ.get("HAT_NO_CAB_SWEEP"):
                        hb["fold"]["sq"] = ops.pack_cab_squeeze(sd[p + ".conv_block.cab.0.weight"], sd[p + ".conv_block.cab.0.bias"], dev)
                if not self.fuse_ffn:
                    hb["fc1"] = self._lin(sd, p + ".mlp.fc1.weight", p + ".mlp.fc1.bias")
                    hb["fc2"] = self._lin(sd, p + ".mlp.fc2.weight", p + ".mlp.fc2.bias")
                hid2 = sd[p + ".mlp.dw.weight"].shape[0]     # (HATX: the first half of the SGFN's hidden width)
                hb["dw_w"] = sd[p + ".mlp.dw.weight"].detach().to(**f32).reshape(hid2, 9).t().contiguous()  # [9][2*hid]
                hb["dw_b"] = vec(p + ".mlp.dw.bias")
                if self.fuse_ffn:
                    fw = [sd[p + k] for k in (".mlp.fc1.weight", ".mlp.fc1.bias", ".mlp.dw.weight", ".mlp.dw.bias",
                                              ".mlp.fc2.weight", ".mlp.fc2.bias")]
                    # hat_ffn2 (fp16 hidden tensor, depthwise conv on the packed-fp16 VALU) where it is built; HAT_FFN_V1=1
                    # keeps the first-generation kernel for A/B runs
                    # ... unless this block's weights could drive its FP16 hidden tensor past the FP16 range for SOME input
                    # (pack-time worst-case bound, ops.ffn_fp16_range_bound): then the bf16-hidden kernel stays
                    fp16_ok = ops.ffn_fp16_range_bound(fw[0], fw[1], fw[2], fw[3], *hb["n2"]) < ops.FP16_SAFE
                    if not fp16_ok:
                        self.fp16_fallbacks = getattr(self, "fp16_fallbacks", 0) + 1
                    if ops.ffn2_supported(C, hid2 // 2, dt) and os.environ.get("HAT_FFN_V1") != "1" and fp16_ok:
                        hb["ffn"] = ops.pack_ffn2(*fw, dev)
                    else:
                        hb["ffn"] = ops.pack_ffn(*fw, dt, dev)
                    hb["tail"] = ("fold" in hb and hb["esc"].pdim == 16 and ops.hab_tail_supported(hb["ffn"], hb["esc"].aggr, w2raw.shape[1], dt)
                                  and os.environ.get("HAT_NO_HAB_TAIL") != "1")
                    # third-generation tail (activation-stationary fc1, weights shared through LDS): its own packing;
                    # HAT_TAIL_V2=1 keeps hat_hab_tail for A/B runs
                    if hb["tail"] and hb["ffn"].khalf == "v2" and os.environ.get("HAT_TAIL_V2") != "1":
                        hb["ffn3"] = ops.pack_ffn3(*fw, *hb["n2"], dev)
                    # embed_dim 180 (HAT / HAT-L): hat_hab_tail3 with the CAB's c2 as a map (their squeeze is 60 wide: no fold)
                    if ("fold" not in hb and C == 180 and fp16_ok and ops.tail3_supported(C, hid2 // 2, dt) and hb["esc"].pdim == 16
                            and hb["esc"].aggr.frag and os.environ.get("HAT_NO_HAB_TAIL") != "1" and os.environ.get("HAT_TAIL_V2") != "1"):
                        f3 = ops.pack_ffn3(*fw, *hb["n2"], dev)
                        if ops.hab_tail_supported(f3, hb["esc"].aggr, w2raw.shape[1], dt):
                            hb["ffn3"], hb["tail180"] = f3, True
                            b256 = torch.zeros(256, **f32)
                            b256[:C] = vec(hb["esc"].aggr_keys[1])
                            hb["bias256"] = b256
                L["habs"].append(hb)
            p = f"layers.{g}.residual_group.overlap_attn"
            d = C // heads
            qscale = cfg.get("qk_scale") or d ** -0.5
            # the tuned OCAB kernel of the embed_dim-144 models takes its queries in log2 units: fold log2(e) into the q
            # projection too (before the weights are rounded), not into the kernel (HAT_NO_ATTN_LOG2=1: the round-2 kernel)
            qlog2 = (ops.ocab_attention_log2_supported(C, heads, self.ws, self.wse, dt) and not (self.focus or self.topk < 1.0)
                     and os.environ.get("HAT_NO_ATTN_LOG2") != "1" and d % 2 == 0 and _r8(C) == C)
            if qlog2:
                qscale = qscale * ops.LOG2E
            table = sd[p + ".relative_position_bias_table"].detach().to(torch.float32).cpu()  # (M*M, heads)
            oc = {
                "n1": (vec(p + ".norm1.weight"), vec(p + ".norm1.bias")),
                "n2": (vec(p + ".norm2.weight"), vec(p + ".norm2.bias")),
                # q * scale (hat_arch.py:375) is folded into the projection
                "q": self._lin(sd, p + ".q_proj.weight", p + ".q_proj.bias", scale=qscale),
                "kv": self._lin(sd, p + ".kv_proj.weight", p + ".kv_proj.bias"),
                "proj": self._lin(sd, p + ".proj.weight", p + ".proj.bias"),
                "mlp0": self._lin(sd, p + ".mlp.0.weight", p + ".mlp.0.bias"),
                "mlp2": self._lin(sd, p + ".mlp.2.weight", p + ".mlp.2.bias"),
                # both MLP layers in one launch where hat_ocab_mlp is built (HAT_NO_OCAB_MLP=1: the two hat_linear launches)
                "mlpf": (ops.pack_ocab_mlp(sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"], sd[p + ".mlp.2.weight"], sd[p + ".mlp.2.bias"], dev)
                         if ops.ocab_mlp_supported(C, sd[p + ".mlp.0.weight"].shape[0], dt) and _r8(C) == C
                         and not os.environ.get("HAT_NO_OCAB_MLP") else None),
                "bias_rot": table[rot].t().contiguous().to(dev),  # [heads][M*M]
                "qlog2": qlog2,
            }
            # q and kv projections in one launch (both read LayerNorm1's output) where hat_ocab_qkv is built and the OCAB has
            # no ESC on its key / value path (HAT_NO_OCAB_QKV=1: two hat_linear launches on two streams)
            oc["qkvf"] = (ops.pack_ocab_qkv(sd[p + ".q_proj.weight"], sd.get(p + ".q_proj.bias"), sd[p + ".kv_proj.weight"],
                                            sd.get(p + ".kv_proj.bias"), qscale, dev)
                          if C == 144 and dt == ops.HAT_BF16 and not cfg.get("ocab_esc_enable", False)
                          and not os.environ.get("HAT_NO_OCAB_QKV") else None)
            if cfg.get("ocab_esc_enable", False):
                oc["esc"] = _ESC(sd, p + ".esc_core", p + ".esc_plk", cfg["ocab_esc_pdim"], cfg["ocab_esc_kernel"], C, dt, dev)
                oc["esc"].aggr = self._lin(sd, *oc["esc"].aggr_keys)
            if self.focus:  # saliency head: 1x1 C -> C/4, GELU, 1x1 -> 1                     hatx_arch.py:357-361
                oc["fh0"] = ops.pack_conv_weight(sd[p + ".focus_head.0.weight"], sd[p + ".focus_head.0.bias"], dt, dev)
                oc["fh2"] = ops.pack_conv_weight(sd[p + ".focus_head.2.weight"], sd[p + ".focus_head.2.bias"], dt, dev)
            L["ocab"] = oc
            L["conv"] = None if self.identity else P(f"layers.{g}.conv.weight", f"layers.{g}.conv.bias")
            self.layers.append(L)
        if any(e.npad > 16 for L in self.layers for e in [hb["esc"] for hb in L["habs"]] + ([L["ocab"]["esc"]] if "esc" in L["ocab"] else [])):
            for L in self.layers:       # the fused tail reads the ESC conv output as 16-channel rows
                for hb in L["habs"]:
                    hb["tail"] = False
        self.norm = (vec("norm.weight"), vec("norm.bias"))
        self.conv_after_body = None if self.identity else P("conv_after_body.weight", "conv_after_body.bias")
        self.conv_before_up = P("conv_before_upsample.0.weight", "conv_before_upsample.0.bias")
        self.ups = []
        s = self.scale
        if s & (s - 1) == 0:
            for i in range(int(math.log2(s))):
                self.ups.append((self._pack_ps(sd, f"upsample.{2 * i}", 2), 2))
        elif s == 3:
            self.ups.append((self._pack_ps(sd, "upsample.0", 3), 3))
        else:
            raise ValueError(f"scale {s} is not supported. Supported scales: 2^n and 3.")
        self.conv_last = P("conv_last.weight", "conv_last.bias")
        wl = sd["conv_last.weight"]
        self.conv_last_sweep = None   # row-sweep kernel (no LDS) for the 64 -> 3 conv at output resolution
        if ops.conv3x3_to_planes_supported(wl.shape[0], wl.shape[1], 16, dt) and not os.environ.get("HAT_NO_CAB_SWEEP"):
            self.conv_last_sweep = ops.pack_cab_squeeze(wl, sd["conv_last.bias"], dev) + (wl.shape[0],)

    def _pack_ps(self, sd, key, r):
        """Conv feeding nn.PixelShuffle(r) (hat_arch.py:598-602): output channel c*r^2 + i*r + j is stored
        as packed row (i*r + j)*Cps + c so one lane's 4 consecutive channels land in ONE output pixel."""
        w = sd[key + ".weight"]
        o = w.shape[0]
        cps = o // (r * r)
        n = torch.arange(o)
        perm = (n % cps) * (r * r) + n // cps  # packed row n' = ij*cps + c  <-  original channel c*r^2 + ij
        return ops.pack_conv_weight(w, sd[key + ".bias"], self.dtype, self.dev, out_perm=perm)

    # ------------------------------------------------------------------------------------------
    def _workspace(self, B, H, W, tag=None):
        """tag: a row band of a sharded frame gets a workspace of its own (two bands of one shape must not share buffers)."""
        key = (B, H, W) if tag is None else (B, H, W, tag)
        ws = self._ws_cache.get(key)
        if ws is not None:
            self._ws_cache.move_to_end(key)
            return ws
        C, dev, T = self.C, self.dev, self.tdt
        N = H * W
        mid = self.layers[0]["habs"][0]["cab0"].nout if self.layers and self.layers[0]["habs"] else 8
        hid2 = 2 * int(C * self.cfg["mlp_ratio"])  # fc1 width of GatedDconvFFN (hat_arch.py:99-100)
        z = lambda *shape, dtype=T: torch.zeros(*shape, dtype=dtype, device=dev)
        f = torch.float32
        escs = [hb["esc"] for L in self.layers for hb in L["habs"]] + [L["ocab"]["esc"] for L in self.layers if "esc" in L["ocab"]]
        yw = max([16] + [e.npad for e in escs])      # channels of the ESC conv output / floats per GAP partial block
        w = {
            "f0": z(B, N, C, dtype=f), "tA": z(B, N, C, dtype=f), "tB": z(B, N, C, dtype=f), "tC": z(B, N, C, dtype=f),
            # the residual stream BETWEEN the fused tails of a group as FP16 rows (hat_hab_tail3 reads and writes it in both types)
            "hB": (z(B, N, C, dtype=torch.float16) if self.t16 else None), "hC": (z(B, N, C, dtype=torch.float16) if self.t16 else None),
            "n": z(B, N, _r8(C)), "n2b": z(B, N, _r8(C)), "c1": z(B, N, _r8(mid)), "c2": z(B, N, _r8(C)), "m2": z(B, N, ops.ffn_m_ld(C)),
            "y16": z(B, N, yw), "n16": z(B, N, 16), "u": z(B, N, _r8(max(hid2, 2 * C))), "g": z(B, N, _r8(max(hid2 // 2, 2 * C))),
            "q": z(B, N, _r8(C)), "kv": z(B, N, _r8(2 * C)), "ao": z(B, N, _r8(C)),
            "fb": z(B, N, 64),
            "gap": z(B, max(ops.layernorm_blocks(), -(-H // 4) * -(-W // 16)), yw, dtype=f),
            "scale": z(B, 256, dtype=f), "eca_tmp": z(B, 32, 256, dtype=f),
        }
        if any(L["ocab"].get("qkvf") is not None for L in self.layers):
            w["qkv"] = z(B, N, 432)      # [q | k | v] rows of the fused projection
        esc0 = self.layers[0]["habs"][0]["esc"] if self.layers and self.layers[0]["habs"] else None
        kpad = max([esc0.kpad if esc0 else 0] + [L["ocab"]["esc"].kpad for L in self.layers if "esc" in L["ocab"]])
        w["weff"] = z(B, yw, max(kpad, 64))
        if any("esc" in L["ocab"] for L in self.layers):
            w["yesc"] = z(B, N, _r8(C))
        if self.focus:
            w["fh"], w["sal"] = z(B, N, _r8(C // 4)), z(B, N, 8, dtype=f)   # (the saliency map stays fp32: the keys are ranked on it)
        if self.focus or self.topk < 1.0:
            w["kb"] = z(B, (H // self.ws) * (W // self.ws), -(-(self.wse * self.wse) // 16) * 16, dtype=f)   # rows of whole key tiles
        cab2 = self.layers[0]["habs"][0]["cab2"] if self.layers and self.layers[0]["habs"] else None
        if cab2 is not None:
            tiles = ops.conv3x3_small_groups(cab2, B, H, W, self.dtype) if cab2.frag else ops.conv_tiles(cab2, H, W, self.dtype)
            w["tiles"] = tiles
            w["colsum"] = z(B, tiles, cab2.npad, dtype=f)
        hab0 = self.layers[0]["habs"][0] if self.layers and self.layers[0]["habs"] else None
        if hab0 is not None and "fold" in hab0:
            cab0 = hab0["cab0"]
            w["sweep"] = "sq" in hab0["fold"] and W % 16 == 0 and cab0.npad == 16
            w["tiles1"] = ops.cab_squeeze_units(H, W) if w["sweep"] else ops.conv_tiles(cab0, H, W, self.dtype)
            w["colsum1"] = z(B, w["tiles1"], cab0.npad, dtype=f)
            w["wf"] = z(B, hab0["esc"].aggr.nt * 3 * 512)
            w["bias_b"] = z(B, hab0["esc"].aggr.npad, dtype=f)
        if tag is not None:   # pooled sums of this band's own rows (local) and of the whole frame (global): SURVEY §8 f4
            w["gstat_l"], w["gstat_g"] = z(B, yw, dtype=f), z(B, yw, dtype=f)
            w["cstat_l"], w["cstat_g"] = z(B, 72, dtype=f), z(B, 72, dtype=f)
            w["estat_l"], w["estat_g"] = z(B, 256, dtype=f), z(B, 256, dtype=f)
            w["rs_tmp"], w["rs_cnt"] = z(B, 64, 256, dtype=f), torch.zeros(B, dtype=torch.int32, device=dev)
        h, wd = H, W
        w["ups"] = []
        for _, r in self.ups:
            h, wd = h * r, wd * r
            w["ups"].append(z(B, h * wd, 64))
        w["bytes"] = sum(t.numel() * t.element_size() for v in w.values() for t in (v if isinstance(v, list) else [v])
                         if isinstance(t, torch.Tensor))
        self._ws_cache[key] = w
        self.ws_allocations = getattr(self, "ws_allocations", 0) + 1
        while len(self._ws_cache) > 1 and (len(self._ws_cache) > self._ws_max
                                           or sum(v["bytes"] for v in self._ws_cache.values()) > self._ws_max_bytes):
            self._ws_cache.popitem(last=False)   # graphs captured on an evicted workspace keep their own reference
        return w

    # ------------------------------------------------------------------------------------------
    def _esc_w(self, esc: _ESC, w, B, H, W, nblk, gap=None):
        """The per-sample 13x13 weights of the ESC conv (static filter + dynamic depthwise kernel) -> w['weff'].
        gap: one block of frame-wide sums over H * W pixels (band-sharded frames) instead of w['gap']'s nblk partial blocks."""
        ops.esc_weights(w["gap"] if gap is None else gap, nblk, H * W, esc.w1, esc.b1, esc.w2, esc.b2, esc.plk, w["weff"], B=B,
                        pdim=esc.pdim, ksize=esc.ksize, kpad=esc.kpad, dtype=self.dtype)

    def _esc_conv(self, esc: _ESC, w, n, B, H, W, n16=None):
        """n16: a compact (B,N,16) copy of n's first 16 channels when the producer wrote one (the fused HAB tail does)."""
        if ops.esc_conv13_supported(esc.pdim, esc.ksize, self.dtype) and os.environ.get("HAT_NO_ESC13") != "1":
            src, ldx = (n16, 16) if n16 is not None else (n, _r8(self.C))
            ops.esc_conv13(src, w["weff"], w["y16"], B=B, H=H, W=W, ldx=ldx, kpad=esc.kpad, dtype=self.dtype)
            return
        pw = ops.PackedConv(w["weff"], esc.zero_bias, esc.ksize, esc.pdim, esc.kpad, 1, esc.npad // 16, esc.pdim,
                            w_bstride=esc.npad * esc.kpad)
        ops.conv(pw, n, w["y16"], B=B, H=H, W=W, dtype=self.dtype, ldx=_r8(self.C), ldo=w["y16"].shape[2], n_store=_r4(esc.pdim))

    def _esc_lk(self, esc: _ESC, w, n, B, H, W, nblk):
        """ESC large-kernel + dynamic depthwise conv on the first pdim channels of `n` -> w['y16']."""
        self._esc_w(esc, w, B, H, W, nblk)
        self._esc_conv(esc, w, n, B, H, W)

    def _side_stream(self):
        if os.environ.get("HAT_ONE_STREAM") == "1":
            return torch.cuda.current_stream(self.dev)
        if getattr(self, "_s1", None) is None:
            self._s1 = torch.cuda.Stream(device=self.dev)
        return self._s1

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("HAT forward needs a device tensor: the HIP path is the only path")
        if x.device != self.dev:
            raise RuntimeError(f"input is on {x.device} but this engine's weights and workspace live on {self.dev}")
        # kernels are enqueued on the CURRENT stream of the engine's device: make that device current for the launches
        # (the C side never switches devices), and serialise callers: workspace and side stream are per-engine state
        with self._lock, torch.cuda.device(self.dev):
            return self._forward(x)

    def ocab_only(self, t: torch.Tensor, group: int, H: int, W: int) -> torch.Tensor:
        """Run only the OCAB of residual group `group` on tokens t (B, H*W, C) fp32 -> (B, H*W, C) fp32 (used by the tests)."""
        x = torch.zeros(t.shape[0], self.cfg["in_chans"], H, W, device=self.dev)
        with self._lock, torch.cuda.device(self.dev):
            return self._forward(x, only_ocab=(t.to(self.dev, torch.float32).contiguous(), group))

    def _forward(self, x: torch.Tensor, only_ocab=None) -> torch.Tensor:
        gen = self._forward_gen(x, only_ocab=only_ocab)
        try:
            req = next(gen)
        except StopIteration as done:
            return done.value
        raise RuntimeError(f"the unsharded forward must not reach an exchange point (got {req[0]!r})")

    def _forward_gen(self, x: torch.Tensor, only_ocab=None, band=None):
        """The forward as a generator.  Unsharded (band=None) it never yields and returns the output.  For one ROW BAND of a
        sharded frame (band: tile_parallel.Band, x = the band's rows + ghost rows of the LR frame) it yields at
        every point where bands must exchange: ("halo", [(tensor, depth) ...]) — refresh `depth` ghost rows above and below
        from the neighbours that own them — and ("reduce", local, glob, n) — glob[:, :n] = sum over bands of local[:, :n] (the
        global average pools of ECA, hat_arch.py:69-73, and of the ESC dynamic kernel, esc_arch.py:96,121) — and returns the
        band's output rows (ghost rows included; the driver keeps the owned ones)."""
        if x.dim() != 4 or x.shape[1] != self.cfg["in_chans"]:
            raise RuntimeError(f"expected (B,{self.cfg['in_chans']},H,W), got {tuple(x.shape)}")
        B, _, H, W = x.shape
        ws = self.ws
        if H % ws or W % ws:  # the reference raises from calculate_mask's view (hat_arch.py:815), SURVEY F4
            raise RuntimeError(f"input size ({H},{W}) is not a multiple of window_size {ws}")
        x = x.to(torch.float32).contiguous()
        cfg, dt, C = self.cfg, self.dtype, self.C
        N, ldc = H * W, _r8(C)
        bd = band
        w = dict(self._workspace(B, H, W, tag=(None if bd is None else ("band", bd.idx, bd.n))))   # (a shallow copy: the forward swaps the two LayerNorm-output buffers locally)
        if bd is not None:
            if (self.hatx or self.identity or self.ape is not None or self.pe_norm is None or self.conv_after_body is None
                    or any("esc" in L["ocab"] for L in self.layers) or (bd.e1 - bd.e0) != H):
                raise NotImplementedError("band-sharded forward: plain HAT with resi_connection='1conv', patch_norm, no ape, no OCAB-ESC")
            lo, own, npix_full = bd.lo, bd.own, bd.Hfull * W

            def pooled(src, ld, C_, dst, off=0, r0=None, r1=None, c0=0, c1=None):
                ops.rect_sum(src, dst, w["rs_tmp"], w["rs_cnt"], B=B, W=W, ld=ld, C_=C_, r0=(lo if r0 is None else r0),
                             r1=(lo + own if r1 is None else r1), c0=c0, c1=c1, out_off=off)
        s = self.scale
        y = torch.empty(B, cfg["in_chans"], H * s, W * s, dtype=torch.float32, device=self.dev)
        mean = RGB_MEAN if cfg["in_chans"] == 3 else (0.0,) * 4
        r = float(cfg.get("img_range", 1.0))
        geo = dict(B=B, H=H, W=W, dtype=dt)
        ln = lambda src, dst, gb, out_f32=False, gap_c=0: ops.layernorm(
            src, dst, gb[0], gb[1], B=B, npix=N, C_=C, ldy=(C if out_f32 else ldc), out_f32=out_f32, dtype=dt,
            gap=(w["gap"] if gap_c else None), gap_c=gap_c)

        tA, tB, tC = w["tA"], w["tB"], w["tC"]
        LNB = ops.layernorm_blocks()

        def run_ocab(L, t, have_n, nblk, as_conv_input=False):
            """OCAB of residual group L on the residual stream t -> the buffer holding the result   hat_arch.py:326-393
            as_conv_input: the only consumer is the group's 3x3 conv, which reads its input as T (bf16) rows anyway: the last
            linear then stores its fp32 result (+ residual) as T rows into w["ao"] and the fp32 stream is not written at all
            (same values as the conv's own staging conversion; 650 B/px less traffic per group)."""
            oc = L["ocab"]
            esc = oc.get("esc")  # OCAB                                                    :326-393
            if bd is not None:   # key / value windows reach (wse - ws) / 2 rows into the neighbours' bands             :359-360
                yield ("halo", [((w["n"] if have_n else t), (self.wse - ws + 1) // 2)])
            if not have_n:
                ln(t, w["n"], oc["n1"], gap_c=(esc.pdim if esc else 0))
                nblk = LNB
            kv_src = w["n"]
            if esc is not None:  # K/V from ESC(LN(x))                                     :336-344
                self._esc_lk(esc, w, w["n"], B, H, W, nblk)
                self._run_lin(esc.aggr, w["n"], w["yesc"], **geo, ldx=ldc, ldo=ldc, x0=w["y16"], c_split=esc.pdim, ldx0=w["y16"].shape[2])
                kv_src = w["yesc"]
            qbuf, kvbuf, ldq, ldkv = w["q"], w["kv"], ldc, w["kv"].shape[2]
            if oc.get("qkvf") is not None and esc is None:
                ops.ocab_qkv(oc["qkvf"], w["n"], w["qkv"], B=B, H=H, W=W, ldx=ldc, ldo=432, dtype=dt)
                qbuf, kvbuf, ldq, ldkv = w["qkv"], w["qkv"].view(-1)[144:], 432, 432
            else:
                s0 = torch.cuda.current_stream(self.dev)   # q and kv projections are independent
                s1 = s0 if bd is not None else self._side_stream()
                s1.wait_stream(s0)
                with torch.cuda.stream(s1):
                    self._run_lin(oc["q"], w["n"], w["q"], **geo, ldx=ldc, ldo=ldc)
                self._run_lin(oc["kv"], kv_src, w["kv"], **geo, ldx=ldc, ldo=w["kv"].shape[2])
                s0.wait_stream(s1)
            if self.focus or self.topk < 1.0:   # HATX: focus bias on the logits and / or top-k key pruning   hatx_arch.py:421-449
                nk, pad = self.wse * self.wse, (self.wse - ws + 1) // 2
                if self.focus:
                    ops.conv(oc["fh0"], kv_src, w["fh"], **geo, ldx=ldc, ldo=w["fh"].shape[2], act=ACT_GELU, n_store=_r4(C // 4))
                    ops.conv(oc["fh2"], w["fh"], w["sal"], **geo, ldx=w["fh"].shape[2], ldo=8, n_store=4, out_mode=O_NHWC_F32)
                k_keep = max(1, int(self.topk * nk)) if self.topk < 1.0 else nk
                ops.ocab_keybias(w["sal"] if self.focus else None, kvbuf, w["kb"], B=B, H=H, W=W, C_=C, ws=ws, wse=self.wse, pad=pad,
                                 k_keep=k_keep, ldsal=(-8 if dt == ops.HAT_BF16 else 8), ldkv=ldkv, dtype=dt)
                ops.ocab_attention_kb(qbuf, kvbuf, oc["bias_rot"], w["kb"], w["ao"], B=B, H=H, W=W, C_=C, heads=L["heads"], ws=ws,
                                      wse=self.wse, pad=pad, ldq=ldq, ldkv=ldkv, ldo=ldc, dtype=dt)
            else:
                ops.ocab_attention(qbuf, kvbuf, oc["bias_rot"], w["ao"], B=B, H=H, W=W, C_=C, heads=L["heads"], ws=ws,
                                   wse=self.wse, ldq=ldq, ldkv=ldkv, ldo=ldc, dtype=dt, q_log2=oc["qlog2"])
            tout = tB if t is tA else t  # never write the RHAG input buffer
            if oc["proj"].frag:  # norm2 (:306) rides on the projection's epilogue
                self._run_lin(oc["proj"], w["ao"], tout, **geo, ldx=ldc, ldo=C, out_mode=O_NHWC_F32, r1=t, ldr1=C,
                              ln=oc["n2"], ln_out=w["n"], ld_ln=ldc)
            else:
                self._run_lin(oc["proj"], w["ao"], tout, **geo, ldx=ldc, ldo=C, out_mode=O_NHWC_F32, r1=t, ldr1=C)
                ln(tout, w["n"], oc["n2"])
            if oc["mlpf"] is not None:   # fc1 + GELU + fc2 + residual fused: the hidden tensor never reaches HBM
                dst = w["ao"] if as_conv_input else tout
                ops.ocab_mlp(oc["mlpf"], w["n"], tout, dst, B=B, H=H, W=W, ldx=ldc, ldr1=C, ldo=(ldc if as_conv_input else C),
                             out_f32=not as_conv_input, dtype=dt)
                return dst
            self._run_lin(oc["mlp0"], w["n"], w["g"], **geo, ldx=ldc, ldo=w["g"].shape[2], act=ACT_GELU)
            if as_conv_input:
                self._run_lin(oc["mlp2"], w["g"], w["ao"], **geo, ldx=w["g"].shape[2], ldo=ldc, out_mode=O_NHWC_T, r1=tout, ldr1=C)
                return w["ao"]
            self._run_lin(oc["mlp2"], w["g"], tout, **geo, ldx=w["g"].shape[2], ldo=C, out_mode=O_NHWC_F32, r1=tout, ldr1=C)
            return tout

        if only_ocab is not None:   # test hook: one OCAB on a given residual stream (block-level parity against reference goldens)
            t_in, gidx = only_ocab
            w["tB"].copy_(t_in.reshape(B, N, C))
            return (yield from run_ocab(self.layers[gidx], w["tB"], False, ops.layernorm_blocks())).clone()
        # (x - mean) * img_range ; conv_first                                           :849-853
        ops.conv(self.conv_first, x, w["f0"], **geo, ldx=0, ldo=C, x_mode=X_NCHW_F32_MEAN, out_mode=O_NHWC_F32,
                 in_scale=r, mean=mean)
        if self.pe_norm is not None:  # patch_embed + LN                                 :836
            ln(w["f0"], tA, self.pe_norm, out_f32=True)
        else:
            tA.copy_(w["f0"])        # device-to-device copy on the current stream
        if self.ape is not None:      # x + absolute_pos_embed (1, num_patches, C)          :837-838
            if self.ape.numel() != N * C:
                raise RuntimeError(f"absolute_pos_embed holds {self.ape.numel() // C} positions but the input has {N} "
                                   f"pixels (ape=True fixes the input size to img_size, hat_arch.py:699-702)")
            ops.add_f32(tA, self.ape, tA, B=B, n=N * C, c_bstride=0)
        # the group conv's epilogue emits the LayerNorm its consumer starts with (the next group's first norm1, or HAT.norm)
        conv_ln = dt == ops.HAT_BF16 and not os.environ.get("HAT_NO_CONV_LN")
        grp_n = grp_n16 = False   # ... so w["n"] (and w["n16"]) are already valid when a group starts
        grp_nblk = LNB
        for gi, L in enumerate(self.layers):
            t = tA            # current value of the residual stream (tA must survive until the RHAG tail)
            have_n = grp_n    # w["n"] already holds the next LayerNorm of t (emitted by the fused FFN / the group conv)
            have_n16 = grp_n16  # ... and w["n16"] a compact copy of its first 16 channels
            nblk = grp_nblk   # number of GAP partial blocks currently in w["gap"]
            grp_n = grp_n16 = False
            oc = L["ocab"]
            for i, hb in enumerate(L["habs"]):  # HAB                                     :217-238
                esc = hb["esc"]
                if bd is not None and not have_n:
                    yield ("halo", [(t, 7)])          # (LayerNorm1 is recomputed on the ghost rows from the refreshed stream)
                if not have_n:
                    ln(t, w["n"], hb["n1"], gap_c=esc.pdim)
                    nblk, have_n16 = LNB, False
                elif bd is not None:
                    # what this block reads beyond the band's own rows: t one row (depthwise 3x3 of the FFN), LayerNorm1's
                    # output three rows (the two CAB convs under it) and its first pdim channels ksize / 2 + 1 = seven rows
                    # (13x13 conv under the FFN's halo row)
                    er = esc.ksize // 2 + 1
                    yield ("halo", [(t, 1)] + ([(w["n"], 3), (w["n16"], er)] if have_n16 else [(w["n"], max(3, er))]))
                gapb = None
                if bd is not None:   # the ESC pool (esc_arch.py:96,121) over the whole FRAME: this band's share, then the sum
                    if have_n16:
                        pooled(w["n16"], 16, esc.pdim, w["gstat_l"])
                    else:
                        pooled(w["n"], ldc, esc.pdim, w["gstat_l"])
                    gapb = w["gstat_g"]    # (summed over the bands together with the CAB pool below: ONE reduce per HAB)
                mid = hb["cab0"].nout
                if "fold" in hb:
                    # c2 = conv3x3(c1) never exists: its ECA pooling follows from the sums of c1 (hat_cab_fold) and the
                    # scaled expand conv is three more k-steps of the aggregation GEMM (hat_aggr_cab).
                    # The two tiny per-sample kernels (one workgroup each, latency-bound) run on a side stream next to the
                    # convs they do not depend on: esc_weights beside the CAB squeeze conv, cab_fold beside the 13x13 conv.
                    fo = hb["fold"]
                    s0 = torch.cuda.current_stream(self.dev)
                    s1 = s0 if bd is not None else self._side_stream()
                    # The critical chain — ESC weight kernel -> 13x13 conv -> tail — stays on ONE stream: every hop between
                    # streams costs an event wait of ~12 us on this runtime (kernel trace: 2 hops per block = 0.9 ms per
                    # frame when the 13x13 conv sat on the side stream).  The CAB squeeze conv and its fold are shorter and
                    # go to the side stream (HAT_ESC_SIDE=1: the round-2 arrangement, for A/B).
                    def squeeze_chain():
                        if w["sweep"]:
                            ops.cab_squeeze(w["n"], fo["sq"][0], fo["sq"][1], w["c1"], w["colsum1"], B=B, H=H, W=W, C_=C, ldx=ldc, dtype=dt)
                        else:
                            ops.conv(hb["cab0"], w["n"], w["c1"], **geo, ldx=ldc, ldo=8, act=ACT_GELU, n_store=8, colsum=w["colsum1"])
                        if bd is None:
                            ops.cab_fold(w["c1"], w["colsum1"], w["tiles1"], hb["cab0"].npad, fo["w2"], fo["b2"], hb["eca_w"],
                                         hb["eca_w"].numel(), fo["ba"], float(cfg["conv_scale"]), w["scale"], w["wf"], w["bias_b"],
                                         w["eca_tmp"], B=B, H=H, W=W, C_=C, mid=mid, dtype=dt, w2f=fo["w2f"])
                    if bd is not None:
                        # band-sharded: the sums hat_cab_fold takes from c1 (hat_arch.py:73 via the linearity of the expand conv)
                        # are this band's share of [total | first row | last row | first column | last column | 4 corners]
                        squeeze_chain()
                        st = w["cstat_l"]
                        pooled(w["c1"], 8, 8, st, 0)
                        pooled(w["c1"], 8, 8, st, 24, c0=0, c1=1)
                        pooled(w["c1"], 8, 8, st, 32, c0=W - 1, c1=W)
                        if bd.r0 == 0:
                            pooled(w["c1"], 8, 8, st, 8, r0=lo, r1=lo + 1)
                            pooled(w["c1"], 8, 8, st, 40, r0=lo, r1=lo + 1, c0=0, c1=1)
                            pooled(w["c1"], 8, 8, st, 48, r0=lo, r1=lo + 1, c0=W - 1, c1=W)
                        if bd.r1 == bd.Hfull:
                            pooled(w["c1"], 8, 8, st, 16, r0=lo + own - 1, r1=lo + own)
                            pooled(w["c1"], 8, 8, st, 56, r0=lo + own - 1, r1=lo + own, c0=0, c1=1)
                            pooled(w["c1"], 8, 8, st, 64, r0=lo + own - 1, r1=lo + own, c0=W - 1, c1=W)
                        yield ("reduce", [(w["gstat_l"], w["gstat_g"], esc.pdim), (st, w["cstat_g"], 72)])
                        ops.cab_fold(None, None, 1, hb["cab0"].npad, fo["w2"], fo["b2"], hb["eca_w"], hb["eca_w"].numel(), fo["ba"],
                                     float(cfg["conv_scale"]), w["scale"], w["wf"], w["bias_b"], None, B=B, H=bd.Hfull, W=W, C_=C,
                                     mid=mid, dtype=dt, stats=w["cstat_g"], w2f=fo["w2f"])
                        self._esc_w(esc, w, B, bd.Hfull, W, 1, gap=gapb)
                        self._esc_conv(esc, w, w["n"], B, H, W, n16=(w["n16"] if have_n16 else None))
                    elif os.environ.get("HAT_ESC_SIDE") == "1":
                        self._esc_w(esc, w, B, H, W, nblk)
                        s1.wait_stream(s0)                              # n and the 13x13 weights are ready
                        with torch.cuda.stream(s1):                     # chain 2: 13x13 conv
                            self._esc_conv(esc, w, w["n"], B, H, W, n16=(w["n16"] if have_n16 else None))
                        squeeze_chain()
                    else:
                        s1.wait_stream(s0)                              # n is ready
                        with torch.cuda.stream(s1):
                            squeeze_chain()
                        self._esc_w(esc, w, B, H, W, nblk)
                        self._esc_conv(esc, w, w["n"], B, H, W, n16=(w["n16"] if have_n16 else None))
                    s0.wait_stream(s1)                              # both chains are done
                    if hb.get("tail"):   # (also the faster choice on small frames: 64x64 HAT-S 3.97 vs 6.19 ms per forward)
                        # aggregation + folded CAB + residuals + the whole FFN in ONE launch: tB never exists in HBM
                        if i + 1 < len(L["habs"]):
                            nxt, gap_c = L["habs"][i + 1]["n1"], L["habs"][i + 1]["esc"].pdim
                        else:
                            nxt, gap_c = oc["n1"], (oc["esc"].pdim if "esc" in oc else 0)
                        # Between two fused tails of a group the stream is FP16 rows (only this kernel reads and writes it there:
                        # 4C of a pixel's 12.3C bytes less, 43.7 -> 43.6 dB at 720p by emulation); the group's first
                        # tail reads fp32 (group conv / first LayerNorm), its last one writes fp32 (the OCAB's linears).
                        nh = L["habs"][i + 1] if i + 1 < len(L["habs"]) else None
                        out16 = bool(self.t16 and "ffn3" in hb and hb["ffn3"].C == 144 and nh is not None and nh.get("tail")
                                     and "ffn3" in nh and nh["ffn3"].C == 144)
                        if out16:
                            tout = w["hB"] if t is not w["hB"] else w["hC"]
                        else:
                            tout = tB if t is not tB else tC
                        ops.hab_tail(hb.get("ffn3", hb["ffn"]), esc.aggr, t, tout, hb["n2"][0], hb["n2"][1], n=w["n"], ldn_in=ldc, y16=w["y16"],
                                     c1=w["c1"], wf=w["wf"], bias_b=w["bias_b"], B=B, H=H, W=W, dtype=dt, ln1=nxt, n_out=w["n2b"],
                                     ldn=ldc, gap_out=w["gap"], gap_c=gap_c, n16_out=(w["n16"] if self.use_n16 else None))
                        w["n"], w["n2b"] = w["n2b"], w["n"]      # the kernel reads n with a halo: its output n' is another buffer
                        if _EMU_T16 is not None and not out16:   # measurement only (tools/residual16_psnr.py): what a 16-bit residual stream would cost
                            tout.copy_(tout.to(_EMU_T16).to(torch.float32))
                        t, have_n, nblk = tout, True, ops.ffn_tiles(hb["ffn"], H, W, dt)
                        have_n16 = self.use_n16
                        continue
                    ops.aggr_cab(esc.aggr, w["n"], tB, w["c1"], w["wf"], w["bias_b"], **geo, ldx=ldc, ldo=C, x0=w["y16"],
                                 c_split=esc.pdim, ldx0=w["y16"].shape[2], r1=t, ldr1=C)
                    pre_ln = False
                else:
                    c3 = ops.conv3x3_small if hb["cab0"].frag else ops.conv
                    c3(hb["cab0"], w["n"], w["c1"], **geo, ldx=ldc, ldo=_r8(mid), act=ACT_GELU, n_store=_r4(mid))
                    c3 = ops.conv3x3_small if hb["cab2"].frag else ops.conv
                    c3(hb["cab2"], w["c1"], w["c2"], **geo, ldx=_r8(mid), ldo=ldc, colsum=w["colsum"])
                    if bd is None:
                        ops.eca_scale(w["colsum"], w["tiles"], hb["cab2"].npad, N, hb["eca_w"], hb["eca_w"].numel(),
                                      float(cfg["conv_scale"]), w["eca_tmp"], w["scale"], B=B, C_=C)
                        self._esc_lk(esc, w, w["n"], B, H, W, nblk)
                    else:   # ECA pool of c2 (hat_arch.py:73) over the whole frame: this band's rows, then the sum over the bands
                        # (rows of npad floats, like the per-tile column sums: hat_eca_scale writes `scale` with that stride and
                        # the aggregation reads it so)
                        npd = hb["cab2"].npad
                        el, eg = (w[k].view(-1)[:B * npd].view(B, npd) for k in ("estat_l", "estat_g"))
                        pooled(w["c2"], ldc, C, el)
                        yield ("reduce", [(w["gstat_l"], w["gstat_g"], esc.pdim), (el, eg, _r4(C))])
                        ops.eca_scale(eg, 1, npd, npix_full, hb["eca_w"], hb["eca_w"].numel(),
                                      float(cfg["conv_scale"]), w["eca_tmp"], w["scale"], B=B, C_=C)
                        self._esc_w(esc, w, B, bd.Hfull, W, 1, gap=gapb)
                        self._esc_conv(esc, w, w["n"], B, H, W)
                    if hb.get("tail180") and not any(e.npad > 16 for e in [esc]):
                        # embed_dim 180: aggregation + scaled c2 + residuals + the whole FFN in one launch (hat_hab_tail3)
                        if i + 1 < len(L["habs"]):
                            nxt, gap_c = L["habs"][i + 1]["n1"], L["habs"][i + 1]["esc"].pdim
                        else:
                            nxt, gap_c = oc["n1"], (oc["esc"].pdim if "esc" in oc else 0)
                        if gap_c <= 16:
                            tout = tB if t is not tB else tC
                            ops.hab_tail(hb["ffn3"], esc.aggr, t, tout, hb["n2"][0], hb["n2"][1], n=w["n"], ldn_in=ldc, y16=w["y16"],
                                         bias_b=hb["bias256"], B=B, H=H, W=W, dtype=dt, ln1=nxt, n_out=w["n2b"], ldn=ldc, gap_out=w["gap"],
                                         gap_c=gap_c, n16_out=(w["n16"] if self.use_n16 else None), r2=w["c2"], ldr2=ldc,
                                         r2scale=w["scale"], r2scale_bstride=hb["cab2"].npad)
                            w["n"], w["n2b"] = w["n2b"], w["n"]
                            t, have_n, nblk = tout, True, -(-H // 8) * -(-W // 16)
                            have_n16 = self.use_n16
                            continue
                    # t = t + aggr(cat(y16, n[pdim:])) + conv_scale * eca * c2                :236
                    # hat_linear can emit LayerNorm2 of its result as hat_ffn's m_in, turning the FFN's stage 0 into a copy.
                    # Measured at 720p HAT-S: FFN -0.034 ms, aggr +0.070 ms per block (320 more bytes per pixel to write,
                    # and the FFN's stage 0 was already hidden behind its other workgroup) — a net loss, so it stays off.
                    pre_ln = False and "ffn" in hb and esc.aggr.frag
                    lnkw = dict(ln=hb["n2"], ln_out=w["m2"], ld_ln=w["m2"].shape[2], ln_ones=True) if pre_ln else {}
                    self._run_lin(esc.aggr, w["n"], tB, **geo, ldx=ldc, ldo=C, out_mode=O_NHWC_F32, x0=w["y16"],
                                  c_split=esc.pdim, ldx0=w["y16"].shape[2], r1=t, ldr1=C, r2=w["c2"], ldr2=ldc, r2scale=w["scale"],
                                  r2scale_bstride=hb["cab2"].npad, **lnkw)
                if "ffn" in hb:  # fused LN2 + fc1 + dw3x3 + gate + fc2 + residual (+ the next block's LayerNorm)
                    if i + 1 < len(L["habs"]):
                        nxt, gap_c = L["habs"][i + 1]["n1"], L["habs"][i + 1]["esc"].pdim
                    else:
                        nxt, gap_c = oc["n1"], (oc["esc"].pdim if "esc" in oc else 0)
                    mkw = dict(m_in=w["m2"], ldm_in=w["m2"].shape[2]) if pre_ln else {}
                    if gap_c > 16:   # (the fused kernels pool at most 16 channels: a wider ESC gets its LayerNorm + pool from hat_layernorm)
                        ops.ffn(hb["ffn"], tB, tC, hb["n2"][0], hb["n2"][1], B=B, H=H, W=W, dtype=dt, **mkw)
                        t, have_n, have_n16 = tC, False, False
                    else:
                        ops.ffn(hb["ffn"], tB, tC, hb["n2"][0], hb["n2"][1], B=B, H=H, W=W, dtype=dt, ln1=nxt, n_out=w["n"],
                                ldn=ldc, gap_out=w["gap"], gap_c=gap_c, **mkw)
                        # (hat_ffn / hat_ffn2 emit the LayerNorm rows only: w["n16"] still holds an OLDER block's compact copy —
                        # the group conv's or a fused tail's — and must not be handed to the next 13x13 conv)
                        t, have_n, have_n16, nblk = tC, True, False, ops.ffn_tiles(hb["ffn"], H, W, dt)
                else:
                    ln(tB, w["n"], hb["n2"])
                    hid2 = hb["fc1"].nout
                    self._run_lin(hb["fc1"], w["n"], w["u"], **geo, ldx=ldc, ldo=w["u"].shape[2])
                    if self.hatx:   # SGFN: [dw(a) * silu(b) | b], hid2 channels in and out            hatx_arch.py:165-177
                        ops.sgfn_gate(w["u"], hb["dw_w"], hb["dw_b"], w["g"], B=B, H=H, W=W, half=hid2 // 2, ldu=w["u"].shape[2],
                                      ldo=w["g"].shape[2], dtype=dt)
                    else:
                        ops.dwconv_gate(w["u"], hb["dw_w"], hb["dw_b"], w["g"], B=B, H=H, W=W, hid=hid2 // 2, ldu=w["u"].shape[2],
                                        ldo=w["g"].shape[2], dtype=dt)
                    self._run_lin(hb["fc2"], w["g"], tB, **geo, ldx=w["g"].shape[2], ldo=C, out_mode=O_NHWC_F32, r1=tB, ldr1=C)
                    t, have_n, have_n16 = tB, False, False
            to_conv = (L["conv"] is not None and dt == ops.HAT_BF16 and ldc == C and L["ocab"]["mlp2"].frag
                       and not os.environ.get("HAT_NO_BF16_CONV_IN"))
            tout = yield from run_ocab(L, t, have_n, nblk, as_conv_input=to_conv)
            if bd is not None and L["conv"] is not None:   # the group's 3x3 conv reads one row beyond the band's own
                yield ("halo", [(tout, 1)])
            # RHAG tail: conv3x3 + group residual, written over the group input             :556
            if L["conv"] is None:  # resi_connection == 'identity': group(x) + x                 :545-546
                ops.add_f32(tout, tA, tA, B=B, n=N * C)
            else:
                lnkw = {}
                if conv_ln and L["conv"].n_slices == 1 and L["conv"].nout == L["conv"].nt * 16 and tout is not w["n"]:
                    if gi + 1 < len(self.layers):
                        nh = self.layers[gi + 1]["habs"]
                        nxt, gap_c = (nh[0]["n1"], nh[0]["esc"].pdim) if nh else (None, 0)
                    else:
                        nxt, gap_c = (self.norm, 0) if self.conv_after_body is not None else (None, 0)
                    if nxt is not None and gap_c in (0, 4, 8, 12, 16):
                        lnkw = dict(ln=nxt, ln_out=w["n"], ld_ln=ldc, gap_out=w["gap"], gap_c=gap_c,
                                    n16_out=(w["n16"] if self.use_n16 and gap_c else None))
                        grp_n, grp_n16, grp_nblk = True, bool(self.use_n16 and gap_c), ops.conv_tiles(L["conv"], H, W, dt)
                if to_conv:
                    ops.conv(L["conv"], tout, tA, **geo, ldx=ldc, ldo=C, x_mode=X_NHWC_T, out_mode=O_NHWC_F32, r1=tA, ldr1=C, **lnkw)
                else:
                    ops.conv(L["conv"], tout, tA, **geo, ldx=C, ldo=C, x_mode=X_NHWC_F32, out_mode=O_NHWC_F32, r1=tA, ldr1=C, **lnkw)
                if _EMU_T16 is not None:
                    tA.copy_(tA.to(_EMU_T16).to(torch.float32))
        # final LN; conv_after_body + f0 ; conv_before_upsample + LeakyReLU                :844, :854-855
        if self.conv_after_body is None:   # nn.Identity: LN(t) + f0 in fp32, read as such by the next conv      :748
            ln(tA, tB, self.norm, out_f32=True)
            ops.add_f32(tB, w["f0"], tB, B=B, n=N * C)
            ops.conv(self.conv_before_up, tB, w["fb"], **geo, ldx=C, ldo=64, x_mode=X_NHWC_F32, act=ACT_LRELU)
        else:
            if bd is not None:
                # conv_after_body, conv_before_upsample, the Upsample convs and conv_last are five 3x3 convs, the last two at
                # 2x / 4x resolution: their receptive field is < 4 LR rows.  ONE refresh of 8 rows here, then the band computes
                # its ghost rows redundantly (f0 = conv_first(x) is exact there: the band's x carries the ghost rows).
                yield ("halo", [((w["n"] if grp_n else tA), 8)])
            if not grp_n:
                ln(tA, w["n"], self.norm)
            ops.conv(self.conv_after_body, w["n"], w["c2"], **geo, ldx=ldc, ldo=ldc, r1=w["f0"], ldr1=C)
            ops.conv(self.conv_before_up, w["c2"], w["fb"], **geo, ldx=ldc, ldo=64, act=ACT_LRELU)
        src, h, wd = w["fb"], H, W
        for (pw, rr), dst in zip(self.ups, w["ups"]):  # conv + PixelShuffle                :593-605
            ops.conv(pw, src, dst, B=B, H=h, W=wd, dtype=dt, ldx=64, ldo=64, out_mode=O_PIXSHUF_T, ps_r=rr)
            src, h, wd = dst, h * rr, wd * rr
        # conv_last ; / img_range + mean                                                   :856-858
        if self.conv_last_sweep is not None and wd % 16 == 0:
            wpk, b8, nout = self.conv_last_sweep
            ops.conv3x3_to_planes(src, wpk, b8, y, B=B, H=h, W=wd, C_=64, ldx=64, n_out=nout, out_scale=1.0 / r, mean=mean, dtype=dt)
        else:
            ops.conv(self.conv_last, src, y, B=B, H=h, W=wd, dtype=dt, ldx=64, ldo=0, out_mode=O_NCHW_F32,
                     out_scale=1.0 / r, mean=mean)
        return y
