"""Tile-parallel inference of one large frame over the GPUs of a node.

Semantics are the reference's `HATModel.tile_process` (hat/models/hat_model.py:40-108): every
tile is an independent forward on `LR[core +- tile_pad]` (clamped to the frame), and only the
core region (x scale) of its output is kept.  Tiles are independent units, so ranks never
exchange features: each rank runs the tiles it owns and ONE collective — an all-gather of the
output cores over RCCL/xGMI — assembles the frame on every rank.

Two tilings:
  * `reference_tiles(H, W, tile_size, tile_pad)`: exactly the reference grid (square tiles);
  * `balanced_tiles(H, W, n, window, tile_pad)`: a gy x gx grid with n tiles whose core sizes are
    multiples of the window size and as equal as possible (720x1280 on 8 GPUs: 2 x 4 tiles of
    368|352 x 320), so that every rank gets the same amount of work.  It reproduces the
    reference semantics tile by tile; the oracle for it is the same loop over the same rectangles.
"""
from __future__ import annotations

import collections
import math
from typing import Callable, List, NamedTuple, Optional, Sequence

import torch


class Tile(NamedTuple):
    y0: int
    y1: int
    x0: int
    x1: int  # core region [y0,y1) x [x0,x1) of the LR frame
    py0: int
    py1: int
    px0: int
    px1: int  # padded input region


def _mk(y0, y1, x0, x1, H, W, pad) -> Tile:
    return Tile(y0, y1, x0, x1, max(y0 - pad, 0), min(y1 + pad, H), max(x0 - pad, 0), min(x1 + pad, W))


def reference_tiles(H: int, W: int, tile_size: int, tile_pad: int) -> List[Tile]:
    """hat_model.py:52-71: row-major grid of `tile_size` squares, last row/column smaller."""
    out = []
    for y in range(math.ceil(H / tile_size)):
        for x in range(math.ceil(W / tile_size)):
            y0, x0 = y * tile_size, x * tile_size
            out.append(_mk(y0, min(y0 + tile_size, H), x0, min(x0 + tile_size, W), H, W, tile_pad))
    return out


def _split(n_units: int, parts: int) -> List[int]:
    base, rem = divmod(n_units, parts)
    return [base + (1 if i < rem else 0) for i in range(parts)]


def balanced_tiles(H: int, W: int, n: int, window: int, tile_pad: int) -> List[Tile]:
    """n tiles on a gy x gx grid (gy*gx == n), core sizes multiples of `window`, chosen to
    minimise the largest padded tile (the critical path of a one-tile-per-rank step)."""
    if H % window or W % window:
        raise RuntimeError(f"frame {H}x{W} is not a multiple of window_size {window} (pad it first, hat_model.py:16-26)")
    uh, uw = H // window, W // window
    best = None
    for gy in range(1, n + 1):
        if n % gy:
            continue
        gx = n // gy
        if gy > uh or gx > uw:
            continue
        hs, ws_ = _split(uh, gy), _split(uw, gx)
        tiles, y = [], 0
        for hh in hs:
            x = 0
            for ww in ws_:
                tiles.append(_mk(y * window, (y + hh) * window, x * window, (x + ww) * window, H, W, tile_pad))
                x += ww
            y += hh
        cost = max((t.py1 - t.py0) * (t.px1 - t.px0) for t in tiles)
        if best is None or cost < best[0]:
            best = (cost, tiles)
    if best is None:
        raise RuntimeError(f"cannot cut a {H}x{W} frame into {n} window-aligned tiles")
    return best[1]


def assign(tiles: Sequence[Tile], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of tile indices to ranks."""
    order = sorted(range(len(tiles)), key=lambda i: -(tiles[i].py1 - tiles[i].py0) * (tiles[i].px1 - tiles[i].px0))
    load, owned = [0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        owned[r].append(i)
        load[r] += (tiles[i].py1 - tiles[i].py0) * (tiles[i].px1 - tiles[i].px0)
    return [sorted(o) for o in owned]


def run_tile(img: torch.Tensor, net: Callable, t: Tile, scale: int) -> torch.Tensor:
    """One tile: forward on the padded crop, keep the core (hat_model.py:77-108)."""
    o = net(img[:, :, t.py0:t.py1, t.px0:t.px1].contiguous())
    oy, ox = (t.y0 - t.py0) * scale, (t.x0 - t.px0) * scale
    return o[:, :, oy:oy + (t.y1 - t.y0) * scale, ox:ox + (t.x1 - t.x0) * scale]


def tile_forward(img: torch.Tensor, net: Callable, scale: int, tiles: Sequence[Tile]) -> torch.Tensor:
    """Single-process loop over `tiles` (the reference's tile_process for the same rectangles)."""
    b, c, h, w = img.shape
    out = img.new_zeros((b, c, h * scale, w * scale))
    for t in tiles:
        out[:, :, t.y0 * scale:t.y1 * scale, t.x0 * scale:t.x1 * scale] = run_tile(img, net, t, scale)
    return out


_XBUF = collections.OrderedDict()
_XBUF_MAX = 4   # staging pairs kept: a caller that alternates between a few frame sizes allocates each of them once


def _exchange_buffers(shape, world, dtype, device):
    """send / recv staging of the all-gather, allocated once per (shape, world, dtype, device) and reused by every step
    (cores smaller than the slot leave stale padding behind them; the unpack below never reads it)."""
    key = (tuple(shape), world, dtype, str(device))
    buf = _XBUF.get(key)
    if buf is None:
        send = torch.zeros(shape, dtype=dtype, device=device)
        recv = torch.empty((world * shape[0],) + tuple(shape[1:]), dtype=dtype, device=device)
        buf = _XBUF[key] = (send, recv)
        while len(_XBUF) > _XBUF_MAX:
            _XBUF.popitem(last=False)     # least recently used first
    else:
        _XBUF.move_to_end(key)
    return buf


def tile_parallel_forward(img: torch.Tensor, net: Callable, scale: int, tiles: Sequence[Tile], group=None,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Every rank holds the full LR frame `img` (11 MB at 720p) and the full weights; rank r runs the
    tiles `assign(tiles, world)[r]`; one all-gather of the output cores assembles the frame on all ranks."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    owned = assign(tiles, world)
    b, c, h, w = img.shape
    slots = max(len(o) for o in owned)
    mh = max(t.y1 - t.y0 for t in tiles) * scale
    mw = max(t.x1 - t.x0 for t in tiles) * scale
    send, recv = _exchange_buffers((slots, b, c, mh, mw), world, img.dtype, img.device)
    for s, i in enumerate(owned[rank]):
        o = run_tile(img, net, tiles[i], scale)
        send[s, :, :, :o.shape[2], :o.shape[3]] = o
    dist.all_gather_into_tensor(recv, send, group=group)  # concatenation along dim 0, rank-major
    recv = recv.view((world, slots) + tuple(send.shape[1:]))
    if out is None:
        out = img.new_zeros((b, c, h * scale, w * scale))
    for r in range(world):
        for s, i in enumerate(owned[r]):
            t = tiles[i]
            th, tw = (t.y1 - t.y0) * scale, (t.x1 - t.x0) * scale
            out[:, :, t.y0 * scale:t.y1 * scale, t.x0 * scale:t.x1 * scale] = recv[r, s, :, :, :th, :tw]
    return out
