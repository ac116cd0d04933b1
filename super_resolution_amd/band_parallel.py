"""Exact full-frame sharding of one HAT forward into ROW BANDS.

The reference has no such mode: its tile loop (`hat_model.py:40-108`, `tile_parallel.py` here) makes every tile an
independent forward, so a tiled frame differs from the full-frame forward — the ECA pool (`hat_arch.py:69,73`)
and the ESC dynamic-kernel pool (`esc_arch.py:96,121`) are means over whatever a forward sees, and every 3x3 / 13x13 conv and
the OCAB's 24x24 key windows read across tile borders.  Here the FRAME is cut into n window-aligned row bands, one per rank,
and the bands together compute exactly the unsharded forward:

  * a band's buffers hold its own rows plus GHOST = 16 rows above and below (one window: the OCAB's windows and the 8-row
    tiles of the fused HAB tail stay aligned); every kernel runs on the whole extended band, as if it were a frame;
  * before a layer reads across the band border, the ghost rows it will read are REFRESHED from the neighbour that owns them
    (`("halo", [(tensor, depth)...])`): per HAB one row of the fp32 residual stream (depthwise 3x3 of the FFN), three rows of
    LayerNorm1's output (CAB squeeze conv under the folded expand conv) and seven rows of its first 16 channels (13x13 conv
    under the aggregation's halo row) = 2.1 MB per side at 720p; per group four rows for the OCAB's key windows and one for
    the group conv; eight rows once before the five convs that end the network;
  * the two global pools become sums over the band's OWN rows (`hat_rect_sum`) added over the bands
    (`("reduce", [(local, glob, n) ...])`): 16 + 72 floats per HAB in ONE reduce (fp32 / embed_dim-180 path: 16 + C).

`HATEngine._forward_gen(x_band, band=...)` is the per-band forward as a generator that yields those two requests; this
module holds the two drivers that answer them:

  * `forward_bands_local`   — all n bands in ONE process on one GPU, in lockstep: a halo refresh is a device copy, a reduce
                              is `hat_add_f32`.  What a one-GPU box can verify: equal to the unsharded forward.
  * `forward_band_distributed` — one band per rank of a torch.distributed group: a halo refresh is a batch of point-to-point
                              send / recv with the two neighbours (RCCL over xGMI), a reduce is one all-reduce; the owned output
                              rows are all-gathered so that every rank ends with the full frame.
"""
from __future__ import annotations

from typing import Callable, Generator, List, NamedTuple, Optional, Sequence

import torch

GHOST = 16


class Band(NamedTuple):
    idx: int
    n: int
    r0: int       # owned rows [r0, r1) of the frame
    r1: int
    e0: int       # buffer rows [e0, e1) of the frame: owned + ghost, clamped to the frame
    e1: int
    Hfull: int

    @property
    def lo(self) -> int:      # ghost rows above the owned ones in the buffer
        return self.r0 - self.e0

    @property
    def own(self) -> int:
        return self.r1 - self.r0

    @property
    def hi(self) -> int:
        return self.e1 - self.r1


def make_bands(H: int, n: int, window: int = 16, ghost: int = GHOST) -> List[Band]:
    """n row bands of a frame of H rows: heights are multiples of `window`, as equal as possible."""
    if H % window:
        raise RuntimeError(f"frame height {H} is not a multiple of window_size {window} (pad it first, hat_model.py:16-26)")
    if ghost % window or ghost < 8:
        raise ValueError("ghost must be a multiple of the window size and cover the 8 rows of the last refresh")
    units = H // window
    if n < 1 or n > units:
        raise RuntimeError(f"cannot cut {H} rows into {n} bands of whole {window}-row windows")
    base, rem = divmod(units, n)
    out, r = [], 0
    for i in range(n):
        h = (base + (1 if i < rem else 0)) * window
        out.append(Band(i, n, r, r + h, max(r - ghost, 0), min(r + h + ghost, H), H))
        r += h
    if n > 1 and min(b.own for b in out) < 8:
        raise RuntimeError("bands thinner than the deepest halo (8 rows)")
    return out


def _pairs(req):
    """A reduce request's (local, glob, n) triples: ("reduce", [(local, glob, n), ...]) or the single-vector ("reduce", local, glob, n)."""
    return req[1] if isinstance(req[1], (list, tuple)) else [(req[1], req[2], req[3])]


def _rows(t: torch.Tensor, B: int, hb: int) -> torch.Tensor:
    """(B, hb, row_elems) view of a channel-last map (B, hb*W, ld) or (B, hb*W*ld)."""
    return t.reshape(B, hb, -1)


# ------------------------------------------------------------------------------------------------------------------
# driver 1: every band in this process, one GPU
# ------------------------------------------------------------------------------------------------------------------
def run_lockstep(gens: Sequence[Generator], bands: Sequence[Band], B: int, add: Callable):
    """Advance the band generators in lockstep and answer their requests locally.  `add(a, c, out, n)`: out[:, :n] = a[:, :n] +
    c[:, :n] on (B, m) fp32 tensors.  Returns the list of values the generators return."""
    n = len(gens)
    reqs, done = [None] * n, [None] * n
    for i, g in enumerate(gens):
        try:
            reqs[i] = next(g)
        except StopIteration as e:
            done[i] = (e.value,)
    while any(d is None for d in done):
        if any(d is not None for d in done):
            raise RuntimeError("band generators left lockstep (one finished while another still asks for an exchange)")
        kinds = {r[0] for r in reqs}
        if len(kinds) != 1:
            raise RuntimeError(f"band generators left lockstep: {sorted(kinds)}")
        kind = kinds.pop()
        if kind == "halo":
            for k in range(len(reqs[0][1])):
                depth = reqs[0][1][k][1]
                views = []
                for i, b in enumerate(bands):
                    t, d = reqs[i][1][k]
                    if d != depth:
                        raise RuntimeError("band generators ask for different halo depths")
                    views.append(_rows(t, B, b.e1 - b.e0))
                for i, b in enumerate(bands):
                    if i > 0:          # rows [r0 - depth, r0) live at the bottom of the upper neighbour's owned rows
                        u = bands[i - 1]
                        views[i][:, b.lo - depth:b.lo].copy_(views[i - 1][:, u.lo + u.own - depth:u.lo + u.own])
                    if i + 1 < n:      # rows [r1, r1 + depth) live at the top of the lower neighbour's owned rows
                        l = bands[i + 1]
                        views[i][:, b.lo + b.own:b.lo + b.own + depth].copy_(views[i + 1][:, l.lo:l.lo + depth])
        elif kind == "reduce":
            for k in range(len(_pairs(reqs[0]))):
                loc0, glob0, m = _pairs(reqs[0])[k]
                add(loc0, _pairs(reqs[1])[k][0], glob0, m) if n > 1 else glob0[:, :m].copy_(loc0[:, :m])
                for i in range(2, n):
                    add(glob0, _pairs(reqs[i])[k][0], glob0, m)
                for i in range(1, n):
                    _pairs(reqs[i])[k][1][:, :m].copy_(glob0[:, :m])
        else:
            raise RuntimeError(f"unknown band request {kind!r}")
        for i, g in enumerate(gens):
            try:
                reqs[i] = g.send(None)
            except StopIteration as e:
                done[i] = (e.value,)
    return [d[0] for d in done]


def forward_bands_local(engine, x: torch.Tensor, n: int) -> torch.Tensor:
    """The exact full-frame forward of `x` (B,3,H,W) computed as n row bands on the engine's GPU (stage 1 of §8 f4)."""
    from . import ops
    B, _, H, W = x.shape
    bands = make_bands(H, n, engine.ws)
    s = engine.scale

    def add(a, c, out, m):
        if m % 4:
            raise RuntimeError("reduce lengths are multiples of 4")
        for b in range(B):       # (rows of the (B, ld) buffers are ld apart: one launch per sample, n = m)
            ops.add_f32(a[b], c[b], out[b], B=1, n=m)

    with engine._lock, torch.cuda.device(engine.dev):
        x = x.to(torch.float32)
        gens = [engine._forward_gen(x[:, :, b.e0:b.e1].contiguous(), band=b) for b in bands]
        ys = run_lockstep(gens, bands, B, add)
        out = torch.empty(B, x.shape[1], H * s, W * s, dtype=torch.float32, device=x.device)
        for b, y in zip(bands, ys):
            out[:, :, b.r0 * s:b.r1 * s] = y[:, :, b.lo * s:(b.lo + b.own) * s]
    return out


# ------------------------------------------------------------------------------------------------------------------
# driver 2: one band per rank
# ------------------------------------------------------------------------------------------------------------------
def run_distributed(gen: Generator, band: Band, bands: Sequence[Band], B: int, group=None, stage_on_host: bool = False):
    """Answer one band generator's requests with collectives: halos = batched point-to-point send / recv with the two
    neighbouring ranks, reduces = all-reduce (sum).  stage_on_host: move payloads through host memory (a `gloo` group with
    device tensors — the one-GPU rehearsal; RCCL takes the device buffers directly)."""
    import torch.distributed as dist
    rank, n = band.idx, band.n
    up = bands[rank - 1] if rank > 0 else None
    down = bands[rank + 1] if rank + 1 < n else None
    hb = band.e1 - band.e0
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    try:
        req = next(gen)
    except StopIteration as e:
        return e.value
    while True:
        if req[0] == "halo":
            ops_, recvs = [], []
            for t, depth in req[1]:
                v = _rows(t, B, hb)
                if up is not None:
                    send = v[:, band.lo:band.lo + depth].contiguous()                # my top rows -> the upper rank's bottom ghost
                    recv = torch.empty_like(v[:, band.lo - depth:band.lo])
                    if stage_on_host:
                        send, recv = send.cpu(), recv.cpu()
                    ops_ += [dist.P2POp(dist.isend, send, peer(rank - 1), group), dist.P2POp(dist.irecv, recv, peer(rank - 1), group)]
                    recvs.append((v, band.lo - depth, band.lo, recv))
                if down is not None:
                    send = v[:, band.lo + band.own - depth:band.lo + band.own].contiguous()
                    recv = torch.empty_like(v[:, band.lo + band.own:band.lo + band.own + depth])
                    if stage_on_host:
                        send, recv = send.cpu(), recv.cpu()
                    ops_ += [dist.P2POp(dist.isend, send, peer(rank + 1), group), dist.P2POp(dist.irecv, recv, peer(rank + 1), group)]
                    recvs.append((v, band.lo + band.own, band.lo + band.own + depth, recv))
            if ops_:
                for w in dist.batch_isend_irecv(ops_):
                    w.wait()
            for v, a, b_, recv in recvs:
                v[:, a:b_].copy_(recv)
        elif req[0] == "reduce":     # all of the request's vectors in ONE all-reduce
            pairs = _pairs(req)
            buf = torch.cat([loc[:, :m] for loc, _, m in pairs], dim=1).contiguous()
            if stage_on_host:
                buf = buf.cpu()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            o = 0
            for _, glob, m in pairs:
                glob[:, :m].copy_(buf[:, o:o + m])
                o += m
        else:
            raise RuntimeError(f"unknown band request {req[0]!r}")
        try:
            req = gen.send(None)
        except StopIteration as e:
            return e.value


def forward_band_distributed(engine, x: torch.Tensor, group=None, stage_on_host: Optional[bool] = None) -> torch.Tensor:
    """Rank r of `group` computes band r of the frame `x` (every rank holds the whole LR frame, 11 MB at 720p, and the whole
    weights) and all ranks end with the full output frame (one all-gather of the owned rows)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if stage_on_host is None:
        stage_on_host = dist.get_backend(group) == "gloo" and x.is_cuda
    B, Cin, H, W = x.shape
    bands = make_bands(H, world, engine.ws)
    band, s = bands[rank], engine.scale
    with engine._lock, torch.cuda.device(engine.dev):
        gen = engine._forward_gen(x.to(torch.float32)[:, :, band.e0:band.e1].contiguous(), band=band)
        y = run_distributed(gen, band, bands, B, group, stage_on_host)
        mh = max(b.own for b in bands) * s
        send = torch.zeros(B, Cin, mh, W * s, dtype=torch.float32, device=x.device)
        send[:, :, :band.own * s] = y[:, :, band.lo * s:(band.lo + band.own) * s]
        recv = torch.empty((world,) + tuple(send.shape), dtype=torch.float32, device=x.device)
        if stage_on_host:
            rh = recv.cpu()
            dist.all_gather_into_tensor(rh.view(-1, *send.shape[1:]), send.cpu(), group=group)
            recv.copy_(rh)
        else:
            dist.all_gather_into_tensor(recv.view(-1, *send.shape[1:]), send, group=group)
        out = torch.empty(B, Cin, H * s, W * s, dtype=torch.float32, device=x.device)
        for b in bands:
            out[:, :, b.r0 * s:b.r1 * s] = recv[b.idx, :, :, :b.own * s]
    return out
