"""The multi-GPU path (tile-parallel inference) on CPU: tiling logic against the oracle's
tile loop (itself pinned to reference goldens), and a world_size-2 `gloo` run of the all-gather path.
The per-tile network here is the CPU oracle — the distributed logic does not depend on what runs a tile."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import hat_oracle as O
from super_resolution_amd import synth, tile_parallel as tp
from helpers import X_SEED, golden, max_abs, oracle_sd


def test_reference_grid_equals_oracle_tile_loop_and_golden():
    g = golden("tiled_tiny_x2.npz")
    cfg, sd = oracle_sd("tiny_x2")
    net = lambda z: O.hat_forward(z, sd, cfg)
    x = synth.synth_input(X_SEED, tuple(g["x_shape"]))
    img, ph, pw = O.pre_process(x, cfg["window_size"])
    ts, pad = (int(v) for v in g["tile"])
    tiles = tp.reference_tiles(img.shape[2], img.shape[3], ts, pad)
    y = O.post_process(tp.tile_forward(img, net, 2, tiles), ph, pw, 2)
    assert max_abs(y, g["y"]) <= 1e-5  # the golden was produced through the reference network itself
    assert torch.equal(tp.tile_forward(img, net, 2, tiles), O.tile_process(img, net, 2, ts, pad))


@pytest.mark.parametrize("H,W,n,win", [(720, 1280, 8, 16), (720, 1280, 4, 16), (720, 1280, 2, 16), (96, 80, 6, 8), (64, 64, 1, 16)])
def test_balanced_tiles_partition_the_frame(H, W, n, win):
    tiles = tp.balanced_tiles(H, W, n, win, 32)
    assert len(tiles) == n
    cover = torch.zeros(H, W, dtype=torch.int32)
    for t in tiles:
        assert (t.y1 - t.y0) % win == 0 and (t.x1 - t.x0) % win == 0 and t.y0 % win == 0 and t.x0 % win == 0
        assert t.py0 == max(t.y0 - 32, 0) and t.py1 == min(t.y1 + 32, H) and t.px0 == max(t.x0 - 32, 0) and t.px1 == min(t.x1 + 32, W)
        assert (t.py1 - t.py0) % win == 0 and (t.px1 - t.px0) % win == 0  # padded tiles stay window multiples (pad % win == 0)
        cover[t.y0:t.y1, t.x0:t.x1] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    owned = tp.assign(tiles, n)
    assert sorted(i for o in owned for i in o) == list(range(n)) and all(len(o) == 1 for o in owned)
    if (H, W, n) == (720, 1280, 8):
        assert sorted({(t.y1 - t.y0, t.x1 - t.x0) for t in tiles}) == [(352, 320), (368, 320)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg, sd = oracle_sd("tiny_x2")
    net = lambda z: O.hat_forward(z, sd, cfg)
    x = synth.synth_input(X_SEED, (1, 3, 48, 64))
    tiles = tp.balanced_tiles(48, 64, 3, 8, 16)  # 3 tiles on 2 ranks: uneven ownership
    y = tp.tile_parallel_forward(x, net, 2, tiles)
    ref = tp.tile_forward(x, net, 2, tiles)
    q.put((rank, float((y - ref).abs().max()), tuple(y.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_parallel_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    for _, err, shape in res:
        assert err == 0.0 and shape == (1, 3, 96, 128)  # every rank ends with the identical full frame
