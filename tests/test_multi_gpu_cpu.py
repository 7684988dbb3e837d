"""The N-GPU tile exchange on CPU: gloo ranks, each owning the interleaved scanline tiles
rt_render_tiles would give it, send their tiles to rank 0 through raytracer_amd.render.gather_frame
(point-to-point sends of each tile's rows straight into rank 0's frame -- the layout the C++ driver's
grouped ncclSend / ncclRecv uses, both from rth_tile_span).

No pixel is traced here (the trace loop has no CPU path): every rank fills the rows it owns
with a row-coded pattern, so the test checks exactly the exchange -- which rows go where,
unequal tile counts per rank, and the clipped last tile."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracer_amd.render import gather_frame, tile_spans, tiles_of_rank  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _pattern(H, W):
    y = torch.arange(H, dtype=torch.int64).view(H, 1, 1)
    x = torch.arange(W, dtype=torch.int64).view(1, W, 1)
    c = torch.arange(4, dtype=torch.int64).view(1, 1, 4)
    return ((y * 7 + x * 3 + c * 11) % 251).to(torch.uint8)


def _worker(rank, world, port, H, W, tile_rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    want = _pattern(H, W)
    frame = torch.zeros((H, W, 4), dtype=torch.uint8)
    for y0, y1 in tiles_of_rank(H, tile_rows, rank, world):      # film rows; frame row = H-1-y
        for y in range(y0, y1):
            frame[H - 1 - y] = want[H - 1 - y]
    gather_frame(None, frame, tile_rows, rank, world, dist=dist, torch=torch)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W,tile_rows,world", [(70, 33, 8, 2), (64, 16, 16, 2), (37, 9, 5, 3), (50, 7, 16, 4)])
def test_gather_rebuilds_the_frame(tmp_path, H, W, tile_rows, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), H, W, tile_rows, out), nprocs=world, join=True)
    got = np.load(out)
    assert np.array_equal(got, _pattern(H, W).numpy())


def test_tile_ownership_is_a_partition():
    H, T = 2160, 16
    for world in (1, 2, 4, 8):
        seen = np.zeros(H, dtype=np.int32)
        for rank in range(world):
            for y0, y1 in tiles_of_rank(H, T, rank, world):
                seen[y0:y1] += 1
        assert (seen == 1).all()


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("H,W,T", [(2160, 3840, 16), (4320, 7680, 16), (1000, 1000, 16), (37, 9, 5), (50, 7, 16), (16, 4, 16), (1, 1, 16)])
def test_tile_spans_map_tiles_to_frame_rows(world, H, W, T):
    """rth_tile_span (what the C++ driver's ncclSend / ncclRecv exchange and TileGather both use) against
    the definition: tile k = film rows [kT, min((k+1)T, H)) of rank k % N, film row y = frame row H-1-y.
    Ragged heights included: the last tile is clipped, ranks own unequal numbers of tiles."""
    spans = tile_spans(W, H, T, world)
    assert len(spans) == (H + T - 1) // T
    covered = np.zeros(H, dtype=np.int32)
    for k, (owner, a, b) in enumerate(spans):
        y0, y1 = k * T, min((k + 1) * T, H)
        assert owner == k % world
        assert (a, b) == (H - y1, H - y0)                      # contiguous frame rows, top row first
        assert (y0, y1) in tiles_of_rank(H, T, owner, world)   # the rows rt_render_tiles(rank = owner) traces
        covered[a:b] += 1
    assert (covered == 1).all()
    # no such tile: an error, not a span
    import ctypes as C
    from raytracer_amd import abi
    off, nb = C.c_size_t(), C.c_size_t()
    assert abi.host_lib().rth_tile_span(W, H, T, world, len(spans), C.byref(off), C.byref(nb)) < 0
