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


def _worker(rank, world, port, H, W, tile_rows, out_path, kind="tiles"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    want = _pattern(H, W)
    frame = torch.zeros((H, W, 4), dtype=torch.uint8)
    for y0, y1 in tiles_of_rank(H, tile_rows, rank, world):      # film rows; frame row = H-1-y
        for y in range(y0, y1):
            frame[H - 1 - y] = want[H - 1 - y]
    gather_frame(None, frame, tile_rows, rank, world, dist=dist, torch=torch, kind=kind)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["tiles", "packed"])
@pytest.mark.parametrize("H,W,tile_rows,world", [(70, 33, 8, 2), (64, 16, 16, 2), (37, 9, 5, 3), (50, 7, 16, 4)])
def test_gather_rebuilds_the_frame(tmp_path, H, W, tile_rows, world, kind):
    """Both exchanges (one transfer per tile / one packed transfer per peer) leave the whole frame on rank 0."""
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), H, W, tile_rows, out, kind), nprocs=world, join=True)
    got = np.load(out)
    assert np.array_equal(got, _pattern(H, W).numpy())


def test_packed_exchange_with_a_rank_that_owns_nothing(tmp_path):
    """Fewer tiles than ranks: the ranks without a tile send nothing and rank 0 expects nothing from them."""
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(4, _free_port(), 20, 5, 16, out, "packed"), nprocs=4, join=True)
    assert np.array_equal(np.load(out), _pattern(20, 5).numpy())


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


class _StandInRenderer:
    """What FramePipeline needs of a Renderer, on CPU tensors: render call c writes the rows this rank owns with
    a pattern that encodes c, into the buffer bound at the time of the call; rt_join is recorded."""

    def __init__(self, H, W, tile_rows):
        self.H, self.W, self.tile_rows = H, W, tile_rows
        self.bound, self.calls, self.in_flight, self.joined_up_to, self.log = None, 0, 1, 0, []

    def set_option(self, option, value):
        from raytracer_amd import abi
        assert option == abi.RT_OPT_FRAMES_IN_FLIGHT
        self.in_flight = value
        self.joined_up_to = self.calls          # (the library waits for queued work when the option changes)

    def bind_frame(self, frame):
        self.bound = frame

    def render_tiles(self, rank, tile_rows, world, stream=None):
        want = (_pattern(self.H, self.W).to(torch.int64) + 5 * self.calls) % 251
        for y0, y1 in tiles_of_rank(self.H, tile_rows, rank, world):
            self.bound[self.H - y1:self.H - y0] = want[self.H - y1:self.H - y0].to(torch.uint8)
        self.log.append(("render", self.calls, self.bound.data_ptr()))
        self.calls += 1
        if self.in_flight == 1:
            self.joined_up_to = self.calls      # (one call at a time is joined into the caller's stream by itself)

    def join(self, stream=None, keep_latest=0):
        self.joined_up_to = max(self.joined_up_to, self.calls - keep_latest)
        self.log.append(("join", keep_latest))


def _pipeline_worker(rank, world, port, H, W, tile_rows, in_flight, n_frames, out_path, kind="tiles"):
    from raytracer_amd.render import FramePipeline, make_exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = _StandInRenderer(H, W, tile_rows)
    frames = [torch.zeros((H, W, 4), dtype=torch.uint8) for _ in range(in_flight)]
    gather = make_exchange(kind, H, W, tile_rows, rank, world, "cpu", dist=dist, torch=torch)
    got, order = [], []

    def exchange(frame):
        # the frame handed to the exchange must be one whose render call has been joined
        call = [c for kind, c, ptr in (e for e in r.log if e[0] == "render") if ptr == frame.data_ptr()][-1]
        assert call < r.joined_up_to, (call, r.joined_up_to)
        order.append(call)
        return gather(frame)

    pipe = FramePipeline(r, frames, exchange, rank, world, tile_rows, on_frame=lambda i, f: got.append(f.clone()))
    pipe.step()                                   # one call at a time first, as the benchmark does
    pipe.set_in_flight(in_flight)
    for _ in range(n_frames):
        pipe.step()
    assert len(pipe.waiting) == min(in_flight - 1, n_frames)
    pipe.drain()
    pipe.set_in_flight(1)
    pipe.step()
    assert order == list(range(n_frames + 2)) and pipe.done == n_frames + 2 and not pipe.waiting
    if rank == 0:
        np.save(out_path, torch.stack(got).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["tiles", "packed"])
@pytest.mark.parametrize("H,W,tile_rows,world,in_flight,n_frames", [(70, 9, 8, 2, 3, 7), (37, 5, 5, 3, 2, 4), (64, 4, 16, 2, 4, 2),
                                                                     (50, 7, 16, 2, 1, 3)])
def test_frames_in_flight_are_each_exchanged_once_in_order(tmp_path, H, W, tile_rows, world, in_flight, n_frames, kind):
    """FramePipeline on gloo ranks with a stand-in renderer: with k calls in flight the exchange of frame i is
    issued k - 1 steps after its render call and behind an rt_join that covers it; rank 0 ends up with every
    frame, whole, in order -- also when fewer frames than lanes are rendered."""
    out = str(tmp_path / "frames.npy")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), H, W, tile_rows, in_flight, n_frames, out, kind), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape[0] == n_frames + 2
    for c in range(n_frames + 2):
        want = ((_pattern(H, W).to(torch.int64) + 5 * c) % 251).to(torch.uint8).numpy()
        assert np.array_equal(got[c], want), c


def test_one_rank_without_a_consumer_joins_only_at_the_end():
    """One rank and nobody to hand frames to: FramePipeline must put nothing on the caller's stream per frame (what
    is enqueued there waits behind a lane's kernels when the two share a hardware queue) -- one join, in drain().
    With a consumer the per-frame joins are back."""
    from raytracer_amd.render import FramePipeline
    H, W, T = 32, 4, 8
    for consumer in (False, True):
        r = _StandInRenderer(H, W, T)
        frames = [torch.zeros((H, W, 4), dtype=torch.uint8) for _ in range(4)]
        seen = []
        pipe = FramePipeline(r, frames, lambda f: f, 0, 1, T, on_frame=(lambda i, f: seen.append(i)) if consumer else None)
        pipe.set_in_flight(4)
        for _ in range(9):
            pipe.step()
        joins_before_drain = [e for e in r.log if e[0] == "join"]
        pipe.drain()
        joins = [e for e in r.log if e[0] == "join"]
        assert pipe.done == 9 and not pipe.waiting
        if consumer:
            assert joins_before_drain == [("join", 3)] * 6 and joins[-1] == ("join", 0) and seen == list(range(9))
        else:
            assert joins_before_drain == [] and joins == [("join", 0)]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("H,W,T", [(2160, 3840, 16), (1000, 10, 16), (37, 9, 5), (50, 7, 16), (16, 4, 16), (1, 1, 16)])
def test_packed_share_describes_a_ranks_tiles_as_one_strided_region(world, H, W, T):
    """rth_packed_share (what `rt_render -packed` feeds hipMemcpy2DAsync with): packing every rank's share with
    exactly that strided copy and unpacking it on the other side rebuilds the frame; a rank's packed buffer holds
    exactly its tiles' bytes."""
    import ctypes as C
    from raytracer_amd import abi
    lib = abi.host_lib()
    frame = np.random.default_rng(5).integers(0, 256, size=H * W * 4, dtype=np.uint8)
    rebuilt = np.zeros_like(frame)
    covered = 0
    for rank in range(world):
        n_full = C.c_int()
        first, tb, stride, clip_off, clip_bytes = (C.c_size_t() for _ in range(5))
        assert lib.rth_packed_share(W, H, T, world, rank, C.byref(n_full), C.byref(first), C.byref(tb), C.byref(stride),
                                    C.byref(clip_off), C.byref(clip_bytes)) == 0
        assert tb.value == T * W * 4 and stride.value == world * T * W * 4
        # the 2-D copy: n_full rows of tile_bytes, source pitch = stride
        packed = [frame[first.value + i * stride.value: first.value + i * stride.value + tb.value] for i in range(n_full.value)]
        if clip_bytes.value:
            packed.append(frame[clip_off.value: clip_off.value + clip_bytes.value])
        buf = np.concatenate(packed) if packed else np.zeros(0, dtype=np.uint8)
        own = sum((b - a) * W * 4 for o, a, b in tile_spans(W, H, T, world) if o == rank)
        assert buf.size == own
        covered += buf.size
        # ... and back
        for i in range(n_full.value):
            rebuilt[first.value + i * stride.value: first.value + i * stride.value + tb.value] = buf[i * tb.value:(i + 1) * tb.value]
        if clip_bytes.value:
            rebuilt[clip_off.value: clip_off.value + clip_bytes.value] = buf[n_full.value * tb.value:]
        # the strided region is exactly this rank's spans
        mine = np.zeros(H * W * 4, dtype=bool)
        for o, a, b in tile_spans(W, H, T, world):
            if o == rank:
                mine[a * W * 4: b * W * 4] = True
        got = np.zeros_like(mine)
        for i in range(n_full.value):
            got[first.value + i * stride.value: first.value + i * stride.value + tb.value] = True
        got[clip_off.value: clip_off.value + clip_bytes.value] = True
        assert np.array_equal(got, mine)
    assert covered == frame.size and np.array_equal(rebuilt, frame)
