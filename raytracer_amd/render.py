"""Device-side renderer handle over libraytracer_amd.so (the C ABI of include/rt_abi.h)
and the one-process-per-GPU tile driver.

The Python layer is plumbing only: it owns no arithmetic of the trace loop.
Every pixel comes out of rt_render_tile / rt_render_tiles.
"""
import ctypes as C

import numpy as np

from . import abi
from .scene import Scene


class DeviceError(RuntimeError):
    pass


def _check(lib, rc, what):
    if rc != 0:
        raise DeviceError(f"{what} failed ({rc}): {lib.rt_last_error().decode()}")


class Renderer:
    """One rt_ctx on the current HIP device.

    Stands where the reference calls scene->render() (raytracer.cpp:857):
    construct from a parsed Scene, render rows or interleaved tiles, download
    the RGBA8 frame that Film::writeFile would hand to lodepng.
    """

    def __init__(self, scene: Scene):
        self._lib = abi.device_lib()
        self._ctx = C.c_void_p()
        self.width, self.height = scene.width, scene.height
        _check(self._lib, self._lib.rt_upload_scene(C.byref(self._ctx), scene.desc_ptr), "rt_upload_scene")
        self._bound = None

    def set_option(self, option, value):
        _check(self._lib, self._lib.rt_set_option(self._ctx, int(option), int(value)), "rt_set_option")

    def render_tile(self, y0, rows, stream=None):
        _check(self._lib, self._lib.rt_render_tile(self._ctx, int(y0), int(rows), C.c_void_p(stream or 0)),
               "rt_render_tile")

    def render_tile_noalloc(self, y0, rows, stream=None):
        """rt_render_tile through the library's static workspace: the call allocates nothing."""
        _check(self._lib, self._lib.rt_render_tile_noalloc(self._ctx, int(y0), int(rows), C.c_void_p(stream or 0)),
               "rt_render_tile_noalloc")

    def render_tiles(self, first_tile, tile_rows, tile_step, stream=None):
        _check(self._lib, self._lib.rt_render_tiles(self._ctx, int(first_tile), int(tile_rows), int(tile_step),
                                                    C.c_void_p(stream or 0)), "rt_render_tiles")

    def join(self, stream=None, keep_latest=0):
        """RT_OPT_FRAMES_IN_FLIGHT > 1: `stream` waits (on the device) for all render calls issued so far but the
        `keep_latest` most recent ones."""
        _check(self._lib, self._lib.rt_join(self._ctx, C.c_void_p(stream or 0), int(keep_latest)), "rt_join")

    def render_frame(self, stream=None):
        self.render_tile(0, self.height, stream)

    def bind_frame(self, tensor):
        """Use a torch uint8 CUDA tensor of height*width*4 bytes as the frame."""
        if tensor is None:
            _check(self._lib, self._lib.rt_bind_frame(self._ctx, None, 0), "rt_bind_frame")
            self._bound = None
            return
        nbytes = tensor.numel() * tensor.element_size()
        _check(self._lib, self._lib.rt_bind_frame(self._ctx, C.c_void_p(tensor.data_ptr()), nbytes), "rt_bind_frame")
        self._bound = tensor   # keep it alive

    def download(self):
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        _check(self._lib, self._lib.rt_download(self._ctx, out.ctypes.data_as(C.c_void_p), out.nbytes), "rt_download")
        return out

    def primary_hits(self, y0=0, rows=None):
        """Nearest hit of every camera ray of film rows [y0, y0 + rows): (object index or -1,
        float metric d) per pixel, shape [rows, width], row 0 = film row y0."""
        rows = self.height - y0 if rows is None else rows
        keys = np.empty((rows, self.width), dtype=np.uint64)
        _check(self._lib, self._lib.rt_primary_hits(self._ctx, int(y0), int(rows), keys.ctypes.data_as(C.c_void_p),
                                                    keys.size), "rt_primary_hits")
        miss = keys == np.uint64(0x7F7FFFFFFFFFFFFF)
        obj = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
        obj[miss] = -1
        d = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
        return obj.astype(np.int32), d

    def bounce_probe(self, n_lights, y0=0, rows=None):
        """Per pixel of film rows [y0, y0 + rows): the camera ray's hit point, the reflected direction, the
        occlusion words of the hit's shadow rays, and (object index or -1, metric d) of the reflected ray's
        nearest hit."""
        rows = self.height - y0 if rows is None else rows
        words = max(1, (n_lights + 31) // 32)
        n = rows * self.width
        point = np.zeros((rows, self.width, 3)); refl = np.zeros((rows, self.width, 3))
        occl = np.zeros((rows, self.width, words), dtype=np.uint32)
        keys = np.empty((rows, self.width), dtype=np.uint64)
        _check(self._lib, self._lib.rt_bounce_probe(self._ctx, int(y0), int(rows), point.ctypes.data_as(C.c_void_p),
                                                    refl.ctypes.data_as(C.c_void_p), occl.ctypes.data_as(C.c_void_p),
                                                    keys.ctypes.data_as(C.c_void_p), n), "rt_bounce_probe")
        miss = keys == np.uint64(0x7F7FFFFFFFFFFFFF)
        obj = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
        obj[miss] = -1
        d = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
        return point, refl, occl, obj.astype(np.int32), d

    def stats(self):
        st = abi.RtStats()
        _check(self._lib, self._lib.rt_get_stats(self._ctx, C.byref(st)), "rt_get_stats")
        return st.as_dict()

    def device_info(self):
        """(compute units, engine clock in kHz) of the device this renderer lives on."""
        cu, khz = C.c_int32(), C.c_int32()
        _check(self._lib, self._lib.rt_device_info(self._ctx, C.byref(cu), C.byref(khz)), "rt_device_info")
        return cu.value, khz.value

    def reset_stats(self):
        _check(self._lib, self._lib.rt_reset_stats(self._ctx), "rt_reset_stats")

    def close(self):
        if self._ctx:
            self._lib.rt_free(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tiles_of_rank(height, tile_rows, rank, world):
    """Film-row ranges [y0, y1) of the interleaved scanline tiles owned by `rank`:
    tile k covers rows [k*tile_rows, (k+1)*tile_rows) and belongs to rank k % world."""
    n_tiles = (height + tile_rows - 1) // tile_rows
    return [(k * tile_rows, min((k + 1) * tile_rows, height)) for k in range(rank, n_tiles, world)]


def tile_spans(width, height, tile_rows, world):
    """[(owner rank, first frame row, last frame row + 1)] of every tile, from the host library's
    rth_tile_span (the same function the C++ driver's exchange uses): a tile's rows are contiguous in
    the frame (row 0 = top scanline, so film row y is frame row height - 1 - y)."""
    lib = abi.host_lib()
    out = []
    off, nbytes = C.c_size_t(), C.c_size_t()
    row_bytes = width * 4
    for k in range((height + tile_rows - 1) // tile_rows):
        owner = lib.rth_tile_span(width, height, tile_rows, world, k, C.byref(off), C.byref(nbytes))
        if owner < 0:
            raise DeviceError(f"rth_tile_span failed: {lib.rth_last_error().decode()}")
        out.append((owner, off.value // row_bytes, (off.value + nbytes.value) // row_bytes))
    return out


class TileGather:
    """End-of-frame exchange, set up once per (frame size, tiling, rank): every rank sends the rows
    of each of its tiles -- a contiguous slice of its frame -- to rank 0, which receives them straight
    into the same rows of its own frame: one batch of point-to-point operations
    (torch.distributed.batch_isend_irecv: grouped ncclSend / ncclRecv over xGMI with the nccl = RCCL
    backend, gloo on CPU), no packing, no staging copy, no scatter afterwards.  `host_staging` moves the
    slices through host memory (for a backend without device-memory transfers, i.e. gloo with the frame
    on a GPU: a rehearsal of the N-rank flow on a box with fewer GPUs than ranks)."""

    def __init__(self, height, width, tile_rows, rank, world, device, dist=None, torch=None, host_staging=False):
        self.rank, self.world, self.dist, self.torch = rank, world, dist, torch
        self.host_staging = host_staging
        self.spans = [(o, a, b) for o, a, b in tile_spans(width, height, tile_rows, world) if o != 0] if world > 1 else []

    def __call__(self, frame):
        if self.world == 1:
            return frame
        dist = self.dist
        ops, landed = [], []
        for owner, a, b in self.spans:
            if self.rank == 0:
                buf = frame[a:b]
                if self.host_staging:
                    buf = self.torch.empty(buf.shape, dtype=buf.dtype, device="cpu")
                    landed.append((a, b, buf))
                ops.append(dist.P2POp(dist.irecv, buf, owner))
            elif self.rank == owner:
                ops.append(dist.P2POp(dist.isend, frame[a:b].cpu() if self.host_staging else frame[a:b], 0))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for a, b, buf in landed:
            frame[a:b].copy_(buf)
        return frame


class PackedGather:
    """The same exchange with ONE message per peer.  Every rank other than 0 packs the rows of its tiles
    into a contiguous buffer (one row-gather kernel: torch.index_select over the frame's rows) and sends it
    to rank 0 in one point-to-point operation; rank 0 receives the peers' buffers side by side into one
    staging buffer and puts all rows where they belong with one row-scatter kernel (index_copy_).  Against
    TileGather: N - 1 receives per frame on rank 0 instead of one per tile (118 of them for a 4K frame on
    8 ranks), two small kernels and one staging copy of the frame more.  Both produce the same bytes; which
    one is faster is a matter of the interconnect and is left to a measurement on the machine.

    Buffers are made once.  A call waits for its transfers before it returns the frame (on the device for
    the nccl backend: the current stream waits), so the next call may pack into the same buffer."""

    def __init__(self, height, width, tile_rows, rank, world, device, dist=None, torch=None, host_staging=False):
        self.rank, self.world, self.dist, self.torch = rank, world, dist, torch
        self.host_staging = host_staging
        self.counts = [0] * world
        if world == 1:
            return
        rows_of = [[] for _ in range(world)]
        for owner, a, b in tile_spans(width, height, tile_rows, world):
            rows_of[owner].extend(range(a, b))
        self.counts = [len(r) for r in rows_of]
        wire = "cpu" if host_staging else device
        if rank == 0:
            peers = [p for p in range(1, world) if rows_of[p]]
            self.peers = peers
            all_rows = [y for p in peers for y in rows_of[p]]
            self.index = torch.tensor(all_rows, dtype=torch.int64, device=device)
            self.staging = torch.empty((len(all_rows), width, 4), dtype=torch.uint8, device=wire)
            self.landed = torch.empty_like(self.staging, device=device) if host_staging else self.staging
        else:
            self.index = torch.tensor(rows_of[rank], dtype=torch.int64, device=device)
            self.packed = torch.empty((len(rows_of[rank]), width, 4), dtype=torch.uint8, device=device)

    def __call__(self, frame):
        if self.world == 1:
            return frame
        dist, torch = self.dist, self.torch
        ops = []
        if self.rank == 0:
            at = 0
            for p in self.peers:
                ops.append(dist.P2POp(dist.irecv, self.staging[at:at + self.counts[p]], p))
                at += self.counts[p]
        elif self.counts[self.rank]:
            torch.index_select(frame, 0, self.index, out=self.packed)
            ops.append(dist.P2POp(dist.isend, self.packed.cpu() if self.host_staging else self.packed, 0))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.rank == 0 and self.index.numel():
            if self.host_staging:
                self.landed.copy_(self.staging)
            frame.index_copy_(0, self.index, self.landed)
        return frame


def make_exchange(kind, height, width, tile_rows, rank, world, device, dist=None, torch=None, host_staging=False):
    """`kind`: "tiles" (TileGather: one transfer per tile, straight into place) or "packed" (PackedGather: one
    per peer through a packed buffer)."""
    cls = {"tiles": TileGather, "packed": PackedGather}[kind]
    return cls(height, width, tile_rows, rank, world, device, dist=dist, torch=torch, host_staging=host_staging)


class FramePipeline:
    """A sequence of frames on one rank of N with up to `in_flight` render calls in flight
    (RT_OPT_FRAMES_IN_FLIGHT): frame i is traced into buffer i mod in_flight, and its exchange is issued
    in_flight - 1 steps later -- behind rt_join(stream, in_flight - 1) on the caller's stream, beside the
    tracing of the frames after it.  drain() issues what is left; step() K times + drain() = K frames
    traced and exchanged, each exactly once and in order.  With in_flight = 1 every step traces and
    exchanges its frame at once, into frames[0].

    `renderer` needs set_option, bind_frame, render_tiles and join (Renderer has them); `exchange(frame)`
    is a TileGather; `on_frame(index, frame)`, if given, is called right behind the exchange of each frame
    (work it enqueues on the caller's stream sees the exchanged frame)."""

    def __init__(self, renderer, frames, exchange, rank, world, tile_rows, stream=None, on_frame=None):
        self.r, self.frames, self.exchange = renderer, list(frames), exchange
        self.rank, self.world, self.tile_rows, self.stream = rank, world, tile_rows, stream
        self.on_frame = on_frame
        self.lanes = 1
        self.calls = 0            # render calls issued
        self.done = 0             # frames exchanged
        self.waiting = []         # frames traced (or being traced) and not yet exchanged, oldest first
        self.r.bind_frame(self.frames[0])

    def set_in_flight(self, k):
        """Exchanges what is waiting, then switches to k calls in flight (1 <= k <= len(frames))."""
        if not 1 <= k <= len(self.frames):
            raise ValueError(f"{k} calls in flight need {k} frame buffers, {len(self.frames)} given")
        self.drain()
        self.r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, k)
        self.lanes = k
        self.r.bind_frame(self.frames[0])

    def _hand_on(self, frame):
        self.exchange(frame)
        if self.on_frame is not None:
            self.on_frame(self.done, frame)
        self.done += 1

    def step(self):
        k = self.lanes
        frame = self.frames[self.calls % k]
        if k > 1:
            self.r.bind_frame(frame)
        self.calls += 1
        self.r.render_tiles(self.rank, self.tile_rows, self.world, self.stream)
        if k == 1:
            self._hand_on(frame)
            return
        self.waiting.append(frame)
        if len(self.waiting) == k:
            # (one rank and nobody to hand the frame to: nothing to order on the caller's stream until drain())
            if self.world > 1 or self.on_frame is not None:
                self.r.join(self.stream, k - 1)
            self._hand_on(self.waiting.pop(0))

    def drain(self):
        if self.waiting:
            self.r.join(self.stream, 0)
            for frame in self.waiting:
                self._hand_on(frame)
            self.waiting = []

    def last_frame(self):
        """The buffer of the most recent render call."""
        return self.frames[(self.calls - 1) % self.lanes] if self.calls else self.frames[0]


def gather_frame(renderer, frame, tile_rows, rank, world, dist=None, torch=None, kind="tiles"):
    """One-shot form of TileGather / PackedGather."""
    g = make_exchange(kind, frame.shape[0], frame.shape[1], tile_rows, rank, world, frame.device, dist=dist, torch=torch)
    return g(frame)
