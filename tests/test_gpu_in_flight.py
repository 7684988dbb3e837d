"""Several render calls in flight (RT_OPT_FRAMES_IN_FLIGHT, rt_join).

A call in flight runs on streams and in a workspace of its own; nothing about the pixels may
change: the frames must be byte-identical to those of one call at a time, the counters must add
up, pixels written by several calls must land in call order, and rt_join must make the
caller's stream see exactly the calls it names.
"""
import ctypes as C
import os

import numpy as np
import pytest

from raytracer_amd import Scene, abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTS = ("rays_primary", "rays_reflect", "rays_shadow", "hits", "closest_tests", "anyhit_tests")


def one_at_a_time(sc, bvh=0):
    from raytracer_amd.render import Renderer
    r = Renderer(sc)
    if bvh:
        r.set_option(abi.RT_OPT_USE_BVH, bvh)
    r.render_frame()
    img, st = r.download(), r.stats()
    r.close()
    return img, st


@pytest.mark.parametrize("name,w,h,depth,bvh", [
    ("teapot.scene", 640, 360, 5, 0), ("teapot.scene", 640, 360, 5, 1), ("mixed_xf.scene", 320, 240, 6, 0),
    ("inputs/input-06", 200, 200, 6, 0), ("campanile_teddy.scene", 256, 144, 3, 1),
])
@pytest.mark.parametrize("in_flight", [2, 3, 4])
def test_frames_in_flight_give_the_same_frame(name, w, h, depth, bvh, in_flight):
    from raytracer_amd.render import Renderer
    sc = Scene.load(os.path.join(ROOT, "scenes", name), w, h, depth)
    want, st1 = one_at_a_time(sc, bvh)
    r = Renderer(sc)
    if bvh:
        r.set_option(abi.RT_OPT_USE_BVH, bvh)
    r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, in_flight)
    n = 7
    for _ in range(n):
        r.render_frame()
    got, st = r.download(), r.stats()
    assert np.array_equal(got, want), int((got != want).any(axis=-1).sum())
    for k in COUNTS:
        assert st[k] == n * st1[k], k
    if not bvh:
        assert st["in_plane_origin_hits"] == n * st1["in_plane_origin_hits"]
    else:
        assert st["bvh_node_tests"] == n * st1["bvh_node_tests"] and st["bvh_leaf_tests"] == n * st1["bvh_leaf_tests"]
    # back to one call at a time on the same context: the same frame again
    r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 1)
    r.render_frame()
    assert np.array_equal(r.download(), want)
    r.close()


class Hip:
    """The few HIP runtime calls the hand-over test needs, from the copy of the runtime that the library itself
    uses in this process (whichever libamdhip64 was mapped first: a second copy must not be loaded beside it)."""

    def __init__(self):
        from raytracer_amd import abi as _abi
        _abi.device_lib()
        mapped = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln]
        assert mapped, "the device library is loaded, so a HIP runtime must be mapped"
        self.lib = C.CDLL(mapped[0])
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipStreamDestroy.argtypes = [C.c_void_p]
        self.lib.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.lib.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        self.lib.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def ok(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def malloc(self, n):
        p = C.c_void_p()
        self.ok(self.lib.hipMalloc(C.byref(p), n))
        return p.value

    def stream(self):
        s = C.c_void_p()
        self.ok(self.lib.hipStreamCreate(C.byref(s)))
        return s.value


def test_a_frame_buffer_per_call_and_join():
    """Three frame buffers, three calls in flight, each handed on (copied on the caller's stream) while the next
    ones are traced: rt_join(stream, keep_latest) must order the copy behind the call that wrote the buffer, and a
    call must come behind what the caller's stream held when it was issued (here: a fill of its buffer)."""
    from raytracer_amd.render import Renderer
    hip = Hip()
    sc = Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 960, 540, 5)
    want, _ = one_at_a_time(sc)
    nbytes = want.nbytes
    r = Renderer(sc)
    k, calls = 3, 8
    r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, k)
    frames = [hip.malloc(nbytes) for _ in range(k)]
    handed_on = [hip.malloc(nbytes) for _ in range(calls)]
    stream = hip.stream()
    lib = r._lib
    pending, out = [], 0
    for i in range(calls):
        f = frames[i % k]
        hip.ok(hip.lib.hipMemsetAsync(f, 7, nbytes, stream))
        assert lib.rt_bind_frame(r._ctx, C.c_void_p(f), nbytes) == 0
        r.render_frame(stream)
        pending.append(f)
        if len(pending) == k:
            r.join(stream, keep_latest=k - 1)
            hip.ok(hip.lib.hipMemcpyAsync(handed_on[out], pending.pop(0), nbytes, 3, stream))      # device to device
            out += 1
    r.join(stream, 0)
    for f in pending:
        hip.ok(hip.lib.hipMemcpyAsync(handed_on[out], f, nbytes, 3, stream))
        out += 1
    hip.ok(hip.lib.hipStreamSynchronize(stream))
    assert out == calls
    got = np.empty_like(want)
    for i in range(calls):
        hip.ok(hip.lib.hipMemcpyAsync(got.ctypes.data_as(C.c_void_p), handed_on[i], nbytes, 2, stream))   # device to host
        hip.ok(hip.lib.hipStreamSynchronize(stream))
        assert np.array_equal(got, want), i
    assert lib.rt_bind_frame(r._ctx, None, 0) == 0
    r.close()
    for p in frames + handed_on:
        hip.ok(hip.lib.hipFree(p))
    hip.ok(hip.lib.hipStreamDestroy(stream))


def test_calls_on_different_rows_land_in_call_order():
    """Tiles of one frame as separate calls in flight, then the whole frame once more with another depth bound to
    the same buffer: the last call's pixels must be the ones that stay."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 480, 272, 5)
    want, _ = one_at_a_time(sc)
    r = Renderer(sc)
    r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 3)
    for rank in range(5):
        r.render_tiles(rank, 16, 5)
    assert np.array_equal(r.download(), want)
    # the same rows again and again: every call clears and rewrites them, the frame stays the frame
    for _ in range(4):
        r.render_tile(100, 60)
    assert np.array_equal(r.download(), want)
    # event timing runs one call at a time; calls in flight before it must come first
    r.render_frame()
    r.set_option(abi.RT_OPT_PROFILE, 1)
    r.render_frame()
    r.set_option(abi.RT_OPT_PROFILE, 0)
    r.render_frame()
    assert np.array_equal(r.download(), want)
    with pytest.raises(Exception):
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 9)
    r.close()


def test_small_passes_in_flight():
    """Calls that need several passes each (a small RT_OPT_PASS_PIXELS) keep their passes in their own lane."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(os.path.join(ROOT, "scenes", "mixed_xf.scene"), 320, 240, 6)
    want, st1 = one_at_a_time(sc)
    r = Renderer(sc)
    r.set_option(abi.RT_OPT_PASS_PIXELS, 320 * 50)
    r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 2)
    for _ in range(5):
        r.render_frame()
    got, st = r.download(), r.stats()
    assert np.array_equal(got, want)
    for k in COUNTS:
        assert st[k] == 5 * st1[k], k
    r.close()


def test_no_two_lanes_share_a_hardware_queue(tmp_path):
    """RT_LANE_REPORT: the queue probe must find the process's hardware queues (several: HIP's default is 4), and
    the lanes' main streams must sit on different ones whenever there are at least as many queues as lanes; two
    lanes on four queues also get a side stream each, on the two queues left."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    assert os.path.exists(exe), "run `make cli`"
    for k in (2, 3, 4):
        env = dict(os.environ, RT_LANE_REPORT="1")
        res = subprocess.run([exe, os.path.join(ROOT, "scenes", "mixed_xf.scene"), str(tmp_path / "o.png"), "-w", "160", "-h", "120",
                              "-depth", "3", "-frames", "6", "-inflight", str(k)], capture_output=True, text=True, env=env)
        assert res.returncode == 0, res.stderr
        found = int(re.search(r"lanes: (\d+) hardware queue", res.stderr).group(1))
        assert found >= 2, res.stderr
        mains = [int(q) for q in re.findall(r"main stream on queue (\d+)", res.stderr)]
        sides = [int(q) for q in re.findall(r"side stream on queue (\d+)", res.stderr)]
        assert len(mains) == k
        if found >= k:
            assert len(set(mains)) == k, res.stderr
        if found >= 2 * k:
            assert len(sides) == k and len(set(mains + sides)) == 2 * k, res.stderr
        else:
            assert not sides, res.stderr


def test_the_deal_of_the_lanes_is_in_the_statistics():
    """rt_stats says how the calls in flight were dealt onto hardware queues (the probe compares device-clock
    intervals noted by short kernels, so it does not depend on host timing): with HIP's default of four queues, up
    to four lanes never share one; before any call has been made in lanes nothing is claimed."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(os.path.join(ROOT, "scenes", "mixed_xf.scene"), 160, 120, 3)
    r = Renderer(sc)
    r.render_frame()
    st = r.stats()
    assert st["lanes"] == 0 and st["caller_stream_shares_a_queue"] == -1
    for k in (2, 3, 4):
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, k)
        for _ in range(k + 1):
            r.render_frame()
        st = r.stats()
        assert st["lanes"] == k
        assert st["lane_hw_queues_found"] >= 2
        if st["lane_hw_queues_found"] >= k:
            assert st["lanes_share_a_queue"] == 0
        assert st["caller_stream_shares_a_queue"] in (0, 1)
    r.close()


def test_launch_geometry_hints_change_nothing(monkeypatch):
    """The streaming kernels of a pass are launched with grids sized for the queue lengths of the last pass of the
    same size (PassHint); frames and counters must be the same with the hints on and off, also when a hint is stale
    (a pass of the same size over other rows, whose queues have other lengths)."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 320, 180, 5)
    results = {}
    for hints in ("0", "1"):
        monkeypatch.setenv("RAYTRACER_AMD_GRID_HINTS", hints)
        r = Renderer(sc)
        frames = []
        for _ in range(3):                       # (the second and third frame are traced with the first one's lengths)
            r.render_frame()
            frames.append(r.download())
        # the same number of rows elsewhere in the frame: the hint of the previous pass does not fit this one
        r.render_tile(0, 60)
        r.render_tile(120, 60)
        r.render_tile(60, 60)
        frames.append(r.download())
        results[hints] = (frames, r.stats())
        r.close()
    for a, b in zip(results["0"][0], results["1"][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(results["1"][0][0], results["1"][0][2]) and np.array_equal(results["1"][0][0], results["1"][0][3])
    for k in COUNTS + ("exact_tests", "flagged_rays"):
        assert results["0"][1][k] == results["1"][1][k], k


def test_launches_leave_their_counters_and_flags_clean(monkeypatch, tmp_path):
    """No fill runs in front of a trace launch: the launches of a kind take three headers and two flag arrays in turn and
    the last kernel of each clears what the ones behind it will use.  With a pair list of no entries every ray that has a
    candidate is flagged for the full reference-precision scan -- a deterministic set -- so a flag that is left over
    from an earlier launch, or cleared while it is needed, shows in `flagged_rays`, and a lost one in the frame.  Held
    against the same frames with the fill forced in front of every launch, one call at a time and four in flight."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 160, 90, 3)
    ref = Renderer(sc)
    ref.render_frame()
    want = ref.download()
    ref.close()
    monkeypatch.setenv("RAYTRACER_AMD_PAIR_CAP", "0")
    lone = tmp_path / "lone.scene"
    lone.write_text("cam 0 0 2  -1 -1 1  1 -1 1  -1 1 1  1 1 1\nmat 0.1 0.1 0.1  0.6 0.5 0.4  0.3 0.3 0.3  16  0 0 0\n"
                    "tri -0.3 -0.25 -3  0.3 -0.25 -3  0 0.25 -3\nlta 0.2 0.2 0.2\nltp 2 2 2  1 1 1\n")
    seen = {}
    for clear in ("1", "0"):
        monkeypatch.setenv("RAYTRACER_AMD_CLEAR_EVERY_LAUNCH", clear)
        r = Renderer(sc)
        counts = []
        for k in range(4):
            r.reset_stats()
            r.render_frame()
            assert np.array_equal(r.download(), want), (clear, k)
            counts.append(r.stats()["flagged_rays"])
        assert counts[0] > 0 and len(set(counts)) == 1, (clear, counts)
        # Depth 0: one launch of each kind per pass, so the pass after the next meets the flag array of this one.  Bands
        # that cut the scene's only triangle differently, with empty bands in between: a band that flags some rays has
        # its last kernel look at every flag, and would count those the band before the last left behind.
        r0 = Renderer(Scene.load(str(lone), 160, 90, 0))
        band = []
        for y0 in (30, 0, 45, 0, 30, 45, 40, 60, 45):
            r0.reset_stats()
            r0.render_tile(y0, 30)
            band.append(r0.stats()["flagged_rays"])
        r0.close()
        assert band[0] > band[2] > 0 and band[1] == band[3] == band[7] == 0, (clear, band)
        assert band[4] == band[0] and band[5] == band[8] == band[2], (clear, band)
        r.reset_stats()
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 4)
        for k in range(8):
            r.render_frame()
        assert np.array_equal(r.download(), want), clear
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 1)
        seen[clear] = (counts[0], r.stats()["flagged_rays"], band)
        r.close()
    assert seen["0"] == seen["1"] and seen["0"][1] == 8 * seen["0"][0], seen
