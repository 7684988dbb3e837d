"""rt_render_tile_noalloc (the entry point that allocates nothing: its workspace is the library's
static 64 MiB array) against rt_render_tile, through the C ABI.  Bit-exact frames and equal ray
counts, on two calls in a row, the second with a smaller input (the array then still holds the
first call's contents at other offsets)."""
import os

import numpy as np
import pytest

from raytracer_amd import Scene, abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTS = ("rays_primary", "rays_reflect", "rays_shadow", "hits")


def scene_file(name):
    for cand in (f"scenes/inputs/{name}", f"scenes/{name}.scene"):
        p = os.path.join(ROOT, cand)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


def render(sc, y0, rows, noalloc, bvh=False):
    from raytracer_amd.render import Renderer
    r = Renderer(sc)
    if bvh:
        r.set_option(abi.RT_OPT_USE_BVH, 1)
    if noalloc:
        r.render_tile_noalloc(y0, rows)
    else:
        r.render_tile(y0, rows)
    img, st = r.download(), r.stats()
    r.close()
    return img, st


@pytest.mark.parametrize("name,w,h,depth,bvh", [("mixed_xf", 160, 120, 4, False), ("teapot", 192, 108, 5, False),
                                               ("teapot", 192, 108, 5, True)])
def test_noalloc_equals_render_tile_twice_in_a_row(name, w, h, depth, bvh):
    from raytracer_amd.render import Renderer
    sc = Scene.load(scene_file(name), w, h, depth)
    want_full, st_full = render(sc, 0, h, noalloc=False, bvh=bvh)
    want_part, st_part = render(sc, h // 4, h // 3, noalloc=False, bvh=bvh)
    r = Renderer(sc)
    if bvh:
        r.set_option(abi.RT_OPT_USE_BVH, 1)
    # first call: the whole frame
    r.render_tile_noalloc(0, h)
    got = r.download()
    st = r.stats()
    assert np.array_equal(got, want_full)
    for k in COUNTS:
        assert st[k] == st_full[k], k
    r.close()
    # second call, smaller input, on a fresh context (so that the rows it does not render stay black):
    # the array still holds the first call's data, cut differently
    r = Renderer(sc)
    if bvh:
        r.set_option(abi.RT_OPT_USE_BVH, 1)
    r.render_tile_noalloc(h // 4, h // 3)
    got = r.download()
    st = r.stats()
    assert np.array_equal(got, want_part)
    for k in COUNTS:
        assert st[k] == st_part[k], k
    # ... and a third on the same context: the rest of the frame in two more calls gives the whole frame
    r.render_tile_noalloc(0, h // 4)
    r.render_tile_noalloc(h // 4 + h // 3, h - h // 4 - h // 3)
    assert np.array_equal(r.download(), want_full)
    r.close()


def test_noalloc_rejects_a_tile_beyond_the_static_workspace():
    from raytracer_amd.render import Renderer, DeviceError
    sc = Scene.load(scene_file("teapot"), 3840, 2160, 5)
    r = Renderer(sc)
    with pytest.raises(DeviceError) as e:
        r.render_tile_noalloc(0, 2160)
    assert "(-3)" in str(e.value) and "at most" in str(e.value)
    # what the message allows does fit, and equals the allocating entry point's rows
    rows = int(str(e.value).split("at most ")[1].split(" rows")[0])
    assert 1 <= rows < 2160
    r.render_tile_noalloc(1000, rows)
    got = r.download()
    r.close()
    want, _ = render(sc, 1000, rows, noalloc=False)
    assert np.array_equal(got, want)


def test_noalloc_call_leaves_device_memory_as_it_found_it():
    """What the entry point exists for: between the moment the call is made and the moment its work has finished,
    the device's free memory does not move (no hipMalloc / hipFree, no workspace growth) -- also with event timing
    requested (RT_OPT_PROFILE), which the allocating entry point serves by creating events."""
    import ctypes as C
    from raytracer_amd.render import Renderer
    from tests.test_gpu_in_flight import Hip
    hip = Hip()
    hip.lib.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    hip.lib.hipDeviceSynchronize.argtypes = []

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        hip.ok(hip.lib.hipDeviceSynchronize())
        hip.ok(hip.lib.hipMemGetInfo(C.byref(free), C.byref(total)))
        return free.value

    sc = Scene.load(scene_file("teapot"), 192, 108, 5)
    want, _ = render(sc, 0, 108, noalloc=False)
    for profile in (0, 1):
        r = Renderer(sc)
        r.set_option(abi.RT_OPT_PROFILE, profile)
        r.render_tile_noalloc(0, 54)             # (warm-up: code objects are loaded at a kernel's first launch)
        before = free_bytes()
        r.render_tile_noalloc(0, 108)
        r.render_tile_noalloc(20, 30)
        after = free_bytes()
        assert after == before, (profile, before - after)
        assert np.array_equal(r.download(), want)
        r.close()
