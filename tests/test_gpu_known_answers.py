"""Per-ray known answers: the nearest hit of every camera ray (object index and the bit pattern
of the reference's float metric d, raytracer.cpp:516-529) from the GPU against the CPU oracle,
in every configuration of the trace kernels -- brute force with either filter form, exact-only,
and the LBVH.  This is the nearest-hit parity gate: identical (hit index, distance bits) per ray.
"""
import os

import numpy as np
import pytest

from raytracer_amd import Scene, abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scene_file(name):
    for cand in (f"scenes/inputs/{name}", f"scenes/{name}.scene"):
        p = os.path.join(ROOT, cand)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


def oracle_hits(oracle, sc):
    obj = np.empty((sc.height, sc.width), dtype=np.int32)
    d = np.empty((sc.height, sc.width), dtype=np.float32)
    for y in range(sc.height):
        o, v = oracle.camera_row(sc, y)
        hit, _, dist, _ = oracle.closest(sc, o, v)
        obj[y], d[y] = hit, dist
    return obj, d


CONFIGS = [("static", {"FILTER_FORM": 1}), ("per-test", {"FILTER_FORM": 2}), ("exact", {"EXACT_ONLY": 1}),
           ("lbvh-static", {"FILTER_FORM": 1, "USE_BVH": 1}), ("lbvh-per-test", {"FILTER_FORM": 2, "USE_BVH": 1}),
           ("no-shared", {"SHARED_CONSTANTS": 0})]


@pytest.mark.parametrize("name,w,h", [("teapot", 96, 54), ("mixed_xf", 80, 60), ("campanile_teddy", 64, 36),
                                     ("scene1", 64, 64), ("input-09", 64, 64), ("input-07", 64, 64),
                                     ("synthetic_small", 80, 45), ("normals", 64, 48),
                                     # sizes that leave partial tiles in the queue's tile order
                                     ("teapot", 75, 41), ("scene1", 37, 16), ("teapot", 15, 33), ("teapot", 131, 7)])
def test_nearest_hit_index_and_distance_bits(name, w, h, oracle):
    from raytracer_amd.render import Renderer
    sc = Scene.load(scene_file(name), w, h, 1)
    want_obj, want_d = oracle_hits(oracle, sc)
    for label, opts in CONFIGS:
        r = Renderer(sc)
        for key, value in opts.items():
            r.set_option(getattr(abi, "RT_OPT_" + key), value)
        obj, d = r.primary_hits()
        r.close()
        assert np.array_equal(obj, want_obj), (label, int((obj != want_obj).sum()))
        hit = want_obj >= 0
        assert np.array_equal(d[hit].view(np.uint32), want_d[hit].view(np.uint32)), label


def test_rows_and_passes_of_the_hook():
    from raytracer_amd.render import Renderer
    sc = Scene.load(scene_file("teapot"), 128, 72, 1)
    r = Renderer(sc)
    full_obj, full_d = r.primary_hits()
    r.set_option(abi.RT_OPT_PASS_PIXELS, 128 * 5)                   # several passes
    part_obj, part_d = r.primary_hits(20, 31)
    r.close()
    assert np.array_equal(part_obj, full_obj[20:51])
    assert np.array_equal(part_d.view(np.uint32), full_d[20:51].view(np.uint32))
