"""LBVH mode against the brute-force scan.

The hierarchy only decides WHICH (ray, triangle) pairs reach the reference-precision test,
so a frame rendered through it must be byte-identical to the brute-force frame, and every
count that depends on hit decisions must agree.  (The brute-force frame itself is checked
against the oracle in test_gpu_parity.py / test_gpu_configs.py.)  The scenes used have no
in-plane-origin hits -- the one reference behaviour a box hierarchy cannot find -- and the
tests assert that through rt_stats.in_plane_origin_hits.
"""
import os

import numpy as np
import pytest

from raytracer_amd import Scene, abi

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def render(sc, bvh):
    from raytracer_amd.render import Renderer
    r = Renderer(sc)
    if bvh:
        r.set_option(abi.RT_OPT_USE_BVH, int(bvh))      # 1: four-wide nodes, 2: the binary tree
    r.render_frame()
    img, st = r.download(), r.stats()
    r.close()
    return img, st


@pytest.mark.parametrize("name,w,h,depth", [
    ("teapot.scene", 640, 360, 5), ("mixed_xf.scene", 320, 240, 6), ("inputs/input-09", 300, 300, 6),
    ("scene1.scene", 200, 200, 6), ("normals.scene", 240, 180, 4), ("synthetic_small.scene", 320, 180, 8),
    ("campanile_teddy.scene", 256, 144, 3), ("s3.scene", 200, 150, 6), ("inputs/input-05", 200, 200, 6),
])
def test_bvh_frame_equals_brute_force(name, w, h, depth):
    sc = Scene.load(os.path.join(ROOT, "scenes", name), w, h, depth)
    a, sa = render(sc, bvh=False)
    assert sa["in_plane_origin_hits"] == 0
    for mode in (1, 2):
        b, sb = render(sc, bvh=mode)
        assert np.array_equal(a, b), (mode, int((a != b).any(axis=-1).sum()))
        for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
            assert sa[k] == sb[k], (k, mode)
        assert sb["flagged_rays"] == 0, mode          # (no ray fell back to the full scan)
        if sa["n_triangles"] - sa["n_irregular"] > 1:
            assert sb["bvh_node_tests"] > 0 and sb["bvh_leaf_tests"] > 0


def test_bvh_teapot_4k_equals_brute_force():
    sc = Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 3840, 2160, 5)
    a, sa = render(sc, bvh=False)
    for mode in (1, 2):
        b, sb = render(sc, bvh=mode)
        assert np.array_equal(a, b), mode
        assert sa["hits"] == sb["hits"] and sa["rays_shadow"] == sb["rays_shadow"]
        # the hierarchy must prune: far fewer leaf tests than the brute-force pair count
        assert sb["bvh_leaf_tests"] * 20 < sa["closest_tests"] + sa["anyhit_tests"]
