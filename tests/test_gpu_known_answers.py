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


def oracle_bounce(oracle, sc):
    """Per pixel: level-0 hit, hit point, reflected direction, occlusion words; level-1 hit and metric."""
    H, W = sc.height, sc.width
    words = max(1, (sc.desc.n_lights + 31) // 32)
    hit0 = np.empty((H, W), dtype=np.int32)
    point = np.zeros((H, W, 3)); refl = np.zeros((H, W, 3))
    occl = np.zeros((H, W, words), dtype=np.uint32)
    hit1 = np.full((H, W), -1, dtype=np.int32); d1 = np.zeros((H, W), dtype=np.float32)
    for y in range(H):
        o, v = oracle.camera_row(sc, y)
        hit0[y], point[y], refl[y], occl[y] = oracle.bounce(sc, o, v)
        h1, _, dist, _ = oracle.closest(sc, point[y], refl[y])
        on = hit0[y] >= 0
        hit1[y][on] = h1[on]; d1[y][on] = dist[on]
    return hit0, point, refl, occl, hit1, d1


@pytest.mark.parametrize("name,w,h", [("teapot", 96, 54), ("mixed_xf", 80, 60), ("campanile_teddy", 64, 36),
                                     ("scene1", 64, 64), ("input-09", 48, 48), ("input-06", 64, 48),
                                     ("synthetic_small", 80, 45), ("normals", 64, 48), ("input-11", 64, 48)])
def test_shadow_bits_and_reflected_rays(name, w, h, oracle):
    """Beyond the camera rays: the hit point and the reflected direction of every camera ray (bit for bit: they
    feed the next level), which lights its shadow rays find blocked (Scene::traceShadowRay,
    raytracer.cpp:551-558), and the nearest hit of the reflected ray (index and metric bits), against the
    oracle's answers for the same rays, in every configuration of the trace kernels."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(scene_file(name), w, h, 1)
    hit0, point, refl, occl, hit1, d1 = oracle_bounce(oracle, sc)
    on = hit0 >= 0
    for label, opts in CONFIGS:
        r = Renderer(sc)
        for key, value in opts.items():
            r.set_option(getattr(abi, "RT_OPT_" + key), value)
        g_point, g_refl, g_occl, g_hit1, g_d1 = r.bounce_probe(sc.desc.n_lights)
        r.close()
        assert np.array_equal(g_point[on].view(np.uint64), point[on].view(np.uint64)), label
        assert np.array_equal(g_refl[on].view(np.uint64), refl[on].view(np.uint64)), label
        assert not g_point[~on].any() and not g_occl[~on].any()
        assert np.array_equal(g_occl[on], occl[on]), (label, int((g_occl[on] != occl[on]).sum()))
        assert np.array_equal(g_hit1[on], hit1[on]), (label, int((g_hit1[on] != hit1[on]).sum()))
        assert (g_hit1[~on] == -1).all()
        both = on & (hit1 >= 0)
        assert np.array_equal(g_d1[both].view(np.uint32), d1[both].view(np.uint32)), label


def test_deeper_levels_colour_against_the_oracle(oracle):
    """The colour a camera ray returns at depth 0, 1, 2 and 3 (Scene::traceRay's recursion, raytracer.cpp:539-544)
    from whole frames, pixel by pixel against the oracle's per-ray trace: the quantised value (raytracer.cpp:174-178)
    of the oracle's double colour must be the GPU's byte except where pow's last ulp sits on a quantisation step."""
    from raytracer_amd.render import Renderer
    for depth in (0, 1, 2, 3):
        sc = Scene.load(scene_file("input-11"), 64, 48, depth)
        r = Renderer(sc)
        r.render_frame()
        img = r.download()
        r.close()
        differing = 0
        for y in range(sc.height):
            o, v = oracle.camera_row(sc, y)
            rgb = oracle.trace(sc, o, v, depth)
            want = np.minimum(np.trunc(rgb * np.float32(255.0)), 255).astype(np.int64)
            got = img[sc.height - 1 - y, :, :3].astype(np.int64)
            differing += int((np.abs(want - got).max(axis=-1) > 0).sum())
            assert np.abs(want - got).max() <= 1, (depth, y)
        assert differing <= 2, (depth, differing)


IN_PLANE_SCENE = """cam 0 0 5  -1 -1 3  1 -1 3  -1 1 3  1 1 3
lta 0.5 0.5 0.5
ltp 0 4 4  0.8 0.8 0.8
mat 0.2 0.9 0.2  0.5 0.5 0.5  0 0 0  1  0 0 0
tri 10 0 -20  12 0 -20  11 0 -22
mat 0.9 0.2 0.2  0.5 0.5 0.5  0.3 0.3 0.3  8  0.2 0.2 0.2
sph 0 0 0 0.6
tri -3 -0.8 -3  3 -0.8 -3  0 -0.8 3
"""


def test_in_plane_origin_branch(oracle, tmp_path):
    """Triangle::intersection returns r = 0 / b -- a "hit" at the ray's own origin -- when the origin lies exactly in
    the triangle's plane (a == 0) and the ray grazes it (|b| < eps), wherever the triangle is (raytracer.cpp:319-321).
    Here the eye lies in the plane y = 0 of a triangle far outside the view: the scanlines next to the horizon
    take that branch for every pixel and show that triangle instead of the sphere in front of the camera.  Every
    configuration must reproduce the oracle's frame and nearest hits -- the LBVH too, whose boxes the rays never
    reach: the scan behind its traversal finds those pairs (RT_OPT_BVH_PLANE_GUARD); with that scan switched off
    the LBVH differs in exactly those pixels, which is what the scan is for."""
    from raytracer_amd.render import Renderer
    from tests.oracle_api import compare
    path = tmp_path / "in_plane.scene"
    path.write_text(IN_PLANE_SCENE)
    sc = Scene.load(str(path), 80, 80, 2)
    want, _ = oracle.render(sc)
    want_obj, want_d = oracle_hits(oracle, sc)
    rows = [y for y in range(80) if (want_obj[y] == 0).any()]
    # (the scanlines whose |dir.y| rounds below eps = 0.05f: two on either side of the horizon line y = 40, where
    # b = 0 makes r = 0 / 0 a NaN that is no hit, raytracer.cpp:525)
    assert rows == [38, 39, 41, 42] and (want_obj[rows] == 0).all() and (want_d[rows] == 0).all()
    for label, opts in CONFIGS + [("lbvh-binary", {"USE_BVH": 2})]:
        r = Renderer(sc)
        for key, value in opts.items():
            r.set_option(getattr(abi, "RT_OPT_" + key), value)
        obj, d = r.primary_hits()
        r.render_frame()
        img, st = r.download(), r.stats()
        r.close()
        assert np.array_equal(obj, want_obj), label
        assert np.array_equal(d[want_obj >= 0].view(np.uint32), want_d[want_obj >= 0].view(np.uint32)), label
        worst, ndiff, _ = compare(img, want)
        assert worst <= 1 and ndiff <= 2, (label, worst, ndiff)
        if label != "exact":
            assert st["in_plane_origin_hits"] >= len(rows) * 80, label
    r = Renderer(sc)
    r.set_option(abi.RT_OPT_USE_BVH, 1)
    r.set_option(abi.RT_OPT_BVH_PLANE_GUARD, 0)
    obj, _ = r.primary_hits()
    r.close()
    differs = obj != want_obj
    assert differs.any() and not differs[[y for y in range(80) if y not in rows]].any()


# The same branch taken by shadow rays and by reflected rays.  Every number is chosen so that the reference's
# arithmetic is exact: a 64 x 64 frame (u, v and 1 - u, 1 - v are exact floats), an image plane at x = 1 seen from
# the origin (every camera direction is (1, y, z)), a mirror wall of two triangles in the plane x = 2 whose edge
# cross products are powers of two (unit normal exactly (-1, 0, 0)): every hit point is (2, 2y, 2z) exactly.  A small
# triangle in the same plane, a thousand units away, then has a == 0 for every ray that leaves the wall; the
# reflected rays of the outer scanlines graze the plane (|b| = 1 / |(1, y, z)| < 0.05), and so does every shadow
# ray towards the directional light and towards the point light from far enough: the reference lets the far
# triangle (and the wall's other half) "hit" them at their own origin.
IN_PLANE_SECONDARY_SCENE = """cam 0 0 0  1 -40 -1  1 -40 1  1 40 -1  1 40 1
lta 0.3 0.3 0.3
ltd 0.02 -1 0  0.7 0.7 0.7
ltp 1.5 0 0  0.6 0.6 0.6
mat 0.1 0.9 0.1  0.4 0.4 0.4  0 0 0  1  0 0 0
tri 2 1000 0  2 1000 2  2 1002 0
mat 0.2 0.2 0.9  0.5 0.5 0.5  0.2 0.2 0.2  4  0.5 0.5 0.5
tri 2 -128 -128  2 -128 128  2 128 -128
tri 2 128 128  2 128 -128  2 -128 128
mat 0.9 0.2 0.2  0.5 0.5 0.5  0.3 0.3 0.3  8  0 0 0
sph 1.2 0 0 0.2
"""


def test_in_plane_origin_branch_of_shadow_and_reflected_rays(oracle, tmp_path):
    from raytracer_amd.render import Renderer
    from tests.oracle_api import compare
    path = tmp_path / "in_plane_secondary.scene"
    path.write_text(IN_PLANE_SECONDARY_SCENE)
    sc = Scene.load(str(path), 64, 64, 1)
    hit0, point, refl, occl, hit1, d1 = oracle_bounce(oracle, sc)
    on = hit0 >= 0
    wall = (hit0 == 1) | (hit0 == 2)
    # the construction works: wall hits lie exactly in the plane, the outer scanlines' reflected rays "hit" the far
    # triangle (object 0) at distance 0, and the far triangle blocks shadow rays of wall points
    assert wall.sum() > 3000 and (point[wall][:, 0] == 2.0).all()
    far_hits = wall & (hit1 == 0)
    assert far_hits.sum() > 1000 and (d1[far_hits] == 0).all()
    assert (occl[wall][:, 0] & 1).all()                      # (the directional light grazes the wall everywhere)
    sc2 = Scene.load(str(path), 64, 64, 2)
    want, counts = oracle.render(sc2)
    for label, opts in [c for c in CONFIGS if c[0] != "no-shared"] + [("lbvh-binary", {"USE_BVH": 2})]:
        r = Renderer(sc)
        for key, value in opts.items():
            r.set_option(getattr(abi, "RT_OPT_" + key), value)
        g_point, g_refl, g_occl, g_hit1, g_d1 = r.bounce_probe(sc.desc.n_lights)
        r.close()
        assert np.array_equal(g_point[on].view(np.uint64), point[on].view(np.uint64)), label
        assert np.array_equal(g_occl[on], occl[on]), (label, int((g_occl[on] != occl[on]).sum()))
        assert np.array_equal(g_hit1[on], hit1[on]), (label, int((g_hit1[on] != hit1[on]).sum()))
        both = on & (hit1 >= 0)
        assert np.array_equal(g_d1[both].view(np.uint32), d1[both].view(np.uint32)), label
        r = Renderer(sc2)
        for key, value in opts.items():
            r.set_option(getattr(abi, "RT_OPT_" + key), value)
        r.render_frame()
        img, st = r.download(), r.stats()
        r.close()
        worst, ndiff, _ = compare(img, want)
        assert worst <= 1 and ndiff <= 2, (label, worst, ndiff)
        for key in ("rays_reflect", "rays_shadow", "hits"):
            assert st[key] == counts[key], (label, key)
    # without the scan behind the traversal the LBVH misses exactly such hits
    r = Renderer(sc)
    r.set_option(abi.RT_OPT_USE_BVH, 1)
    r.set_option(abi.RT_OPT_BVH_PLANE_GUARD, 0)
    _, _, g_occl, g_hit1, _ = r.bounce_probe(sc.desc.n_lights)
    r.close()
    assert (g_hit1[on] != hit1[on]).any() or (g_occl[on] != occl[on]).any()


def test_in_plane_origin_question_is_decided_by_the_frame(tmp_path):
    """RT_OPT_IN_PLANE_ORIGIN = 2 (default): the first render call first traces the whole frame with the question asked of
    every ray, and keeps asking iff some pair took the branch.  A scene without such a pair is then traced without the
    naming machinery -- same frame, same counts, fewer pairs for the re-test -- and a scene with one keeps it (the
    tests above run in this mode).  Until a frame has been traced the question is asked; rt_primary_hits does not
    start the calibration; the explicit settings win."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 160, 90, 4)
    always = Renderer(sc)
    always.set_option(abi.RT_OPT_IN_PLANE_ORIGIN, 1)
    always.render_frame()
    want, st_always = always.download(), always.stats()
    always.close()
    assert st_always["bvh_plane_guard"] == 1 and st_always["in_plane_origin_hits"] == 0
    for bvh in (0, 1):
        r = Renderer(sc)
        assert r.stats()["bvh_plane_guard"] == 1          # (not decided yet: asked)
        r.primary_hits()
        assert r.stats()["bvh_plane_guard"] == 1
        if bvh:
            r.set_option(abi.RT_OPT_USE_BVH, 1)             # (the calibration frame is traced here, through the LBVH)
            assert r.stats()["bvh_plane_guard"] == 0
        r.reset_stats()
        r.render_frame()
        got, st = r.download(), r.stats()
        assert st["bvh_plane_guard"] == 0
        assert np.array_equal(got, want)
        for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits", "closest_tests", "anyhit_tests"):
            assert st[k] == st_always[k], k                 # (one frame's counts: the calibration frame's are not in them)
        if not bvh:
            assert st["exact_tests"] <= st_always["exact_tests"]
        r.set_option(abi.RT_OPT_IN_PLANE_ORIGIN, 1)
        r.render_frame()
        assert r.stats()["bvh_plane_guard"] == 1 and np.array_equal(r.download(), want)
        r.set_option(abi.RT_OPT_IN_PLANE_ORIGIN, 2)
        assert r.stats()["bvh_plane_guard"] == 0          # (decided once per context)
        r.close()
    path = tmp_path / "in_plane.scene"
    path.write_text(IN_PLANE_SCENE)
    r = Renderer(Scene.load(str(path), 80, 80, 2))
    r.render_frame()
    st = r.stats()
    assert st["bvh_plane_guard"] == 1 and st["in_plane_origin_hits"] >= 4 * 80
    r.close()


# A sphere over a floor of two triangles, two lights, and one triangle of no area (two corners coincide) far behind
# the camera.
VOID_NORMAL_SCENE = """cam 0 1 6  -2 -1 3  2 -1 3  -2 2 3  2 2 3
lta 0.2 0.25 0.3
ltp 3 5 4  1 1 1
ltd 0.2 -1 -0.3  0.6 0.6 0.6
mat 0.3 0.2 0.1  0.6 0.5 0.4  0.5 0.5 0.5  20  0.4 0.4 0.4
sph 0 0.2 0 1
mat 0.2 0.3 0.2  0.5 0.6 0.5  0.2 0.2 0.2  8  0.3 0.3 0.3
tri -6 -0.8 -6  -6 -0.8 6  6 -0.8 6
tri -6 -0.8 -6  6 -0.8 6  6 -0.8 -6
obj teapot.obj
mat 0.5 0.5 0.5  0.5 0.5 0.5  0.5 0.5 0.5  2  0 0 0
tri 0 0 50  0 0 50  1 1 50
"""


def test_a_triangle_of_no_area_shadows_every_point(oracle, tmp_path, monkeypatch):
    """The unit normal of a triangle without area is 0 / 0 (raytracer.cpp:313), Triangle::intersection then returns a
    NaN for every ray, and NaN != MAXFLOAT makes the triangle an occluder of every shadow ray (raytracer.cpp:553) while
    it is never a nearest hit (raytracer.cpp:525): the frame is the ambient term plus reflections of ambient terms.
    The library answers that at upload -- no shadow ray is tested against anything -- and must give the frame of
    the long way round (RAYTRACER_AMD_TEST_VOID_NORMALS=1: the triangle is tested against every ray) and the oracle's
    shadow bits, in every configuration."""
    import shutil
    from raytracer_amd.render import Renderer
    from tests.oracle_api import compare
    shutil.copy(os.path.join(ROOT, "scenes", "teapot.obj"), tmp_path / "teapot.obj")
    path = tmp_path / "void_normal.scene"
    path.write_text(VOID_NORMAL_SCENE)
    sc = Scene.load(str(path), 96, 72, 3)
    want, _ = oracle.render(sc)
    hit0, _, _, want_occl, _, _ = oracle_bounce(oracle, sc)
    on = hit0 >= 0
    frames, stats = {}, {}
    for long_way in ("1", "0"):
        monkeypatch.setenv("RAYTRACER_AMD_TEST_VOID_NORMALS", long_way)
        for label, opts in CONFIGS:
            r = Renderer(sc)
            for key, value in opts.items():
                r.set_option(getattr(abi, "RT_OPT_" + key), value)
            r.render_frame()
            frames[long_way, label], stats[long_way, label] = r.download(), r.stats()
            occl = r.bounce_probe(sc.desc.n_lights)[2]
            r.close()
            assert np.array_equal(occl[on], want_occl[on]) and (want_occl[on] == 3).all() and on.any(), (long_way, label)
            worst, ndiff, _ = compare(frames[long_way, label], want)
            assert worst <= 1, (long_way, label, worst, ndiff)
    for label, _ in CONFIGS:
        assert np.array_equal(frames["1", label], frames["0", label]), label
        a, b = stats["1", label], stats["0", label]
        for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
            assert a[k] == b[k] and a[k] > 0, (label, k)
        if label != "exact":
            assert b["exact_tests"] < a["exact_tests"], label
    # every point in shadow: no light's diffuse or specular term anywhere, so a frame without lights is the same frame
    dark = tmp_path / "dark.scene"
    dark.write_text("\n".join(l for l in VOID_NORMAL_SCENE.splitlines() if not l.startswith(("ltp", "ltd"))) + "\n")
    r = Renderer(Scene.load(str(dark), 96, 72, 3))
    r.render_frame()
    assert np.array_equal(r.download(), frames["0", "static"])
    r.close()


def many_spheres_scene():
    """48 spheres and ellipsoids (rotated, stretched up to 5 : 1, some far away, one huge, one with a negative scale)
    over a floor, two point lights and a directional one; no random numbers."""
    lines = ["cam 0 2 12  -4 -1.5 7  4 -1.5 7  -4 3 7  4 3 7", "lta 0.1 0.1 0.12",
             "ltp 6 9 8  40 38 36  2", "ltp -7 5 -2  1 1 1", "ltd 0.4 -1 -0.2  0.5 0.5 0.5"]
    for k in range(48):
        a = 0.61803398875 * k
        x, z = 5.5 * np.cos(2 * np.pi * a) * (0.3 + (k % 7) / 7), -3 + 6.0 * np.sin(2 * np.pi * a) * (0.3 + (k % 5) / 5)
        y = -0.4 + 0.35 * (k % 9)
        lines.append("mat 0.1 0.1 0.1  %.2f %.2f %.2f  0.5 0.5 0.5  %d  %.1f %.1f %.1f"
                     % (0.3 + 0.1 * (k % 6), 0.3 + 0.1 * (k % 4), 0.3 + 0.1 * (k % 5), 4 + k, 0.3 * (k % 3), 0.3 * (k % 3), 0.3 * (k % 3)))
        if k % 3 == 0:
            lines.append("sph %.4f %.4f %.4f %.3f" % (x, y, z, 0.25 + 0.05 * (k % 4)))
        else:
            lines += ["xft %.4f %.4f %.4f" % (x, y, z), "xfr %d %d %d" % (17 * k % 360, 29 * k % 360, 41 * k % 360),
                      "xfs %.2f %.2f %.2f" % (0.2 + 0.2 * (k % 5), (-1 if k == 7 else 1) * (0.25 + 0.05 * (k % 3)), 0.2 + 0.15 * (k % 4)),
                      "sph 0 0 0 1" if k % 2 else "sph 0.3 -0.2 0.1 0.8", "xfz"]
    lines += ["mat 0.1 0.1 0.1  0.4 0.4 0.5  0.3 0.3 0.3  8  0.5 0.5 0.5",
              "sph 0 -60 -40 55", "sph 300 80 -900 40",
              "mat 0.1 0.1 0.1  0.5 0.5 0.4  0.2 0.2 0.2  8  0.4 0.4 0.4",
              "tri -9 -1 -9  -9 -1 9  9 -1 9", "tri -9 -1 -9  9 -1 9  9 -1 -9"]
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("name,w,h,depth", [("many_spheres", 160, 90, 4), ("many_spheres", 61, 37, 2), ("ellipse.scene", 128, 72, 5),
                                            ("mixed_xf.scene", 120, 90, 4), ("synthetic_small.scene", 96, 54, 3), ("scene1.scene", 64, 64, 3)])
def test_spheres_dropped_wave_by_wave(name, w, h, depth, oracle, tmp_path, monkeypatch):
    """k_brute tests every ray against every sphere (Scene::intersection, raytracer.cpp:521-531); in a scene of many it first
    asks, per wave of 64 neighbouring rays, which spheres the cone around those rays can meet at all.  What it drops must be
    misses: the frame, the counts and the shadow bits with the question asked of every scene with a sphere
    (RAYTRACER_AMD_SPHERE_CULL=1) equal those without it (=0), brute force and through the LBVH, and the oracle's."""
    from raytracer_amd.render import Renderer
    from tests.oracle_api import compare
    if name == "many_spheres":
        path = tmp_path / "many_spheres.scene"
        path.write_text(many_spheres_scene())
        path = str(path)
    else:
        path = os.path.join(ROOT, "scenes", name)
    sc = Scene.load(path, w, h, depth)
    assert sc.desc.n_objects > 0
    want, _ = oracle.render(sc)
    got = {}
    for cull in ("0", "1"):
        monkeypatch.setenv("RAYTRACER_AMD_SPHERE_CULL", cull)
        for bvh in (0, 1):
            r = Renderer(sc)
            r.set_option(abi.RT_OPT_USE_BVH, bvh)
            r.reset_stats()
            r.render_frame()
            st = r.stats()
            got[cull, bvh] = (r.download(), tuple(st[k] for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits")),
                              r.bounce_probe(sc.desc.n_lights), st["exact_tests"])
            r.close()
    for bvh in (0, 1):
        a, b = got["0", bvh], got["1", bvh]
        assert np.array_equal(a[0], b[0]), (bvh, int((a[0] != b[0]).any(axis=-1).sum()))
        assert a[1] == b[1], bvh
        for x, y in zip(a[2], b[2]):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), bvh
        if bvh == 0:            # (through the LBVH the spheres are then asked ahead of the traversal, also by rays a triangle shadows)
            assert b[3] <= a[3]
    worst, ndiff, frac1 = compare(got["1", 0][0], want)
    assert worst <= 4 and frac1 >= 0.999, (worst, ndiff, frac1)      # (per-channel difference <= 1 on 99.9 % of the pixels, <= 4 everywhere)
