"""GPU trace loop (through the C ABI) against the CPU oracle and the reference's goldens.

Bars, from the task's tolerance: per-channel 8-bit difference <= 1 on >= 99.9 % of pixels
and <= 4 everywhere.  Every accept/reject decision and every hit distance is computed in
the reference's own arithmetic on the GPU, so the tests also demand that at most a
handful of pixels differ at all (libm pow is the only operation that is not bit-defined).
Ray and hit COUNTS are integer work and must match the oracle exactly.
"""
import glob
import os
import re

import numpy as np
import pytest

from raytracer_amd import Scene, abi
from tests import oracle_api

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "ref_renders", "*.png")))


def scene_file(name):
    for cand in (f"scenes/inputs/{name}", f"scenes/{name}.scene"):
        p = os.path.join(ROOT, cand)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


def gpu_render(sc, exact_only=False, pass_pixels=None, tiles=None, filter_form=None, bvh=False, shared=None,
               bounding=None, scan_copies=None):
    from raytracer_amd.render import Renderer
    r = Renderer(sc)
    if scan_copies is not None:
        r.set_option(abi.RT_OPT_SCAN_COPIES, scan_copies)
    if shared is not None:
        r.set_option(abi.RT_OPT_SHARED_CONSTANTS, shared)
    if bounding is not None:
        r.set_option(abi.RT_OPT_BOUNDING_TESTS, bounding)
    if filter_form is not None:
        r.set_option(abi.RT_OPT_FILTER_FORM, filter_form)
    if bvh:
        r.set_option(abi.RT_OPT_USE_BVH, 1)
    if exact_only:
        r.set_option(abi.RT_OPT_EXACT_ONLY, 1)
    if pass_pixels:
        r.set_option(abi.RT_OPT_PASS_PIXELS, pass_pixels)
    if tiles is None:
        r.render_frame()
    else:
        tile_rows, world = tiles
        for rank in range(world):
            r.render_tiles(rank, tile_rows, world)
    img = r.download()
    st = r.stats()
    r.close()
    return img, st


def assert_parity(img, want, max_diff_pixels):
    worst, ndiff, frac1 = oracle_api.compare(img, want)
    assert worst <= 4 and frac1 >= 0.999, (worst, ndiff, frac1)
    assert ndiff <= max_diff_pixels, (worst, ndiff, frac1)
    assert np.array_equal(img[..., 3], want[..., 3])


def parse_gold(path):
    m = re.match(r"(.+)_(\d+)x(\d+)_d(\d+)\.png$", os.path.basename(path))
    return m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))


@pytest.mark.parametrize("gold", GOLD, ids=[os.path.basename(g)[:-4] for g in GOLD])
def test_gpu_equals_reference_golden(gold):
    name, w, h, depth = parse_gold(gold)
    sc = Scene.load(scene_file(name), w, h, depth)
    img, _ = gpu_render(sc)
    assert_parity(img, oracle_api.load_png(gold), max_diff_pixels=2)


@pytest.mark.parametrize("name,w,h,depth", [("mixed_xf", 160, 120, 4), ("input-09", 100, 80, 6), ("s3", 120, 90, 6),
                                           ("normals", 120, 90, 3), ("teapot", 64, 36, 2),
                                           ("campanile_teddy", 48, 27, 1),
                                           # spheres and ellipsoids: the FP32 pre-test of k_brute
                                           ("ellipse", 192, 108, 6), ("input-06", 128, 96, 6), ("journal2", 128, 96, 6),
                                           ("synthetic_small", 96, 54, 4), ("input-07", 96, 96, 6), ("input-08", 96, 96, 6)])
def test_exact_only_mode_equals_filtered_mode(name, w, h, depth):
    """The FP32 filter may only drop pairs the reference-precision test would reject:
    switching it off must not change a single byte, nor any count."""
    sc = Scene.load(scene_file(name), w, h, depth)
    b, sb = gpu_render(sc, exact_only=True)
    # both forms of the filter (static margins / per-test margins), brute force and LBVH
    for form in (1, 2):
        for bvh in (False, True):
            a, sa = gpu_render(sc, filter_form=form, bvh=bvh)
            assert sa["filter_form"] == form
            assert np.array_equal(a, b), (form, bvh)
            for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
                assert sa[k] == sb[k], (k, form, bvh)
            assert sb["exact_tests"] >= sa["exact_tests"]


@pytest.mark.parametrize("name,w,h,depth", [("mixed_xf", 160, 120, 4), ("teapot", 96, 54, 3), ("input-06", 128, 96, 6),
                                           ("campanile_teddy", 64, 36, 2)])
def test_shared_origin_and_direction_constants_change_nothing(name, w, h, depth):
    """Camera rays share their origin and the shadow rays towards a directional light their
    direction; the brute-force filter then reads host-computed per-triangle constants instead
    of recomputing them per ray.  Same frame, same counts, with and without."""
    sc = Scene.load(scene_file(name), w, h, depth)
    for form in (1, 2):
        a, sa = gpu_render(sc, filter_form=form, shared=1)
        b, sb = gpu_render(sc, filter_form=form, shared=0)
        assert np.array_equal(a, b), form
        for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
            assert sa[k] == sb[k], (k, form)


@pytest.mark.parametrize("name,w,h,depth", [("teapot", 1280, 720, 5), ("mixed_xf", 640, 480, 6), ("input-06", 512, 384, 6),
                                           ("campanile_teddy", 480, 270, 3), ("synthetic_small", 480, 270, 4),
                                           ("normals", 320, 240, 4), ("journal2", 320, 240, 6), ("scene1", 320, 240, 5),
                                           ("s2", 320, 240, 5)])
def test_scan_copies_change_nothing(name, w, h, depth):
    """The kernels that make a ray also write the FP32 values the bounded scans hold for it (RT_OPT_SCAN_COPIES);
    the scans load those instead of converting the ray's doubles.  The two must be the same bits: same frame,
    same ray and hit counts, and the same pairs listed for the reference-precision re-test, in both forms of
    the filter (scenes with identity and non-identity transforms, point and directional lights)."""
    sc = Scene.load(scene_file(name), w, h, depth)
    for form in (1, 2):
        a, sa = gpu_render(sc, filter_form=form, scan_copies=1)
        b, sb = gpu_render(sc, filter_form=form, scan_copies=0)
        assert np.array_equal(a, b), form
        # (a pair list that runs full sends rays through the full scan instead; how many pairs were listed before
        # that depends on the order in which the waves arrive: the listed pairs are comparable only without it)
        full = sa["flagged_rays"] or sb["flagged_rays"]
        for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits") + (() if full else ("exact_tests", "in_plane_origin_hits")):
            assert sa[k] == sb[k], (k, form, sa[k], sb[k])


@pytest.mark.parametrize("name,w,h,depth", [("teapot", 1280, 720, 5), ("mixed_xf", 640, 480, 6), ("input-06", 512, 384, 6),
                                           ("campanile_teddy", 480, 270, 3), ("synthetic_small", 480, 270, 4),
                                           ("normals", 320, 240, 4), ("journal2", 320, 240, 6)])
def test_bounding_tests_change_nothing(name, w, h, depth):
    """Where the rays of a launch pass through one point (camera rays, shadow rays towards a point
    light) or are parallel (a directional light) the scan asks a bounding-sphere question before
    the triangle test.  Same frame and counts with and without, in both forms of the filter; the
    triangle test itself must see no more pairs than without (the question can only remove some)."""
    sc = Scene.load(scene_file(name), w, h, depth)
    for form in (1, 2):
        a, sa = gpu_render(sc, filter_form=form, bounding=1)
        b, sb = gpu_render(sc, filter_form=form, bounding=0)
        assert np.array_equal(a, b), form
        for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits", "closest_tests", "anyhit_tests", "in_plane_origin_hits"):
            assert sa[k] == sb[k], (k, form)
        d, sd = gpu_render(sc, filter_form=form, bounding=2)      # only where the rays share a point or a direction
        assert np.array_equal(a, d), form
        assert sd["in_plane_origin_hits"] == sb["in_plane_origin_hits"]
        if sa["flagged_rays"] == sd["flagged_rays"] == sb["flagged_rays"] == 0:
            # A bounding test can only take pairs away from the list.  (How many depends on which
            # rays share a lane, i.e. on the order in which the waves of k_hit appended their hits,
            # so the two bounded counts are not comparable with each other; a ray whose pairs
            # overflow the list is scanned in full, which counts every triangle.)
            assert sa["exact_tests"] <= sb["exact_tests"] and sd["exact_tests"] <= sb["exact_tests"]
    c, _ = gpu_render(sc, exact_only=True)
    assert np.array_equal(a, c)


@pytest.mark.parametrize("name,w,h,depth,bvh", [("teapot", 320, 180, 5, False), ("mixed_xf", 320, 240, 6, False),
                                               ("teapot", 320, 180, 5, True), ("input-06", 256, 192, 6, False)])
def test_two_streams_give_the_frame_of_one(name, w, h, depth, bvh):
    """A level's shadow launches run on an internal stream beside the next level's nearest-hit
    launches; with RT_OPT_OVERLAP off everything runs on the caller's stream.  Same frame and
    counts either way, also over several passes and repeated frames."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(scene_file(name), w, h, depth)
    frames = []
    for overlap in (1, 0):
        r = Renderer(sc)
        r.set_option(abi.RT_OPT_OVERLAP, overlap)
        r.set_option(abi.RT_OPT_PASS_PIXELS, w * 40)        # several passes per frame
        if bvh:
            r.set_option(abi.RT_OPT_USE_BVH, 1)
        for _ in range(3):
            r.render_frame()
        frames.append((r.download(), r.stats()))
        r.close()
    assert np.array_equal(frames[0][0], frames[1][0])
    for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
        assert frames[0][1][k] == frames[1][1][k], k


@pytest.mark.parametrize("name,w,h,depth", [("teapot", 640, 360, 5), ("mixed_xf", 320, 240, 4), ("input-11", 200, 160, 6)])
def test_work_counters_change_nothing_and_add_up(name, w, h, depth):
    """RT_OPT_WORK_COUNTERS runs the scans in instances that count what they execute.  Same frame, same ray
    counts; the counts are consistent with each other and with the nominal pair counts."""
    from raytracer_amd.render import Renderer
    sc = Scene.load(scene_file(name), w, h, depth)
    base, sb = gpu_render(sc)
    for bounding in (1, 0):
        r = Renderer(sc)
        r.set_option(abi.RT_OPT_BOUNDING_TESTS, bounding)
        r.set_option(abi.RT_OPT_WORK_COUNTERS, 1)
        r.render_frame()
        img, st = r.download(), r.stats()
        r.close()
        assert np.array_equal(img, base)
        for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
            assert st[k] == sb[k], k
        work = st["scan_work"]
        n_tri = st["n_triangles"] - st["n_irregular"]
        if n_tri == 0:
            continue
        for kind, wk in work.items():
            assert wk["tri_tests_wanted"] <= wk["tri_tests"], kind
            assert wk["groups_live"] <= wk["groups"], kind
            assert wk["batches_tested"] <= wk["batches"] or kind.startswith("plain"), kind
        if bounding == 0:
            # every pair takes the triangle test: whole waves of 64 lanes x 8 ray slots per triangle, so the
            # executed count is the nominal one rounded up to whole tasks
            rays0 = st["rays_primary"] + st["rays_reflect"]
            assert work["plain_nearest"]["tri_tests"] >= rays0 * n_tri
            assert work["plain_shadow"]["tri_tests"] >= st["rays_shadow"] * n_tri
            assert work["camera"]["tasks"] == work["reflection"]["tasks"] == 0
        else:
            assert work["plain_nearest"]["tasks"] == 0 and work["camera"]["tasks"] > 0
            # a bounded scan executes fewer triangle tests than there are pairs
            if n_tri >= 256:
                assert work["camera"]["tri_tests"] < st["rays_primary"] * n_tri


def test_filter_form_follows_the_scene():
    """Large smooth triangles take the cheap static margins; a mesh of triangles that are tiny
    against its scene, or that has walls of coplanar triangles, takes per-test margins and then
    lists only a few pairs per ray instead of overflowing into the full scan.  In between -- the teapot, whose static
    margins are some per cent of a triangle -- the choice follows the configuration: per-test margins where bounding
    tests decide most pairs (fewer pairs for the re-test), static ones where every pair takes the triangle test."""
    from raytracer_amd.render import Renderer
    r = Renderer(Scene.load(scene_file("teapot"), 96, 54, 2))
    r.render_frame()
    bounded, st = r.download(), r.stats()
    assert st["filter_form"] == 2 and st["flagged_rays"] == 0
    listed_per_test = st["exact_tests"]
    r.set_option(abi.RT_OPT_BOUNDING_TESTS, 0)
    r.reset_stats()
    r.render_frame()
    st = r.stats()
    assert st["filter_form"] == 1 and st["flagged_rays"] == 0 and np.array_equal(r.download(), bounded)
    assert st["exact_tests"] > listed_per_test
    r.set_option(abi.RT_OPT_FILTER_FORM, 1)          # ... and the explicit choice stays when the configuration changes
    r.set_option(abi.RT_OPT_BOUNDING_TESTS, 1)
    r.render_frame()
    assert r.stats()["filter_form"] == 1 and np.array_equal(r.download(), bounded)
    r.set_option(abi.RT_OPT_FILTER_FORM, 0)
    r.render_frame()
    assert r.stats()["filter_form"] == 2 and np.array_equal(r.download(), bounded)
    r.close()
    _, st = gpu_render(Scene.load(scene_file("s1"), 96, 54, 2))                  # two large triangles
    assert st["filter_form"] == 1 and st["flagged_rays"] == 0
    _, st = gpu_render(Scene.load(scene_file("campanile_teddy"), 192, 108, 2))
    assert st["filter_form"] == 2 and st["flagged_rays"] == 0
    rays = st["rays_primary"] + st["rays_reflect"] + st["rays_shadow"]
    # exact tests: the 50-odd irregular triangles for every ray, plus a few listed pairs
    assert st["exact_tests"] < rays * (st["n_irregular"] + 16)


@pytest.mark.parametrize("name,w,h,depth", [("input-11", 100, 80, 6), ("mixed_xf", 80, 60, 6), ("teapot", 48, 27, 5),
                                           ("scene1", 64, 48, 6), ("input-06", 128, 96, 6)])
def test_counts_match_oracle(name, w, h, depth, oracle):
    sc = Scene.load(scene_file(name), w, h, depth)
    img, st = gpu_render(sc)
    want, c = oracle.render(sc)
    assert_parity(img, want, max_diff_pixels=2)
    assert st["rays_primary"] == c["rays_primary"] == w * h
    assert st["rays_reflect"] == c["rays_reflect"]
    assert st["rays_shadow"] == c["rays_shadow"]
    assert st["hits"] == c["hits"]
    assert st["closest_tests"] == c["closest_tests"]


# every scene / render pair the reference's authors left behind (1000 x 1000, depth 6)
AUTHORS = [("input-%02d" % k, "image-%02d" % k) for k in range(1, 12)] + [
    ("scene1", "scene1"), ("sphere", "sphere2"), ("s1", "output"), ("s2", "output-01")]


@pytest.mark.parametrize("scene,image", AUTHORS, ids=[a[0] for a in AUTHORS])
def test_gpu_against_authors_renders(scene, image):
    sc = Scene.load(scene_file(scene), 1000, 1000)
    img, _ = gpu_render(sc)
    want = oracle_api.load_png(os.path.join(ROOT, "tests", "golden", "images", image + ".png"))
    assert_parity(img, want, max_diff_pixels=3)


def test_tiles_and_small_passes_give_the_same_frame():
    sc = Scene.load(scene_file("mixed_xf"), 96, 70, 4)      # 70 rows: last tile is clipped
    whole, s0 = gpu_render(sc)
    tiled, s1 = gpu_render(sc, tiles=(8, 3))               # 3 "ranks", interleaved 8-row tiles
    small, s2 = gpu_render(sc, pass_pixels=96 * 5)         # many passes of 5 rows
    assert np.array_equal(whole, tiled)
    assert np.array_equal(whole, small)
    for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
        assert s0[k] == s1[k] == s2[k], k


@pytest.mark.parametrize("w,h", [(75, 41), (33, 16), (15, 50), (131, 7), (1, 1), (16, 16), (17, 17)])
def test_frame_sizes_with_partial_queue_tiles(w, h, oracle):
    # the ray queue is in 16 x 16 tile order; these sizes leave partial tiles in both directions
    sc = Scene.load(scene_file("mixed_xf"), w, h, 4)
    img, st = gpu_render(sc)
    want, c = oracle.render(sc)
    assert_parity(img, want, max_diff_pixels=2)
    for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
        assert st[k] == c[k], k
    tiled, _ = gpu_render(sc, tiles=(3, 2))
    assert np.array_equal(img, tiled)


def test_partial_render_leaves_other_rows_black():
    from raytracer_amd.render import Renderer
    sc = Scene.load(scene_file("input-05"), 64, 48)
    r = Renderer(sc)
    r.render_tile(10, 20)
    part = r.download()
    r.render_tile(0, 48)
    full = r.download()
    r.close()
    assert np.array_equal(part[18:38], full[18:38])         # film rows 10..29 = frame rows 18..37
    assert not part[:18].any() and not part[38:].any()


def test_edge_cases():
    from raytracer_amd.render import Renderer, DeviceError
    # empty scene: black frame, alpha 255 where rendered
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "e.scene")
        open(p, "w").write("cam 0 0 2 -1 -1 1 1 -1 1 -1 1 1 1 1 1\nlta 1 1 1\n")
        sc = Scene.load(p, 16, 9)
        img, st = gpu_render(sc)
        assert not img[..., :3].any() and (img[..., 3] == 255).all()
        assert st["rays_primary"] == 144 and st["hits"] == 0
        # objects but no lights: ambient only
        open(p, "w").write("cam 0 0 2 -1 -1 1 1 -1 1 -1 1 1 1 1 1\nlta .5 .5 .5\n"
                           "mat 1 .5 .25 1 1 1 1 1 1 5 0 0 0\nsph 0 0 0 .5\n")
        sc = Scene.load(p, 32, 32, 0)
        img, st = gpu_render(sc)
        assert set(np.unique(img[..., 0])) == {0, 127} and st["rays_shadow"] == 0
    sc = Scene.load(scene_file("input-01"), 8, 8)
    r = Renderer(sc)
    with pytest.raises(DeviceError):
        r.render_tile(4, 8)          # outside the frame
    with pytest.raises(DeviceError):
        r.set_option(99, 1)
    r.close()


def test_workspace_that_cannot_be_sized_fails_every_time_and_launches_nothing(tmp_path):
    """One row of shadow rays beyond 2^31 (a very wide frame under very many lights) cannot be numbered:
    the render call fails with RT_ERR_RANGE, and so does the next one -- the context must not come to
    believe that its workspace exists.  A context whose workspace does fit still renders afterwards."""
    from raytracer_amd.render import Renderer, DeviceError
    lines = ["cam 0 0 2 -1 -1 1 1 -1 1 -1 1 1 1 1 1", "lta 1 1 1", "mat 1 1 1 1 1 1 1 1 1 5 0 0 0", "sph 0 0 0 .5"]
    lines += ["ltd 0 -1 %d 0.001 0.001 0.001" % (k + 1) for k in range(2048)]
    path = os.path.join(str(tmp_path), "wide.scene")
    open(path, "w").write("\n".join(lines) + "\n")
    r = Renderer(Scene.load(path, 1 << 20, 1, 0))
    for _ in range(2):
        with pytest.raises(DeviceError) as e:
            r.render_tile(0, 1)
        assert "(-3)" in str(e.value)
    with pytest.raises(DeviceError):
        r.primary_hits(0, 1)
    r.close()
    img, st = gpu_render(Scene.load(path, 64, 4, 0))
    assert st["rays_shadow"] == 2048 * st["hits"] and st["hits"] > 0


def test_transform_that_is_not_affine_is_refused():
    """The margins of the FP32 filter are derived without a perspective divide: a transform whose last
    row is not 0 0 0 1 is an argument error, not a wrong picture."""
    import ctypes as C
    from raytracer_amd.render import Renderer, DeviceError
    sc = Scene.load(scene_file("mixed_xf"), 16, 12, 1)
    assert sc.desc.n_transforms > 1
    keep = sc.desc.transforms[1].m[12]
    sc.desc.transforms[1].m[12] = 0.25
    with pytest.raises(DeviceError) as e:
        Renderer(sc)
    assert "(-1)" in str(e.value) and "affine" in str(e.value)
    sc.desc.transforms[1].m[12] = keep
    Renderer(sc).close()


def test_more_than_32_lights(oracle, tmp_path):
    """Occlusion bits take one 32-bit word per 32 lights: 40 point and directional lights over a
    small mesh and a sphere, brute force and LBVH, against the oracle."""
    lines = ["cam 0 1 8  -3.2 -1.4 4  3.2 -1.4 4  -3.2 2.2 4  3.2 2.2 4", "lta 0.05 0.05 0.05"]
    for k in range(40):
        ang = 2 * np.pi * k / 40
        if k % 5 == 4:
            lines.append(f"ltd {np.cos(ang):.4f} -1 {np.sin(ang):.4f}  0.03 0.03 0.03")
        else:
            lines.append(f"ltp {6 * np.cos(ang):.4f} {3 + (k % 3)} {6 * np.sin(ang):.4f}  0.06 0.05 0.04  {k % 3}")
    lines += ["mat 0.1 0.1 0.1  0.6 0.5 0.4  0.5 0.5 0.5  20  0.3 0.3 0.3", "sph -1.2 0.4 0.5 0.7",
              "xft 1.0 0 0", "xfs 0.8 0.8 0.8", "obj %s" % os.path.join(ROOT, "scenes", "dodec.obj"), "xfz",
              "tri -6 -1 -6  -6 -1 6  6 -1 6", "tri -6 -1 -6  6 -1 6  6 -1 -6"]
    path = os.path.join(str(tmp_path), "lights.scene")
    open(path, "w").write("\n".join(lines) + "\n")
    sc = Scene.load(path, 96, 72, 3)
    want, c = oracle.render(sc, threads=16)
    for bvh in (False, True):
        img, st = gpu_render(sc, bvh=bvh)
        assert_parity(img, want, max_diff_pixels=3)
        assert st["rays_shadow"] == c["rays_shadow"] == 40 * c["hits"]


def test_cli_driver_writes_the_reference_image(tmp_path):
    """The C++ host driver (the reference's main) end to end: scene file -> PNG."""
    import subprocess
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    assert os.path.exists(exe), "run `make cli`"
    out = tmp_path / "out.png"
    gold = os.path.join(ROOT, "tests", "golden", "ref_renders", "mixed_xf_160x120_d4.png")
    res = subprocess.run([exe, os.path.join(ROOT, "scenes", "mixed_xf.scene"), str(out), "-w", "160", "-h", "120",
                          "-depth", "4", "-stats"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert_parity(oracle_api.load_png(str(out)), oracle_api.load_png(gold), max_diff_pixels=2)
    # defaults are the reference's: 1000 x 1000, depth 6, eps 0.05 (raytracer.cpp:849-851, 477)
    out2 = tmp_path / "d.png"
    res = subprocess.run([exe, os.path.join(ROOT, "scenes", "inputs", "input-03"), str(out2)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    want = oracle_api.load_png(os.path.join(ROOT, "tests", "golden", "images", "image-03.png"))
    assert_parity(oracle_api.load_png(str(out2)), want, max_diff_pixels=3)


def test_cli_driver_n_gpu_path_with_one_gpu(tmp_path):
    """`rt_render -gpus 1`: the N-GPU path of the C++ driver (ncclCommInitAll, one host thread per GPU,
    interleaved tiles into a caller-owned frame, the grouped ncclSend / ncclRecv exchange -- empty for one
    rank) must write the picture of the single-GPU path."""
    import subprocess
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    assert os.path.exists(exe), "run `make cli`"
    scene = os.path.join(ROOT, "scenes", "mixed_xf.scene")
    one, multi = tmp_path / "one.png", tmp_path / "multi.png"
    common = ["-w", "160", "-h", "117", "-depth", "4"]           # 117 rows: the last 8-row tile is clipped
    res = subprocess.run([exe, scene, str(one)] + common, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    res = subprocess.run([exe, scene, str(multi), "-gpus", "1", "-tile", "8", "-stats"] + common, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert '"gpus": 1' in res.stdout
    assert np.array_equal(oracle_api.load_png(str(one)), oracle_api.load_png(str(multi)))
    # more GPUs than the box has: a clean error
    res = subprocess.run([exe, scene, str(multi), "-gpus", "63"] + common, capture_output=True, text=True)
    assert res.returncode == 1 and "visible" in res.stderr



def test_cli_driver_frames_in_flight(tmp_path):
    """`rt_render -frames F -inflight K`: the C++ driver's frame loop (a buffer per call in flight, rt_join before
    each frame's exchange) must write the picture of a single render, and count F frames' rays."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    assert os.path.exists(exe), "run `make cli`"
    scene = os.path.join(ROOT, "scenes", "teapot.scene")
    common = ["-w", "320", "-h", "181", "-depth", "5", "-stats"]
    one = tmp_path / "one.png"
    res = subprocess.run([exe, scene, str(one)] + common, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    rays1 = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{"render_s"')][0])
    for frames, in_flight, extra in ((7, 3, []), (2, 4, ["-bvh"]), (5, 1, ["-tile", "8"]), (1, 2, [])):
        out = tmp_path / f"f{frames}_{in_flight}.png"
        res = subprocess.run([exe, scene, str(out), "-frames", str(frames), "-inflight", str(in_flight)] + extra + common,
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        st = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{"render_s"')][0])
        assert st["frames"] == frames and st["frames_in_flight"] == in_flight and st["gpus"] == 1
        for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits"):
            assert st[k] == frames * rays1[k], (k, frames, in_flight)
        assert np.array_equal(oracle_api.load_png(str(out)), oracle_api.load_png(str(one))), (frames, in_flight)
    res = subprocess.run([exe, scene, str(one), "-inflight", "9"] + common, capture_output=True, text=True)
    assert res.returncode == 2


def test_cli_driver_a_failing_rank_ends_the_run(tmp_path):
    """`rt_render -gpus N`: a GPU that fails -- during set-up, or inside the frame loop -- must end the run with the
    failing GPU's message and status 1 in bounded time (the ranks agree after set-up; a failure inside the loop
    aborts the communicators), never leave the process waiting in an exchange.  RT_RENDER_FAIL forces the failure."""
    import subprocess
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    scene = os.path.join(ROOT, "scenes", "mixed_xf.scene")
    out = tmp_path / "never.png"
    for where in ("setup", "frame"):
        env = dict(os.environ, RT_RENDER_FAIL=f"0:{where}")
        res = subprocess.run([exe, scene, str(out), "-gpus", "1", "-w", "160", "-h", "120", "-depth", "3", "-frames", "3", "-inflight", "2"],
                             capture_output=True, text=True, env=env, timeout=120)
        assert res.returncode == 1, (where, res.returncode, res.stderr)
        assert "GPU 0: failure forced by RT_RENDER_FAIL" in res.stderr and not out.exists()
    # (a failure forced on a rank that does not exist changes nothing)
    res = subprocess.run([exe, scene, str(out), "-gpus", "1", "-w", "160", "-h", "120", "-depth", "3"], capture_output=True, text=True,
                         env=dict(os.environ, RT_RENDER_FAIL="5:setup"), timeout=120)
    assert res.returncode == 0 and out.exists()


def test_cli_driver_packed_exchange_option(tmp_path):
    """`rt_render -gpus 1 -packed`: the option is accepted and the one-rank frame is the plain one (with one rank
    neither form of the exchange moves anything; the span arithmetic of both is checked on CPU)."""
    import subprocess
    exe = os.path.join(ROOT, "raytracer_amd", "rt_render")
    scene = os.path.join(ROOT, "scenes", "mixed_xf.scene")
    a, b = tmp_path / "a.png", tmp_path / "b.png"
    common = ["-w", "160", "-h", "117", "-depth", "4", "-tile", "8"]
    assert subprocess.run([exe, scene, str(a), "-gpus", "1"] + common, capture_output=True, text=True).returncode == 0
    assert subprocess.run([exe, scene, str(b), "-gpus", "1", "-packed"] + common, capture_output=True, text=True).returncode == 0
    assert np.array_equal(oracle_api.load_png(str(a)), oracle_api.load_png(str(b)))


def test_slot_order_changes_nothing(tmp_path, monkeypatch):
    """The order of the triangles inside a (transform, axis) run is free: file order (kept for meshes written patch by
    patch), or the order along a space-filling curve through the centroids (taken for meshes whose faces come in no
    spatial order).  A teapot with its faces shuffled must give the same frame and counts in either order as in the
    reference-precision scan of everything, and must be given the curve order by the upload's own choice."""
    import random
    src = open(os.path.join(ROOT, "scenes", "teapot.obj")).read().splitlines()
    faces = [ln for ln in src if ln.startswith("f ")]
    rest = [ln for ln in src if not ln.startswith("f ")]
    random.Random(7).shuffle(faces)
    (tmp_path / "teapot.obj").write_text("\n".join(rest + faces) + "\n")
    scene_text = open(os.path.join(ROOT, "scenes", "teapot.scene")).read()
    (tmp_path / "shuffled.scene").write_text(scene_text)
    monkeypatch.chdir(tmp_path)                      # (obj paths are relative to the working directory, raytracer.cpp:802-805)
    sc = Scene.load(str(tmp_path / "shuffled.scene"), 192, 108, 3)
    want, st_want = gpu_render(sc, exact_only=True)
    got = {}
    for order in ("file", "curve", ""):
        if order:
            monkeypatch.setenv("RAYTRACER_AMD_SLOT_ORDER", order)
        else:
            monkeypatch.delenv("RAYTRACER_AMD_SLOT_ORDER")
        img, st = gpu_render(sc)
        assert np.array_equal(img, want), order
        for key in ("rays_reflect", "rays_shadow", "hits"):
            assert st[key] == st_want[key], (order, key)
        got[order] = st["exact_tests"]
    # the upload's own choice for a shuffled mesh is the curve (compact clusters list the pairs the curve order lists)
    assert got[""] == got["curve"]


@pytest.mark.parametrize("name,w,h,depth", [
    ("teapot.scene", 320, 180, 5), ("mixed_xf.scene", 200, 150, 7), ("scene3.scene", 160, 120, 2), ("teapot.scene", 160, 90, 9),
])
def test_the_two_folds_agree(name, w, h, depth, monkeypatch):
    """The colours are folded deepest level first (raytracer.cpp:542-543) either along each camera ray's chain of
    reflected hits in one launch (scenes of at most eight levels) or level by level; the same additions and
    multiplications in the same order, so the same frame -- also with several calls in flight and through the LBVH.
    (Depth 9 has ten levels and takes the level-by-level fold either way: that leg pins the fallback.)"""
    from raytracer_amd.render import Renderer
    sc = Scene.load(os.path.join(ROOT, "scenes", name), w, h, depth)
    frames = {}
    for by_level in ("1", "0"):
        monkeypatch.setenv("RAYTRACER_AMD_FOLD_BY_LEVEL", by_level)
        r = Renderer(sc)
        r.render_frame()
        frames[by_level] = [r.download()]
        st = r.stats()
        r.set_option(abi.RT_OPT_USE_BVH, 1)
        r.render_frame()
        frames[by_level].append(r.download())
        r.set_option(abi.RT_OPT_USE_BVH, 0)
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 3)
        for _ in range(4):
            r.render_frame()
        frames[by_level].append(r.download())
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 1)
        frames[by_level].append((st["rays_primary"], st["rays_reflect"], st["rays_shadow"], st["hits"]))
        r.close()
    for a, b in zip(frames["1"][:3], frames["0"][:3]):
        assert np.array_equal(a, b), int((a != b).any(axis=-1).sum())
    assert frames["1"][3] == frames["0"][3]
    assert np.array_equal(frames["0"][0], frames["0"][2])


@pytest.mark.parametrize("name,w,h,depth", [
    ("teapot.scene", 320, 180, 5), ("mixed_xf.scene", 200, 150, 6), ("campanile_teddy.scene", 160, 90, 4),
    ("synthetic_small.scene", 160, 90, 4), ("normals.scene", 128, 96, 3), ("scene3.scene", 160, 120, 2),
])
def test_scan_instances_without_the_in_plane_origin_machinery(name, w, h, depth, monkeypatch):
    """Launches that do not ask the in-plane-origin question run instances of the shadow and reflection scans from which
    the machinery for it is compiled out (k_filter's PLANE = false).  What is compiled out only ever added candidates
    under a comparison that is false for such a launch: the frame, the ray counts and the number of listed pairs that
    reach the reference's arithmetic equal those of the full instances (RAYTRACER_AMD_LEAN_SCANS=0), one call at a time
    and three in flight, in both filter forms."""
    from raytracer_amd.render import Renderer
    monkeypatch.setenv("RAYTRACER_AMD_TEST_VOID_NORMALS", "1")        # (campanile: shadow rays are scanned, not answered at upload)
    sc = Scene.load(os.path.join(ROOT, "scenes", name), w, h, depth)
    got = {}
    for lean in ("0", "1"):
        monkeypatch.setenv("RAYTRACER_AMD_LEAN_SCANS", lean)
        for form in (1, 2):
            r = Renderer(sc)
            r.set_option(abi.RT_OPT_FILTER_FORM, form)
            r.render_frame()                   # (the calibration frame runs here; the counts below are one frame's)
            r.reset_stats()
            r.render_frame()
            st = r.stats()
            one = r.download()
            r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 3)
            for _ in range(4):
                r.render_frame()
            several = r.download()
            r.close()
            got[lean, form] = (one, several, st["bvh_plane_guard"],
                               tuple(st[k] for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits", "exact_tests", "flagged_rays")))
    for form in (1, 2):
        a, b = got["0", form], got["1", form]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(b[0], b[1]), form
        assert a[2] == b[2] and a[3] == b[3], (form, a[2:], b[2:])
    if name == "teapot.scene":
        assert got["1", 2][2] == 0              # (no ray of this frame takes the branch: the lean instances are the ones that ran)
