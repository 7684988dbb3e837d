"""Seeded random scenes: every configuration of the trace loop against the exact-only mode.

The FP32 candidate filter, its two forms, the shared-origin / shared-direction constants and
the LBVH are all supposed to be invisible: each may only skip work whose result the
reference-precision test would not have used.  RT_OPT_EXACT_ONLY switches all of them off
(every primitive is tested in reference arithmetic for every ray), so its frame is the
yardstick; one scene per family is also checked against the CPU oracle.

The generator aims at the awkward cases: slivers and tiny triangles next to huge ones, grids
of coplanar triangles, axis-aligned planes through the camera or through hit points (the
reference's in-plane-origin branch, raytracer.cpp:319-321), grazing directional lights with a
zero component, transformed groups with strong anisotropic scaling, lights inside the geometry.
"""
import os

import numpy as np
import pytest

from raytracer_amd import Scene, abi
from tests import oracle_api

pytestmark = pytest.mark.gpu


def fmt(v):
    return " ".join(f"{x:.6g}" for x in v)


def random_scene(seed, family):
    """Returns the text of a .scene file (triangles and spheres inline, no .obj)."""
    rng = np.random.default_rng(seed)
    L = []
    eye = np.array([rng.uniform(-1, 1), rng.uniform(0.5, 3), rng.uniform(6, 9)])
    if family == "axis":
        eye = np.array([0.0, 1.0, 8.0])            # x = 0: centre-column camera rays have a zero component
    z = eye[2] - 4.0
    L.append(f"cam {fmt(eye)}  -3.2 -1.6 {z:.6g}  3.2 -1.6 {z:.6g}  -3.2 2.0 {z:.6g}  3.2 2.0 {z:.6g}")
    L.append("lta 0.1 0.1 0.1")
    for _ in range(int(rng.integers(1, 3))):
        L.append(f"ltp {fmt(rng.uniform(-6, 6, 3) + [0, 4, 0])}  {fmt(rng.uniform(0.3, 1.0, 3))}  {int(rng.integers(0, 3))}")
    d = rng.uniform(-1, 1, 3)
    d[1] = -abs(d[1]) - 0.2
    if family == "axis":
        d[0] = 0.0                                  # shadow rays with an exactly-zero x component
    L.append(f"ltd {fmt(d)}  {fmt(rng.uniform(0.2, 0.8, 3))}")
    if family == "grazing":
        L.append("ltd 1 -0.001 0.3  0.4 0.4 0.4")   # nearly parallel to the floor

    def mat():
        kr = rng.uniform(0, 0.7)
        L.append(f"mat {fmt(rng.uniform(0, 0.3, 3))}  {fmt(rng.uniform(0.1, 0.8, 3))}  {fmt(rng.uniform(0, 0.9, 3))}  "
                 f"{rng.uniform(2, 60):.4g}  {kr:.4g} {kr:.4g} {kr:.4g}")

    def soup(n, centre, extent, size):
        for _ in range(n):
            c = centre + rng.uniform(-1, 1, 3) * extent
            s = size * 10 ** rng.uniform(-1.5, 0.3)
            a, b = rng.normal(size=3) * s, rng.normal(size=3) * s
            if rng.random() < 0.15:
                b = a * rng.uniform(0.5, 2.0) + rng.normal(size=3) * s * 1e-3      # sliver
            L.append(f"tri {fmt(c)}  {fmt(c + a)}  {fmt(c + b)}")

    def grid(n, y, half):
        """n x n quads of coplanar triangles in the plane y = const."""
        xs = np.linspace(-half, half, n + 1)
        for i in range(n):
            for j in range(n):
                p00, p10 = (xs[i], y, xs[j]), (xs[i + 1], y, xs[j])
                p01, p11 = (xs[i], y, xs[j + 1]), (xs[i + 1], y, xs[j + 1])
                L.append(f"tri {fmt(p00)}  {fmt(p01)}  {fmt(p11)}")
                L.append(f"tri {fmt(p00)}  {fmt(p11)}  {fmt(p10)}")

    mat()
    if family == "soup":
        soup(300, np.array([0, 0.5, 0.0]), np.array([3, 1.5, 3]), 0.5)
        L.append(f"sph {fmt(rng.uniform(-2, 2, 3))} {rng.uniform(0.3, 0.9):.4g}")
    elif family == "coplanar":
        grid(12, -1.0, 5.0)
        mat()
        grid(6, 0.25, 1.5)                           # a second, smaller sheet above the first
        mat()
        soup(40, np.array([0, 1.0, 0.0]), np.array([2, 0.8, 2]), 0.4)
    elif family == "axis":
        grid(4, 0.0, 6.0)                            # the plane y = 0 ...
        mat()
        L.append("tri 0 -1 -3  0 3 -3  0 1 2")       # ... and a triangle in the plane x = 0 of the camera
        L.append("tri -3 1 -2  3 1 -2  0 1 3")       # the plane y = 1 contains the eye
        soup(60, np.array([0, 1.0, 0.0]), np.array([3, 1.0, 3]), 0.6)
        L.append("sph 1.5 0.8 0.5 0.8")
    elif family == "grazing":
        grid(10, -0.5, 6.0)
        mat()
        soup(80, np.array([0, 0.0, 0.0]), np.array([4, 0.5, 4]), 0.3)
    elif family == "transformed":
        soup(60, np.array([0, 0.3, 0.0]), np.array([3, 1.0, 3]), 0.5)
        for _ in range(3):
            L.append(f"xft {fmt(rng.uniform(-2, 2, 3))}")
            L.append(f"xfr {fmt(rng.uniform(-60, 60, 3))}")
            L.append(f"xfs {fmt(10 ** rng.uniform(-1.3, 0.5, 3))}")
            mat()
            ext = rng.uniform(1, 15)
            soup(50, np.zeros(3), np.array([ext, ext, ext]), ext * 0.3)
            L.append(f"sph {fmt(rng.uniform(-1, 1, 3) * ext)} {ext * rng.uniform(0.1, 0.4):.4g}")
            L.append("xfz")
        mat()
        grid(3, -1.5, 7.0)
    elif family == "tiny":
        # thousands of triangles that are tiny against the scene: the static margins are useless here
        n = 40
        xs = np.linspace(-0.4, 0.4, n + 1)
        for i in range(n):
            for j in range(n):
                def h(x, zz):
                    return 0.05 * np.sin(20 * x) * np.cos(17 * zz)
                q = [(xs[i], h(xs[i], xs[j]), xs[j]), (xs[i + 1], h(xs[i + 1], xs[j]), xs[j]),
                     (xs[i], h(xs[i], xs[j + 1]), xs[j + 1]), (xs[i + 1], h(xs[i + 1], xs[j + 1]), xs[j + 1])]
                L.append(f"tri {fmt(q[0])}  {fmt(q[2])}  {fmt(q[3])}")
                L.append(f"tri {fmt(q[0])}  {fmt(q[3])}  {fmt(q[1])}")
        mat()
        L.append("tri -60 -0.3 -60  -60 -0.3 60  60 -0.3 60")
        L.append("tri -60 -0.3 -60  60 -0.3 60  60 -0.3 -60")
    return "\n".join(L) + "\n"


def render(sc, **opts):
    from raytracer_amd.render import Renderer
    r = Renderer(sc)
    for key, value in opts.items():
        r.set_option(getattr(abi, "RT_OPT_" + key), value)
    r.render_frame()
    img, st = r.download(), r.stats()
    r.close()
    return img, st


FAMILIES = ["soup", "coplanar", "axis", "grazing", "transformed", "tiny"]
SEEDS = range(1, 1 + int(os.environ.get("RT_FUZZ_SEEDS", "4")))      # raise for a longer hunt
CASES = [(f, s) for f in FAMILIES for s in SEEDS]


@pytest.mark.parametrize("family,seed", CASES, ids=[f"{f}-{s}" for f, s in CASES])
def test_every_configuration_equals_exact_only(family, seed, tmp_path):
    path = os.path.join(str(tmp_path), "fuzz.scene")
    with open(path, "w") as f:
        f.write(random_scene(1000 * seed + FAMILIES.index(family), family))
    w, h = (96, 54) if family != "tiny" else (64, 36)
    if family == "tiny":
        # look straight at the small patch so that most rays cross it
        text = open(path).read().splitlines()
        text[0] = "cam 0 0.6 1.2  -0.5 -0.2 0.2  0.5 -0.2 0.2  -0.5 0.35 0.2  0.5 0.35 0.2"
        open(path, "w").write("\n".join(text) + "\n")
    sc = Scene.load(path, w, h, 4)
    want, sw = render(sc, EXACT_ONLY=1)
    in_plane = None
    for form in (1, 2):
        # shared constants with and without the bounding tests in front of the triangle test, then neither
        for more in ({"SHARED_CONSTANTS": 1}, {"SHARED_CONSTANTS": 1, "BOUNDING_TESTS": 2},
                     {"SHARED_CONSTANTS": 1, "BOUNDING_TESTS": 0}, {"SHARED_CONSTANTS": 0},
                     {"SHARED_CONSTANTS": 0, "BOUNDING_TESTS": 0}):
            img, st = render(sc, FILTER_FORM=form, **more)
            assert np.array_equal(img, want), (form, more, int((img != want).any(axis=-1).sum()))
            for key in ("rays_reflect", "rays_shadow", "hits"):
                assert st[key] == sw[key], (key, form, more)
            in_plane = st["in_plane_origin_hits"] if in_plane is None else max(in_plane, st["in_plane_origin_hits"])
        # The LBVH cannot see the reference's in-plane-origin "hits" (a box test has no notion of
        # an infinite plane); the brute-force counter says whether this scene has any.
        if in_plane == 0:
            img, st = render(sc, FILTER_FORM=form, USE_BVH=1)
            assert np.array_equal(img, want), ("bvh", form, int((img != want).any(axis=-1).sum()))
            for key in ("rays_reflect", "rays_shadow", "hits"):
                assert st[key] == sw[key], (key, form, "bvh")


@pytest.mark.parametrize("family", FAMILIES)
def test_random_scene_against_the_oracle(family, oracle, tmp_path):
    path = os.path.join(str(tmp_path), "fuzz.scene")
    with open(path, "w") as f:
        f.write(random_scene(77 + FAMILIES.index(family), family))
    sc = Scene.load(path, 80, 45, 3)
    img, st = render(sc)
    want, c = oracle.render(sc, threads=16)
    worst, ndiff, frac1 = oracle_api.compare(img, want)
    assert worst <= 4 and frac1 >= 0.999 and ndiff <= 3, (worst, ndiff, frac1)
    for key in ("rays_reflect", "rays_shadow", "hits"):
        assert st[key] == c[key], key


@pytest.mark.parametrize("eps", [1e-7, 1e-3, 0.5])
@pytest.mark.parametrize("family", ["soup", "coplanar", "transformed"])
def test_unusual_epsilons(family, eps, tmp_path):
    """The filter's thresholds are derived from Scene::eps: a tiny eps leaves nothing certain
    about grazing rays (the static form then sends every triangle to the reference-precision
    test), a large one rejects nearby hits.  Every configuration must still equal exact-only."""
    path = os.path.join(str(tmp_path), "fuzz.scene")
    with open(path, "w") as f:
        f.write(random_scene(4242 + FAMILIES.index(family), family))
    sc = Scene.load(path, 64, 36, 3, eps=eps)
    want, sw = render(sc, EXACT_ONLY=1)
    in_plane = 0
    for opts in ({}, {"FILTER_FORM": 1}, {"FILTER_FORM": 2}, {"FILTER_FORM": 2, "USE_BVH": 1}):
        if opts.get("USE_BVH") and in_plane:
            continue                       # the LBVH cannot see in-plane-origin hits (counted by brute force)
        img, st = render(sc, **opts)
        in_plane = max(in_plane, st["in_plane_origin_hits"])
        assert np.array_equal(img, want), (opts, int((img != want).any(axis=-1).sum()))
        for key in ("rays_reflect", "rays_shadow", "hits"):
            assert st[key] == sw[key], (key, opts)


LARGE_SEEDS = range(11, 11 + int(os.environ.get("RT_FUZZ_LARGE_SEEDS", "2")))


@pytest.mark.parametrize("family,seed", [(f, s) for f in FAMILIES for s in LARGE_SEEDS],
                         ids=[f"{f}-{s}" for f in FAMILIES for s in LARGE_SEEDS])
def test_bounding_tests_on_larger_frames(family, seed, tmp_path):
    """The bounding tests work on bundles of hundreds of neighbouring rays; on frames large enough
    that a bundle is a small part of the picture they must still change nothing: default against
    RT_OPT_BOUNDING_TESTS = 0 (every pair takes the triangle test), frames and counts."""
    path = os.path.join(str(tmp_path), "fuzz.scene")
    with open(path, "w") as f:
        f.write(random_scene(5000 * seed + FAMILIES.index(family), family))
    sc = Scene.load(path, 640, 360, 4)
    want, sw = render(sc, BOUNDING_TESTS=0)
    for more in ({}, {"BOUNDING_TESTS": 2}, {"FILTER_FORM": 1}, {"FILTER_FORM": 2}):
        if "FILTER_FORM" in more:
            want, sw = render(sc, BOUNDING_TESTS=0, **more)
        img, st = render(sc, **more)
        assert np.array_equal(img, want), (more, int((img != want).any(axis=-1).sum()))
        for key in ("rays_reflect", "rays_shadow", "hits", "in_plane_origin_hits"):
            # (in-plane-origin hits are counted where listed pairs are re-tested: a ray whose pairs did not fit
            # the pair list takes the full scan instead, which counts none -- comparable only without such rays)
            if key == "in_plane_origin_hits" and (st["flagged_rays"] or sw["flagged_rays"]):
                continue
            assert st[key] == sw[key], (key, more)


@pytest.mark.parametrize("family", FAMILIES)
def test_calls_in_flight_on_random_scenes(family, tmp_path):
    """A render call in flight beside others cuts its scans into other tasks (fewer, of half as many blocks) and
    runs in a workspace of its own: frames and counts of four calls in flight must be those of one call, brute
    force and LBVH."""
    from raytracer_amd.render import Renderer
    path = os.path.join(str(tmp_path), "fuzz.scene")
    with open(path, "w") as f:
        f.write(random_scene(9000 + FAMILIES.index(family), family))
    sc = Scene.load(path, 640, 360, 4)
    want, sw = render(sc)
    for bvh in (0, 1):
        if bvh and sw["in_plane_origin_hits"]:
            continue                       # the LBVH cannot see in-plane-origin hits (counted by brute force)
        r = Renderer(sc)
        r.set_option(abi.RT_OPT_USE_BVH, bvh)
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 4)
        for _ in range(6):
            r.render_frame()
        img, st = r.download(), r.stats()
        r.close()
        assert np.array_equal(img, want), (bvh, int((img != want).any(axis=-1).sum()))
        for key in ("rays_reflect", "rays_shadow", "hits") + (() if bvh else ("in_plane_origin_hits",)):
            if key == "in_plane_origin_hits" and (st["flagged_rays"] or sw["flagged_rays"]):
                continue
            assert st[key] == 6 * sw[key], (key, bvh)
