"""What k_brute's wave-by-wave sphere question drops, and whether any of it would have hit.  Needs the diagnostic build
(make variant NAME=cullcheck FLAGS=-DRT_CHECK_SPHERE_CULL; run with RAYTRACER_AMD_LIB=variants/libraytracer_amd_cullcheck.so
and RAYTRACER_AMD_SPHERE_CULL=1), which runs the reference's arithmetic on every dropped (ray, sphere) pair: per scene the
share of the pairs that were dropped and the number of dropped pairs that hit (must be 0)."""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import Scene
from raytracer_amd.render import Renderer
assert "cullcheck" in os.environ.get("RAYTRACER_AMD_LIB", ""), __doc__
tmp = tempfile.mkdtemp(prefix="rt_cull_")
bad = 0
try:
    from test_gpu_known_answers import many_spheres_scene
    open(os.path.join(tmp, "many_spheres.scene"), "w").write(many_spheres_scene())
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic.py"), tmp], check=True, stdout=subprocess.DEVNULL)
    cases = [(os.path.join(tmp, "many_spheres.scene"), 640, 360, 5), (os.path.join(tmp, "synthetic.scene"), 480, 270, 8)]
    cases += [(os.path.join(ROOT, "scenes", n), 480, 270, 5) for n in ("ellipse.scene", "mixed_xf.scene", "synthetic_small.scene", "scene1.scene", "scene2.scene", "scene3.scene", "sphere.scene")]
    for path, w, h, d in cases:
        r = Renderer(Scene.load(path, w, h, d))
        r.reset_stats()
        r.render_frame()
        st = r.stats()
        pairs = (st["rays_primary"] + st["rays_reflect"] + st["rays_shadow"]) * st["n_spheres"]
        print("%-24s %4d spheres  %12d pairs  dropped %5.1f %%  dropped pairs that hit: %d"
              % (os.path.basename(path), st["n_spheres"], pairs, 100.0 * st["bvh_leaf_tests"] / max(1, pairs), st["bvh_node_tests"]), flush=True)
        bad += st["bvh_node_tests"]
        r.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
sys.exit(1 if bad else 0)
