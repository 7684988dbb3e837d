"""Renders N frames of the roofline-stress scene (1,000,000 triangles + 64 spheres, 16 lights, depth 8, written by
tools/make_synthetic.py into a temporary directory) through the LBVH, for rocprofv3.
usage: config5_loop.py [frames] [width height]"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w, h = (int(x) for x in sys.argv[2:4]) if len(sys.argv) > 3 else (7680, 4320)
tmp = tempfile.mkdtemp(prefix="rt_config5_")
try:
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synthetic.py"), tmp], check=True, stdout=subprocess.DEVNULL)
    r = Renderer(Scene.load(os.path.join(tmp, "synthetic.scene"), w, h, 8))
    r.set_option(abi.RT_OPT_USE_BVH, 1)
    r.render_frame()
    r.download()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(n):
        r.render_frame()
    r.download()
    dt = time.perf_counter() - t0
    st = r.stats()
    print({k: st[k] for k in ("rays_primary", "rays_reflect", "rays_shadow", "hits", "exact_tests", "flagged_rays", "bvh_plane_guard")}, "%.1f ms per frame" % (dt / n * 1e3))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
