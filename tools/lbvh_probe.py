"""LBVH frame time of the benchmark scene (teapot 4K depth 5), two streams, N frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
r = Renderer(Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 3840, 2160, 5))
r.set_option(abi.RT_OPT_USE_BVH, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(3):
    r.render_frame()
r.stats()
t0 = time.perf_counter()
for _ in range(20):
    r.render_frame()
st = r.stats()
dt = (time.perf_counter() - t0) / 20
print(f"lbvh {dt * 1e3:.3f} ms/frame  box tests/ray {st['bvh_node_tests'] / (st['rays_primary'] + st['rays_reflect'] + st['rays_shadow']):.1f}")
