"""With the diagnostic build `make variant NAME=util FLAGS=-DRT_BVH_UTIL` (RAYTRACER_AMD_LIB=variants/libraytracer_amd_util.so):
how full the waves of the LBVH traversal are -- lane-steps worked over lane-steps spent, nearest-hit and shadow launches
together, 4K teapot frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
for depth in (0, 1, 5):
    r = Renderer(Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 3840, 2160, depth))
    r.set_option(abi.RT_OPT_USE_BVH, 1)
    r.render_frame()
    st = r.stats()
    print(f"depth {depth}: lane-steps worked {st['bvh_node_tests']}, spent {st['bvh_leaf_tests']}: "
          f"{st['bvh_node_tests'] / max(1, st['bvh_leaf_tests']):.3f} of the lanes busy; "
          f"{st['exact_tests'] / max(1, st['bvh_leaf_tests']):.3f} of the steps have lanes at a leaf and lanes at a node", flush=True)
    r.close()
