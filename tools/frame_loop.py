"""Renders N frames of a scene (default: the benchmark's teapot 4K depth 5) on one stream, for rocprofv3.
OVERLAP=1: the library's two streams; OPTS=BOUNDING_TESTS=0,USE_BVH=1,...: rt_set_option calls before the first frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scene = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "scenes", "teapot.scene")
w, h, d = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (3840, 2160, 5)
r = Renderer(Scene.load(scene, w, h, d))
r.set_option(abi.RT_OPT_OVERLAP, int(os.environ.get("OVERLAP", "0")))     # (OVERLAP=1: the library's two streams)
for item in filter(None, os.environ.get("OPTS", "").split(",")):
    key, value = item.split("=")
    r.set_option(getattr(abi, "RT_OPT_" + key), int(value))
for _ in range(n):
    r.render_frame()
r.download()
