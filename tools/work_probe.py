"""Executed-work counters of one frame (default: the benchmark's teapot 4K depth 5), by kind of scan."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes", "teapot.scene")
w, h, d = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (3840, 2160, 5)
r = Renderer(Scene.load(scene, w, h, d))
r.render_frame()
r.reset_stats()
r.set_option(abi.RT_OPT_PROFILE, 1)
for _ in range(3):
    r.render_frame()
st = r.stats()
ms = {k: v / 3 for k, v in st["scan_ms"].items()}
ms["exact"] = st["exact_ms"] / 3
r.set_option(abi.RT_OPT_PROFILE, 0)
r.reset_stats()
r.set_option(abi.RT_OPT_WORK_COUNTERS, 1)
r.render_frame()
st = r.stats()
print(json.dumps({"ms": ms, "work": {k: v for k, v in st["scan_work"].items() if v["tasks"]},
                  "exact_tests": st["exact_tests"], "flagged": st["flagged_rays"]}, indent=1))
