"""Frame time (default configuration, two streams) and serialised scan times with one option switched:
usage: python tools/opt_ab.py OPTION_NAME VALUE [VALUE ...]   (e.g. SCAN_COPIES 1 0)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
opt = getattr(abi, "RT_OPT_" + sys.argv[1])
sc = Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 3840, 2160, 5)
for v in sys.argv[2:]:
    r = Renderer(sc)
    r.set_option(opt, int(v))
    for _ in range(2):
        r.render_frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r.render_frame()
    torch.cuda.synchronize()
    frame = (time.perf_counter() - t0) / 5 * 1e3
    r.reset_stats()
    r.set_option(abi.RT_OPT_PROFILE, 1)
    for _ in range(5):
        r.render_frame()
    st = r.stats()
    print(json.dumps({sys.argv[1]: int(v), "frame_ms": round(frame, 3), "ms": {k: round(x / 5, 3) for k, x in st["scan_ms"].items() if x},
                      "exact": round(st["exact_ms"] / 5, 3)}))
    r.close()
