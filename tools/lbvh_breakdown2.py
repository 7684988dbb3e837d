"""LBVH frame of the benchmark scene: traversal kernels (HIP events, streams serialised) against the frame."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
r = Renderer(Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 3840, 2160, 5))
r.set_option(abi.RT_OPT_USE_BVH, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(3):
    r.render_frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r.render_frame()
torch.cuda.synchronize()
frame = (time.perf_counter() - t0) / 10 * 1e3
r.set_option(abi.RT_OPT_OVERLAP, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r.render_frame()
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / 10 * 1e3
r.reset_stats()
r.set_option(abi.RT_OPT_PROFILE, 1)
for _ in range(5):
    r.render_frame()
st = r.stats()
print(json.dumps({"frame_ms": round(frame, 3), "one_stream_ms": round(serial, 3),
                  "traversal_nearest_ms": round(st["scan_ms"]["plain_nearest"] / 5, 3),
                  "traversal_shadow_ms": round(st["scan_ms"]["plain_shadow"] / 5, 3), "other_trace_kernels_ms": round(st["exact_ms"] / 5, 3)}))
