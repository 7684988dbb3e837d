"""Milliseconds per frame by kind of scan (HIP events, streams serialised) and for the whole frame in the
default configuration (two streams), teapot 4K depth 5 unless another scene is named.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
world = int(os.environ.get("WORLD", "1"))          # WORLD=N: rank 0's share of an N-rank frame (interleaved 16-row tiles)
scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes", "teapot.scene")
w, h, d = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (3840, 2160, 5)
n = 5
r = Renderer(Scene.load(scene, w, h, d))
if os.environ.get("FORM"):                       # FORM=1|2: the static or the per-test margins of the filter, whatever the upload chose
    r.set_option(abi.RT_OPT_FILTER_FORM, int(os.environ["FORM"]))
for _ in range(2):
    r.render_tiles(0, 16, world)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    r.render_tiles(0, 16, world)
torch.cuda.synchronize()
frame = (time.perf_counter() - t0) / n * 1e3
# ... and with four render calls in flight (RT_OPT_FRAMES_IN_FLIGHT)
r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 4)
frames = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(4)]      # (a buffer per call in flight)
for i in range(4):
    r.bind_frame(frames[i % 4]); r.render_tiles(0, 16, world)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(4 * n):
    r.bind_frame(frames[i % 4]); r.render_tiles(0, 16, world)
torch.cuda.synchronize()
frame3 = (time.perf_counter() - t0) / (4 * n) * 1e3
r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 1)
r.bind_frame(None)
r.reset_stats()
r.set_option(abi.RT_OPT_PROFILE, 1)
for _ in range(n):
    r.render_tiles(0, 16, world)
st = r.stats()
ms = {k: round(v / n, 3) for k, v in st["scan_ms"].items() if v}
ms["exact"] = round(st["exact_ms"] / n, 3)
work = None
if os.environ.get("WORK"):
    # one more frame through the counting instances: executed work per kind of scan
    r.set_option(abi.RT_OPT_PROFILE, 0)
    r.reset_stats()
    r.set_option(abi.RT_OPT_WORK_COUNTERS, 1)
    r.render_tiles(0, 16, world)
    sw = r.stats()["scan_work"]
    work = {k: {f: v[f] for f in ("tasks", "clusters", "groups", "groups_live", "batches", "batches_tested", "listed")} for k, v in sw.items() if v["tasks"]}
print(json.dumps({"world": world, "order": os.environ.get("RAYTRACER_AMD_SLOT_ORDER", "curve"), "work": work, "lib": os.path.basename(os.environ.get("RAYTRACER_AMD_LIB", "default")), "frame_ms": round(frame, 3), "frame_ms_4_in_flight": round(frame3, 3), "ms": ms,
                  "exact_tests": st["exact_tests"] // n}))
