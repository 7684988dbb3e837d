"""Where a wave's time goes in the scan kernels, from a diagnostic build of the library
(make variant NAME=phase FLAGS=-DRT_PHASE_CLOCK; run with RAYTRACER_AMD_LIB=variants/libraytracer_amd_phase.so):
shader cycles by phase of a task, summed over waves, as shares of the waves' lifetimes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
share = 1
if "--share" in sys.argv:                      # the share of rank 0 of an N-GPU job (interleaved 16-row tiles) instead of the frame
    i = sys.argv.index("--share")
    share = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes", "teapot.scene")
w, h, d = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (3840, 2160, 5)
r = Renderer(Scene.load(scene, w, h, d))
def frame():
    if share == 1:
        r.render_frame()
    else:
        r.render_tiles(0, 16, share)
frame()
r.reset_stats()
r.set_option(abi.RT_OPT_PROFILE, 1)
r.set_option(abi.RT_OPT_WORK_COUNTERS, 1)
frame()
st = r.stats()
names = ["tasks", "wait_task", "rays", "judge", "batches", "flush", "task_total", "wave_life", "waves"]
out = {}
for kind, v in st["scan_work"].items():
    f = list(v.values())
    if kind in ("plain_nearest", "plain_shadow"):
        out["task_cycles_histogram_" + ("reflection" if kind == "plain_nearest" else "shadow")] = {f"2^{14 + b}": f[b] for b in range(11)}
        continue
    if not f[0]:
        continue
    life = max(1, f[7])
    row = {"tasks": f[0], "waves_with_work": f[8], "ms": st["scan_ms"][kind],
           "cycles_per_task": f[6] / f[0], "wave_life_cycles": life / max(1, f[8])}
    for i in range(1, 6):
        row[names[i] + "_share_of_tasks"] = round(f[i] / max(1, f[6]), 3)
    row["ray_loop_share_of_tasks"] = round(f[9] / max(1, f[6]), 3)       # (part of the rays phase)
    row["tasks_share_of_life"] = round(f[6] / life, 3)
    out[kind] = row
print(json.dumps(out, indent=1))
