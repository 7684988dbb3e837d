"""Does the chip have room beside a frame?  K contexts render interleaved tile sets of the same frame
concurrently (one stream pair each); compares the time with one context rendering the whole frame."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
scene = os.path.join(ROOT, "scenes", "teapot.scene")
w, h, d = 3840, 2160, 5
sc = Scene.load(scene, w, h, d)
out = {}
for k in (1, 2, 3, 4):
    rs = [Renderer(sc) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def frame():
        for i, r in enumerate(rs):
            if k == 1:
                r.render_frame(streams[i].cuda_stream)
            else:
                r.render_tiles(i, 16 * 8, k, streams[i].cuda_stream)
    for _ in range(2):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        frame()
    torch.cuda.synchronize()
    out[k] = round((time.perf_counter() - t0) / n * 1e3, 3)
    for r in rs:
        r.close()
print(json.dumps(out))
