"""One rank's share of an 8-GPU frame rendered as K concurrent sub-shares (K contexts, a stream pair each,
interleaved tiles rank, rank + 8, ... dealt round-robin to the contexts) against the share in one context."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load(os.path.join(ROOT, "scenes", "teapot.scene"), 3840, 2160, 5)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out = {}
for k in (1, 2, 3, 4):
    rs = [Renderer(sc) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def frame():
        for i, r in enumerate(rs):
            r.render_tiles(i * world, 16, world * k, streams[i].cuda_stream)      # rank 0's tiles 0, 8, 16, ... dealt to k contexts
    for _ in range(2):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        frame()
    torch.cuda.synchronize()
    out[k] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
    for r in rs:
        r.close()
print(json.dumps({"world": world, "ms_per_share_by_contexts": out}))
