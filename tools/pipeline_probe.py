"""Would frame-level pipelining pay?  K contexts on K streams render consecutive whole frames (or one
rank's share of them) at the same time; compared with one context rendering them one after the other.
Usage: pipeline_probe.py [world ...]"""
import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
worlds = [int(a) for a in sys.argv[1:]] or [1, 8]
K = 3
frames = [torch.zeros((2160, 3840, 4), dtype=torch.uint8, device='cuda') for _ in range(K)]
ctx = [Renderer(sc) for _ in range(K)]
for c, f in zip(ctx, frames):
    c.bind_frame(f)
streams = [torch.cuda.Stream() for _ in range(K)]


def timed(fn, steps=12):
    fn(0); fn(1); fn(2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for bvh in (0, 1):
    for c in ctx:
        c.set_option(abi.RT_OPT_USE_BVH, bvh)
    for world in worlds:
        res = []
        for k in range(1, K + 1):
            res.append(timed(lambda i: ctx[i % k].render_tiles(0, 16, world, streams[i % k].cuda_stream)))
        print(f'bvh {bvh} rank 0 of {world}: ms per frame with 1 / 2 / 3 frames in flight: ' + ' / '.join(f'{t:.2f}' for t in res), flush=True)
