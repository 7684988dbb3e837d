"""Host time to enqueue one render call (a frame, or one rank's share of it) against the time the GPU takes for it."""
import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
r = Renderer(sc)
stream = torch.cuda.current_stream().cuda_stream
for bvh in (0, 1):
    r.set_option(abi.RT_OPT_USE_BVH, bvh)
    for k in (1, 3):
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, k)
        for world in (1, 8):
            for _ in range(4): r.render_tiles(0, 16, world, stream)
            torch.cuda.synchronize()
            n = 6
            t0 = time.perf_counter()
            for _ in range(n): r.render_tiles(0, 16, world, stream)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f'bvh {bvh} in flight {k} rank 0 of {world}: host enqueue {(t1 - t0) / n * 1e3:.2f} ms per call, '
                  f'GPU {(t2 - t0) / n * 1e3:.2f} ms per call', flush=True)
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 1)
