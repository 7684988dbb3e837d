import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
for pp in (1 << 21, 1 << 22, 1 << 23, 1 << 24):
    r = Renderer(sc)
    r.set_option(abi.RT_OPT_PASS_PIXELS, pp)
    for bvh in (0, 1):
        r.set_option(abi.RT_OPT_USE_BVH, bvh)
        r.render_frame(); torch.cuda.synchronize(); r.reset_stats()
        t0 = time.perf_counter()
        for _ in range(5): r.render_frame()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        s = r.stats(); rays = (s['rays_primary'] + s['rays_reflect'] + s['rays_shadow']) / 5
        print(f'pass_pixels {pp:9d} bvh {bvh}: {dt*1e3:6.2f} ms  {rays/dt/1e6:7.1f} Mrays/s', flush=True)
    r.close()
