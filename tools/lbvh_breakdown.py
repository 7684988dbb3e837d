import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
r = Renderer(sc)
r.set_option(abi.RT_OPT_USE_BVH, 1)
for prof in (0, 1):
    r.set_option(abi.RT_OPT_PROFILE, prof)
    r.render_frame(); torch.cuda.synchronize(); r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(5): r.render_frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    s = r.stats()
    print(f'profile {prof}: {dt*1e3:.2f} ms/frame; trace<0> {s["closest_ms"]/5:.2f} ms, trace<1> {s["anyhit_ms"]/5:.2f} ms, brute+pairs {s["exact_ms"]/5:.2f} ms')
