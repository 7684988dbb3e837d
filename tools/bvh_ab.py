"""LBVH frame time of the 4K teapot (one at a time / four in flight) for the library named by RAYTRACER_AMD_LIB."""
import os, sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
r = Renderer(sc)
r.set_option(abi.RT_OPT_USE_BVH, 1)
frames = [torch.zeros((2160, 3840, 4), dtype=torch.uint8, device='cuda') for _ in range(4)]
res = []
for k in (1, 4):
    r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, k)
    for i in range(4):
        r.bind_frame(frames[i % k]); r.render_frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(12):
        r.bind_frame(frames[i % k]); r.render_frame()
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 12 * 1e3)
print(os.path.basename(os.environ.get("RAYTRACER_AMD_LIB", "default")), "LBVH ms per frame: %.2f one at a time, %.2f four in flight" % tuple(res))
