"""Host time of one render call (the call returns when its launches are enqueued) against the frame time at that many calls
in flight, for one rank's share of the 4K teapot frame: is the share bound by the host thread that submits it?
usage: enqueue_cost.py [world ...]"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
worlds = [int(a) for a in sys.argv[1:]] or [1, 8, 128]
r = Renderer(Scene.load('scenes/teapot.scene', 3840, 2160, 5))
stream = torch.cuda.current_stream().cuda_stream
K = 8
frames = [torch.zeros((2160, 3840, 4), dtype=torch.uint8, device='cuda') for _ in range(K)]
r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, K)
for world in worlds:
    n = [0]
    def call():
        r.bind_frame(frames[n[0] % K]); n[0] += 1
        r.render_tiles(0, 16, world, stream)
    for _ in range(2 * K): call()
    r.join(stream); torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(40):
        a = time.perf_counter(); call(); host += time.perf_counter() - a
    r.join(stream); torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("rank 0 of %3d, %d calls in flight: %.3f ms per frame, of which the host spends %.3f ms inside the call" % (world, K, wall / 40 * 1e3, host / 40 * 1e3))
r.close()
