"""Milliseconds per 4K teapot frame (or one rank's share of it) against RT_OPT_FRAMES_IN_FLIGHT, brute force and LBVH.
Usage: in_flight_probe.py [world ...]"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
worlds = [int(a) for a in sys.argv[1:]] or [1, 8]
r = Renderer(sc)
if os.environ.get('SCAN_COPIES'):
    r.set_option(abi.RT_OPT_SCAN_COPIES, 1)
if os.environ.get('OVERLAP'):
    r.set_option(abi.RT_OPT_OVERLAP, int(os.environ['OVERLAP']))
stream = torch.cuda.current_stream().cuda_stream
frames = [torch.zeros((2160, 3840, 4), dtype=torch.uint8, device='cuda') for _ in range(8)]
calls = [0]


def render(world, k, own_frames):
    if own_frames:
        r.bind_frame(frames[calls[0] % k])
        calls[0] += 1
    r.render_tiles(0, 16, world, stream)


def timed(fn, steps=12):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for own_frames in (1,):
  r.bind_frame(frames[0])
  print('a frame buffer per call in flight' if own_frames else 'one frame buffer (folds in call order)')
  for order in ((4, 8),):
    for bvh in (0,):
        r.set_option(abi.RT_OPT_USE_BVH, bvh)
        for world in worlds:
            res = []
            for k in order:
                r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, k)
                res.append(timed(lambda: render(world, k, own_frames)))
            r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 1)
            print(f'bvh {bvh} rank 0 of {world}: ms per frame with {order} frames in flight: ' + ' / '.join(f'{t:.2f}' for t in res), flush=True)
