"""What one rank of an N-GPU job has to do, timed on one GPU (no exchange): the interleaved
16-row tiles of rank 0 and of rank N-1 of the 4K teapot frame, with the shadow/nearest-hit
overlap on and off.  usage: python tools/sim_world.py [N ...]"""
import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
r = Renderer(sc)
import os
if os.environ.get('RT_SCAN_COPIES'):
    r.set_option(abi.RT_OPT_SCAN_COPIES, int(os.environ['RT_SCAN_COPIES']))
frame = torch.zeros((2160, 3840, 4), dtype=torch.uint8, device='cuda')
r.bind_frame(frame)
st = torch.cuda.current_stream().cuda_stream


def timed(rank, world, steps=5):
    r.render_tiles(rank, 16, world, st); torch.cuda.synchronize()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_tiles(rank, 16, world, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    s = r.stats()
    return dt, (s['rays_primary'] + s['rays_reflect'] + s['rays_shadow']) / steps


for world in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    for rank in ((0,) if world == 1 else (0, world - 1)):
        out = []
        for overlap in (1, 0):
            r.set_option(abi.RT_OPT_OVERLAP, overlap)
            dt, rays = timed(rank, world)
            out.append(f'overlap {overlap}: {dt*1e3:6.2f} ms/frame {rays/dt/1e6:6.1f} Mrays/s (x{world} = {world*rays/dt/1e6:5.0f})')
        print(f'world {world} rank {rank}: ' + '   '.join(out), flush=True)
