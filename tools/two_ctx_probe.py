"""Would pass-level concurrency pay?  Two contexts on two streams render the two halves of a
rank's interleaved tiles at the same time; compared with one context rendering all of them."""
import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
frame = torch.zeros((2160, 3840, 4), dtype=torch.uint8, device='cuda')
one = Renderer(sc); one.bind_frame(frame)
a = Renderer(sc); a.bind_frame(frame)
b = Renderer(sc); b.bind_frame(frame)
s0 = torch.cuda.current_stream().cuda_stream
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, steps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for world in (1, 2, 4, 8):
    t1 = timed(lambda: one.render_tiles(0, 16, world, s0))
    def both():
        a.render_tiles(0, 16, 2 * world, sa.cuda_stream)
        b.render_tiles(world, 16, 2 * world, sb.cuda_stream)
    t2 = timed(both)
    print(f'rank 0 of {world}: one context {t1:.2f} ms, two contexts side by side {t2:.2f} ms', flush=True)
