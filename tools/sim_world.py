import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 3840, 2160, 5)
r = Renderer(sc)
frame = torch.zeros((2160, 3840, 4), dtype=torch.uint8, device='cuda')
r.bind_frame(frame)
st = torch.cuda.current_stream().cuda_stream
for world in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    for rank in ((0,) if world == 1 else (0, world - 1)):
        r.render_tiles(rank, 16, world, st); torch.cuda.synchronize()
        r.reset_stats(); r.set_option(abi.RT_OPT_PROFILE, 1)
        t0 = time.perf_counter()
        for _ in range(5):
            r.render_tiles(rank, 16, world, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        s = r.stats()
        rays = (s['rays_primary'] + s['rays_reflect'] + s['rays_shadow']) / 5
        r.set_option(abi.RT_OPT_PROFILE, 0)
        print(f'[filter {(s["closest_ms"]+s["anyhit_ms"])/5:.2f} ms ({s["closest_launches"]//5}+{s["anyhit_launches"]//5} launches), exact {s["exact_ms"]/5:.2f} ms] ', end='')
        print(f'world {world} rank {rank}: {dt*1e3:.2f} ms/frame, {rays/1e6:.2f} Mrays, {rays/dt/1e6:.1f} Mrays/s per GPU -> x{world} = {world*rays/dt/1e6:.0f}', flush=True)
