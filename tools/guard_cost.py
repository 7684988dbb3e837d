"""What the in-plane-origin scan behind the LBVH traversal costs (RT_OPT_BVH_PLANE_GUARD on / off): milliseconds
per frame, one frame at a time and four in flight, for the 4K teapot, the 8K campanile + teddy and (argument
"c5": the 8K frame of the 1,000,000-triangle scene)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer


def timed(fn, steps):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def probe(name, sc, steps):
    r = Renderer(sc)
    r.set_option(abi.RT_OPT_USE_BVH, 1)
    stream = torch.cuda.current_stream().cuda_stream
    frames = [torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device='cuda') for _ in range(4)]
    calls = [0]

    def render(k):
        r.bind_frame(frames[calls[0] % k]); calls[0] += 1
        r.render_frame(stream)
    for guard in (1, 0, 1, 0):
        r.set_option(abi.RT_OPT_BVH_PLANE_GUARD, guard)
        res = []
        for k in (1, 4):
            r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, k)
            res.append(timed(lambda: render(k), steps))
        r.set_option(abi.RT_OPT_FRAMES_IN_FLIGHT, 1)
        r.reset_stats(); r.render_frame(stream); st = r.stats()
        print(f'{name}: guard {guard}: {res[0]:.2f} ms one at a time, {res[1]:.2f} ms with 4 in flight; '
              f'exact re-tests per frame {st["exact_tests"]}, in-plane hits {st["in_plane_origin_hits"]}', flush=True)
    r.close()


probe('teapot 4K d5', Scene.load('scenes/teapot.scene', 3840, 2160, 5), 12)
probe('campanile+teddy 8K d6', Scene.load('scenes/campanile_teddy.scene', 7680, 4320, 6), 3)
if 'c5' in sys.argv:
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([sys.executable, 'tools/make_synthetic.py', tmp], check=True, stdout=subprocess.DEVNULL)
        probe('synthetic 1M 8K d8', Scene.load(os.path.join(tmp, 'synthetic.scene'), 7680, 4320, 8), 2)
