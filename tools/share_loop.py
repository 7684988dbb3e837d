"""Renders the share of rank R of an N-GPU job (interleaved 16-row tiles of the 4K teapot frame) F times, for rocprofv3.
usage: python tools/share_loop.py N [R] [F] [overlap]"""
import sys, time
sys.path.insert(0, '.')
import torch
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 10
overlap = int(sys.argv[4]) if len(sys.argv) > 4 else 1
r = Renderer(Scene.load('scenes/teapot.scene', 3840, 2160, 5))
r.set_option(abi.RT_OPT_OVERLAP, overlap)
st = torch.cuda.current_stream().cuda_stream
r.render_tiles(rank, 16, world, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    r.render_tiles(rank, 16, world, st)
torch.cuda.synchronize()
print(f"world {world} rank {rank} overlap {overlap}: {(time.perf_counter() - t0) / frames * 1e3:.3f} ms per share")
