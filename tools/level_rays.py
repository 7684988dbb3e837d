"""Rays per reflection level of rank 0's share of a frame (teapot 4K), from runs at depth 0..5."""
import sys
sys.path.insert(0, '.')
from raytracer_amd import Scene
from raytracer_amd.render import Renderer
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prev = (0, 0)
for d in range(6):
    r = Renderer(Scene.load('scenes/teapot.scene', 3840, 2160, d))
    r.render_tiles(0, 16, world)
    r.download()
    s = r.stats(); r.close()
    closest = s['rays_primary'] + s['rays_reflect']
    print(f'level {d}: closest rays {closest - prev[0]:9d}  shadow rays {s["rays_shadow"] - prev[1]:9d}')
    prev = (closest, s['rays_shadow'])
