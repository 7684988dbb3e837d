import sys; sys.path.insert(0,'.')
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
sc = Scene.load('scenes/teapot.scene', 1920, 1080, 5)
for bvh in (0,1):
    r = Renderer(sc)
    if bvh: r.set_option(abi.RT_OPT_USE_BVH, 1)
    r.render_frame(); st = r.stats(); r.close()
    print(bvh, {k: st[k] for k in ('rays_primary','rays_reflect','rays_shadow','flagged_rays','exact_tests','bvh_node_tests','bvh_leaf_tests')})
