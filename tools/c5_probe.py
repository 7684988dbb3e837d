"""Config-5 probe: the 1,000,000-triangle synthetic scene (+64 spheres, 16 lights, depth 8).
Generates the scene into a scratch directory, times load/upload/BVH build, renders a small
frame brute force and through the LBVH, and checks the two frames are byte-identical."""
import os, subprocess, sys, time
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import Scene, abi
from raytracer_amd.render import Renderer
out = sys.argv[1] if len(sys.argv) > 1 else '/tmp/c5'
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 144)
modes = (1,) if len(sys.argv) > 4 and sys.argv[4] == 'bvh' else (1, 0)
t0 = time.time(); subprocess.run([sys.executable, 'tools/make_synthetic.py', out, '--nx', os.environ.get('C5_NX', '1000'), '--nz', os.environ.get('C5_NZ', '500')], check=True); t1 = time.time()
sc = Scene.load(os.path.join(out, 'synthetic.scene'), W, H, 8); t2 = time.time()
print(f'generate {t1-t0:.1f}s  load {t2-t1:.1f}s  objects {sc.desc.n_objects} lights {sc.desc.n_lights}', flush=True)
res = {}
for bvh in modes:
    t3 = time.time(); r = Renderer(sc); t4 = time.time()
    if bvh: r.set_option(abi.RT_OPT_USE_BVH, 1)
    r.set_option(abi.RT_OPT_PROFILE, 1)
    t5 = time.time(); r.render_frame(); img = r.download(); t6 = time.time()
    st = r.stats(); r.close()
    rays = st['rays_primary'] + st['rays_reflect'] + st['rays_shadow']
    print(f'bvh={bvh}: upload {t4-t3:.2f}s build {t5-t4:.2f}s render {t6-t5:.2f}s rays {rays} -> {rays/(t6-t5)/1e6:.1f} Mrays/s '
          f'flagged {st["flagged_rays"]} exact {st["exact_tests"]} box/ray {st["bvh_node_tests"]/max(1,rays):.1f} leaf/ray {st["bvh_leaf_tests"]/max(1,rays):.2f} irregular {st["n_irregular"]} '
          f'closest_ms {st["closest_ms"]:.1f} anyhit_ms {st["anyhit_ms"]:.1f} exact_ms {st["exact_ms"]:.1f}', flush=True)
    if not bvh:
        tests = st['closest_tests'] + st['anyhit_tests']
        ms = st['closest_ms'] + st['anyhit_ms']
        tri_share = st['n_triangles'] / (st['n_triangles'] + st['n_spheres'])
        print(f'    brute force: {tests * tri_share * 42 / (ms * 1e-3) / 1e12:.1f} TFLOP/s in the filter kernels (form {st["filter_form"]})', flush=True)
    res[bvh] = img
if len(res) == 2:
    print('identical:', np.array_equal(res[0], res[1]), 'differing pixels:', int((res[0] != res[1]).any(axis=-1).sum()))
