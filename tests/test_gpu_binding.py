"""The reference program with its render() statement swapped for the MI355X library.

oracle/_ref/ref_main_gpu is the reference's own parser and PNG encoder around the reference-side binding
(oracle/ref_main_gpu.cpp, compiled from the reference's sources where they lie by oracle/Makefile);
oracle/_ref/ref_render is the unmodified reference.  Same scene, same size: the PNG the former writes must be the
image the latter renders, to the bars of tests/test_gpu_parity.py.

Every scene used here has an `lta` line: without one the reference reads its ambient colour from memory it never
initialised (raytracer.cpp:473-474), and two different programs around the same parser need not find the same
bytes there."""
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_api

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINDING = os.path.join(ROOT, "oracle", "_ref", "ref_main_gpu")


@pytest.mark.skipif(not (os.path.exists(BINDING) and oracle_api.have_reference_binary()),
                    reason="oracle/_ref is built only where the reference sources are present (make -C oracle ref)")
@pytest.mark.parametrize("scene,w,h,depth", [("scenes/inputs/input-10", 160, 120, 6), ("scenes/inputs/input-11", 120, 120, 6),
                                            ("scenes/mixed_xf.scene", 160, 120, 4), ("scenes/normals.scene", 96, 72, 3),
                                            ("scenes/teapot.scene", 96, 54, 2)])
def test_reference_main_with_render_swapped(scene, w, h, depth, tmp_path):
    path = os.path.join(ROOT, scene)
    out = tmp_path / "gpu.png"
    res = subprocess.run([BINDING, path, str(out), str(w), str(h), str(depth)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    got = oracle_api.load_png(str(out))
    want = oracle_api.reference_render(path, w, h, depth, tmp=str(tmp_path))
    worst, ndiff, frac1 = oracle_api.compare(got, want)
    assert worst <= 4 and frac1 >= 0.999 and ndiff <= 2, (worst, ndiff, frac1)
    assert np.array_equal(got[..., 3], want[..., 3])
