"""Host loader (librt_host.so) against the reference's own parse.

tests/golden/rtsd/*.rtsd hold the reference's parsed Scene objects, flattened by the
reference-side binding (oracle/ref_wrapper.cpp) -- matrices, inverses, vertices,
materials, lights, object order.  The loader must reproduce them byte for byte.
"""
import glob
import os

import numpy as np
import pytest

from raytracer_amd import Scene, write_png
from tests import oracle_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOBS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "rtsd", "*.rtsd")))


def scene_path_of(blob):
    name = os.path.basename(blob)[:-5]
    for cand in (f"scenes/inputs/{name}", f"scenes/{name}.scene"):
        p = os.path.join(ROOT, cand)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


@pytest.mark.parametrize("blob", BLOBS, ids=[os.path.basename(b) for b in BLOBS])
def test_loader_matches_reference_parse(blob, tmp_path):
    sc = Scene.load(scene_path_of(blob), 100, 80)
    out = tmp_path / "mine.rtsd"
    sc.save_blob(str(out))
    assert out.read_bytes() == open(blob, "rb").read()


def test_blob_round_trip(tmp_path):
    blob = os.path.join(ROOT, "tests", "golden", "rtsd", "mixed_xf.rtsd")
    sc = Scene.load_blob(blob)
    out = tmp_path / "again.rtsd"
    sc.save_blob(str(out))
    assert out.read_bytes() == open(blob, "rb").read()
    d = sc.desc
    assert (d.width, d.height, d.max_depth) == (100, 80, 6)
    assert d.n_lights == 3 and d.n_objects == 2 + 36 + 2 + 8 + 1


def test_defaults_and_errors(tmp_path):
    sc = Scene.load(os.path.join(ROOT, "scenes", "inputs", "input-01"), 10, 10)
    assert sc.desc.max_depth == 6                      # raytracer.cpp:477
    assert np.float32(sc.desc.eps) == np.float32(0.05) and np.float32(sc.desc.shadow_eps) == np.float32(0.1)
    assert list(sc.desc.ambient) == [0.0, 0.0, 0.0]    # no lta line: ambient reads as zero
    with pytest.raises(Exception):
        Scene.load(str(tmp_path / "missing.scene"), 10, 10)
    bad = tmp_path / "bad.scene"
    bad.write_text("mat 1 1 1 1 1 1 1 1 1 1 0 0 0\nsph 0 0 0 1\n")
    with pytest.raises(Exception):                     # no cam line
        Scene.load(str(bad), 10, 10)
    empty = tmp_path / "empty.scene"
    empty.write_text("cam 0 0 2 -1 -1 1 1 -1 1 -1 1 1 1 1 1\n# nothing else\n\n")
    sc = Scene.load(str(empty), 10, 10)
    assert sc.desc.n_objects == 0 and sc.desc.n_lights == 0


def test_obj_grammar(tmp_path):
    # v / vn / f a b c / f a//n b//n c//n / a/b/c tokens, '#' lines, unknown lines are skipped
    (tmp_path / "m.obj").write_text(
        "# comment\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvn 0 1 0\nvn 1 0 0\n"
        "g grp\nf 1 2 3\nf 1//1 2//2 4//3\nf 2/9/9 3/9/9 4/9/9\n")
    (tmp_path / "s.scene").write_text(
        "cam 0 0 2 -1 -1 1 1 -1 1 -1 1 1 1 1 1\nmat .1 .1 .1 1 1 1 0 0 0 1 0 0 0\nxft 1 2 3\nobj m.obj\n")
    sc = Scene.load(str(tmp_path / "s.scene"), 8, 8)
    d = sc.desc
    assert d.n_objects == 3 and d.n_transforms == 1 and d.n_materials == 1
    assert [d.obj_interp[i] for i in range(3)] == [0, 1, 0]
    geom = np.ctypeslib.as_array(d.obj_geom, shape=(3, 9))
    assert geom[0].tolist() == [0, 0, 0, 1, 0, 0, 0, 1, 0]
    assert geom[2].tolist() == [1, 0, 0, 0, 1, 0, 0, 0, 1]
    nrm = np.ctypeslib.as_array(d.obj_normals, shape=(3, 9))
    assert nrm[1].tolist() == [0, 0, 1, 0, 1, 0, 1, 0, 0]
    m = list(d.transforms[0].m)
    assert m[3] == 1.0 and m[7] == 2.0 and m[11] == 3.0
    inv = list(d.transforms[0].inv)
    assert inv[3] == -1.0 and inv[7] == -2.0 and inv[11] == -3.0


def test_png_writer_round_trip(tmp_path):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(37, 53, 4), dtype=np.uint8)
    p = tmp_path / "x.png"
    write_png(str(p), img.tobytes(), 53, 37)
    back = oracle_api.load_png(str(p))
    assert np.array_equal(back, img)


def test_png_writer_bands_round_trip(tmp_path):
    """The PNG writer deflates bands of rows on several threads and joins them into one zlib
    stream; sizes around the band boundaries (64-row minimum, 16 bands) must decode to the input."""
    import numpy as np
    from raytracer_amd import write_png
    from tests import oracle_api
    rng = np.random.default_rng(3)
    for w, h in ((1, 1), (7, 63), (7, 64), (5, 127), (33, 128), (16, 1025), (64, 2161)):
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        img[: h // 2] = (img[: h // 2] // 64) * 64          # half compressible, half noise
        path = os.path.join(str(tmp_path), f"t_{w}x{h}.png")
        write_png(path, img.tobytes(), w, h)
        assert np.array_equal(oracle_api.load_png(path), img), (w, h)


def test_loader_survives_malformed_input(tmp_path):
    """Mutated scene and OBJ text (missing or surplus arguments, non-numbers, huge indices,
    unknown commands, blank lines): the loader either returns a scene or an error, never
    crashes, and what it returns can be saved and loaded again.  (The same corpus runs clean
    under AddressSanitizer and UBSan when the loader is built with them.)"""
    import random
    from raytracer_amd.scene import SceneError
    rnd = random.Random(11)
    base = open(os.path.join(ROOT, "scenes", "mixed_xf.scene")).read().splitlines()
    for name in ("dodec.obj", "oct.obj"):
        with open(os.path.join(str(tmp_path), name), "w") as f:
            f.write(open(os.path.join(ROOT, "scenes", name)).read())
    tokens = ["nan", "inf", "1e400", "abc", "1/2/3", "//", "-1", "2147483648", "missing.obj", "dodec.obj", "", "7"]
    loaded = 0
    for k in range(120):
        lines = list(base)
        for _ in range(rnd.randint(1, 10)):
            i = rnd.randrange(len(lines))
            t = lines[i].split()
            op = rnd.random()
            if t and op < 0.3:
                del t[rnd.randrange(len(t))]
            elif t and op < 0.6:
                t[rnd.randrange(len(t))] = rnd.choice(tokens)
            elif op < 0.8:
                t += [rnd.choice(tokens) for _ in range(rnd.randint(1, 12))]
            else:
                lines.insert(i, rnd.choice(["bogus 1 2", "\t ", "f 1 2", "tri", "sph 1", "obj"]))
            lines[i] = " ".join(t)
        path = os.path.join(str(tmp_path), f"m{k}.scene")
        with open(path, "w") as f:
            f.write("\n".join(lines) + "\n")
        try:
            sc = Scene.load(path, 16, 12, 2)
        except SceneError:
            continue
        loaded += 1
        blob = path + ".rtsd"
        sc.save_blob(blob)
        again = Scene.load_blob(blob)
        assert again.desc.n_objects == sc.desc.n_objects
    assert loaded > 0
