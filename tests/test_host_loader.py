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
from raytracer_amd.scene import SceneError
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


def _model_obj_parse(text):
    """Line-by-line model of the .obj grammar the loader accepts (raytracer.cpp:624-703):
    first token decides; numbers go text -> atof -> float; a face takes its first three tokens,
    the last token decides whether it carries normals; an index that does not name an
    earlier vertex reads as (0, 0, 0)."""
    import re
    num = re.compile(r"^[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)")

    def atof(tok):
        m = num.match(tok)
        with np.errstate(over="ignore"):                             # 80 digits: atof -> 1e79 -> float inf
            return float(np.float32(float(m.group(0)))) if m else 0.0

    def atol(tok):
        m = re.match(r"^[+-]?\d+", tok)
        return int(m.group(0)) if m else 0

    verts, norms, faces = [(0.0, 0.0, 0.0)], [(0.0, 0.0, 0.0)], []
    for line in text.split("\n"):
        toks = line.replace("\r", " ").replace("\t", " ").split(" ")
        toks = [t for t in toks if t]
        if not toks or toks[0] == "#":
            continue
        if toks[0] in ("v", "vn"):
            a = [atof(t) for t in toks[1:4]] + [0.0] * 3
            (verts if toks[0] == "v" else norms).append(tuple(a[:3]))
        elif toks[0] == "f":
            vi, ni, with_n = [0, 0, 0], [0, 0, 0], False
            for n, t in enumerate(toks[1:]):
                pos = t.find("//")
                with_n = pos >= 0
                if n < 3:
                    if with_n:
                        vi[n], ni[n] = atol(t[:pos]), atol(t[pos + 2:])
                    else:
                        vi[n] = atol(t)
            g = [verts[i] if 0 <= i < len(verts) else (0.0, 0.0, 0.0) for i in vi]
            nn = [norms[i] if with_n and 0 <= i < len(norms) else (0.0, 0.0, 0.0) for i in ni]
            faces.append((sum(map(list, g), []), sum(map(list, nn), []), int(with_n)))
    return faces


def test_blob_with_counts_beyond_its_size_is_refused(tmp_path):
    """A truncated or corrupt RTSD header must not size anything from its counts: an error, no abort,
    no allocation of gigabytes."""
    import struct
    sc = Scene.load(os.path.join(ROOT, "scenes", "inputs", "input-05"), 8, 8)
    good = tmp_path / "good.rtsd"
    sc.save_blob(str(good))
    data = bytearray(open(good, "rb").read())
    # counts sit after "RTSD", i32 version/width/height/depth, f32 eps/shadow_eps
    for bad in ((2 ** 31 - 1, 1, 1, 1), (1, 2 ** 31 - 1, 1, 1), (1, 1, 2 ** 31 - 1, 1), (1, 1, 1, 2 ** 31 - 1)):
        b = bytearray(data)
        b[28:44] = struct.pack("<4i", *bad)
        p = tmp_path / "bad.rtsd"
        open(p, "wb").write(bytes(b))
        with pytest.raises(SceneError):
            Scene.load_blob(str(p))
    p = tmp_path / "short.rtsd"
    open(p, "wb").write(bytes(data[:60]))
    with pytest.raises(SceneError):
        Scene.load_blob(str(p))
    Scene.load_blob(str(good)).close()


def test_obj_chunks_parse_like_one_pass(tmp_path, monkeypatch):
    # The .obj text is cut at line ends and parsed concurrently; faces may name vertices of
    # earlier chunks, must not see later ones, and come out in file order.
    lines = ["# head", "#tight", "v 0 0 0", "v 1 0 0\r", "v\t0\t1\t0", "v 1 2", "v 1 2 3 4", "vn 0 0 1", "vn 0 1",
             "f 1 2 3", "f 1//1 2//2 3//", "f //1 2 3", "f 1/2/3 2/5/6 3/8/9 10", "f 1//1 2//2 3", "f 1 2 3//1",
             "f 7 8 9", "f -1 2 3", "f 1 2", "f", "   ", "", "\r", "v 1e-3 +2 -.5", "v 1.5abc 2,3 4", "g group",
             "v " + "1" * 80 + " 2 3", "f 3 2 1 # trailing", "\tf 1 2 3", "f 1\t2\t3\t"]
    for k in range(1500):
        lines.append(f"v {k * 0.001:.4f} {k % 7} {-k}")
        if k % 3 == 2:
            lines.append(f"f {k} {k + 1} {k + 2}")                   # k + 2 is not defined yet
        if k % 50 == 0:
            lines += ["vn 0 1 0", f"f {k}//{k // 50} {k - 1}//1 {k - 2}//2"]
        if k % 97 == 0:
            lines.append("bogus line")
    text = "\n".join(lines) + "\nf 1 2 3"                            # no newline at the end
    (tmp_path / "m.obj").write_text(text, newline="")
    (tmp_path / "s.scene").write_text(
        "cam 0 0 2 -1 -1 1 1 -1 1 -1 1 1 1 1 1\nmat .1 .1 .1 1 1 1 0 0 0 1 0 0 0\nsph 0 0 0 1\nobj m.obj\nsph 1 1 1 2\n")
    model = _model_obj_parse(text)
    blobs = []
    for chunk in (None, "64", "1000", "20000"):
        if chunk is None:
            monkeypatch.delenv("RTH_OBJ_CHUNK_BYTES", raising=False)
        else:
            monkeypatch.setenv("RTH_OBJ_CHUNK_BYTES", chunk)
        sc = Scene.load(str(tmp_path / "s.scene"), 8, 8)
        out = tmp_path / f"c{chunk}.rtsd"
        sc.save_blob(str(out))
        blobs.append(out.read_bytes())
        d = sc.desc
        assert d.n_objects == len(model) + 2
        geom = np.ctypeslib.as_array(d.obj_geom, shape=(d.n_objects, 9))
        nrm = np.ctypeslib.as_array(d.obj_normals, shape=(d.n_objects, 9))
        kinds = [d.obj_kind[i] for i in (0, d.n_objects - 1)]
        assert kinds == [1, 1]                                       # the spheres stand before and after the mesh
        for j, (g, n, w) in enumerate(model):
            assert geom[1 + j].tolist() == g, (chunk, j)
            assert nrm[1 + j].tolist() == n, (chunk, j)
            assert d.obj_interp[1 + j] == w, (chunk, j)
    assert all(b == blobs[0] for b in blobs)
