#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the UNMODIFIED reference.

Needs oracle/_ref/ref_render (built by `make -C oracle ref` where the reference
sources are present).  Outputs are data only:
  ref_renders/<scene>_<W>x<H>_d<depth>.png   RGBA image exactly as Film::writeFile lays it out
  rtsd/<scene>.rtsd                          the reference's parsed Scene, flattened by the
                                             reference-side binding in oracle/ref_wrapper.cpp
The committed 1000x1000 PNGs under images/ are the reference authors' own renders and
are not produced here.
"""
import os
import subprocess
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_render")

# (scene path relative to scenes/, width, height, depth)
RENDERS = [
    ("inputs/input-01", 160, 120, 6), ("inputs/input-02", 160, 120, 6), ("inputs/input-03", 160, 120, 6),
    ("inputs/input-04", 160, 120, 6), ("inputs/input-05", 160, 120, 6), ("inputs/input-06", 160, 120, 6),
    ("inputs/input-07", 160, 120, 6), ("inputs/input-08", 160, 120, 6), ("inputs/input-09", 160, 120, 6),
    ("inputs/input-10", 160, 120, 6), ("inputs/input-11", 160, 120, 6),
    ("ellipse.scene", 192, 108, 6), ("scene1.scene", 160, 120, 6), ("scene2.scene", 160, 120, 6),
    ("scene3.scene", 64, 48, 6), ("s3.scene", 160, 120, 6), ("sphere.scene", 160, 120, 6),
    ("journal.scene", 160, 120, 6), ("journal2.scene", 160, 120, 6),
    ("teapot.scene", 96, 54, 5), ("teapot.scene", 64, 36, 0),
    ("campanile_teddy.scene", 64, 36, 2), ("mixed_xf.scene", 160, 120, 4), ("normals.scene", 160, 120, 3),
    ("synthetic_small.scene", 96, 54, 3),
]
BLOBS = ["inputs/input-05", "inputs/input-07", "inputs/input-08", "inputs/input-09", "inputs/input-11",
         "ellipse.scene", "scene1.scene", "s3.scene", "mixed_xf.scene", "normals.scene"]


def name_of(scene):
    return os.path.basename(scene).replace(".scene", "")


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_render is missing: run `make -C oracle ref` next to the reference sources")
    for scene, w, h, depth in RENDERS:
        path = os.path.join(ROOT, "scenes", scene)
        raw = os.path.join(HERE, "tmp.rgba")
        subprocess.run([REF, path, str(w), str(h), str(depth), raw, "8"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        img = np.fromfile(raw, dtype=np.uint8).reshape(h, w, 4)
        os.remove(raw)
        out = os.path.join(HERE, "ref_renders", f"{name_of(scene)}_{w}x{h}_d{depth}.png")
        Image.fromarray(img, "RGBA").save(out, optimize=True)
        print("wrote", os.path.relpath(out, ROOT))
    for scene in BLOBS:
        path = os.path.join(ROOT, "scenes", scene)
        out = os.path.join(HERE, "rtsd", name_of(scene) + ".rtsd")
        subprocess.run([REF, path, "100", "80", "6", "-", "1", "--dump", out, "--no-render"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        print("wrote", os.path.relpath(out, ROOT))


if __name__ == "__main__":
    main()
