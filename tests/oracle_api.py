"""ctypes access to oracle/librt_oracle.so -- the CPU restatement used as the CHECKER.

Test infrastructure only: nothing under raytracer_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "librt_oracle.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_render")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays_primary", "rays_reflect", "rays_shadow", "hits", "closest_tests", "shadow_tests")]


class Oracle:
    def __init__(self):
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} missing: run `make -C oracle oracle`")
        self.lib = C.CDLL(LIB)
        self.lib.rto_render.restype = C.c_int
        self.lib.rto_closest.restype = C.c_int
        self.lib.rto_occluded.restype = C.c_int
        self.lib.rto_trace.restype = C.c_int
        self.lib.rto_camera_row.restype = C.c_int
        self.lib.rto_bounce.restype = C.c_int

    def render(self, scene, y0=0, y1=None, threads=8):
        """Returns (rgba[H, W, 4] uint8 in Film::writeFile layout, counters dict)."""
        d = scene.desc
        y1 = d.height if y1 is None else y1
        out = np.zeros((d.height, d.width, 4), dtype=np.uint8)
        cnt = Counters()
        rc = self.lib.rto_render(scene.desc_ptr, C.c_int(y0), C.c_int(y1), C.c_int(threads),
                                 out.ctypes.data_as(C.c_void_p), C.byref(cnt))
        assert rc == 0
        return out, {n: getattr(cnt, n) for n, _ in Counters._fields_}

    def camera_row(self, scene, y):
        w = scene.desc.width
        o = np.zeros((w, 3)); d = np.zeros((w, 3))
        self.lib.rto_camera_row(scene.desc_ptr, C.c_int(y), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
        return o, d

    def closest(self, scene, orig, dirs):
        n = len(orig)
        orig = np.ascontiguousarray(orig, dtype=np.float64); dirs = np.ascontiguousarray(dirs, dtype=np.float64)
        hit = np.zeros(n, dtype=np.int32); t = np.zeros(n, dtype=np.float32); d = np.zeros(n, dtype=np.float32)
        pt = np.zeros((n, 3))
        self.lib.rto_closest(scene.desc_ptr, orig.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p),
                             C.c_int(n), hit.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                             d.ctypes.data_as(C.c_void_p), pt.ctypes.data_as(C.c_void_p))
        return hit, t, d, pt


    def _bounce(self, scene, orig, dirs):
        n = len(orig)
        words = max(1, (scene.desc.n_lights + 31) // 32)
        orig = np.ascontiguousarray(orig, dtype=np.float64); dirs = np.ascontiguousarray(dirs, dtype=np.float64)
        hit = np.zeros(n, dtype=np.int32); pt = np.zeros((n, 3)); refl = np.zeros((n, 3))
        occl = np.zeros((n, words), dtype=np.uint32)
        self.lib.rto_bounce(scene.desc_ptr, orig.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p), C.c_int(n),
                            hit.ctypes.data_as(C.c_void_p), pt.ctypes.data_as(C.c_void_p), refl.ctypes.data_as(C.c_void_p),
                            occl.ctypes.data_as(C.c_void_p))
        return hit, pt, refl, occl

    def bounce(self, scene, orig, dirs):
        """One bounce: (hit index or -1, hit point, reflected direction, occlusion words per light) per ray."""
        return self._bounce(scene, orig, dirs)

    def occluded(self, scene, orig, dirs, source):
        n = len(orig)
        orig = np.ascontiguousarray(orig, dtype=np.float64); dirs = np.ascontiguousarray(dirs, dtype=np.float64)
        source = np.ascontiguousarray(source, dtype=np.int32)
        out = np.zeros(n, dtype=np.int32)
        self.lib.rto_occluded(scene.desc_ptr, orig.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p),
                              source.ctypes.data_as(C.c_void_p), C.c_int(n), out.ctypes.data_as(C.c_void_p))
        return out

    def trace(self, scene, orig, dirs, depth):
        n = len(orig)
        orig = np.ascontiguousarray(orig, dtype=np.float64); dirs = np.ascontiguousarray(dirs, dtype=np.float64)
        rgb = np.zeros((n, 3))
        self.lib.rto_trace(scene.desc_ptr, orig.ctypes.data_as(C.c_void_p), dirs.ctypes.data_as(C.c_void_p), C.c_int(n),
                           C.c_int(depth), rgb.ctypes.data_as(C.c_void_p))
        return rgb


def have_reference_binary():
    return os.path.exists(REF_BIN)


def reference_render(scene_path, width, height, depth, threads=8, tmp="/tmp"):
    """Runs the unmodified reference (oracle/_ref) and returns rgba[H, W, 4]."""
    raw = os.path.join(tmp, f"ref_{os.getpid()}.rgba")
    subprocess.run([REF_BIN, scene_path, str(width), str(height), str(depth), raw, str(threads)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    img = np.fromfile(raw, dtype=np.uint8).reshape(height, width, 4)
    os.remove(raw)
    return img


def load_png(path):
    from PIL import Image
    return np.array(Image.open(path).convert("RGBA"))


def compare(a, b):
    """(max abs channel difference, number of differing pixels, fraction of pixels within 1 level)."""
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))[..., :3]
    per_px = d.max(axis=-1)
    return int(d.max()), int((per_px > 0).sum()), float((per_px <= 1).mean())
