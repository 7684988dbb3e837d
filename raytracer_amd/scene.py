"""Host-side scene handle: wraps librt_host.so's loader (rth_*)."""
import ctypes as C
import os

from . import abi


class SceneError(RuntimeError):
    pass


class Scene:
    """A parsed scene; `.desc` is the rt_scene_desc handed to rt_upload_scene.

    Mirrors what parseSceneFile returns in the reference (raytracer.cpp:705):
    camera, objects in file order, lights, ambient, film size, eps, depth.
    """

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self._lib = abi.host_lib()

    @classmethod
    def load(cls, path, width=1000, height=1000, max_depth=-1, eps=0.0):
        lib = abi.host_lib()
        h = C.c_void_p()
        rc = lib.rth_load_scene(C.byref(h), os.fsencode(path), int(width), int(height), int(max_depth), float(eps))
        if rc != 0:
            raise SceneError(f"rth_load_scene({path}) failed ({rc}): {lib.rth_last_error().decode()}")
        return cls(h.value)

    @classmethod
    def load_blob(cls, path):
        lib = abi.host_lib()
        h = C.c_void_p()
        rc = lib.rth_load_blob(C.byref(h), os.fsencode(path))
        if rc != 0:
            raise SceneError(f"rth_load_blob({path}) failed ({rc}): {lib.rth_last_error().decode()}")
        return cls(h.value)

    def save_blob(self, path):
        rc = self._lib.rth_save_blob(self._h, os.fsencode(path))
        if rc != 0:
            raise SceneError(f"rth_save_blob({path}) failed ({rc}): {self._lib.rth_last_error().decode()}")

    @property
    def desc_ptr(self):
        return self._lib.rth_desc(self._h)

    @property
    def desc(self):
        return self.desc_ptr.contents

    @property
    def width(self):
        return self.desc.width

    @property
    def height(self):
        return self.desc.height

    def close(self):
        if self._h:
            self._lib.rth_free_scene(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_png(path, rgba, width, height):
    """rgba: bytes-like of width*height*4 bytes, row 0 = top scanline."""
    lib = abi.host_lib()
    buf = (C.c_uint8 * (width * height * 4)).from_buffer_copy(bytes(rgba))
    rc = lib.rth_write_png(os.fsencode(path), buf, int(width), int(height))
    if rc != 0:
        raise SceneError(f"rth_write_png({path}) failed ({rc})")
