"""ctypes mirror of include/rt_abi.h and include/rt_host.h.

Only declarations live here: structure layouts, function prototypes and the
loaders for the two in-tree shared libraries.  There is no fallback: a missing
library raises, it is never replaced by Python code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

RT_OK, RT_ERR_ARG, RT_ERR_DEVICE, RT_ERR_RANGE, RT_ERR_STATE = 0, -1, -2, -3, -4
RT_OPT_EXACT_ONLY, RT_OPT_PROFILE, RT_OPT_PASS_PIXELS, RT_OPT_USE_BVH, RT_OPT_FILTER_FORM, RT_OPT_SHARED_CONSTANTS, RT_OPT_OVERLAP = 1, 2, 3, 4, 5, 6, 7
RT_OPT_BOUNDING_TESTS = 8
RT_OPT_WORK_COUNTERS = 9
RT_OPT_SCAN_COPIES = 10
RT_OPT_FRAMES_IN_FLIGHT = 11
RT_OPT_IN_PLANE_ORIGIN = 12
RT_OPT_BVH_PLANE_GUARD = 12          # (earlier name)
SCAN_KINDS = ["plain_nearest", "camera", "shadow_point", "shadow_directional", "reflection", "plain_shadow"]


class RtMaterial(C.Structure):
    _fields_ = [("ka", C.c_double * 3), ("kd", C.c_double * 3), ("ks", C.c_double * 3),
                ("kr", C.c_double * 3), ("sp", C.c_double)]


class RtTransform(C.Structure):
    _fields_ = [("m", C.c_double * 16), ("inv", C.c_double * 16), ("inv_t", C.c_double * 16)]


class RtLight(C.Structure):
    _fields_ = [("color", C.c_double * 3), ("pos", C.c_double * 3),
                ("falloff", C.c_int32), ("is_point", C.c_int32)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32),
        ("max_depth", C.c_int32), ("eps", C.c_float), ("shadow_eps", C.c_float),
        ("cam_eye", C.c_double * 3), ("cam_ll", C.c_double * 3), ("cam_lr", C.c_double * 3),
        ("cam_ul", C.c_double * 3), ("cam_ur", C.c_double * 3), ("ambient", C.c_double * 3),
        ("n_objects", C.c_int32), ("n_materials", C.c_int32),
        ("n_transforms", C.c_int32), ("n_lights", C.c_int32),
        ("obj_kind", C.POINTER(C.c_int32)), ("obj_material", C.POINTER(C.c_int32)),
        ("obj_transform", C.POINTER(C.c_int32)), ("obj_interp", C.POINTER(C.c_int32)),
        ("obj_geom", C.POINTER(C.c_double)), ("obj_normals", C.POINTER(C.c_double)),
        ("materials", C.POINTER(RtMaterial)), ("transforms", C.POINTER(RtTransform)),
        ("lights", C.POINTER(RtLight)),
    ]


class RtScanWork(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("tasks", "groups", "group_tests", "groups_live", "batches", "ray_bound_tests",
                                          "batches_tested", "tri_tests", "tri_tests_wanted", "listed", "clusters")]


class RtStats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_reflect", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("hits", C.c_uint64), ("closest_tests", C.c_uint64), ("anyhit_tests", C.c_uint64),
        ("exact_tests", C.c_uint64), ("bvh_node_tests", C.c_uint64), ("bvh_leaf_tests", C.c_uint64),
        ("in_plane_origin_hits", C.c_uint64), ("flagged_rays", C.c_uint64),
        ("closest_ms", C.c_double), ("anyhit_ms", C.c_double),
        ("exact_ms", C.c_double),
        ("closest_launches", C.c_uint32), ("anyhit_launches", C.c_uint32), ("exact_launches", C.c_uint32),
        ("n_triangles", C.c_int32), ("n_spheres", C.c_int32), ("n_groups", C.c_int32),
        ("n_irregular", C.c_int32), ("filter_form", C.c_int32),
        ("camera_ms", C.c_double), ("camera_launches", C.c_uint32), ("bvh_plane_guard", C.c_uint32),
        ("scan_work", RtScanWork * 6), ("scan_ms", C.c_double * 6), ("scan_launches", C.c_uint32 * 6),
        ("lanes", C.c_uint32), ("lane_hw_queues_found", C.c_uint32), ("lanes_share_a_queue", C.c_int32),
        ("caller_stream_shares_a_queue", C.c_int32),
    ]

    def as_dict(self):
        out = {name: getattr(self, name) for name, _ in self._fields_ if not name.startswith("scan_")}
        out["scan_work"] = {kind: {n: getattr(self.scan_work[k], n) for n, _ in RtScanWork._fields_}
                            for k, kind in enumerate(SCAN_KINDS)}
        out["scan_ms"] = {kind: self.scan_ms[k] for k, kind in enumerate(SCAN_KINDS)}
        out["scan_launches"] = {kind: self.scan_launches[k] for k, kind in enumerate(SCAN_KINDS)}
        return out


HOST_SYMBOLS = ["rth_load_scene", "rth_load_blob", "rth_save_blob", "rth_desc", "rth_free_scene",
                "rth_write_png", "rth_tile_span", "rth_packed_share", "rth_last_error"]
DEVICE_SYMBOLS = ["rt_upload_scene", "rt_set_option", "rt_render_tile", "rt_render_tiles", "rt_join", "rt_render_tile_noalloc",
                  "rt_bind_frame", "rt_download", "rt_primary_hits", "rt_bounce_probe", "rt_queue_pixel", "rt_get_stats", "rt_reset_stats", "rt_device_info",
                  "rt_free", "rt_last_error"]

_host = None
_device = None


def host_lib_path():
    return os.path.join(_HERE, "librt_host.so")


def device_lib_path():
    # RAYTRACER_AMD_LIB: another build of the same library (A/B measurements of kernel variants)
    return os.environ.get("RAYTRACER_AMD_LIB") or os.path.join(_HERE, "libraytracer_amd.so")


def host_lib():
    """librt_host.so: scene loader and PNG writer (CPU only)."""
    global _host
    if _host is None:
        lib = C.CDLL(host_lib_path())
        lib.rth_load_scene.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float]
        lib.rth_load_scene.restype = C.c_int
        lib.rth_load_blob.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        lib.rth_load_blob.restype = C.c_int
        lib.rth_save_blob.argtypes = [C.c_void_p, C.c_char_p]
        lib.rth_save_blob.restype = C.c_int
        lib.rth_desc.argtypes = [C.c_void_p]
        lib.rth_desc.restype = C.POINTER(RtSceneDesc)
        lib.rth_free_scene.argtypes = [C.c_void_p]
        lib.rth_free_scene.restype = None
        lib.rth_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
        lib.rth_write_png.restype = C.c_int
        lib.rth_tile_span.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        lib.rth_tile_span.restype = C.c_int
        lib.rth_packed_share.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)] + [C.POINTER(C.c_size_t)] * 5
        lib.rth_packed_share.restype = C.c_int
        lib.rth_last_error.argtypes = []
        lib.rth_last_error.restype = C.c_char_p
        _host = lib
    return _host


def device_lib():
    """libraytracer_amd.so: the HIP kernels behind the C ABI.  Raises if it was not built."""
    global _device
    if _device is None:
        path = device_lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make hip` (the trace loop has no CPU fallback)")
        lib = C.CDLL(path)
        lib.rt_upload_scene.argtypes = [C.POINTER(C.c_void_p), C.POINTER(RtSceneDesc)]
        lib.rt_upload_scene.restype = C.c_int
        lib.rt_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        lib.rt_set_option.restype = C.c_int
        lib.rt_render_tile.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.rt_render_tile.restype = C.c_int
        lib.rt_render_tile_noalloc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.rt_render_tile_noalloc.restype = C.c_int
        lib.rt_render_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.rt_render_tiles.restype = C.c_int
        lib.rt_join.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.rt_join.restype = C.c_int
        lib.rt_bind_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.rt_bind_frame.restype = C.c_int
        lib.rt_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.rt_download.restype = C.c_int
        lib.rt_primary_hits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        lib.rt_primary_hits.restype = C.c_int
        lib.rt_bounce_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.rt_bounce_probe.restype = C.c_int
        lib.rt_queue_pixel.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.rt_queue_pixel.restype = C.c_int
        lib.rt_get_stats.argtypes = [C.c_void_p, C.POINTER(RtStats)]
        lib.rt_get_stats.restype = C.c_int
        lib.rt_reset_stats.argtypes = [C.c_void_p]
        lib.rt_reset_stats.restype = C.c_int
        lib.rt_device_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.rt_device_info.restype = C.c_int
        lib.rt_free.argtypes = [C.c_void_p]
        lib.rt_free.restype = None
        lib.rt_last_error.argtypes = []
        lib.rt_last_error.restype = C.c_char_p
        _device = lib
    return _device
