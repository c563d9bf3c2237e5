"""ctypes binding of librt_amd.so (the C ABI declared in include/rt_amd.h).

The library holds the HIP kernels; there is no other implementation behind this module.
If the shared object is missing, loading raises with the build command -- nothing falls back
to a CPU path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB names another build of the same library (kernel experiments, counting builds)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "lib", "librt_amd.so")

RT_OK = 0
RNG_REFERENCE, RNG_COUNTER = 0, 1
PIPE_MEGAKERNEL, PIPE_WAVEFRONT = 0, 1
HITS_LEAF_ORDER, HITS_REFERENCE_ORDER, HITS_REFERENCE_WALK = 0, 1, 2
MISS_ID = 0xFFFFFFFF
INF = 10000.0


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("forward", C.c_float * 3), ("tan_x", C.c_float), ("tan_y", C.c_float)]


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("emission", C.c_float * 3), ("type", C.c_int32), ("ior", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("tri_verts", C.c_void_p), ("tri_material", C.c_void_p), ("ntri", C.c_uint32),
                ("materials", C.c_void_p), ("nmat", C.c_uint32), ("camera", Camera), ("background", C.c_float * 3),
                ("prims", C.c_void_p), ("nprim", C.c_uint32)]


class SceneStats(C.Structure):
    _fields_ = [("ntri", C.c_uint32), ("nlights", C.c_uint32), ("nnodes", C.c_uint32), ("tree_height", C.c_uint32),
                ("build_seconds", C.c_double), ("device_bytes", C.c_uint64), ("height_capped_splits", C.c_uint32),
                ("nprims", C.c_uint32), ("ntri_records", C.c_uint32), ("hit_order", C.c_uint32),
                ("ref_tree_nodes", C.c_uint32), ("ref_tree_depth", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("nodes", C.c_uint64), ("tri_tests", C.c_uint64),
                ("light_tests", C.c_uint64), ("node_lane_slots", C.c_uint64), ("tri_lane_slots", C.c_uint64),
                ("retraced", C.c_uint64)]


KERNEL_KINDS = ("extend", "shade", "generate", "accumulate", "mega", "closest_hit", "tonemap", "sort")


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_double * len(KERNEL_KINDS)), ("launches", C.c_uint64 * len(KERNEL_KINDS))]  # RT_KERNEL_KINDS


# name -> (restype, argtypes); exactly the entry points of include/rt_amd.h
SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_gltf_load": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rt_gltf_load_ex": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rt_host_scene_get_desc": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "rt_host_scene_free": (None, [C.c_void_p]),
    "rt_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_scene_get_stats": (C.c_int, [C.c_void_p, C.POINTER(SceneStats)]),
    "rt_scene_set_hit_order": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_flat_get_ref_tree": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_scene_get_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_flat_build": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]),
    "rt_flat_get_light_tree": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_flat_get": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_flat_get_wide": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p]),
    "rt_flat_get_quality": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rt_flat_free": (None, [C.c_void_p]),
    "rt_intersect_rays": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_intersect_rays_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "rt_trace_primary": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_trace_primary_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                            C.c_uint32, C.c_int, C.c_void_p]),
    "rt_render_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "rt_render_noalloc_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_multi_create": (C.c_int, [C.POINTER(SceneDesc), C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "rt_multi_destroy": (None, [C.c_void_p]),
    "rt_multi_device_count": (C.c_int, [C.c_void_p]),
    "rt_multi_scene": (C.c_void_p, [C.c_void_p, C.c_int]),
    "rt_multi_sample_range": (None, [C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rt_multi_render": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_int, C.c_void_p]),
    "rt_scene_reserve": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int]),
    "rt_selftest_rng": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "rt_tonemap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_tonemap_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rt_set_counting": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_counters_reset": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_counters_read": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "rt_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_timing_read": (C.c_int, [C.c_void_p, C.POINTER(KernelTimes)]),
    "rt_counter_seed": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
}

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("librt_amd error %d: %s" % (code, message))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C raytracing_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("RT_AMD_LIB_OLD") and not hasattr(L, name):
                continue   # timing an older build of the library against this one (tools only)
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error().decode("utf-8", "replace"))
