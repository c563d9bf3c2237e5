"""ctypes view of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.  Every function mirrors one extern "C" entry
of oracle/cpu_ref.cpp, which cites the reference lines it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

MISS_ID = 0xFFFFFFFF
INF = np.float32(10000.0)


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("forward", C.c_float * 3), ("tan_x", C.c_float), ("tan_y", C.c_float)]


class Material(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("emission", C.c_float * 3), ("type", C.c_int32),
                ("ior", C.c_float)]


MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("emission", "<f4", 3), ("type", "<i4"), ("ior", "<f4")])
PRIMITIVE_DTYPE = np.dtype([("type", "<i4"), ("position", "<f4", 3), ("rotation", "<f4", 4), ("param", "<f4", 3),
                            ("material", "<u4")])


def build(force=False):
    """Compile the oracle with its Makefile (g++ only, a second or two)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cpu_ref.cpp")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.POINTER(Camera)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_set_background.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.orc_scene_set_primitives.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        L.orc_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_tree.restype = C.c_uint32
        L.orc_scene_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_intersect_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_primary_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_tonemap.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_rng_kat.argtypes = [C.c_uint32] + [C.c_void_p] * 6
        L.orc_rng_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_counter_seed.restype = C.c_uint32
        L.orc_counter_seed.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_triangle_derived.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_lights.restype = C.c_uint32
        L.orc_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


COUNTER_NAMES = ("segments", "nodes", "tri_tests", "box_tests", "light_tests")


class OracleScene:
    """InputData of the reference (main.cpp:689-706) held by the CPU oracle.

    tri_verts: (N, 3, 3) float32 world-space vertices in loader emission order,
    tri_material: (N,) uint32, materials: structured array MATERIAL_DTYPE,
    camera: dict(position, right, up, forward, tan_x, tan_y).
    """

    def __init__(self, tri_verts, tri_material, materials, camera, background=None, prims=None, prims_in_tree=False):
        """prims_in_tree: analytic primitives are placed as BVH::build would place them (planes at the root, the
        others in the tree under their own boxes, main.cpp:552-558) instead of being folded in after the walk."""
        tv = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
        tm = np.ascontiguousarray(tri_material, dtype=np.uint32)
        mats = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        assert tv.shape[0] == tm.shape[0]
        cam = Camera()
        for k in ("position", "right", "up", "forward"):
            getattr(cam, k)[:] = [float(x) for x in camera[k]]
        cam.tan_x = float(camera["tan_x"])
        cam.tan_y = float(camera["tan_y"])
        self.ntri = tv.shape[0]
        self.nprims = 0 if prims is None else len(prims)
        self.prims_in_tree = bool(prims_in_tree) and self.nprims > 0
        self._h = lib().orc_scene_create(_p(tv), _p(tm), self.ntri, _p(mats), mats.shape[0], C.byref(cam))
        if background is not None:
            lib().orc_scene_set_background(self._h, *[float(x) for x in background])
        if prims is not None and len(prims):
            pr = np.ascontiguousarray(prims, dtype=PRIMITIVE_DTYPE)
            lib().orc_scene_set_primitives(self._h, _p(pr), pr.shape[0], int(bool(prims_in_tree)))

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = np.zeros(4, dtype=np.uint64)
        lib().orc_scene_info(self._h, _p(out))
        return dict(nodes=int(out[0]), lights=int(out[1]), unsplittable=int(out[2]), max_depth=int(out[3]))

    def tree(self):
        """BVH::build's result: (nodes (n, 4) u32: left, right, first, count; boxes (n, 12) f32: sizePlusCenter,
        negSizePlusCenter, minPoint, maxPoint; items: the re-ordered primitive list as loader ids)."""
        n = lib().orc_scene_tree(self._h, None, None, None)
        nodes = np.empty((n, 4), dtype=np.uint32)
        boxes = np.empty((n, 12), dtype=np.float32)
        items = np.empty(max(self.ntri + self.nprims, 1), dtype=np.uint32)
        lib().orc_scene_tree(self._h, _p(nodes), _p(boxes), _p(items))
        return nodes, boxes, items[:self.ntri + self.nprims if self.prims_in_tree else self.ntri]

    def intersect(self, o, d, use_bvh=True):
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        ids = np.empty(n, dtype=np.uint32)
        t = np.empty(n, dtype=np.float32)
        tie = np.empty(n, dtype=np.uint8)
        cnt = np.zeros(5, dtype=np.uint64)
        lib().orc_intersect_rays(self._h, n, _p(o), _p(d), _p(ids), _p(t), _p(tie), _p(cnt), int(use_bvh))
        return ids, t, tie, dict(zip(COUNTER_NAMES, (int(x) for x in cnt)))

    def primary_rays(self, W, H):
        o = np.empty((W * H, 3), dtype=np.float32)
        d = np.empty((W * H, 3), dtype=np.float32)
        lib().orc_primary_rays(self._h, W, H, _p(o), _p(d))
        return o, d

    def trace_primary(self, W, H):
        o, d = self.primary_rays(W, H)
        return self.intersect(o, d)

    def render(self, W, H, spp, max_depth=6, rng_mode=0, seed=0, spp_begin=0, nthreads=0):
        out = np.empty((H * W, 3), dtype=np.float32)
        cnt = np.zeros(5, dtype=np.uint64)
        rc = lib().orc_render(self._h, W, H, spp_begin, spp, max_depth, rng_mode, seed, _p(out), _p(cnt),
                              nthreads)
        if rc != 0:
            raise ValueError("orc_render rejected the arguments (rng_mode 0 needs spp_begin == 0)")
        return out, dict(zip(COUNTER_NAMES, (int(x) for x in cnt)))

    def triangle_derived(self):
        out = np.empty((self.ntri, 4), dtype=np.float32)
        lib().orc_triangle_derived(self._h, _p(out))
        return out

    def lights(self):
        out = np.empty(max(self.ntri + self.nprims, 1), dtype=np.uint32)
        n = lib().orc_lights(self._h, _p(out), out.shape[0])
        return out[:n].copy()


def tonemap(radiance_sum, spp):
    """main.cpp:1226-1231 + 840-846: returns (float image in [0,1], uint8 image)."""
    rs = np.ascontiguousarray(radiance_sum, dtype=np.float32).reshape(-1, 3)
    f = np.empty_like(rs)
    b = np.empty(rs.shape, dtype=np.uint8)
    lib().orc_tonemap(_p(rs), rs.shape[0], spp, _p(f), _p(b))
    return f, b


def rng_kat(seed):
    raw = np.zeros(4, dtype=np.uint64)
    uni = np.zeros(3, dtype=np.float32)
    nrm = np.zeros(3, dtype=np.float32)
    nd = np.zeros(1, dtype=np.uint32)
    ui2 = np.zeros(4, dtype=np.uint16)
    ui12 = np.zeros(4, dtype=np.uint16)
    lib().orc_rng_kat(seed, _p(raw), _p(uni), _p(nrm), _p(nd), _p(ui2), _p(ui12))
    return dict(raw=raw, uni=uni, nrm=nrm, nrm_draws=int(nd[0]), ui2=ui2, ui12=ui12)


def rng_stream(seed, n, L):
    uni = np.zeros(n, dtype=np.float32)
    nrm = np.zeros(n, dtype=np.float32)
    ui = np.zeros(n, dtype=np.uint16)
    lib().orc_rng_stream(seed, n, L, _p(uni), _p(nrm), _p(ui))
    return uni, nrm, ui


def counter_seed(pixel, sample, seed):
    return int(lib().orc_counter_seed(pixel, sample, seed))


def num_threads():
    return int(lib().orc_num_threads())
