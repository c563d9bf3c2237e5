"""Host-side mirror of the reference's three seams, on top of the C ABI:

    parse_gltf(path, width, height)  ~ InputData parseGLTF(string&, w, h)      main.cpp:866
    generate_scene(input_data)       ~ SceneFloat generateScene(const InputData&)  main.cpp:1235
    print_image(scene_float, path)   ~ void printImage(SceneFloat&, string&)   main.cpp:856

plus `Scene`, the device-resident scene with the lower-level calls (closest hit, partial
sample ranges for multi-GPU, tone mapping).  All compute runs in librt_amd.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (PIPE_MEGAKERNEL, PIPE_WAVEFRONT, RNG_COUNTER, RNG_REFERENCE, HITS_LEAF_ORDER, HITS_REFERENCE_ORDER, HITS_REFERENCE_WALK,
                   MISS_ID, INF, RtError, check)

MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("emission", "<f4", 3), ("type", "<i4"), ("ior", "<f4")])
# rt_primitive: Plane / Ellipsoid / Box of main.cpp:242-464 (type 0 / 1 / 2; rotation = unit quaternion x, y, z, w;
# param = plane normal / ellipsoid radii / box half extents)
PRIMITIVE_DTYPE = np.dtype([("type", "<i4"), ("position", "<f4", 3), ("rotation", "<f4", 4), ("param", "<f4", 3),
                            ("material", "<u4")])
PRIM_PLANE, PRIM_ELLIPSOID, PRIM_BOX = 0, 1, 2


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _stream_ptr(stream):
    if stream is None:
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)  # torch.cuda.Stream


class HostScene:
    """A parsed scene file on the host: triangle soup, materials, camera (InputData's host half)."""

    def __init__(self, tri_verts, tri_material, materials, camera, background=(0.0, 0.0, 0.0), prims=None):
        self.background = np.asarray(background, dtype=np.float32).reshape(3)  # InputData::backgroundColor, main.cpp:694
        self.prims = np.ascontiguousarray(prims if prims is not None else np.zeros(0, PRIMITIVE_DTYPE), dtype=PRIMITIVE_DTYPE)
        self.tri_verts = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 3, 3)
        self.tri_material = np.ascontiguousarray(tri_material, dtype=np.uint32)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self.camera = {k: (np.asarray(v, dtype=np.float32) if k in ("position", "right", "up", "forward")
                           else np.float32(v)) for k, v in camera.items()}

    def as_dict(self):
        return dict(tri_verts=self.tri_verts, tri_material=self.tri_material, materials=self.materials,
                    camera=self.camera, background=self.background, prims=self.prims)

    def _desc(self):
        d = _lib.SceneDesc()
        d.tri_verts = self.tri_verts.ctypes.data
        d.tri_material = self.tri_material.ctypes.data
        d.ntri = self.tri_material.shape[0]
        d.materials = self.materials.ctypes.data
        d.nmat = self.materials.shape[0]
        for k in ("position", "right", "up", "forward"):
            getattr(d.camera, k)[:] = [float(x) for x in self.camera[k]]
        d.camera.tan_x = float(self.camera["tan_x"])
        d.camera.tan_y = float(self.camera["tan_y"])
        d.background[:] = [float(x) for x in self.background]
        d.prims = self.prims.ctypes.data if self.prims.size else None
        d.nprim = self.prims.shape[0]
        return d


GLTF_STRIDE, GLTF_INDEX_OFFSET, GLTF_DATA_URI, GLTF_HIERARCHY, GLTF_GLB, GLTF_IOR, GLTF_BACKGROUND = 1, 2, 4, 8, 16, 32, 64
GLTF_SPEC, GLTF_ALL = 63, 127


def load_gltf(path, width, height, flags=0):
    """The host part of parseGLTF (main.cpp:866-1104): no GPU involved.  `flags` (GLTF_*) opt into
    glTF 2.0 rules the reference does not follow; 0 = the reference's reading of the file."""
    L = _lib.lib()
    h = C.c_void_p()
    check(L.rt_gltf_load_ex(str(path).encode(), int(width), int(height), int(flags), C.byref(h)))
    try:
        d = _lib.SceneDesc()
        check(L.rt_host_scene_get_desc(h, C.byref(d)))
        n, nm = d.ntri, d.nmat
        tv = np.ctypeslib.as_array(C.cast(d.tri_verts, C.POINTER(C.c_float)), shape=(n * 9,)).copy() if n else \
            np.zeros(0, np.float32)
        tm = np.ctypeslib.as_array(C.cast(d.tri_material, C.POINTER(C.c_uint32)), shape=(n,)).copy() if n else \
            np.zeros(0, np.uint32)
        mats = np.zeros(nm, dtype=MATERIAL_DTYPE)
        if nm:
            C.memmove(mats.ctypes.data, d.materials, nm * MATERIAL_DTYPE.itemsize)
        cam = dict(position=np.array(d.camera.position[:], np.float32), right=np.array(d.camera.right[:], np.float32),
                   up=np.array(d.camera.up[:], np.float32), forward=np.array(d.camera.forward[:], np.float32),
                   tan_x=np.float32(d.camera.tan_x), tan_y=np.float32(d.camera.tan_y))
        bg = np.array(d.background[:], np.float32)
    finally:
        L.rt_host_scene_free(h)
    return HostScene(tv, tm, mats, cam, bg)


def node_boxes(nodes, grid):
    """Decodes the 16-bit node boxes: (n, 2, 6) float64, per child lo xyz then hi xyz
    (plane = origin + q * cell; see rt_flat_get in include/rt_amd.h)."""
    q = nodes[:, :6].reshape(-1, 2, 3)
    origin, cell = grid[:3].astype(np.float64), grid[3:].astype(np.float64)
    lo = origin + (q & 0xFFFF).astype(np.float64) * cell
    hi = origin + (q >> 16).astype(np.float64) * cell
    return np.concatenate([lo, hi], axis=2)


def build_flat(host_scene):
    """Host-only acceleration-structure build (no GPU): returns dict(nodes (n,8) u32, grid (6,) f32,
    tris (m,16) f32, lights (k,) u32 record indices (leaf-order positions of the light triangles, then m + j for
    emissive analytic primitive j), height, wide (w,12) u32 four-child nodes,
    light_tree=dict(nodes, grid, recs (k,16), height))."""
    if isinstance(host_scene, dict):
        host_scene = HostScene(**host_scene)
    L = _lib.lib()
    h = C.c_void_p()
    d = host_scene._desc()
    check(L.rt_flat_build(C.byref(d), C.byref(h)))
    try:
        info = np.zeros(4, dtype=np.uint32)
        check(L.rt_flat_get(h, _p(info), None, None, None, None))
        grid = np.empty(6, dtype=np.float32)
        nodes = np.empty((int(info[0]), 8), dtype=np.uint32)
        tris = np.empty((int(info[1]), 16), dtype=np.float32)
        lights = np.empty(int(info[2]), dtype=np.uint32)
        check(L.rt_flat_get(h, None, _p(grid), _p(nodes), _p(tris), _p(lights) if lights.size else None))
        linfo = np.zeros(2, dtype=np.uint32)
        check(L.rt_flat_get_light_tree(h, _p(linfo), None, None, None))
        lgrid = np.empty(6, dtype=np.float32)
        lnodes = np.empty((int(linfo[0]), 8), dtype=np.uint32)
        ntl = int((lights < tris.shape[0]).sum())      # the light triangles lead the list; emissive analytic primitives follow
        lrecs = np.empty((ntl, 16), dtype=np.float32)
        check(L.rt_flat_get_light_tree(h, None, _p(lgrid), _p(lnodes), _p(lrecs) if ntl else None))
        capped, sah_n, sah_t = C.c_uint32(), C.c_double(), C.c_double()
        check(L.rt_flat_get_quality(h, C.byref(capped), C.byref(sah_n), C.byref(sah_t)))
        rinfo = np.zeros(4, dtype=np.uint32)
        check(L.rt_flat_get_ref_tree(h, _p(rinfo), None, None))
        ref_tree = None
        if rinfo[0]:
            rnodes = np.empty((int(rinfo[1]), 16), dtype=np.uint32)
            rorder = np.empty(max(host_scene.tri_material.shape[0] + host_scene.prims.shape[0], 1), dtype=np.uint32)
            check(L.rt_flat_get_ref_tree(h, None, _p(rnodes), _p(rorder)))
            ref_tree = dict(nodes=rnodes, order=rorder[:host_scene.tri_material.shape[0] + host_scene.prims.shape[0]],
                            depth=int(rinfo[2]), nplanes=int(rinfo[3]))
        nwide = C.c_uint32()
        check(L.rt_flat_get_wide(h, C.byref(nwide), None))
        wide = np.empty((int(nwide.value), 12), dtype=np.uint32)
        check(L.rt_flat_get_wide(h, None, _p(wide)))
    finally:
        L.rt_flat_free(h)
    return dict(nodes=nodes, wide=wide, grid=grid, tris=tris, lights=lights, height=int(info[3]),
                height_capped_splits=int(capped.value), sah_nodes=float(sah_n.value), sah_tris=float(sah_t.value),
                light_tree=dict(nodes=lnodes, grid=lgrid, recs=lrecs, height=int(linfo[1])), ref_tree=ref_tree)


class Scene:
    """Device-resident scene: BVH + triangles + materials + lights + camera on one GPU."""

    def __init__(self, host_scene, device=0):
        if isinstance(host_scene, dict):
            host_scene = HostScene(**host_scene)
        self.host = host_scene
        self.device = int(device)
        self._h = C.c_void_p()
        d = host_scene._desc()
        check(_lib.lib().rt_scene_create(C.byref(d), self.device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- introspection -----------------------------------------------------------------------
    def stats(self):
        s = _lib.SceneStats()
        check(_lib.lib().rt_scene_get_stats(self._h, C.byref(s)))
        return dict(ntri=s.ntri, nlights=s.nlights, nnodes=s.nnodes, tree_height=s.tree_height,
                    build_seconds=s.build_seconds, device_bytes=int(s.device_bytes),
                    height_capped_splits=s.height_capped_splits, nprims=s.nprims, ntri_records=s.ntri_records,
                    hit_order=s.hit_order, ref_tree_nodes=s.ref_tree_nodes, ref_tree_depth=s.ref_tree_depth)

    def set_hit_order(self, order):
        """HITS_REFERENCE_ORDER: results that depend on the reference's traversal order (equal distances above
        all) are resolved as the reference resolves them; HITS_LEAF_ORDER: the kernels' own rule.  Raises
        RtError (RT_ERR_LIMIT) when the scene is beyond what the reference itself can hold."""
        check(_lib.lib().rt_scene_set_hit_order(self._h, int(order)))

    def flat(self):
        st = self.stats()
        grid = np.empty(6, dtype=np.float32)
        nodes = np.empty((st["nnodes"], 8), dtype=np.uint32)
        tris = np.empty((st["ntri_records"], 16), dtype=np.float32)
        check(_lib.lib().rt_scene_get_flat(self._h, _p(grid), _p(nodes), _p(tris)))
        return grid, nodes, tris

    def set_counting(self, enable):
        check(_lib.lib().rt_set_counting(self._h, int(bool(enable))))

    def reset_counters(self, stream=None):
        check(_lib.lib().rt_counters_reset(self._h, _stream_ptr(stream)))

    def counters(self):
        c = _lib.Counters()
        check(_lib.lib().rt_counters_read(self._h, C.byref(c)))
        return dict(segments=int(c.segments), nodes=int(c.nodes), tri_tests=int(c.tri_tests),
                    light_tests=int(c.light_tests), node_lane_slots=int(c.node_lane_slots),
                    tri_lane_slots=int(c.tri_lane_slots), retraced=int(c.retraced))

    def set_timing(self, enable):
        """Record HIP events around every kernel of later launches (clears the totals)."""
        check(_lib.lib().rt_set_timing(self._h, int(bool(enable))))

    def kernel_times(self):
        """{kind: (device ms summed over launches, launches)} since set_timing(True)."""
        t = _lib.KernelTimes()
        check(_lib.lib().rt_timing_read(self._h, C.byref(t)))
        return {k: (float(t.ms[i]), int(t.launches[i])) for i, k in enumerate(_lib.KERNEL_KINDS)}

    # -- closest hit (intersectScene, main.cpp:1113) -----------------------------------------
    def intersect(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        if o.shape != d.shape:
            raise ValueError("origins and directions must have the same shape")
        n = o.shape[0]
        ids = np.empty(n, dtype=np.uint32)
        t = np.empty(n, dtype=np.float32)
        check(_lib.lib().rt_intersect_rays(self._h, n, _p(o), _p(d), _p(ids), _p(t)))
        return ids, t

    def trace_primary(self, width, height):
        n = int(width) * int(height)
        ids = np.empty(n, dtype=np.uint32)
        t = np.empty(n, dtype=np.float32)
        check(_lib.lib().rt_trace_primary(self._h, int(width), int(height), _p(ids), _p(t)))
        return ids, t

    def trace_primary_dev(self, width, height, prim_id_ptr, t_ptr, stream=None):
        check(_lib.lib().rt_trace_primary_dev(self._h, int(width), int(height), C.c_void_p(prim_id_ptr),
                                             C.c_void_p(t_ptr), _stream_ptr(stream)))

    # -- render (generateScene, main.cpp:1235) -----------------------------------------------
    def render(self, width, height, spp, max_depth=6, rng_mode=RNG_REFERENCE, seed=0, spp_begin=0,
               pipeline=PIPE_MEGAKERNEL):
        """Returns the per-pixel radiance SUM over the sample range, shape (H*W, 3) float32."""
        out = np.empty((int(width) * int(height), 3), dtype=np.float32)
        check(_lib.lib().rt_render(self._h, int(width), int(height), int(spp_begin), int(spp), int(max_depth),
                                   int(rng_mode), int(seed), int(pipeline), _p(out)))
        return out

    def render_dev(self, width, height, spp, radiance_sum_ptr, max_depth=6, rng_mode=RNG_COUNTER, seed=0,
                   spp_begin=0, pipeline=PIPE_MEGAKERNEL, stream=None):
        """Enqueue a render into a device buffer of W*H*3 floats (e.g. tensor.data_ptr())."""
        check(_lib.lib().rt_render_dev(self._h, int(width), int(height), int(spp_begin), int(spp), int(max_depth),
                                       int(rng_mode), int(seed), int(pipeline), C.c_void_p(radiance_sum_ptr),
                                       _stream_ptr(stream)))

    def render_noalloc_dev(self, width, height, spp, radiance_sum_ptr, max_depth=6, seed=0, spp_begin=0, stream=None):
        """render_dev with the counter RNG and the wavefront pipeline out of the library's static scratch:
        no device allocation in the call; width * height <= 2^21."""
        check(_lib.lib().rt_render_noalloc_dev(self._h, int(width), int(height), int(spp_begin), int(spp),
                                               int(max_depth), int(seed), C.c_void_p(radiance_sum_ptr),
                                               _stream_ptr(stream)))

    def reserve(self, width, height, spp, max_depth=6, rng_mode=RNG_COUNTER, pipeline=PIPE_WAVEFRONT):
        """Allocate the scratch a render of this shape needs, so that render_dev only enqueues."""
        check(_lib.lib().rt_scene_reserve(self._h, int(width), int(height), int(spp), int(max_depth), int(rng_mode),
                                          int(pipeline)))

    # -- image (main.cpp:1226-1231, 840-846) -------------------------------------------------
    def tonemap(self, radiance_sum, spp):
        rs = np.ascontiguousarray(radiance_sum, dtype=np.float32).reshape(-1, 3)
        f = np.empty_like(rs)
        b = np.empty(rs.shape, dtype=np.uint8)
        check(_lib.lib().rt_tonemap(self._h, _p(rs), rs.shape[0], int(spp), _p(f), _p(b)))
        return f, b

    def tonemap_dev(self, radiance_sum_ptr, npixels, spp, out_float_ptr=None, out_rgb_ptr=None, stream=None):
        check(_lib.lib().rt_tonemap_dev(self._h, C.c_void_p(radiance_sum_ptr), int(npixels), int(spp),
                                       C.c_void_p(out_float_ptr) if out_float_ptr else None,
                                       C.c_void_p(out_rgb_ptr) if out_rgb_ptr else None, _stream_ptr(stream)))


class MultiScene:
    """The scene on several GPUs of this node, driven from this one process: one host BVH build, one
    upload per GPU, an RCCL communicator; render() splits the samples over the GPUs and sum-reduces."""

    def __init__(self, host_scene, devices):
        if isinstance(host_scene, dict):
            host_scene = HostScene(**host_scene)
        self.host = host_scene
        devs = np.ascontiguousarray(list(devices), dtype=np.int32)
        self._h = C.c_void_p()
        d = host_scene._desc()
        check(_lib.lib().rt_multi_create(C.byref(d), _p(devs), int(devs.size), C.byref(self._h)))
        self.ndev = int(_lib.lib().rt_multi_device_count(self._h))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().rt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def scene(self, i=0):
        """GPU i's Scene (borrowed: owned by this MultiScene)."""
        s = Scene.__new__(Scene)
        s.host, s.device, s._h = self.host, None, None
        s._h = C.c_void_p(_lib.lib().rt_multi_scene(self._h, int(i)))
        s.close = lambda: None
        return s

    def render(self, width, height, spp, max_depth=6, seed=0, spp_begin=0, pipeline=PIPE_WAVEFRONT):
        out = np.empty((int(width) * int(height), 3), dtype=np.float32)
        check(_lib.lib().rt_multi_render(self._h, int(width), int(height), int(spp_begin), int(spp), int(max_depth),
                                         int(seed), int(pipeline), _p(out)))
        return out


def multi_sample_range(spp, ndev, g):
    b, c = C.c_uint32(), C.c_uint32()
    _lib.lib().rt_multi_sample_range(int(spp), int(ndev), int(g), C.byref(b), C.byref(c))
    return int(b.value), int(c.value)


class InputData:
    """What parseGLTF returns in the reference (main.cpp:689-706): image size, depth, sample
    count and the scene, here resident on a GPU."""

    def __init__(self, scene, width, height, samples=0, ray_depth=6):
        self.scene = scene
        self.width = int(width)
        self.height = int(height)
        self.ray_depth = int(ray_depth)  # const 6 in the reference (main.cpp:692)
        self.samples = int(samples)


class SceneFloat:
    """main.cpp:44-53: the tone-mapped float image, row-major from the top-left."""

    def __init__(self, width, height, data, rgb8):
        self.width = width
        self.height = height
        self.data = data    # (H*W, 3) float32 in [0, 1]
        self.rgb8 = rgb8    # (H*W, 3) uint8, truncating conversion


def parse_gltf(path, width, height, device=0):
    host = load_gltf(path, width, height)
    return InputData(Scene(host, device=device), width, height)


def generate_scene(input_data, rng_mode=RNG_REFERENCE, pipeline=PIPE_MEGAKERNEL, seed=0):
    s = input_data.scene
    rs = s.render(input_data.width, input_data.height, input_data.samples, max_depth=input_data.ray_depth,
                  rng_mode=rng_mode, seed=seed, pipeline=pipeline)
    f, b = s.tonemap(rs, input_data.samples)
    return SceneFloat(input_data.width, input_data.height, f, b)


def print_image(scene_float, path):
    rgb = np.ascontiguousarray(scene_float.rgb8, dtype=np.uint8)
    check(_lib.lib().rt_write_ppm(str(path).encode(), int(scene_float.width), int(scene_float.height), _p(rgb)))


def selftest_rng(seeds, n, nlights=2, what=0, device=0):
    """Runs the device's RNG restatements: returns (uniforms, normals, uints), each (len(seeds), n)."""
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    shape = (seeds.shape[0], int(n))
    uni, nrm, ui = np.empty(shape, np.float32), np.empty(shape, np.float32), np.empty(shape, np.uint32)
    check(_lib.lib().rt_selftest_rng(int(device), _p(seeds), seeds.shape[0], int(n), int(nlights), int(what), _p(uni),
                                     _p(nrm), _p(ui)))
    return uni, nrm, ui


def counter_seed(pixel, sample, seed):
    return int(_lib.lib().rt_counter_seed(int(pixel), int(sample), int(seed)))


def device_count():
    return int(_lib.lib().rt_device_count())
