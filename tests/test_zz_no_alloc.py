"""rt_render_noalloc_dev: the wavefront render out of the library's static scratch array, for callers that
cannot allocate device memory.  Its sums must equal rt_render_dev's (counter RNG, wavefront pipeline) bit
for bit -- same kernels, and a pixel's samples are added in sample order whatever the pass boundaries --
on two calls in a row, the second with a smaller input (the array keeps the first call's contents)."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


class _Hip:
    """The few HIP runtime calls a C caller of the _dev entry points would make, through ctypes (the
    runtime is the one librt_amd.so is linked against; no torch in this test)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamDestroy.argtypes = [C.c_void_p]
        self.rt.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.rt.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
        self.rt.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        self.rt.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        self.rt.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        self.rt.hipGraphExecDestroy.argtypes = [C.c_void_p]
        self.rt.hipGraphDestroy.argtypes = [C.c_void_p]
        self.rt.hipGraphGetNodes.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]

    def free_bytes(self):
        free, total = self.C.c_size_t(), self.C.c_size_t()
        self._ok(self.rt.hipDeviceSynchronize())
        self._ok(self.rt.hipMemGetInfo(self.C.byref(free), self.C.byref(total)))
        return free.value

    def _ok(self, rc):
        assert rc == 0, "HIP runtime call failed with %d" % rc

    def malloc(self, nbytes, fill=0xFF):
        p = self.C.c_void_p()
        self._ok(self.rt.hipSetDevice(0))
        self._ok(self.rt.hipMalloc(self.C.byref(p), nbytes))
        self._ok(self.rt.hipMemset(p, fill, nbytes))
        return p

    def to_host(self, p, n_floats):
        out = np.empty(n_floats, dtype=np.float32)
        self._ok(self.rt.hipDeviceSynchronize())
        self._ok(self.rt.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), p, n_floats * 4, 2))  # device to host
        return out

    def free(self, p):
        self._ok(self.rt.hipFree(p))


def _both(hip, gpu, W, H, spp, depth, seed, spp_begin):
    import raytracing_amd as rt
    n = W * H * 3
    a, b = hip.malloc(n * 4), hip.malloc(n * 4)
    gpu.render_noalloc_dev(W, H, spp, a.value, max_depth=depth, seed=seed, spp_begin=spp_begin)
    gpu.render_dev(W, H, spp, b.value, max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=seed, spp_begin=spp_begin,
                   pipeline=rt.PIPE_WAVEFRONT)
    got, want = hip.to_host(a, n).reshape(-1, 3), hip.to_host(b, n).reshape(-1, 3)
    hip.free(a)
    hip.free(b)
    return got, want


def test_noalloc_render_equals_the_allocating_one(scene_dir, oracle_mod):
    import raytracing_amd as rt
    hip = _Hip()
    W, H = 256, 192
    path = make_scene(scene_dir, "spheres1m", n_spheres=40, subdiv=2, ground=16)
    host = rt.load_gltf(path, W, H)
    gpu = rt.Scene(host, device=0)
    # first call: 70 samples of a 256x192 frame = 2 passes of the 42 frames the array holds
    gpu.reset_counters()
    got, want = _both(hip, gpu, W, H, 70, 8, 9, 2)
    assert gpu.counters()["segments"] % 2 == 0 and want.sum() > 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # second call right after it, smaller image and fewer samples: stale records of the first call lie around
    got, want = _both(hip, gpu, 96, 72, 5, 8, 10, 0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and want.sum() > 0
    # and against the oracle
    d = host.as_dict()
    orc = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"])
    ref, _ = orc.render(96, 72, 5, max_depth=8, rng_mode=1, seed=10)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    assert rel.max() <= 1e-5
    # a full 1920x1080 frame fits exactly once: 3 samples = 3 passes
    got, want = _both(hip, gpu, 1920, 1080, 3, 6, 4, 5)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and want.sum() > 0
    # zero samples: zeros
    z = hip.malloc(96 * 72 * 3 * 4)
    gpu.render_noalloc_dev(96, 72, 0, z.value)
    assert not hip.to_host(z, 96 * 72 * 3).any()
    # beyond the array: a clear error, nothing launched
    with pytest.raises(rt.RtError) as ei:
        gpu.render_noalloc_dev(2048, 1025, 1, z.value)
    assert ei.value.code == -4 and "2097152" in str(ei.value)
    hip.free(z)
    gpu.close()


def test_noalloc_render_allocates_nothing_and_can_be_captured(scene_dir):
    """What the entry point is named for.  (1) The device's free memory and the scene's own accounting are the
    same before and after a call -- also a call larger than anything the scene has rendered so far, where
    rt_render_dev would grow its scratch.  (2) The call is recorded into a HIP graph on a non-default stream in
    global capture mode, where hipMalloc / hipFree / a synchronisation would fail and invalidate the capture; the
    graph holds one memset and kernels; replayed twice, and once more after another render has overwritten the
    shared scratch, it reproduces rt_render_dev's sums bit for bit."""
    import ctypes as C
    import raytracing_amd as rt
    hip = _Hip()
    W, H, spp, depth, seed = 320, 240, 9, 8, 31
    host = rt.load_gltf(make_scene(scene_dir, "spheres1m", n_spheres=40, subdiv=2, ground=16), W, H)
    gpu = rt.Scene(host, device=0)
    n = W * H * 3
    out, want_d = hip.malloc(n * 4), hip.malloc(n * 4)
    gpu.render_dev(W, H, spp, want_d.value, max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=seed, pipeline=rt.PIPE_WAVEFRONT)
    want = hip.to_host(want_d, n)
    big = hip.malloc(1920 * 1080 * 3 * 4)
    gpu.render_noalloc_dev(64, 48, 1, out.value)             # the code object (and its static array) is resident from here on
    free0, bytes0 = hip.free_bytes(), gpu.stats()["device_bytes"]
    gpu.render_noalloc_dev(W, H, spp, out.value, max_depth=depth, seed=seed)
    gpu.render_noalloc_dev(1920, 1080, 2, big.value, max_depth=depth, seed=seed)     # far beyond the scene's own scratch
    assert hip.free_bytes() == free0 and gpu.stats()["device_bytes"] == bytes0
    assert np.array_equal(hip.to_host(out, n).view(np.uint32), want.view(np.uint32))
    # (2) capture
    stream, graph, exe = C.c_void_p(), C.c_void_p(), C.c_void_p()
    hip._ok(hip.rt.hipStreamCreate(C.byref(stream)))
    hip._ok(hip.rt.hipMemset(out, 0xFF, n * 4))
    hip._ok(hip.rt.hipDeviceSynchronize())
    hip._ok(hip.rt.hipStreamBeginCapture(stream, 0))          # hipStreamCaptureModeGlobal
    gpu.render_noalloc_dev(W, H, spp, out.value, max_depth=depth, seed=seed, stream=stream.value)
    hip._ok(hip.rt.hipStreamEndCapture(stream, C.byref(graph)))
    assert graph.value
    nodes = C.c_size_t()
    hip._ok(hip.rt.hipGraphGetNodes(graph, None, C.byref(nodes)))
    assert nodes.value > 10                                  # a memset and a few kernels per bounce
    assert np.all(hip.to_host(out, n).view(np.uint32) == 0xFFFFFFFF)     # recording ran nothing
    hip._ok(hip.rt.hipGraphInstantiate(C.byref(exe), graph, None, None, 0))
    for k in range(3):
        if k == 2:                                           # someone else used the shared scratch in between
            gpu.render_noalloc_dev(1920, 1080, 1, big.value, max_depth=4, seed=1)
            hip._ok(hip.rt.hipDeviceSynchronize())
        hip._ok(hip.rt.hipMemset(out, 0xFF, n * 4))
        hip._ok(hip.rt.hipGraphLaunch(exe, stream))
        hip._ok(hip.rt.hipStreamSynchronize(stream))
        assert np.array_equal(hip.to_host(out, n).view(np.uint32), want.view(np.uint32)), "replay %d" % k
    hip._ok(hip.rt.hipGraphExecDestroy(exe))
    hip._ok(hip.rt.hipGraphDestroy(graph))
    hip._ok(hip.rt.hipStreamDestroy(stream))
    for p in (out, want_d, big):
        hip.free(p)
    gpu.close()
