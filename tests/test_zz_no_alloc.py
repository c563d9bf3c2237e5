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
