"""rt_render_noalloc_dev: the wavefront render out of the library's static scratch array, for callers that
cannot allocate device memory.  Its sums must equal rt_render_dev's (counter RNG, wavefront pipeline) bit
for bit -- same kernels, and a pixel's samples are added in sample order whatever the pass boundaries --
on two calls in a row, the second with a smaller input (the array keeps the first call's contents)."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _both(gpu, W, H, spp, depth, seed, spp_begin):
    import torch
    import raytracing_amd as rt
    stream = torch.cuda.current_stream()
    a = torch.full((W * H * 3,), -1.0, dtype=torch.float32, device="cuda:0")
    b = torch.full((W * H * 3,), -2.0, dtype=torch.float32, device="cuda:0")
    gpu.render_noalloc_dev(W, H, spp, a.data_ptr(), max_depth=depth, seed=seed, spp_begin=spp_begin, stream=stream)
    gpu.render_dev(W, H, spp, b.data_ptr(), max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=seed, spp_begin=spp_begin,
                   pipeline=rt.PIPE_WAVEFRONT, stream=stream)
    torch.cuda.synchronize()
    return a.cpu().numpy().reshape(-1, 3), b.cpu().numpy().reshape(-1, 3)


def test_noalloc_render_equals_the_allocating_one(scene_dir, oracle_mod):
    import torch
    import raytracing_amd as rt
    torch.cuda.set_device(0)
    W, H = 256, 192
    path = make_scene(scene_dir, "spheres1m", n_spheres=40, subdiv=2, ground=16)
    host = rt.load_gltf(path, W, H)
    gpu = rt.Scene(host, device=0)
    # first call: 70 samples of a 256x192 frame = 2 passes of the 42 frames the array holds
    gpu.reset_counters()
    got, want = _both(gpu, W, H, 70, 8, 9, 2)
    assert gpu.counters()["segments"] % 2 == 0 and want.sum() > 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # second call right after it, smaller image and fewer samples: stale records of the first call lie around
    got, want = _both(gpu, 96, 72, 5, 8, 10, 0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and want.sum() > 0
    # and against the oracle
    d = host.as_dict()
    orc = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"])
    ref, _ = orc.render(96, 72, 5, max_depth=8, rng_mode=1, seed=10)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    assert rel.max() <= 1e-5
    # a full 1920x1080 frame fits exactly once: 3 samples = 3 passes
    got, want = _both(gpu, 1920, 1080, 3, 6, 4, 5)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and want.sum() > 0
    # zero samples: zeros
    z = torch.ones(96 * 72 * 3, dtype=torch.float32, device="cuda:0")
    gpu.render_noalloc_dev(96, 72, 0, z.data_ptr(), stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert not z.any()
    # beyond the array: a clear error, nothing launched
    big = torch.empty(2048 * 1025 * 3, dtype=torch.float32, device="cuda:0")
    with pytest.raises(rt.RtError) as ei:
        gpu.render_noalloc_dev(2048, 1025, 1, big.data_ptr())
    assert ei.value.code == -4 and "2097152" in str(ei.value)
    gpu.close()
