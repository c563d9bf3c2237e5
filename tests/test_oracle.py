"""CPU tests of the oracle (the checker itself) against the committed fixtures and against its
own brute-force / algebraic properties.  No GPU."""
import json
import os

import numpy as np
import pytest

from conftest import make_scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _oracle_scene(oracle_mod, path, W, H):
    from oracle import gltf_ref
    sc = gltf_ref.parse_gltf(path, W, H)
    return sc, oracle_mod.OracleScene(sc["tri_verts"], sc["tri_material"], sc["materials"], sc["camera"])


def test_rng_known_answers(oracle_mod):
    """std::minstd_rand + libstdc++ distributions as instantiated by main.cpp:98-111."""
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    for seed, e in kat.items():
        if seed == "counter_seed":
            for p, s, sd, want in e:
                assert oracle_mod.counter_seed(p, s, sd) == want
            continue
        k = oracle_mod.rng_kat(int(seed))
        assert [int(x) for x in k["raw"]] == e["raw"]
        assert [int(x) for x in k["uni"].view(np.uint32)] == e["uni_bits"]
        assert [int(x) for x in k["nrm"].view(np.uint32)] == e["nrm_bits"]
        assert k["nrm_draws"] == e["nrm_draws"]
        assert [int(x) for x in k["ui2"]] == e["ui2"] and [int(x) for x in k["ui12"]] == e["ui12"]
    # seeds 0 and 1 alias (seed 0 -> state 1), as do s and s mod (2^31 - 1)
    assert kat["0"] == kat["1"] and kat["2147483647"]["raw"] == kat["0"]["raw"]


def test_uniform_formula(oracle_mod):
    """uniform = float(x - 1) / 2^31 with the engine x <- 48271 x mod (2^31 - 1), clamped below 1."""
    uni, _, _ = oracle_mod.rng_stream(12345, 1, 2)
    x = (12345 * 48271) % 2147483647
    assert uni[0] == np.float32(np.float32(x - 1) / np.float32(2147483648.0))


def test_golden_cornell(oracle_mod, scene_dir):
    g = np.load(os.path.join(GOLDEN, "cornell_64x48.npz"))
    sc, orc = _oracle_scene(oracle_mod, make_scene(scene_dir, "cornell"), 64, 48)
    assert np.array_equal(sc["tri_verts"], g["tri_verts"])
    ids, t, tie, _ = orc.trace_primary(64, 48)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(t.view(np.uint32), g["t"].view(np.uint32))
    ref, c0 = orc.render(64, 48, 4, max_depth=6, rng_mode=0)
    ctr, c1 = orc.render(64, 48, 4, max_depth=8, rng_mode=1, seed=0x5EED, spp_begin=2)
    assert np.array_equal(ref.view(np.uint32), g["sum_reference_rng"].view(np.uint32))
    assert np.array_equal(ctr.view(np.uint32), g["sum_counter_rng"].view(np.uint32))
    assert [c0["segments"], c1["segments"]] == [int(x) for x in g["segments"]]


def test_golden_cornell_onfloor(oracle_mod, scene_dir):
    """Coincident faces (the boxes stand on the floor): the fixture pins the order in which the oracle's restatement
    of BVH::build / BVH::intersect resolves equal distances."""
    g = np.load(os.path.join(GOLDEN, "cornell_onfloor_64x48.npz"))
    sc, orc = _oracle_scene(oracle_mod, make_scene(scene_dir, "cornell_onfloor"), 64, 48)
    assert np.array_equal(sc["tri_verts"], g["tri_verts"])
    ids, t, tie, _ = orc.trace_primary(64, 48)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(t.view(np.uint32), g["t"].view(np.uint32))
    ref, c0 = orc.render(64, 48, 4, max_depth=8, rng_mode=0)
    ctr, c1 = orc.render(64, 48, 4, max_depth=8, rng_mode=1, seed=0x5EED, spp_begin=2)
    assert np.array_equal(ref.view(np.uint32), g["sum_reference_rng"].view(np.uint32))
    assert np.array_equal(ctr.view(np.uint32), g["sum_counter_rng"].view(np.uint32))
    assert [c0["segments"], c1["segments"]] == [int(x) for x in g["segments"]]


def test_golden_emissive(oracle_mod, scene_dir):
    g = np.load(os.path.join(GOLDEN, "emissive_48x36.npz"))
    sc, orc = _oracle_scene(oracle_mod, make_scene(scene_dir, "emissive", subdiv=1, lamps=2), 48, 36)
    assert np.array_equal(sc["tri_verts"], g["tri_verts"]) and orc.info()["lights"] == 160
    ids, t, tie, _ = orc.trace_primary(48, 36)
    assert np.array_equal(ids, g["ids"]) and np.array_equal(t.view(np.uint32), g["t"].view(np.uint32))
    ref, c0 = orc.render(48, 36, 2, max_depth=5, rng_mode=0)
    ctr, c1 = orc.render(48, 36, 2, max_depth=5, rng_mode=1, seed=11)
    assert np.array_equal(ref.view(np.uint32), g["sum_reference_rng"].view(np.uint32))
    assert np.array_equal(ctr.view(np.uint32), g["sum_counter_rng"].view(np.uint32))
    assert [c0["segments"], c1["segments"]] == [int(x) for x in g["segments"]]
    assert [c0["light_tests"], c1["light_tests"]] == [int(x) for x in g["light_tests"]]


@pytest.mark.parametrize("name,kw", [("cornell", {}), ("heightfield100k", {"nx": 24, "nz": 24}),
                                     ("glass", {"grid": 2, "subdiv": 1}),
                                     ("spheres1m", {"n_spheres": 12, "subdiv": 1, "ground": 8})])
def test_bvh_traversal_equals_brute_force(oracle_mod, scene_dir, name, kw):
    """BVH::intersect (main.cpp:560-605) returns the minimum-t triangle over ALL triangles."""
    sc, orc = _oracle_scene(oracle_mod, make_scene(scene_dir, name, **kw), 96, 64)
    assert orc.info()["unsplittable"] == 0  # the reference's broken-leaf branch (main.cpp:675-679) is never taken
    o, d = orc.primary_rays(96, 64)
    rng = np.random.RandomState(5)
    o2 = rng.uniform(-0.8, 0.8, size=(2000, 3)).astype(np.float32)
    d2 = rng.normal(size=(2000, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    ids_a, t_a, tie_a, cnt = orc.intersect(o, d, use_bvh=True)
    ids_b, t_b, tie_b, _ = orc.intersect(o, d, use_bvh=False)
    ok = (tie_a == 0) & (tie_b == 0)
    assert np.array_equal(ids_a[ok], ids_b[ok])
    assert np.array_equal(t_a.view(np.uint32), t_b.view(np.uint32))
    assert cnt["nodes"] > 0 and cnt["tri_tests"] < o.shape[0] * sc["tri_verts"].shape[0]


def test_render_independent_of_thread_count(oracle_mod, scene_dir):
    """The reference is deterministic irrespective of OMP thread count (per-pixel RNG, main.cpp:1221)."""
    _, orc = _oracle_scene(oracle_mod, make_scene(scene_dir, "cornell"), 48, 36)
    a, ca = orc.render(48, 36, 3, nthreads=1)
    b, cb = orc.render(48, 36, 3, nthreads=4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ca == cb


def test_counter_mode_ranges_add_up(oracle_mod, scene_dir):
    """Counter RNG: samples are independent of the range they are rendered in, so partial sums of
    disjoint ranges add up to the full sum (up to fp32 summation order)."""
    _, orc = _oracle_scene(oracle_mod, make_scene(scene_dir, "cornell"), 40, 30)
    full, cf = orc.render(40, 30, 6, max_depth=8, rng_mode=1, seed=3)
    a, ca = orc.render(40, 30, 2, max_depth=8, rng_mode=1, seed=3, spp_begin=0)
    b, cb = orc.render(40, 30, 4, max_depth=8, rng_mode=1, seed=3, spp_begin=2)
    assert np.allclose(a + b, full, rtol=1e-5, atol=1e-6)
    assert ca["segments"] + cb["segments"] == cf["segments"]
    with pytest.raises(ValueError):
        orc.render(40, 30, 2, rng_mode=0, spp_begin=1)  # the reference stream cannot start mid-pixel


def test_depth_semantics(oracle_mod, scene_dir):
    """applyLight (main.cpp:1188-1199): depth 0 is black, depth 1 sees emitters only."""
    sc, orc = _oracle_scene(oracle_mod, make_scene(scene_dir, "cornell"), 40, 30)
    d0, c0 = orc.render(40, 30, 2, max_depth=0)
    assert not d0.any() and c0["segments"] == 40 * 30 * 2
    d1, c1 = orc.render(40, 30, 2, max_depth=1)
    assert c1["segments"] == 40 * 30 * 2
    lit = d1.any(axis=1)
    assert 0 < lit.sum() < 0.2 * lit.size  # only the pixels that look at the lamp
    emission = sc["materials"]["emission"][3]
    assert np.allclose(np.unique(d1[lit], axis=0) / emission, np.array([[1, 1, 1], [2, 2, 2]])[:len(np.unique(d1[lit], axis=0))])


def test_tonemap_values(oracle_mod):
    """ACES + gamma + truncation (main.cpp:827-846, 1226-1231) on hand-computed inputs."""
    rs = np.array([[0, 0, 0], [4, 4, 4], [400, 2, 0.04]], dtype=np.float32)
    f, b = oracle_mod.tonemap(rs, 4)
    assert (b[0] == 0).all() and (f[0] == 0).all()
    x = np.float32(1.0)
    want = np.float32((x * (np.float32(2.51) * x + np.float32(0.03))) / (x * (np.float32(2.43) * x + np.float32(0.59)) + np.float32(0.14)))
    assert abs(f[1, 0] - want ** (1 / 2.2)) < 1e-6 and b[1, 0] == int(f[1, 0] * 255)
    assert f[2, 0] == 1.0 and b[2, 0] == 255  # saturated
