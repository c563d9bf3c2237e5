"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bars:
  * closest-hit ids and distances: bit for bit;
  * radiance sums in the reference-RNG mode (the reference's random stream AND its recursion
    order): bit for bit, and equal ray-segment / light-test counts;
  * radiance sums of the throughput kernels (counter RNG, forward accumulation -- the same random
    numbers and branches, products rounded in a different order): every pixel within 1e-5 relative
    (floor 1e-3 absolute on the denominator), equal ray-segment counts;
  * tone-mapped floats and bytes: bit for bit (glibc's powf is restated on the device)."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _load(scene_dir, oracle_mod, name, W, H, **kw):
    import raytracing_amd as rt
    path = make_scene(scene_dir, name, **kw)
    host = rt.load_gltf(path, W, H)
    gpu = rt.Scene(host, device=0)
    d = host.as_dict()
    orc = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"], d["background"])
    return host, gpu, orc


def _compare_hits(ids_g, t_g, ids_o, t_o, tie):
    """ids must match bit for bit except where the oracle saw an exact distance tie between two
    triangles (the winner then depends on traversal order); t must match bit for bit.
    tie=None (a scene traced in the reference's own order): no exception, every id must match, and a
    distance may be NaN on both sides (the reference's zero-determinant "hit", main.cpp:483)."""
    notie = (tie == 0) if tie is not None else np.ones(ids_g.shape, dtype=bool)
    bad = (ids_g != ids_o) & notie
    assert bad.sum() == 0, "hit id mismatches: %d of %d (first at %s)" % (bad.sum(), ids_g.size, np.flatnonzero(bad)[:5])
    tbad = (t_g.view(np.uint32) != t_o.view(np.uint32))
    if tie is None:
        tbad &= ~(np.isnan(t_g) & np.isnan(t_o))
    assert tbad.sum() == 0, "t mismatches: %d (max rel %g)" % (
        tbad.sum(), np.max(np.abs(t_g[tbad] - t_o[tbad]) / np.abs(t_o[tbad])))
    return int((~notie).sum())


def _assert_bits_equal_nan(out, ref):
    """Bit-equal, except that a NaN matches a NaN whatever its sign and payload (the CPU's default NaN has the
    sign bit set, the GPU's does not)."""
    bad = ((out.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(out) & np.isnan(ref))).any(axis=1)
    assert bad.sum() == 0, "radiance sums differ in %d of %d pixels" % (bad.sum(), bad.size)


def _assert_bits_equal(out, ref):
    bad = (out.view(np.uint32) != ref.view(np.uint32)).any(axis=1)
    assert bad.sum() == 0, "radiance sums differ in %d of %d pixels (max rel %g)" % (
        bad.sum(), bad.size, np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-6)))


def _assert_close(out, ref, tol=1e-5, max_nan_pixels=2):
    """Every pixel within `tol` relative.  A pixel that is NaN in the oracle is left out (at most a couple per
    render: the reference's own arithmetic, e.g. 0 * inf at a vertex whose pdf underflows, which the recursion
    L = k * L_next + e turns into NaN while the forward sum L += T * e of the throughput kernels may never add
    the poisoned term); a NaN on the GPU side where the oracle is finite is an error."""
    nan_ref = np.isnan(ref).any(axis=1)
    assert nan_ref.sum() <= max_nan_pixels, "%d NaN pixels in the oracle's render" % nan_ref.sum()
    assert not np.isnan(out[~nan_ref]).any(), "NaN pixels where the oracle is finite"
    rel = np.abs(out[~nan_ref] - ref[~nan_ref]) / np.maximum(np.abs(ref[~nan_ref]), 1e-3)
    assert rel.max() <= tol, "max relative pixel error %g (%d pixels above %g)" % (rel.max(), (rel.max(axis=1) > tol).sum(), tol)


@pytest.mark.parametrize("name,kw,W,H", [
    ("cornell", {}, 320, 240),
    ("cornell_onfloor", {}, 320, 240),
    ("heightfield100k", {"nx": 48, "nz": 48}, 320, 240),
    ("glass", {"grid": 3, "subdiv": 2}, 256, 192),
    ("spheres1m", {"n_spheres": 40, "subdiv": 2, "ground": 16}, 256, 192),
])
def test_primary_hits_bit_exact(scene_dir, oracle_mod, name, kw, W, H):
    """Both hit orders: the kernels' own tree with its own rule for equal distances (those rays are left out of
    the id comparison), and the reference's tree walked in the reference's order (no ray left out)."""
    import raytracing_amd as rt
    host, gpu, orc = _load(scene_dir, oracle_mod, name, W, H, **kw)
    ids_o, t_o, tie, _ = orc.trace_primary(W, H)
    assert gpu.stats()["hit_order"] == rt.HITS_REFERENCE_ORDER and gpu.stats()["ref_tree_nodes"] == orc.info()["nodes"]
    ids_g, t_g = gpu.trace_primary(W, H)
    _compare_hits(ids_g, t_g, ids_o, t_o, None)
    gpu.set_hit_order(rt.HITS_LEAF_ORDER)
    ids_g, t_g = gpu.trace_primary(W, H)
    nties = _compare_hits(ids_g, t_g, ids_o, t_o, tie)
    assert nties < 0.001 * W * H
    assert (ids_g != 0xFFFFFFFF).sum() > 0.2 * W * H  # the camera actually sees the scene


def test_random_rays_bit_exact(scene_dir, oracle_mod):
    """Incoherent rays from inside the Cornell box."""
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", 64, 64)
    rng = np.random.RandomState(7)
    n = 20000
    o = rng.uniform(-0.9, 0.9, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    ids_o, t_o, tie, _ = orc.intersect(o, d)
    ids_g, t_g = gpu.intersect(o, d)
    _compare_hits(ids_g, t_g, ids_o, t_o, None)
    import raytracing_amd as rt
    gpu.set_hit_order(rt.HITS_LEAF_ORDER)
    ids_g, t_g = gpu.intersect(o, d)
    _compare_hits(ids_g, t_g, ids_o, t_o, tie)


def test_reference_order_on_coincident_faces(scene_dir, oracle_mod):
    """The Cornell box with its boxes standing ON the floor: their bottom faces lie in the floor plane, every ray
    that leaves the glass box downwards (and every ray from below the mirror box) meets two triangles at exactly
    the same distance, and the reference keeps whichever its own tree meets first (main.cpp:566-567, 589-590).
    In the reference's hit order the library answers the same: ids and distances of random rays and of the camera
    rays with no ray left out, the reference-stream render bit for bit at depth 8, both counter-RNG pipelines
    within 1e-5 with equal segment counts (the wavefront one with the two-child and the four-child walk, sorted
    and unsorted).  In leaf order the same render differs from the reference's -- which is what the mode is for."""
    import raytracing_amd as rt
    W, H = 160, 120
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell_onfloor", W, H)
    rng = np.random.RandomState(23)
    n = 200000
    o = rng.uniform(-0.95, 0.95, size=(n, 3)).astype(np.float32)
    o[: n // 2, 1] = rng.uniform(-0.99, -0.3, size=n // 2)           # many start inside the boxes, near the floor
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    ids_r, t_r, tie, _ = orc.intersect(o, d)
    assert tie.sum() > 500                                           # the scene does what it is for
    ids_g, t_g = gpu.intersect(o, d)
    _compare_hits(ids_g, t_g, ids_r, t_r, None)
    ids_o, t_o, _, _ = orc.trace_primary(W, H)
    ids_g, t_g = gpu.trace_primary(W, H)
    _compare_hits(ids_g, t_g, ids_o, t_o, None)
    ref0, c0 = orc.render(W, H, 6, max_depth=8, rng_mode=0)
    gpu.reset_counters()
    out0 = gpu.render(W, H, 6, max_depth=8, rng_mode=0)
    _assert_bits_equal_nan(out0, ref0)
    assert gpu.counters()["segments"] == c0["segments"] and gpu.counters()["light_tests"] == c0["light_tests"]
    ref, c_o = orc.render(W, H, 5, max_depth=8, rng_mode=1, seed=17, spp_begin=1)
    for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
        gpu.reset_counters()
        _assert_close(gpu.render(W, H, 5, max_depth=8, rng_mode=1, seed=17, spp_begin=1, pipeline=pipe), ref)
        assert gpu.counters()["segments"] == c_o["segments"]
    gpu.set_hit_order(rt.HITS_LEAF_ORDER)
    leaf = gpu.render(W, H, 6, max_depth=8, rng_mode=0)
    differ = (leaf.view(np.uint32) != ref0.view(np.uint32)).any(axis=1).mean()
    ids_l, _ = gpu.intersect(o, d)
    # (how far the kernels' own rule is from the reference's on this scene: reported, not a bar)
    print("leaf order: %.2f %% of the pixels and %d of the %d tied rays differ from the reference's answer"
          % (100 * differ, (ids_l != ids_r).sum(), tie.sum()))
    gpu.close()


@pytest.mark.parametrize("wide,sort", [("0", "0"), ("1", "1")])
def test_reference_order_in_the_wavefront_variants(scene_dir, oracle_mod, monkeypatch, wide, sort):
    """The rays that met two primitives at the best distance travel through the retry list of wf_extend: with the
    two-child and the four-child walk, with sorted queues (hits stored by slot) and several passes."""
    import raytracing_amd as rt
    W, H = 128, 96
    monkeypatch.setenv("RT_WF_WIDE", wide)
    monkeypatch.setenv("RT_WF_SORT", sort)
    monkeypatch.setenv("RT_WF_PATHS", str(2 * W * H))
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell_onfloor", W, H)
    ref, c_o = orc.render(W, H, 5, max_depth=8, rng_mode=1, seed=3)
    gpu.reset_counters()
    _assert_close(gpu.render(W, H, 5, max_depth=8, rng_mode=1, seed=3, pipeline=rt.PIPE_WAVEFRONT), ref)
    assert gpu.counters()["segments"] == c_o["segments"]
    gpu.close()


def test_far_origin_rays_bit_exact(scene_dir, oracle_mod):
    """Ray origins 10 to 5000 scene extents away (the box test's rounding grows with the distance between
    the origin and the node grid; it widens its slabs accordingly) and rays aimed along box faces."""
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", 64, 64)
    rng = np.random.RandomState(19)
    n = 20000
    target = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
    target[: n // 4, rng.randint(0, 3)] = np.float32(1.0)          # points on a wall plane
    away = rng.normal(size=(n, 3))
    away /= np.linalg.norm(away, axis=1, keepdims=True)
    dist = (10.0 ** rng.uniform(1.3, 3.9, size=(n, 1)))            # 20 .. 8000 units; the box is 2 wide
    o = (target + away * dist).astype(np.float32)
    d = (target - o).astype(np.float64)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    import raytracing_amd as rt
    ids_o, t_o, tie, _ = orc.intersect(o, d, use_bvh=False)
    gpu.set_hit_order(rt.HITS_LEAF_ORDER)                          # the kernels' own tree and box test
    ids_g, t_g = gpu.intersect(o, d)
    _compare_hits(ids_g, t_g, ids_o, t_o, tie)
    assert (ids_g != 0xFFFFFFFF).sum() > 0.5 * n
    gpu.set_hit_order(rt.HITS_REFERENCE_ORDER)                     # the reference's walk: its boxes, its divisions
    ids_r, t_r, _, _ = orc.intersect(o, d, use_bvh=True)
    ids_g, t_g = gpu.intersect(o, d)
    _compare_hits(ids_g, t_g, ids_r, t_r, None)


def test_order_sensitive_random_scenes(oracle_mod):
    """Eight of the scenes of tests/fuzz_reference_order.py (lattices of coincident and abutting triangles,
    duplicated triangles, flat sheets of overlapping triangles whose distances differ by an ulp, random analytic
    primitives with planes): closest-hit ids and distances and the reference-stream render bit for bit in the
    reference's order, no mask; both throughput pipelines within 1e-5 with equal segment counts -- or, where a
    path meets a NaN "hit" that only the reference's own walk sees, closed by RT_HITS_REFERENCE_WALK (at most
    one scene of these)."""
    import fuzz_reference_order as fz
    soft = built = 0
    for seed in (5002, 5007, 5010, 5022, 5026, 5030, 5034, 5053):
        r = fz.check(seed)
        if r is not None:
            built += 1
            soft += r
    assert built >= 6 and soft <= 2, (built, soft)


def test_pruned_walk_and_handed_over_rays_equal_the_full_walk(scene_dir, oracle_mod, monkeypatch):
    """The throughput kernels answer in the reference's order by their fast walk plus, for rays on which two hits
    lay close together or which start far outside the scene, a pruned form of the reference's walk.  Through a test
    switch rt_intersect_rays answers that way too (and with the pruned walk on every ray): against the full walk on
    the Cornell box with analytic primitives (a Box's tree box is half its size; planes send rays thousands of
    units out, from where the reference's ellipsoid test reports hits a unit beside the ellipsoid) -- random rays,
    rays leaving surfaces, rays coming in from 3 to 8000 units away.  Only rays the full walk answers with NaN
    may differ."""
    import raytracing_amd as rt
    base = rt.load_gltf(make_scene(scene_dir, "cornell"), 64, 48)
    types = base.materials["type"]
    emits = np.linalg.norm(base.materials["emission"], axis=1) > 1e-4
    pick = [int(np.flatnonzero((types == t) & ~emits)[0]) for t in (0, 2, 1)]
    rng = np.random.RandomState(12)
    prims = _analytic_prims(rt, rng, pick)
    host = rt.HostScene(base.tri_verts, base.tri_material, base.materials, base.camera, prims=prims)
    full = rt.Scene(host, device=0)
    m = 1500000
    o = rng.uniform(-1.0, 1.0, size=(m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ids0, t0 = full.intersect(o, d)
    ok = (ids0 != 0xFFFFFFFF) & (t0 < 100)
    o2 = (o[ok] + (t0[ok][:, None] - 1e-4) * d[ok]).astype(np.float32)          # leaving the surfaces those rays hit
    d2 = rng.normal(size=o2.shape)
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    dirs = rng.normal(size=(m, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o3 = (rng.uniform(-1.0, 1.0, size=(m, 3)) - dirs * 10.0 ** rng.uniform(0.5, 3.9, size=(m, 1))).astype(np.float32)
    d3 = dirs.astype(np.float32)
    sets = [(o, d), (o2, d2), (o3, d3)]
    want = [full.intersect(a, b) for a, b in sets]
    for how in ("pruned", "ties"):
        monkeypatch.setenv("RT_TEST_INTERSECT_WALK", how)
        with rt.Scene(host, device=0) as other:
            for (a, b), (ids_f, t_f) in zip(sets, want):
                ids_p, t_p = other.intersect(a, b)
                bad = ((ids_p != ids_f) | (t_p.view(np.uint32) != t_f.view(np.uint32))) & ~np.isnan(t_f)
                assert bad.sum() == 0, "%s: %d of %d rays differ from the full walk, first %s" % (how, bad.sum(), len(a), np.flatnonzero(bad)[:5])
    assert sum(np.isnan(t).sum() for _, t in want) < 50
    full.close()


@pytest.mark.parametrize("seed,extent,offset", [(1, 0.05, 0.0), (2, 2.0, 50.0), (3, 150.0, 0.0), (4, 1.0, -99.0)])
def test_random_triangle_soups_bit_exact(oracle_mod, seed, extent, offset):
    """Triangle soups of mixed sizes (edges from 1e-4 to the whole extent, some degenerate, some
    duplicated, some axis-aligned) at different scales and positions inside [-100, 100]^3, against the
    oracle's brute-force loop over all triangles: stresses the 16-bit node grid and its guard cells."""
    import raytracing_amd as rt
    rng = np.random.RandomState(seed)
    n = 3000
    centre = rng.uniform(-0.5, 0.5, size=(n, 1, 3)) * extent + offset
    size = (10.0 ** rng.uniform(-4, 0, size=(n, 1, 1))) * extent
    tri = centre + rng.uniform(-0.5, 0.5, size=(n, 3, 3)) * size
    tri[:100, :, rng.randint(0, 3)] = tri[:100, :1, 0]            # axis-aligned (flat boxes)
    tri[100:130, 2] = tri[100:130, 1]                             # zero area
    tri[130:160] = tri[160:190]                                   # duplicates: exact distance ties
    tri = np.clip(tri, -100.0, 100.0).astype(np.float32)
    mats = np.zeros(1, dtype=rt.api.MATERIAL_DTYPE)
    mats["color"] = 0.5
    cam = dict(position=[0, 0, 0], right=[1, 0, 0], up=[0, 1, 0], forward=[0, 0, -1], tan_x=1, tan_y=1)
    host = rt.HostScene(tri, np.zeros(n, np.uint32), mats, cam)
    gpu = rt.Scene(host, device=0)
    orc = oracle_mod.OracleScene(tri, np.zeros(n, np.uint32), mats, cam)
    m = 30000
    o = (rng.uniform(-0.8, 0.8, size=(m, 3)) * extent + offset).astype(np.float32)
    target = tri[rng.randint(0, n, m)]
    w = rng.dirichlet([1, 1, 1], size=m).astype(np.float32)
    w[: m // 6] = np.eye(3, dtype=np.float32)[rng.randint(0, 3, m // 6)]      # aimed exactly at a vertex
    w[m // 6: m // 3, 0] = 0                                                   # aimed at an edge
    w /= w.sum(axis=1, keepdims=True)
    p = (target * w[:, :, None]).sum(axis=1)
    d = p - o
    d[m // 2:] = rng.normal(size=(m - m // 2, 3))                              # unaimed
    d = (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)).astype(np.float32)
    # the reference's order (its own tree, walked its own way): every ray, duplicates' ties and the NaN distances of
    # zero-determinant "hits" included
    assert gpu.stats()["hit_order"] == rt.HITS_REFERENCE_ORDER and orc.info()["unsplittable"] == 0
    ids_r, t_r, _, _ = orc.intersect(o, d, use_bvh=True)
    ids_g, t_g = gpu.intersect(o, d)
    _compare_hits(ids_g, t_g, ids_r, t_r, None)
    # the kernels' own tree against the brute-force minimum
    gpu.set_hit_order(rt.HITS_LEAF_ORDER)
    ids_o, t_o, tie, _ = orc.intersect(o, d, use_bvh=False)
    ids_g, t_g = gpu.intersect(o, d)
    ok = ~np.isnan(t_o)                                            # zero-determinant "hits" of the reference: see below
    _compare_hits(ids_g[ok], t_g[ok], ids_o[ok], t_o[ok], tie[ok])
    assert (ids_g != 0xFFFFFFFF).sum() > 0.3 * m
    print("leaf order: %d of %d rays answered differently from the reference's walk (ids), %d NaN distances there"
          % ((ids_g != ids_r).sum(), m, np.isnan(t_r).sum()))
    gpu.close()
    orc.close()


def test_axis_parallel_rays_are_finite(scene_dir, oracle_mod):
    """Rays exactly parallel to a triangle's plane make the reference's 3x3 inverse divide by a
    zero determinant; it then accepts a NaN-distance "hit" whenever that triangle is the first
    one met, which depends on its BVH's traversal order (main.cpp:483, 566-571) and cannot be a
    parity target for a kernel with a tree of its own.  In leaf order the GPU treats a NaN distance as a
    miss of that triangle; there it must still return clean results, and agree with the oracle wherever
    the oracle's own answer for the ray is the brute-force one and not NaN.  In the reference's order
    (the reference's tree, walked its way) the GPU reproduces the NaN "hits" themselves."""
    import raytracing_amd as rt
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", 64, 64)
    rng = np.random.RandomState(11)
    n = 600
    o = rng.uniform(-0.9, 0.9, size=(n, 3)).astype(np.float32)
    d = np.zeros((n, 3), dtype=np.float32)
    d[np.arange(n), rng.randint(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    ids_r, t_r, _, _ = orc.intersect(o, d)
    ids_g, t_g = gpu.intersect(o, d)
    _compare_hits(ids_g, t_g, ids_r, t_r, None)
    assert np.isnan(t_r).sum() > 0                                  # the reference does answer NaN on some of these
    gpu.set_hit_order(rt.HITS_LEAF_ORDER)
    ids_g, t_g = gpu.intersect(o, d)
    assert not np.isnan(t_g).any()
    hit = ids_g != 0xFFFFFFFF
    assert (ids_g[hit] < 36).all() and (t_g[hit] < 1e4).all() and (t_g[~hit] == np.float32(1e4)).all()
    ids_o, t_o, tie, _ = orc.intersect(o, d)
    ids_b, t_b, _, _ = orc.intersect(o, d, use_bvh=False)
    clean = ~np.isnan(t_o) & ~np.isnan(t_b) & (t_o == t_b) & (tie == 0)
    assert clean.sum() > 0
    assert (ids_g[clean] == ids_o[clean]).all() and (t_g[clean] == t_o[clean]).all()


@pytest.mark.parametrize("name,kw,W,H,spp,depth", [
    ("cornell", {}, 160, 120, 8, 6),
    ("glass", {"grid": 3, "subdiv": 2}, 128, 96, 4, 8),
    ("spheres1m", {"n_spheres": 40, "subdiv": 2, "ground": 16}, 128, 96, 4, 6),
])
def test_render_reference_stream_bit_exact(scene_dir, oracle_mod, name, kw, W, H, spp, depth):
    """Reference RNG stream (one minstd_rand per pixel carried across samples) evaluated in the
    reference's recursion order: the GPU's radiance sums equal the oracle's bit for bit, and so do
    the number of rays traced and of light-pdf triangle tests."""
    host, gpu, orc = _load(scene_dir, oracle_mod, name, W, H, **kw)
    ref, cnt_o = orc.render(W, H, spp, max_depth=depth, rng_mode=0)
    gpu.reset_counters()
    out = gpu.render(W, H, spp, max_depth=depth, rng_mode=0)
    cnt_g = gpu.counters()
    _assert_bits_equal(out, ref)
    assert cnt_g["segments"] == cnt_o["segments"] and cnt_g["light_tests"] == cnt_o["light_tests"]
    assert ref.sum() > 0


@pytest.mark.parametrize("pipeline", [0, 1])
def test_render_counter_mode_matches_oracle(scene_dir, oracle_mod, pipeline):
    """Counter RNG, both throughput pipelines, a sample sub-range (as one GPU of several renders it)."""
    W, H = 128, 96
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", W, H)
    ref, cnt_o = orc.render(W, H, 6, max_depth=8, rng_mode=1, seed=99, spp_begin=3)
    gpu.reset_counters()
    out = gpu.render(W, H, 6, max_depth=8, rng_mode=1, seed=99, spp_begin=3, pipeline=pipeline)
    cnt_g = gpu.counters()
    _assert_close(out, ref)
    assert cnt_g["segments"] == cnt_o["segments"] and cnt_g["light_tests"] == cnt_o["light_tests"]


def test_tonemap_matches_oracle(scene_dir, oracle_mod):
    """Mean, ACES, gamma, truncation (main.cpp:1226-1231, 827-846): floats and bytes bit for bit.  The
    gamma curve is glibc's powf restated on the device; the oracle calls the host's libm."""
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", 64, 48)
    rng = np.random.RandomState(3)
    n = 400000
    rs = (rng.gamma(0.7, 3.0, size=(n, 3)) * 16).astype(np.float32)
    rs[:50] = 0
    rs[50:20000] *= (10.0 ** rng.uniform(-12, 0, size=(19950, 1))).astype(np.float32)   # dark pixels
    rs[20000:20100] = np.float32(1e15)                                                  # saturated
    f_o, b_o = oracle_mod.tonemap(rs, 16)
    f_g, b_g = gpu.tonemap(rs, 16)
    assert np.array_equal(f_g.view(np.uint32), f_o.view(np.uint32)), \
        "tonemapped floats differ in %d of %d" % ((f_g != f_o).sum(), f_g.size)
    assert np.array_equal(b_g, b_o)


def test_rng_restatements_known_answers(oracle_mod):
    """The device's minstd_rand / uniform_real / normal / uniform_int restatements (main.cpp:98-111)
    against the committed known answers (g++ / libstdc++ 11 outputs) and against the oracle, which links
    the very libstdc++ classes the reference instantiates: 4000 engines x 64 interleaved triples."""
    import json
    import os
    import raytracing_amd as rt
    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rng_kat.json")))
    seeds = np.array(sorted(int(k) for k in kat if k.isdigit()), dtype=np.uint32)
    _, _, raw = rt.selftest_rng(seeds, 4, what=4)
    uni, _, _ = rt.selftest_rng(seeds, 3, what=1)
    _, nrm, _ = rt.selftest_rng(seeds, 3, what=2)
    _, _, ui2 = rt.selftest_rng(seeds, 4, nlights=2, what=3)
    _, _, ui12 = rt.selftest_rng(seeds, 4, nlights=12, what=3)
    for k, seed in enumerate(seeds):
        want = kat[str(int(seed))]
        assert raw[k].tolist() == want["raw"]
        assert uni[k].view(np.uint32).tolist() == want["uni_bits"]
        assert nrm[k].view(np.uint32).tolist() == want["nrm_bits"]
        assert ui2[k].tolist() == want["ui2"] and ui12[k].tolist() == want["ui12"]
    rng = np.random.RandomState(5)
    seeds = np.concatenate([[0, 1, 2, 2147483646, 2147483647, 2147483648, 0xFFFFFFFF],
                            rng.randint(0, 2 ** 32, size=3993, dtype=np.uint64)]).astype(np.uint32)
    for L in (1, 2, 160, 65536):
        uni, nrm, ui = rt.selftest_rng(seeds, 64, nlights=L, what=0)
        for k in range(0, seeds.size, 7 if L != 2 else 1):
            u_o, n_o, i_o = oracle_mod.rng_stream(int(seeds[k]), 64, L)
            assert np.array_equal(uni[k].view(np.uint32), u_o.view(np.uint32)), (L, k)
            assert np.array_equal(nrm[k].view(np.uint32), n_o.view(np.uint32)), (L, k)
            assert np.array_equal(ui[k], i_o.astype(np.uint32)), (L, k)


@pytest.mark.parametrize("name,kw,depth", [("cornell", {}, 6), ("glass", {"grid": 3, "subdiv": 2}, 8)])
def test_converged_image_rmse_against_reference_stream(scene_dir, oracle_mod, name, kw, depth):
    """Converged images: the GPU's throughput path (wavefront pipeline, one hash-seeded engine per pixel and
    sample) against the CPU path with the REFERENCE's random stream (one engine per pixel carried across
    its samples, main.cpp:1220-1225), 96x72 at 256 spp.  The two are independent estimates of the same
    integrand, so their per-pixel RMSE (linear mean radiance, all three channels) can only be judged
    against the noise floor: the RMSE between two CPU renders that differ in nothing but the seed.
    Stated bar: RMSE <= 1.2 x floor, and image means within 2 %.  Measured (CPU restatement both sides, which
    the GPU equals to 1e-5): Cornell 0.063-0.071 against a floor of 0.063-0.072 (mean radiance 0.19);
    glass 0.113-0.120 against 0.112-0.118 (mean 0.43): the hash-seeded streams cost no variance.
    About one pixel per render holds a NaN (the reference's own arithmetic: e.g. a light sample that lands
    on the shading point itself); those pixels are left out on both sides."""
    import raytracing_amd as rt
    W, H, spp = 96, 72, 256
    host, gpu, orc = _load(scene_dir, oracle_mod, name, W, H, **kw)
    ref = orc.render(W, H, spp, max_depth=depth, rng_mode=0)[0] / spp
    got = gpu.render(W, H, spp, max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=11, pipeline=rt.PIPE_WAVEFRONT) / spp
    noise = [orc.render(W, H, spp, max_depth=depth, rng_mode=1, seed=s)[0] / spp for s in (12, 13, 14)]

    def rmse(a, b):
        ok = np.isfinite(a).all(axis=1) & np.isfinite(b).all(axis=1)
        assert ok.sum() >= a.shape[0] - 4
        return float(np.sqrt(np.mean((a[ok] - b[ok]) ** 2)))

    floor = np.mean([rmse(noise[0], noise[1]), rmse(noise[0], noise[2]), rmse(noise[1], noise[2])])
    err = rmse(got, ref)
    print("%s: RMSE vs reference stream %.4f, noise floor %.4f, mean radiance %.4f" % (name, err, floor, np.nanmean(ref)))
    assert err <= 1.2 * floor
    assert abs(np.nanmean(got) - np.nanmean(ref)) < 0.02 * np.nanmean(ref)


@pytest.mark.parametrize("name,kw,W,H,depth", [
    ("cornell", {}, 128, 96, 8),
    ("spheres1m", {"n_spheres": 40, "subdiv": 2, "ground": 16}, 100, 75, 8),   # 75 rows: not a multiple of 8, row-order records
])
def test_wavefront_multi_pass_matches_single_pass_and_oracle(scene_dir, oracle_mod, monkeypatch, name, kw, W, H, depth):
    """The wavefront pipeline renders `slots` frames of samples per pass and loops when a call asks for
    more (second generate at spp_begin + done, accumulate continuing from the stored sums).  With a path
    budget of two frames, 7 samples take 4 passes, the last one partial.  A pixel's samples are added in
    sample order whatever the pass boundaries, so the result is bit-equal to the one-pass render, and
    within 1e-5 of the oracle with equal segment counts."""
    import raytracing_amd as rt
    host, gpu, orc = _load(scene_dir, oracle_mod, name, W, H, **kw)
    ref, c_o = orc.render(W, H, 7, max_depth=depth, rng_mode=1, seed=77, spp_begin=3)
    gpu.reset_counters()
    one = gpu.render(W, H, 7, max_depth=depth, rng_mode=1, seed=77, spp_begin=3, pipeline=rt.PIPE_WAVEFRONT)
    assert gpu.counters()["segments"] == c_o["segments"]
    _assert_close(one, ref)
    monkeypatch.setenv("RT_WF_PATHS", str(2 * W * H))
    gpu2 = rt.Scene(host, device=0)
    gpu2.reset_counters()
    multi = gpu2.render(W, H, 7, max_depth=depth, rng_mode=1, seed=77, spp_begin=3, pipeline=rt.PIPE_WAVEFRONT)
    assert gpu2.counters()["segments"] == c_o["segments"]
    assert gpu2.stats()["device_bytes"] < gpu.stats()["device_bytes"]          # 2 frames of records, not 7
    _assert_bits_equal(multi, one)
    monkeypatch.setenv("RT_WF_PATHS", str(W * H))                              # one frame per pass: 7 passes
    _assert_bits_equal(gpu2.render(W, H, 7, max_depth=depth, rng_mode=1, seed=77, spp_begin=3, pipeline=rt.PIPE_WAVEFRONT), one)
    gpu2.close()


@pytest.mark.parametrize("name,kw,W,H,depth", [
    ("cornell", {}, 128, 96, 8),
    ("spheres1m", {"n_spheres": 40, "subdiv": 2, "ground": 16}, 100, 75, 8),
    ("glass", {"grid": 3, "subdiv": 2}, 96, 72, 16),
])
def test_wavefront_ray_sorting_changes_no_result(scene_dir, oracle_mod, monkeypatch, name, kw, W, H, depth):
    """From the second bounce on the wavefront pipeline can regroup the ray queue by the cell of the ray origin:
    RT_WF_SORT=1 on every bounce, =2 on the bounces a tuner on the device finds it pays for (it times both ways
    over the first passes: the several-pass renders below go through all its states; the default for scenes
    larger than the L2 caches), =0 never.
    Only the order in which rays are traced changes: every path's arithmetic is its own and sums are taken per
    slot, so all three renders are bit-equal, with equal segment and light-test counts, one pass or several."""
    import raytracing_amd as rt
    host, _, orc = _load(scene_dir, oracle_mod, name, W, H, **kw)
    ref, c_o = orc.render(W, H, 6, max_depth=depth, rng_mode=1, seed=21, spp_begin=2)
    outs = {}
    monkeypatch.setenv("RT_WF_SORT_MIN_RAYS", "0")               # the tuned mode ignores short queues otherwise
    for flag in ("0", "1", "2"):
        monkeypatch.setenv("RT_WF_SORT", flag)
        monkeypatch.delenv("RT_WF_PATHS", raising=False)
        with rt.Scene(host, device=0) as gpu:
            gpu.reset_counters()
            gpu.set_timing(True)
            outs[flag] = gpu.render(W, H, 6, max_depth=depth, rng_mode=1, seed=21, spp_begin=2, pipeline=rt.PIPE_WAVEFRONT)
            c = gpu.counters()
            assert c["segments"] == c_o["segments"] and c["light_tests"] == c_o["light_tests"]
            sorts = gpu.kernel_times()["sort"][1]
            assert sorts == (0 if flag == "0" else depth - 2)   # bounces 1 .. depth-2 can hand a sorted queue on
        _assert_close(outs[flag], ref)
    _assert_bits_equal(outs["1"], outs["0"])
    _assert_bits_equal(outs["2"], outs["0"])
    monkeypatch.setenv("RT_WF_PATHS", str(2 * W * H))           # three passes of two frames
    for flag in ("1", "2"):
        monkeypatch.setenv("RT_WF_SORT", flag)
        with rt.Scene(host, device=0) as gpu:
            _assert_bits_equal(gpu.render(W, H, 6, max_depth=depth, rng_mode=1, seed=21, spp_begin=2, pipeline=rt.PIPE_WAVEFRONT), outs["0"])


def test_ray_sorting_tuner_on_paths_deeper_than_its_books(scene_dir, oracle_mod, monkeypatch):
    """The tuner keeps books for 30 bounces; deeper bounces are never sorted, and the decision word of the last
    booked bounce must not change while that bounce's kernels read it.  Closed box, depth 34, three passes (so
    the tuner has measured bounce 29 unsorted and tries it sorted): bit-equal to the unsorted render, equal
    segment counts, and paths really reach the bounces beyond the books."""
    import raytracing_amd as rt
    W, H, depth = 128, 96, 34
    host, _, orc = _load(scene_dir, oracle_mod, "cornell", W, H)
    monkeypatch.setenv("RT_WF_SORT_MIN_RAYS", "0")
    monkeypatch.setenv("RT_WF_PATHS", str(2 * W * H))
    outs, segs = {}, {}
    for flag in ("0", "2", "1"):
        monkeypatch.setenv("RT_WF_SORT", flag)
        with rt.Scene(host, device=0) as gpu:
            gpu.reset_counters()
            outs[flag] = gpu.render(W, H, 6, max_depth=depth, rng_mode=1, seed=4, pipeline=rt.PIPE_WAVEFRONT)
            segs[flag] = gpu.counters()["segments"]
            again = gpu.render(W, H, 6, max_depth=depth, rng_mode=1, seed=4, pipeline=rt.PIPE_WAVEFRONT)
            _assert_bits_equal(again, outs[flag])                  # the tuner's state differs, the sums do not
    _assert_bits_equal(outs["2"], outs["0"])
    _assert_bits_equal(outs["1"], outs["0"])
    assert segs["0"] == segs["1"] == segs["2"]
    ref, c_o = orc.render(W, H, 6, max_depth=depth, rng_mode=1, seed=4)
    _assert_close(outs["0"], ref)
    shallow, c_s = orc.render(W, H, 6, max_depth=30, rng_mode=1, seed=4)
    assert c_o["segments"] > c_s["segments"]                       # some paths are longer than 30 segments
    # One path of this render meets a hit of the kind only the reference's own walk reproduces (a NaN distance:
    # the CPU path goes NaN there and stops being traced as the kernels' own walk traces it, 9 segments of 332 510).
    # With every ray on the reference's walk the megakernel counts what the oracle counts.
    assert abs(segs["0"] - c_o["segments"]) <= 1e-4 * c_o["segments"]
    monkeypatch.delenv("RT_WF_PATHS")
    with rt.Scene(host, device=0) as gpu:
        gpu.set_hit_order(rt.HITS_REFERENCE_WALK)
        gpu.reset_counters()
        _assert_close(gpu.render(W, H, 6, max_depth=depth, rng_mode=1, seed=4, pipeline=rt.PIPE_MEGAKERNEL), ref)
        assert gpu.counters()["segments"] == c_o["segments"]
        # ... and so does the wavefront pipeline, whose trace kernel then only lists the rays for the reference's walk
        gpu.reset_counters()
        _assert_close(gpu.render(W, H, 6, max_depth=depth, rng_mode=1, seed=4, pipeline=rt.PIPE_WAVEFRONT), ref)
        c = gpu.counters()
        assert c["segments"] == c_o["segments"] == c["retraced"]


@pytest.mark.parametrize("name,kw,W,H,depth", [
    ("cornell", {}, 128, 96, 8),
    ("spheres1m", {"n_spheres": 40, "subdiv": 2, "ground": 16}, 100, 75, 8),
    ("glass", {"grid": 3, "subdiv": 2}, 96, 72, 16),
])
def test_wavefront_wide_tree_and_its_retry_path(scene_dir, oracle_mod, monkeypatch, name, kw, W, H, depth):
    """wf_extend walks the four-child tree (8-bit boxes in a frame per node: looser boxes, same triangles) and
    hands a ray whose walk runs out of stack to a second kernel on the two-child tree.  The closest hit does
    not depend on the tree, so: bit-equal radiance sums and equal segment counts with the wide tree, with the
    two-child tree, and with the wide walk cut to 2 or 0 stack entries (most rays / every ray that meets two
    boxes in one node then go through the retry kernel), sorted or not."""
    import raytracing_amd as rt
    host, _, orc = _load(scene_dir, oracle_mod, name, W, H, **kw)
    ref, c_o = orc.render(W, H, 5, max_depth=depth, rng_mode=1, seed=33, spp_begin=1)
    outs = []
    for wide, limit, sort in (("0", None, "0"), ("1", None, "0"), ("1", "2", "1"), ("1", "0", "0")):
        monkeypatch.setenv("RT_WF_WIDE", wide)
        monkeypatch.setenv("RT_WF_SORT", sort)
        if limit is None:
            monkeypatch.delenv("RT_WF_STACK_LIMIT", raising=False)
        else:
            monkeypatch.setenv("RT_WF_STACK_LIMIT", limit)
        with rt.Scene(host, device=0) as gpu:
            gpu.reset_counters()
            outs.append(gpu.render(W, H, 5, max_depth=depth, rng_mode=1, seed=33, spp_begin=1, pipeline=rt.PIPE_WAVEFRONT))
            assert gpu.counters()["segments"] == c_o["segments"]
    _assert_close(outs[0], ref)
    for o in outs[1:]:
        _assert_bits_equal(o, outs[0])


def test_scratch_is_grow_only_and_sized_by_the_call(scene_dir, oracle_mod):
    """Path-record buffers: sized by min(samples of the call, frames the path budget allows), kept when a
    later call needs less, grown when it needs more, and accounted in device_bytes both ways."""
    import raytracing_amd as rt
    W, H = 64, 48
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", W, H)
    base = gpu.stats()["device_bytes"]
    per_frame = W * H * 90
    # counters, work cursors, cell histogram; and, the scene being one the reference can hold, the frames of the
    # reference-order walk that resolves equal distances: 65536 lanes x (tree depth + 1) x 16 bytes
    fixed = 70000 + 65536 * (gpu.stats()["ref_tree_depth"] + 1) * 16
    gpu.render(W, H, 1, rng_mode=1, pipeline=rt.PIPE_WAVEFRONT)
    b1 = gpu.stats()["device_bytes"] - base
    assert per_frame + fixed - 70000 <= b1 < per_frame + fixed      # one frame of 90 bytes per path
    gpu.reserve(W, H, 5, pipeline=rt.PIPE_WAVEFRONT)
    b5 = gpu.stats()["device_bytes"] - base
    assert 5 * per_frame + fixed - 70000 <= b5 < 5 * per_frame + fixed
    out5 = gpu.render(W, H, 5, rng_mode=1, seed=3, pipeline=rt.PIPE_WAVEFRONT)
    gpu.render(W // 2, H // 2, 2, rng_mode=1, pipeline=rt.PIPE_WAVEFRONT)
    assert gpu.stats()["device_bytes"] - base == b5       # smaller call: nothing reallocated
    assert np.array_equal(out5, gpu.render(W, H, 5, rng_mode=1, seed=3, pipeline=rt.PIPE_WAVEFRONT))
    gpu.reserve(W, H, 3, max_depth=6, rng_mode=rt.RNG_REFERENCE, pipeline=rt.PIPE_MEGAKERNEL)
    assert gpu.stats()["device_bytes"] - base > b5        # replay records of the reference-RNG mode


def test_config5_glass_depth16(scene_dir, oracle_mod):
    """BASELINE config 5's content at a size the oracle finishes: the dielectric-heavy scene at path
    depth 16.  Reference stream bit for bit; both throughput pipelines within 1e-5 of the oracle's
    counter mode with equal segment counts."""
    import raytracing_amd as rt
    W, H = 160, 90
    host, gpu, orc = _load(scene_dir, oracle_mod, "glass", W, H, grid=3, subdiv=2)
    ref0, c0 = orc.render(W, H, 3, max_depth=16, rng_mode=0)
    gpu.reset_counters()
    _assert_bits_equal(gpu.render(W, H, 3, max_depth=16, rng_mode=0), ref0)
    assert gpu.counters()["segments"] == c0["segments"]
    assert c0["segments"] > 4.0 * W * H * 3               # paths really are long in this scene
    ref, c_o = orc.render(W, H, 4, max_depth=16, rng_mode=1, seed=0x5EED, spp_begin=1)
    for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
        gpu.reset_counters()
        _assert_close(gpu.render(W, H, 4, max_depth=16, rng_mode=1, seed=0x5EED, spp_begin=1, pipeline=pipe), ref)
        assert gpu.counters()["segments"] == c_o["segments"]


def test_config5_full_size_properties_4k(scene_dir, oracle_mod):
    """3840x2160, depth 16, the bench's glass scene (819 k triangles): at this size a pass holds 32
    frames (2.65e8 path records), so 36 samples take two passes.  Checked through properties:
    sample-range additivity, determinism, agreement of the two pipelines, and segment counts per
    sample consistent with the oracle's on the same scene at 1/12 size."""
    import raytracing_amd as rt
    W, H = 3840, 2160
    host, gpu, orc = _load(scene_dir, oracle_mod, "glass", W, H, grid=6, subdiv=5)
    gpu.reset_counters()
    full = gpu.render(W, H, 36, max_depth=16, rng_mode=1, seed=8, pipeline=rt.PIPE_WAVEFRONT)
    segs = gpu.counters()["segments"]
    assert np.isfinite(full).all() and (full >= 0).all() and full.sum() > 0
    a = gpu.render(W, H, 34, max_depth=16, rng_mode=1, seed=8, spp_begin=0, pipeline=rt.PIPE_WAVEFRONT)
    b = gpu.render(W, H, 2, max_depth=16, rng_mode=1, seed=8, spp_begin=34, pipeline=rt.PIPE_WAVEFRONT)
    assert np.allclose(a + b, full, rtol=1e-5, atol=1e-6)
    assert np.array_equal(full, gpu.render(W, H, 36, max_depth=16, rng_mode=1, seed=8, pipeline=rt.PIPE_WAVEFRONT))
    mega = gpu.render(W, H, 2, max_depth=16, rng_mode=1, seed=8, spp_begin=34, pipeline=rt.PIPE_MEGAKERNEL)
    assert np.allclose(mega, b, rtol=1e-5, atol=1e-6)
    _, c_small = orc.render(W // 12, H // 12, 4, max_depth=16, rng_mode=1, seed=8)
    per_small = c_small["segments"] / float((W // 12) * (H // 12) * 4)
    assert abs(segs / float(W * H * 36) - per_small) < 0.03 * per_small


def test_cornell_1080p_more_samples_than_a_pass(scene_dir, oracle_mod):
    """BASELINE config 3's frame (Cornell box, 1920x1080, depth 8) with 134 samples: more than the 128
    frames a pass holds, so the second pass runs.  Additivity over the pass boundary, determinism, and
    the mean image against the oracle's converged low-resolution render of the same view."""
    import raytracing_amd as rt
    W, H = 1920, 1080
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", W, H)
    full = gpu.render(W, H, 134, max_depth=8, rng_mode=1, seed=5, pipeline=rt.PIPE_WAVEFRONT)
    a = gpu.render(W, H, 128, max_depth=8, rng_mode=1, seed=5, spp_begin=0, pipeline=rt.PIPE_WAVEFRONT)
    b = gpu.render(W, H, 6, max_depth=8, rng_mode=1, seed=5, spp_begin=128, pipeline=rt.PIPE_WAVEFRONT)
    assert np.allclose(a + b, full, rtol=1e-5, atol=1e-6)
    assert np.array_equal(full, gpu.render(W, H, 134, max_depth=8, rng_mode=1, seed=5, pipeline=rt.PIPE_WAVEFRONT))
    assert np.isfinite(full).all() and (full >= 0).all()
    # block means of the frame (8x8 pixels x 134 samples) against the oracle's 240x135 render at 128 spp
    w, h = W // 8, H // 8
    small, _ = orc.render(w, h, 128, max_depth=8, rng_mode=1, seed=6)
    blocks = full.reshape(h, 8, w, 8, 3).mean(axis=(1, 3)) / 134.0
    small = small.reshape(h, w, 3) / 128.0
    assert abs(blocks.mean() - small.mean()) < 0.02 * small.mean()
    # two independent 128-spp oracle renders of this view differ by 4 % in the median pixel; the GPU's
    # 8576-sample blocks against one of them must do no worse (a structural error would)
    assert np.median(np.abs(blocks - small) / np.maximum(small, 1e-3)) < 0.05


# ---------------------------------------------------------------------------------------------
# committed fixtures, edge cases, full-size checks
# ---------------------------------------------------------------------------------------------
def test_golden_cornell_fixture(scene_dir, oracle_mod):
    """The committed oracle outputs (tests/golden/cornell_64x48.npz) as the target."""
    import os
    import raytracing_amd as rt
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_64x48.npz"))
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", 64, 48)
    assert np.array_equal(host.tri_verts, g["tri_verts"])
    assert gpu.stats()["hit_order"] == rt.HITS_REFERENCE_ORDER   # the default for a scene the reference can hold
    ids, t = gpu.trace_primary(64, 48)
    _compare_hits(ids, t, g["ids"], g["t"], None)
    _assert_bits_equal(gpu.render(64, 48, 4, rng_mode=0, max_depth=6), g["sum_reference_rng"])
    for pipe in (0, 1):
        _assert_close(gpu.render(64, 48, 4, rng_mode=1, max_depth=8, seed=0x5EED, spp_begin=2, pipeline=pipe),
                      g["sum_counter_rng"])


def test_golden_cornell_onfloor_fixture(scene_dir, oracle_mod):
    """tests/golden/cornell_onfloor_64x48.npz: the boxes stand ON the floor, 1381 of the render's rays meet two
    faces at exactly the same distance.  Reference order: every id, the reference-stream sums bit for bit, the
    throughput pipelines within 1e-5 -- against the committed vectors, no mask."""
    import os
    import raytracing_amd as rt
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cornell_onfloor_64x48.npz"))
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell_onfloor", 64, 48)
    assert np.array_equal(host.tri_verts, g["tri_verts"])
    assert gpu.stats()["hit_order"] == rt.HITS_REFERENCE_ORDER
    ids, t = gpu.trace_primary(64, 48)
    _compare_hits(ids, t, g["ids"], g["t"], None)
    _assert_bits_equal_nan(gpu.render(64, 48, 4, rng_mode=0, max_depth=8), g["sum_reference_rng"])
    gpu.reset_counters()
    gpu.set_counting(True)
    for pipe in (0, 1):
        _assert_close(gpu.render(64, 48, 4, rng_mode=1, max_depth=8, seed=0x5EED, spp_begin=2, pipeline=pipe),
                      g["sum_counter_rng"])
    gpu.set_counting(False)
    c = gpu.counters()
    assert c["segments"] == 2 * int(g["segments"][1]) and c["retraced"] > 0   # the wavefront render re-traced its ties


def test_golden_emissive_fixture(scene_dir, oracle_mod):
    """tests/golden/emissive_48x36.npz: 160 light triangles, so the light pdf goes through the light tree."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "emissive_48x36.npz"))
    host, gpu, orc = _load(scene_dir, oracle_mod, "emissive", 48, 36, subdiv=1, lamps=2)
    assert np.array_equal(host.tri_verts, g["tri_verts"]) and gpu.stats()["nlights"] == 160
    ids, t = gpu.trace_primary(48, 36)
    _compare_hits(ids, t, g["ids"], g["t"], None)   # reference order: shared edges included
    _assert_bits_equal(gpu.render(48, 36, 2, rng_mode=0, max_depth=5), g["sum_reference_rng"])
    for pipe in (0, 1):
        _assert_close(gpu.render(48, 36, 2, rng_mode=1, max_depth=5, seed=11, pipeline=pipe), g["sum_counter_rng"])


def test_config2_full_size_primary_hits(scene_dir, oracle_mod):
    """BASELINE config 2 at full size: 100 354-triangle mesh, 1920x1080 primary rays, hit ids and
    distances bit for bit against the oracle (which builds its own, different, BVH)."""
    W, H = 1920, 1080
    host, gpu, orc = _load(scene_dir, oracle_mod, "heightfield100k", W, H)
    assert host.tri_verts.shape[0] == 100354
    ids_g, t_g = gpu.trace_primary(W, H)
    ids_o, t_o, tie, _ = orc.trace_primary(W, H)
    nties = _compare_hits(ids_g, t_g, ids_o, t_o, tie)
    assert nties <= 20
    assert (ids_g != 0xFFFFFFFF).mean() > 0.5
    # same rays through the explicit-ray entry point give the same answer
    o, d = orc.primary_rays(W, H)
    sel = np.random.RandomState(0).choice(W * H, 50000, replace=False)
    ids_r, t_r = gpu.intersect(o[sel], d[sel])
    assert np.array_equal(ids_r, ids_g[sel]) and np.array_equal(t_r, t_g[sel])


def test_degenerate_scenes(oracle_mod):
    """Empty scene, one triangle, no lights, only lights."""
    import raytracing_amd as rt
    mats = np.zeros(2, dtype=rt.api.MATERIAL_DTYPE)
    mats["color"] = 0.5
    mats["ior"] = 1.5
    mats["emission"][1] = (3, 2, 1)
    cam = dict(position=[0, 0, 3], right=[1, 0, 0], up=[0, 1, 0], forward=[0, 0, -1], tan_x=0.5, tan_y=0.5)
    W = H = 32
    empty = rt.HostScene(np.zeros((0, 3, 3), np.float32), np.zeros(0, np.uint32), mats, cam)
    with rt.Scene(empty) as s:
        ids, t = s.trace_primary(W, H)
        assert (ids == 0xFFFFFFFF).all() and (t == np.float32(1e4)).all()
        for pipe in (0, 1):
            assert not s.render(W, H, 3, rng_mode=1, pipeline=pipe).any()
    tri = np.array([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]], np.float32)
    for mat in (0, 1):   # a lone diffuse triangle (no lights), a lone emitter
        host = rt.HostScene(tri, np.array([mat], np.uint32), mats, cam)
        orc = oracle_mod.OracleScene(tri, np.array([mat], np.uint32), mats, cam)
        with rt.Scene(host) as s:
            ids, t = s.trace_primary(W, H)
            ids_o, t_o, tie, _ = orc.trace_primary(W, H)
            _compare_hits(ids, t, ids_o, t_o, tie)
            assert 0 < (ids == 0).sum() < W * H
            ref, _ = orc.render(W, H, 4, max_depth=6, rng_mode=0)
            _assert_bits_equal(s.render(W, H, 4, max_depth=6, rng_mode=0), ref)
            assert ref.any() == (mat == 1)


def test_depth_parameter_and_zero_samples(scene_dir, oracle_mod):
    W, H = 48, 36
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", W, H)
    for depth in (0, 1, 2, 16):
        ref, c_o = orc.render(W, H, 3, max_depth=depth, rng_mode=1, seed=4)
        for pipe in (0, 1):
            gpu.reset_counters()
            _assert_close(gpu.render(W, H, 3, max_depth=depth, rng_mode=1, seed=4, pipeline=pipe), ref)
            assert gpu.counters()["segments"] == c_o["segments"], (depth, pipe)
        ref0, c0 = orc.render(W, H, 3, max_depth=depth, rng_mode=0)
        gpu.reset_counters()
        _assert_bits_equal(gpu.render(W, H, 3, max_depth=depth, rng_mode=0), ref0)
        assert gpu.counters()["segments"] == c0["segments"]
    for pipe in (0, 1):
        assert not gpu.render(W, H, 0, rng_mode=1, pipeline=pipe).any()
    import raytracing_amd as rt
    with pytest.raises(rt.RtError):
        gpu.render(W, H, 2, rng_mode=0, spp_begin=1)            # the reference stream cannot start mid-pixel
    with pytest.raises(rt.RtError):
        gpu.render(W, H, 2, rng_mode=0, pipeline=1)             # wavefront needs the counter RNG
    with pytest.raises(rt.RtError):
        gpu.render(W, H, 2, rng_mode=0, max_depth=65)           # the replay kernel records at most 64 vertices


def test_full_size_properties_1080p(scene_dir, oracle_mod):
    """At BASELINE's frame size the oracle is too slow for a full comparison, so the 1080p run is
    checked through properties: (1) partial sums of disjoint sample ranges add up to the full
    range's sum, as the multi-GPU split needs; (2) both pipelines agree; (3) launches are
    deterministic; (4) a crop of the frame agrees with the oracle rendering the same pixels is not
    possible (the oracle renders whole frames), so instead the frame at 1/8 size is compared with
    the oracle in test_render_counter_mode_matches_oracle and the ray counts per sample must agree
    between the two frame sizes within sampling noise."""
    W, H = 1920, 1080
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", W, H)
    gpu.reset_counters()
    full = gpu.render(W, H, 4, max_depth=8, rng_mode=1, seed=21, pipeline=1)
    segs_full = gpu.counters()["segments"]
    a = gpu.render(W, H, 1, max_depth=8, rng_mode=1, seed=21, spp_begin=0, pipeline=1)
    b = gpu.render(W, H, 3, max_depth=8, rng_mode=1, seed=21, spp_begin=1, pipeline=1)
    assert np.allclose(a + b, full, rtol=1e-5, atol=1e-6)
    mega = gpu.render(W, H, 4, max_depth=8, rng_mode=1, seed=21, pipeline=0)
    assert np.allclose(mega, full, rtol=1e-6, atol=1e-7)
    again = gpu.render(W, H, 4, max_depth=8, rng_mode=1, seed=21, pipeline=1)
    assert np.array_equal(again, full)
    assert np.isfinite(full).all() and (full >= 0).all()
    _, c_small = orc.render(W // 8, H // 8, 4, max_depth=8, rng_mode=1, seed=21)
    per_sample_small = c_small["segments"] / float((W // 8) * (H // 8) * 4)
    per_sample_full = segs_full / float(W * H * 4)
    assert abs(per_sample_full - per_sample_small) < 0.03 * per_sample_small
    # the bit-exact mode at full size: deterministic, and consistent with the throughput kernels'
    # estimate of the same integrand (different random numbers, so only statistically)
    # (in the reference's hit order this mode reproduces the reference's NaN "hits" too: a few pixels of the two
    # million are NaN, as they are on the CPU)
    exact = gpu.render(W, H, 2, max_depth=8, rng_mode=0)
    assert np.array_equal(exact.view(np.uint32), gpu.render(W, H, 2, max_depth=8, rng_mode=0).view(np.uint32))
    assert np.isnan(exact).any(axis=1).sum() < 20
    assert abs(np.nanmean(exact) / 2 - full.mean() / 4) < 0.05 * full.mean() / 4


def test_million_triangle_scene_matches_oracle_on_primary_rays(scene_dir, oracle_mod):
    """The bench scene (1 006 600 triangles, mixed materials): primary hits at 480x270 bit for bit,
    and a 2-spp depth-8 render within the radiance tolerances."""
    W, H = 480, 270
    host, gpu, orc = _load(scene_dir, oracle_mod, "spheres1m", W, H)
    assert gpu.stats()["ntri"] == 1006600 and gpu.stats()["tree_height"] <= 24
    ids_g, t_g = gpu.trace_primary(W, H)
    ids_o, t_o, tie, _ = orc.trace_primary(W, H)
    _compare_hits(ids_g, t_g, ids_o, t_o, tie)
    ref, c_o = orc.render(W, H, 2, max_depth=8, rng_mode=1, seed=0x5EED)
    for pipe in (0, 1):
        gpu.reset_counters()
        _assert_close(gpu.render(W, H, 2, max_depth=8, rng_mode=1, seed=0x5EED, pipeline=pipe), ref)
        assert gpu.counters()["segments"] == c_o["segments"]
    ref0, c0 = orc.render(W, H, 1, max_depth=6, rng_mode=0)
    gpu.reset_counters()
    _assert_bits_equal(gpu.render(W, H, 1, max_depth=6, rng_mode=0), ref0)
    assert gpu.counters()["segments"] == c0["segments"]


def test_config4_full_size_frame_properties(scene_dir, oracle_mod, monkeypatch):
    """BASELINE config 4's own frame -- the scene and image size the bench line is measured on: 1 006 600
    triangles, 1920x1080, depth 8, the scene's default ray sorting (tuned on the device), the four-child walk.  With
    a path budget of four frames 12 samples take three passes whose queues stay above 2^20 rays for several
    bounces, so the tuner goes through its states (unsorted, sorted, chosen) inside the call.  The oracle cannot
    render this frame, so it is checked through properties (generatePixel's sum, main.cpp:1220-1225):
    (i) bit-equal to the same render with sorting off; (ii) sample ranges add up across a pass boundary;
    (iii) a second call (the tuner in another state) gives the same bits; (iv) wavefront and megakernel agree to
    1e-5 on 2 samples; (v) the sort kernels did real work; (vi) ray segments per sample within 3 % of the
    oracle's on the same scene at 480x270."""
    import raytracing_amd as rt
    W, H, depth = 1920, 1080, 8
    monkeypatch.delenv("RT_WF_SORT", raising=False)
    monkeypatch.delenv("RT_WF_SORT_MIN_RAYS", raising=False)
    monkeypatch.setenv("RT_WF_PATHS", str(4 * W * H))
    host, gpu, orc = _load(scene_dir, oracle_mod, "spheres1m", W, H)
    assert gpu.stats()["ntri"] == 1006600 and gpu.stats()["hit_order"] == rt.HITS_LEAF_ORDER   # beyond the reference's limits
    gpu.set_timing(True)
    gpu.reset_counters()
    full = gpu.render(W, H, 12, max_depth=depth, rng_mode=1, seed=0x5EED, pipeline=rt.PIPE_WAVEFRONT)
    segs = gpu.counters()["segments"]
    times = gpu.kernel_times()
    gpu.set_timing(False)
    print("kernel times (ms, launches):", times)
    assert times["sort"][1] > 0 and times["sort"][0] > 0.5, times["sort"]     # (v) ms: an idle sort launch takes 0.01
    assert np.isfinite(full).all() and (full >= 0).all() and full.sum() > 0
    again = gpu.render(W, H, 12, max_depth=depth, rng_mode=1, seed=0x5EED, pipeline=rt.PIPE_WAVEFRONT)
    assert np.array_equal(again, full)                                        # (iii)
    a = gpu.render(W, H, 10, max_depth=depth, rng_mode=1, seed=0x5EED, spp_begin=0, pipeline=rt.PIPE_WAVEFRONT)
    b = gpu.render(W, H, 2, max_depth=depth, rng_mode=1, seed=0x5EED, spp_begin=10, pipeline=rt.PIPE_WAVEFRONT)
    assert np.allclose(a + b, full, rtol=1e-5, atol=1e-6)                     # (ii)
    mega = gpu.render(W, H, 2, max_depth=depth, rng_mode=1, seed=0x5EED, spp_begin=10, pipeline=rt.PIPE_MEGAKERNEL)
    assert np.allclose(mega, b, rtol=1e-5, atol=1e-6)                         # (iv)
    monkeypatch.setenv("RT_WF_SORT", "0")
    with rt.Scene(host, device=0) as plain:
        plain.set_timing(True)
        unsorted = plain.render(W, H, 12, max_depth=depth, rng_mode=1, seed=0x5EED, pipeline=rt.PIPE_WAVEFRONT)
        assert plain.kernel_times()["sort"][1] == 0
    assert np.array_equal(unsorted, full)                                     # (i)
    _, c_small = orc.render(W // 4, H // 4, 2, max_depth=depth, rng_mode=1, seed=0x5EED)
    per_small = c_small["segments"] / float((W // 4) * (H // 4) * 2)
    assert abs(segs / float(W * H * 12) - per_small) < 0.03 * per_small         # (vi)


def test_counting_launches_match_oracle_semantics(scene_dir, oracle_mod):
    """The counted launches report nodes / triangle tests per ray for the roofline model.  Segment
    and light-pdf test counts are properties of the estimator and equal the oracle's."""
    W, H = 64, 48
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", W, H)
    ref, c_o = orc.render(W, H, 4, max_depth=6, rng_mode=0)
    gpu.set_counting(True)
    gpu.reset_counters()
    gpu.render(W, H, 4, max_depth=6, rng_mode=0)
    c = gpu.counters()
    gpu.set_counting(False)
    assert c["segments"] == c_o["segments"] and c["light_tests"] == c_o["light_tests"]
    # the reference-stream render of a scene within the reference's limits walks the reference's own tree in
    # its own order: the very same node visits and triangle tests
    assert c["nodes"] == c_o["nodes"] and c["tri_tests"] == c_o["tri_tests"]
    import raytracing_amd as rt
    gpu.set_hit_order(rt.HITS_LEAF_ORDER)
    gpu.set_counting(True)
    gpu.reset_counters()
    gpu.render(W, H, 4, max_depth=6, rng_mode=0)
    c = gpu.counters()
    gpu.set_counting(False)
    assert c["segments"] == c_o["segments"] and c["light_tests"] == c_o["light_tests"]
    assert c["nodes"] >= c["segments"] and c["tri_tests"] > 0
    # the reference's tree (2x inflated boxes, no pruning across subtrees) costs far more node visits than the kernels' own
    assert c["nodes"] < c_o["nodes"]


def test_counting_launches_watch_the_walks(oracle_mod, monkeypatch):
    """A ray's walk ends when its lane holds neither a node nor a leaf, which rests on an invariant of the
    traversal state machine; a state that broke it would spin a kernel forever.  The counting launches bound
    every walk (four times the tree's nodes and leaves) and rt_counters_read reports a walk that hit the bound.
    On triangle soups, every kernel family: no alarm.  With the bound forced to 3 the alarm is raised -- the
    plumbing works -- and the launch still ends."""
    import raytracing_amd as rt
    rng = np.random.RandomState(8)
    n = 4000
    tri = (rng.uniform(-1, 1, size=(n, 1, 3)) + rng.uniform(-0.2, 0.2, size=(n, 3, 3))).astype(np.float32)
    mats = np.zeros(3, dtype=rt.api.MATERIAL_DTYPE)
    mats["color"], mats["type"], mats["ior"] = 0.7, [0, 1, 2], 1.5
    mats["emission"][0] = (2, 2, 2)
    which = rng.randint(0, 3, n).astype(np.uint32)
    cam = dict(position=[0, 0, 4], right=[1, 0, 0], up=[0, 1, 0], forward=[0, 0, -1], tan_x=0.5, tan_y=0.375)
    host = rt.HostScene(tri, which, mats, cam)
    m = 50000
    o = rng.uniform(-1.2, 1.2, size=(m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    for bound, order in ((None, rt.HITS_LEAF_ORDER), (None, rt.HITS_REFERENCE_ORDER), ("3", rt.HITS_LEAF_ORDER)):
        if bound is None:
            monkeypatch.delenv("RT_TEST_STEP_BOUND", raising=False)
        else:
            monkeypatch.setenv("RT_TEST_STEP_BOUND", bound)
        with rt.Scene(host, device=0) as gpu:
            gpu.set_hit_order(order)
            gpu.set_counting(True)
            for what in ("rays", "primary", "mega", "wavefront", "replay"):
                gpu.reset_counters()
                if what == "rays":
                    gpu.intersect(o, d)
                elif what == "primary":
                    gpu.trace_primary(128, 96)
                elif what == "replay":
                    gpu.render(64, 48, 2, max_depth=5, rng_mode=rt.RNG_REFERENCE)
                else:
                    gpu.render(128, 96, 3, max_depth=6, rng_mode=rt.RNG_COUNTER, seed=2,
                               pipeline=rt.PIPE_MEGAKERNEL if what == "mega" else rt.PIPE_WAVEFRONT)
                if bound is None:
                    c = gpu.counters()
                    assert c["nodes"] > 0 and c["tri_tests"] > 0, what
                else:
                    with pytest.raises(rt.RtError) as ei:
                        gpu.counters()
                    assert ei.value.code == -1 and "step bound" in str(ei.value), what


def test_command_line_end_to_end(scene_dir, oracle_mod, tmp_path):
    """The reference's command line (main.cpp:1244-1259): `raytracing scene W H spp out.ppm` writes the
    P6 file the CPU path would write -- same header, same bytes."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, make_scene
    from oracle import gltf_ref
    W, H, spp = 160, 120, 4
    path = make_scene(scene_dir, "cornell")
    sc = gltf_ref.parse_gltf(path, W, H)
    orc = oracle_mod.OracleScene(sc["tri_verts"], sc["tri_material"], sc["materials"], sc["camera"])
    ref, _ = orc.render(W, H, spp, max_depth=6, rng_mode=0)
    _, want = oracle_mod.tonemap(ref, spp)
    header = b"P6\n%d %d\n255\n" % (W, H)
    for cmd in ([os.path.join(ROOT, "raytracing_amd", "bin", "raytracing")],
                [sys.executable, "-m", "raytracing_amd"]):
        out = str(tmp_path / "out.ppm")
        r = subprocess.run(cmd + [path, str(W), str(H), str(spp), out], capture_output=True, text=True, cwd=ROOT,
                           timeout=280)
        assert r.returncode == 0, r.stderr
        assert "Processed input after" in r.stdout and "Finished after" in r.stdout
        data = open(out, "rb").read()
        assert data.startswith(header) and len(data) == len(header) + W * H * 3
        got = np.frombuffer(data[len(header):], dtype=np.uint8).reshape(-1, 3)
        assert np.array_equal(got, want), "%d image bytes differ" % (got != want).sum()
        os.remove(out)


@pytest.mark.parametrize("name", ["cornell", "cornell_onfloor"])
def test_config1_the_reference_cpu_case_at_its_own_size(scene_dir, oracle_mod, name):
    """BASELINE config 1: the reference's plumbing case, a Cornell-box glTF at 640x480 with 16 samples and the
    reference's fixed depth of 6 (main.cpp:1216), its own random stream.  Radiance sums bit for bit, image floats
    and bytes equal, ray-segment count equal -- also with the boxes standing ON the floor (coincident faces)."""
    import raytracing_amd as rt
    W, H, spp = 640, 480, 16
    host, gpu, orc = _load(scene_dir, oracle_mod, name, W, H)
    ref, c_o = orc.render(W, H, spp, max_depth=6, rng_mode=0)
    gpu.reset_counters()
    out = gpu.render(W, H, spp, max_depth=6, rng_mode=rt.RNG_REFERENCE)
    _assert_bits_equal_nan(out, ref)
    assert gpu.counters()["segments"] == c_o["segments"]
    img_o, rgb_o = oracle_mod.tonemap(ref, spp)
    img_g, rgb_g = gpu.tonemap(out, spp)
    assert np.array_equal(rgb_g, rgb_o)
    same = (img_g.view(np.uint32) == img_o.view(np.uint32)) | (np.isnan(img_g) & np.isnan(img_o))
    assert same.all()
    gpu.close()
    orc.close()


@pytest.mark.parametrize("name,kw,W,H,spp,depth", [
    ("emissive", {"subdiv": 2, "lamps": 3}, 96, 72, 2, 5),     # 960 light triangles
    ("cornell", {}, 96, 72, 4, 6),                              # 2 lights, tree forced
])
def test_light_tree_matches_the_loop_over_all_lights(scene_dir, oracle_mod, monkeypatch, name, kw, W, H, spp, depth):
    """LightSurface::pdf (main.cpp:771-783) sums a term for every light triangle the direction hits.
    Scenes with 48 lights or more evaluate it through a tree over the light triangles; the terms
    are added in light-list order there too, so the radiance sums still equal the oracle's (which
    loops over all lights) bit for bit -- and the throughput kernels stay within 1e-5 of it."""
    import raytracing_amd as rt
    monkeypatch.setenv("RT_LIGHT_TREE_MIN", "1")
    host, gpu, orc = _load(scene_dir, oracle_mod, name, W, H, **kw)
    assert gpu.stats()["nlights"] == (960 if name == "emissive" else 2)
    ref, c_o = orc.render(W, H, spp, max_depth=depth, rng_mode=0)
    out = gpu.render(W, H, spp, max_depth=depth, rng_mode=rt.RNG_REFERENCE, pipeline=rt.PIPE_MEGAKERNEL)
    _assert_bits_equal(out, ref)
    assert out.sum() > 0
    ref_c, c_c = orc.render(W, H, spp, max_depth=depth, rng_mode=1, seed=3)
    for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
        gpu.set_counting(True)
        gpu.reset_counters()
        out_c = gpu.render(W, H, spp, max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=3, pipeline=pipe)
        c = gpu.counters()
        gpu.set_counting(False)
        _assert_close(out_c, ref_c)
        assert c["segments"] == c_c["segments"]
        if name == "emissive":   # the point of the tree: far fewer exact light tests than lights x vertices
            assert c["light_tests"] < 0.05 * c_c["light_tests"]
    # the wavefront shading kernel keeps a 16-entry stack for shallow light trees and a 24-entry one otherwise
    monkeypatch.setenv("RT_LIGHT_STACK_FULL", "1")
    with rt.Scene(host, device=0) as gpu2:
        _assert_bits_equal(gpu2.render(W, H, spp, max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=3, pipeline=rt.PIPE_WAVEFRONT), out_c)


@pytest.mark.parametrize("seed,n_emissive", [(5, 6), (6, 200)])
def test_random_material_soup_renders_bit_exact(oracle_mod, seed, n_emissive):
    """A soup of 700 random triangles with all three material types and `n_emissive` emissive ones
    scattered through it: with 200 lights a sampled direction often crosses more than four of them,
    which exercises the light tree's repeated walks (terms must still be added in light-list
    order); with 6 the LDS loop runs.  Reference-RNG sums bit for bit, counter-RNG kernels within 1e-5."""
    import raytracing_amd as rt
    rng = np.random.RandomState(seed)
    n = 700
    centre = rng.uniform(-1.0, 1.0, size=(n, 1, 3))
    tri = (centre + rng.uniform(-0.35, 0.35, size=(n, 3, 3))).astype(np.float32)
    mats = np.zeros(5, dtype=rt.api.MATERIAL_DTYPE)
    mats["color"] = [[0.7, 0.6, 0.5], [0.9, 0.9, 0.9], [0.95, 1.0, 0.97], [0.8, 0.8, 0.8], [0.3, 0.4, 0.8]]
    mats["type"] = [0, 1, 2, 0, 0]
    mats["ior"] = 1.5
    mats["emission"][3] = [4.0, 3.5, 3.0]
    which = rng.choice([0, 1, 2, 4], size=n, p=[0.5, 0.2, 0.2, 0.1]).astype(np.uint32)
    which[rng.choice(n, n_emissive, replace=False)] = 3
    cam = dict(position=[0.0, 0.0, 4.0], right=[1, 0, 0], up=[0, 1, 0], forward=[0, 0, -1], tan_x=0.5, tan_y=0.375)
    host = rt.HostScene(tri, which, mats, cam)
    gpu = rt.Scene(host, device=0)
    orc = oracle_mod.OracleScene(tri, which, mats, cam)
    assert gpu.stats()["nlights"] == n_emissive
    W, H, spp, depth = 64, 48, 3, 6
    ref, _ = orc.render(W, H, spp, max_depth=depth, rng_mode=0)
    _assert_bits_equal(gpu.render(W, H, spp, max_depth=depth, rng_mode=rt.RNG_REFERENCE), ref)
    assert ref.sum() > 0
    ref_c, c_c = orc.render(W, H, spp, max_depth=depth, rng_mode=1, seed=9)
    for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
        gpu.reset_counters()
        out_c = gpu.render(W, H, spp, max_depth=depth, rng_mode=rt.RNG_COUNTER, seed=9, pipeline=pipe)
        _assert_close(out_c, ref_c)
        assert gpu.counters()["segments"] == c_c["segments"]
    gpu.close()
    orc.close()


def test_multi_gpu_entry_point_on_one_gpu(scene_dir, oracle_mod, tmp_path):
    """rt_multi_create / rt_multi_render with one GPU: the whole path runs -- shared host BVH build, upload,
    RCCL communicator (ncclCommInitAll over one device), per-GPU stream, ncclReduce -- and the sums equal
    rt_render's bit for bit (with n GPUs the reduce adds n partial sums per pixel: about n * 2^-24
    relative).  Also through the command line (`raytracing ... --gpus 1`)."""
    import os
    import subprocess
    import raytracing_amd as rt
    from conftest import ROOT
    W, H = 96, 72
    host, gpu, orc = _load(scene_dir, oracle_mod, "cornell", W, H)
    with rt.MultiScene(host, [0]) as multi:
        assert multi.ndev == 1
        for pipe in (rt.PIPE_WAVEFRONT, rt.PIPE_MEGAKERNEL):
            want = gpu.render(W, H, 5, max_depth=8, rng_mode=rt.RNG_COUNTER, seed=21, spp_begin=2, pipeline=pipe)
            got = multi.render(W, H, 5, max_depth=8, seed=21, spp_begin=2, pipeline=pipe)
            _assert_bits_equal(got, want)
        ref, _ = orc.render(W, H, 5, max_depth=8, rng_mode=1, seed=21, spp_begin=2)
        _assert_close(got, ref)
        assert multi.scene(0).stats()["ntri"] == 36
        assert not multi.render(W, H, 0, seed=1).any()
    with pytest.raises(rt.RtError):
        rt.MultiScene(host, [0, 0])
    with pytest.raises(rt.RtError):
        rt.MultiScene(host, [rt.device_count()])
    path = make_scene(scene_dir, "cornell")
    exe = os.path.join(ROOT, "raytracing_amd", "bin", "raytracing")
    outs = []
    for extra in (["--gpus", "1"], []):
        out = str(tmp_path / ("multi%d.ppm" % len(outs)))
        r = subprocess.run([exe, path, str(W), str(H), "4", out, "--rng", "counter", "--pipeline", "wavefront", "--seed", "3"]
                           + extra, capture_output=True, text=True, cwd=ROOT, timeout=280)
        assert r.returncode == 0, r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > W * H * 3
    r = subprocess.run([exe, path, str(W), str(H), "4", str(tmp_path / "x.ppm"), "--gpus", "1"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 2 and "--rng counter" in r.stderr


def test_background_colour_and_per_material_ior(scene_dir, oracle_mod):
    """Two inputs the reference's InputData has but its loader never sets (main.cpp:65, 694): a background
    colour (what applyLight returns for a ray that hits nothing, main.cpp:1190-1192) and an index of refraction
    other than 1.5 (applyLightDielectric, main.cpp:1149-1186).  An open scene (spheres over a ground plane:
    most paths end in the background) with water- and diamond-like glass: reference stream bit for bit,
    both throughput pipelines within 1e-5, equal segment counts."""
    import raytracing_amd as rt
    W, H = 128, 96
    path = make_scene(scene_dir, "spheres1m", n_spheres=40, subdiv=2, ground=16)
    base = rt.load_gltf(path, W, H)
    mats = base.materials.copy()
    glass = np.flatnonzero(mats["type"] == 2)
    assert glass.size >= 1
    mats["ior"][glass] = np.where(np.arange(glass.size) % 2 == 0, 1.33, 2.4).astype(np.float32)
    host = rt.HostScene(base.tri_verts, base.tri_material, mats, base.camera, background=(0.35, 0.5, 0.9))
    gpu = rt.Scene(host, device=0)
    d = host.as_dict()
    orc = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"], d["background"])
    ref0, c0 = orc.render(W, H, 3, max_depth=8, rng_mode=0)
    gpu.reset_counters()
    _assert_bits_equal(gpu.render(W, H, 3, max_depth=8, rng_mode=0), ref0)
    assert gpu.counters()["segments"] == c0["segments"]
    plain, _ = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], base.materials, d["camera"]).render(
        W, H, 3, max_depth=8, rng_mode=0)
    assert ref0.mean() > 2.0 * plain.mean()               # the sky really lights the scene
    ref, c_o = orc.render(W, H, 4, max_depth=8, rng_mode=1, seed=5, spp_begin=1)
    for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
        gpu.reset_counters()
        _assert_close(gpu.render(W, H, 4, max_depth=8, rng_mode=1, seed=5, spp_begin=1, pipeline=pipe), ref)
        assert gpu.counters()["segments"] == c_o["segments"]
    with pytest.raises(rt.RtError):
        rt.Scene(rt.HostScene(base.tri_verts, base.tri_material, mats, base.camera, background=(-1, 0, 0)))


def _analytic_prims(rt, rng, mat_ids):
    """A plane, ellipsoids and boxes at random poses: rt_primitive records (main.cpp:242-464)."""
    def unit_quat():
        q = rng.normal(size=4)
        return (q / np.linalg.norm(q)).astype(np.float32)
    prims = np.zeros(7, dtype=rt.PRIMITIVE_DTYPE)
    prims["type"] = [rt.PRIM_PLANE, rt.PRIM_ELLIPSOID, rt.PRIM_ELLIPSOID, rt.PRIM_BOX, rt.PRIM_BOX, rt.PRIM_ELLIPSOID, rt.PRIM_BOX]
    prims["position"] = [[0, -0.95, 0], [-0.45, -0.55, 0.3], [0.5, 0.2, -0.3], [0.35, -0.6, 0.45], [-0.5, 0.45, -0.5],
                         [0.0, 0.55, 0.2], [0.0, -0.2, -0.6]]
    prims["rotation"] = [[0, 0, 0, 1]] + [unit_quat() for _ in range(6)]
    prims["param"] = [[0.02, 1.0, -0.03], [0.3, 0.2, 0.25], [0.18, 0.3, 0.15], [0.2, 0.15, 0.1], [0.12, 0.25, 0.2],
                      [0.15, 0.15, 0.15], [0.3, 0.05, 0.2]]
    prims["material"] = [mat_ids[k % len(mat_ids)] for k in range(7)]
    return prims


def test_analytic_primitives(scene_dir, oracle_mod):
    """Plane / Ellipsoid / Box (main.cpp:242-464; no loader path of the reference constructs them, so the
    target is the oracle's restatement of their intersectPrimitive): inside the Cornell box with diffuse,
    metallic and dielectric materials.  Closest-hit ids (ntri + k) and distances bit for bit for camera rays
    and for random rays (many start inside an ellipsoid or a box); reference-stream render bit for bit; both
    throughput pipelines within 1e-5 with equal segment counts.  Then the same primitives alone under a sky
    (no triangles at all)."""
    import raytracing_amd as rt
    W, H = 128, 96
    base = rt.load_gltf(make_scene(scene_dir, "cornell"), W, H)
    types = base.materials["type"]
    emits = np.linalg.norm(base.materials["emission"], axis=1) > 1e-4
    pick = [int(np.flatnonzero((types == t) & ~emits)[0]) for t in (0, 2, 1)]        # diffuse, glass, metal
    rng = np.random.RandomState(42)
    prims = _analytic_prims(rt, rng, pick)
    for with_tris in (True, False):
        if with_tris:
            host = rt.HostScene(base.tri_verts, base.tri_material, base.materials, base.camera, prims=prims)
        else:
            host = rt.HostScene(np.zeros((0, 3, 3), np.float32), np.zeros(0, np.uint32), base.materials, base.camera,
                                background=(0.6, 0.7, 0.9), prims=prims)
        gpu = rt.Scene(host, device=0)
        d = host.as_dict()
        # the reference's structure: planes tested at the root, ellipsoids and boxes in the tree under their own
        # (for a box: half-size, doubled again by the slabs) boxes, main.cpp:552-558, 358-364, 457-463
        orc_in = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"], d["background"], d["prims"],
                                        prims_in_tree=True)
        assert gpu.stats()["hit_order"] == rt.HITS_REFERENCE_ORDER and gpu.stats()["ref_tree_nodes"] == orc_in.info()["nodes"]
        ids_o, t_o, _, _ = orc_in.trace_primary(W, H)
        ids_g, t_g = gpu.trace_primary(W, H)
        _compare_hits(ids_g, t_g, ids_o, t_o, None)
        n = 40000
        o = rng.uniform(-0.9, 0.9, size=(n, 3)).astype(np.float32)
        o[: n // 4] = (prims["position"][rng.randint(1, 7, n // 4)] + rng.normal(size=(n // 4, 3)) * 0.05).astype(np.float32)
        dirs = rng.normal(size=(n, 3))
        dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
        ids_o, t_o, _, _ = orc_in.intersect(o, dirs, use_bvh=True)
        ids_g, t_g = gpu.intersect(o, dirs)
        _compare_hits(ids_g, t_g, ids_o, t_o, None)
        ref0, c0 = orc_in.render(W, H, 3, max_depth=8, rng_mode=0)
        gpu.reset_counters()
        _assert_bits_equal_nan(gpu.render(W, H, 3, max_depth=8, rng_mode=0), ref0)
        assert gpu.counters()["segments"] == c0["segments"]
        # the throughput pipelines in reference order: fast walk, ties and beyond-INF plane hits re-traced
        ref, c_o = orc_in.render(W, H, 4, max_depth=8, rng_mode=1, seed=6, spp_begin=1)
        for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
            gpu.reset_counters()
            _assert_close(gpu.render(W, H, 4, max_depth=8, rng_mode=1, seed=6, spp_begin=1, pipeline=pipe), ref)
            assert gpu.counters()["segments"] == c_o["segments"]
        orc_in.close()
        # the kernels' own structure: primitives folded in after the walk of the triangle tree
        gpu.set_hit_order(rt.HITS_LEAF_ORDER)
        orc = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"], d["background"], d["prims"])
        ntri = host.tri_material.shape[0]
        assert gpu.stats()["nprims"] == 7
        ids_o, t_o, tie, _ = orc.trace_primary(W, H)
        ids_g, t_g = gpu.trace_primary(W, H)
        _compare_hits(ids_g, t_g, ids_o, t_o, tie)
        seen = set(np.unique(ids_g[ids_g != 0xFFFFFFFF]) - ntri) & set(range(7))
        assert len(seen) >= 5, seen                                                  # the camera sees most of them
        for use_bvh in (True, False):
            ids_o, t_o, tie, _ = orc.intersect(o, dirs, use_bvh=use_bvh)
            ids_g, t_g = gpu.intersect(o, dirs)
            _compare_hits(ids_g, t_g, ids_o, t_o, tie)
        assert (ids_g >= ntri)[ids_g != 0xFFFFFFFF].mean() > 0.2
        ref0, c0 = orc.render(W, H, 3, max_depth=8, rng_mode=0)
        gpu.reset_counters()
        _assert_bits_equal(gpu.render(W, H, 3, max_depth=8, rng_mode=0), ref0)
        assert gpu.counters()["segments"] == c0["segments"] and ref0.sum() > 0
        ref, c_o = orc.render(W, H, 4, max_depth=8, rng_mode=1, seed=6, spp_begin=1)
        for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
            gpu.reset_counters()
            _assert_close(gpu.render(W, H, 4, max_depth=8, rng_mode=1, seed=6, spp_begin=1, pipeline=pipe), ref)
            assert gpu.counters()["segments"] == c_o["segments"]
        gpu.close()
        orc.close()


@pytest.mark.parametrize("with_tris,light_tree", [(True, False), (False, False), (True, True)])
def test_emissive_analytic_primitives(scene_dir, oracle_mod, monkeypatch, with_tris, light_tree):
    """Ellipsoids and boxes that emit (main.cpp:218-239, 320-356, 425-455: samplePoint, pdf, intersectFull,
    pdfFull; dead code in the reference, whose loader constructs triangles only, so the target is the oracle's
    restatement, bug for bug: the ellipsoid's second density taken at the first intersection point, a missed
    second intersection dividing by zero).  They join the light list behind the light triangles.  Inside the
    Cornell box (whose lamp stays) and alone in the dark: reference-stream render bit for bit in both hit orders
    with equal segment and light-test counts; both throughput pipelines within 1e-5 with equal segment counts; the
    analytic lights really light the scene."""
    import raytracing_amd as rt
    W, H = 112, 84
    if light_tree:      # the light triangles' share of the density through the light tree, the primitives' share behind it
        monkeypatch.setenv("RT_LIGHT_TREE_MIN", "1")
    base = rt.load_gltf(make_scene(scene_dir, "cornell"), W, H)
    mats = np.concatenate([base.materials, base.materials[:2]])
    glow_a, glow_b = len(base.materials), len(base.materials) + 1
    mats["type"][glow_a:], mats["color"][glow_a:] = 0, 0.8
    mats["emission"][glow_a], mats["emission"][glow_b] = (9.0, 6.0, 3.0), (2.0, 5.0, 8.0)
    types = base.materials["type"]
    emits = np.linalg.norm(base.materials["emission"], axis=1) > 1e-4
    diffuse, glass, metal = [int(np.flatnonzero((types == t) & ~emits)[0]) for t in (0, 2, 1)]
    rng = np.random.RandomState(12)
    prims = _analytic_prims(rt, rng, [diffuse, glass, metal])
    prims["material"][1], prims["material"][3], prims["material"][5] = glow_a, glow_b, glow_b   # ellipsoid, box, ellipsoid
    if with_tris:
        host = rt.HostScene(base.tri_verts, base.tri_material, mats, base.camera, prims=prims)
    else:
        host = rt.HostScene(np.zeros((0, 3, 3), np.float32), np.zeros(0, np.uint32), mats, base.camera, prims=prims)
    gpu = rt.Scene(host, device=0)
    d = host.as_dict()
    ntri = host.tri_material.shape[0]
    assert gpu.stats()["nlights"] == (2 if with_tris else 0) + 3
    for order, in_tree in ((rt.HITS_REFERENCE_ORDER, True), (rt.HITS_LEAF_ORDER, False)):
        gpu.set_hit_order(order)
        orc = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"], d["background"], d["prims"],
                                     prims_in_tree=in_tree)
        assert orc.lights().tolist()[-3:] == [ntri + 1, ntri + 3, ntri + 5]
        ref0, c0 = orc.render(W, H, 4, max_depth=6, rng_mode=0)
        gpu.reset_counters()
        _assert_bits_equal_nan(gpu.render(W, H, 4, max_depth=6, rng_mode=0), ref0)
        c = gpu.counters()
        assert c["segments"] == c0["segments"]
        assert c["light_tests"] == c0["light_tests"] or light_tree        # (the tree walk tests fewer light triangles)
        assert np.nansum(ref0) > 0
        # the throughput pipelines: in reference order their fast walk hands ties and beyond-INF plane hits to the
        # reference's walk, so they follow the oracle's tree placement too
        ref, c_o = orc.render(W, H, 5, max_depth=6, rng_mode=1, seed=8, spp_begin=2)
        for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
            gpu.reset_counters()
            _assert_close(gpu.render(W, H, 5, max_depth=6, rng_mode=1, seed=8, spp_begin=2, pipeline=pipe), ref, max_nan_pixels=8)
            assert gpu.counters()["segments"] == c_o["segments"]
        orc.close()
    if with_tris:   # against the same scene with the primitives dark
        dark = mats.copy()
        dark["emission"][glow_a:] = 0
        plain, _ = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], dark, d["camera"], d["background"], d["prims"]).render(
            W, H, 4, max_depth=6, rng_mode=0)
        assert np.nanmean(ref0) > 1.2 * np.nanmean(plain)
    gpu.close()


def test_analytic_primitive_ids_with_pad_records(scene_dir, oracle_mod):
    """A scene whose triangle array holds pad records (two-triangle leaves start on even records), plus analytic
    primitives: on the device a primitive's index lies behind the RECORDS, the caller must still see it as
    (loader triangle count) + k, and triangles as their loader ids.  Hit ids and distances bit for bit against
    the oracle for camera rays and random rays; a counter-mode render on both pipelines within 1e-5."""
    import raytracing_amd as rt
    W, H = 96, 72
    base = rt.load_gltf(make_scene(scene_dir, "spheres1m", n_spheres=10, subdiv=2, ground=8), W, H)
    types = base.materials["type"]
    emits = np.linalg.norm(base.materials["emission"], axis=1) > 1e-4
    pick = [int(np.flatnonzero((types == t) & ~emits)[0]) for t in (0, 2, 1)]
    rng = np.random.RandomState(7)
    prims = _analytic_prims(rt, rng, pick)
    host = rt.HostScene(base.tri_verts, base.tri_material, base.materials, base.camera, prims=prims)
    gpu = rt.Scene(host, device=0)
    gpu.set_hit_order(rt.HITS_LEAF_ORDER)
    d = host.as_dict()
    orc = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"], d["background"], d["prims"])
    ntri = host.tri_material.shape[0]
    st = gpu.stats()
    assert st["ntri"] == ntri and st["ntri_records"] > ntri and st["nprims"] == 7
    ids_o, t_o, tie, _ = orc.trace_primary(W, H)
    ids_g, t_g = gpu.trace_primary(W, H)
    _compare_hits(ids_g, t_g, ids_o, t_o, tie)
    n = 30000
    o = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
    dirs = rng.normal(size=(n, 3))
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    ids_o, t_o, tie, _ = orc.intersect(o, dirs, use_bvh=True)
    ids_g, t_g = gpu.intersect(o, dirs)
    _compare_hits(ids_g, t_g, ids_o, t_o, tie)
    hit = ids_g != 0xFFFFFFFF
    assert (ids_g[hit] < ntri + 7).all() and (ids_g[hit] >= ntri).any() and (ids_g[hit] < ntri).any()
    ref, c_o = orc.render(W, H, 3, max_depth=6, rng_mode=1, seed=9)
    for pipe in (rt.PIPE_MEGAKERNEL, rt.PIPE_WAVEFRONT):
        gpu.reset_counters()
        _assert_close(gpu.render(W, H, 3, max_depth=6, rng_mode=1, seed=9, pipeline=pipe), ref)
        assert gpu.counters()["segments"] == c_o["segments"]
    gpu.close()
    orc.close()
