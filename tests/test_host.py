"""CPU tests of the host side of the product: the C ABI library loads and exports what the header
declares, the C++ scene loader agrees bit for bit with the numpy restatement of the reference
loader, the BVH builder produces a valid flattened tree, error paths fail loudly.  No GPU."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_scene


def test_library_exports_every_declared_symbol():
    from raytracing_amd import _lib
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), "librt_amd.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), "python binding and header disagree: %s" % (declared ^ set(_lib.SIGNATURES))


def test_no_product_code_touches_the_oracle():
    """The product path must never route through the CPU oracle."""
    for base in ("raytracing_amd",):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or fn == "Makefile":
                    text = open(os.path.join(dirpath, fn)).read()
                    assert "oracle" not in text.lower(), "%s mentions the oracle" % os.path.join(dirpath, fn)


@pytest.mark.parametrize("name,kw", [("cornell", {}), ("heightfield100k", {"nx": 300, "nz": 120}),
                                     ("spheres1m", {"n_spheres": 30, "subdiv": 2, "ground": 8}),
                                     ("glass", {"grid": 3, "subdiv": 2})])
def test_loader_matches_reference_restatement(scene_dir, name, kw):
    import raytracing_amd as rt
    from oracle import gltf_ref
    path = make_scene(scene_dir, name, **kw)
    a = rt.load_gltf(path, 1920, 1080).as_dict()
    b = gltf_ref.parse_gltf(path, 1920, 1080)
    assert a["tri_verts"].shape == b["tri_verts"].shape and a["tri_verts"].shape[0] > 0
    assert a["tri_verts"].tobytes() == b["tri_verts"].tobytes()
    assert np.array_equal(a["tri_material"], b["tri_material"])
    assert a["materials"].tobytes() == b["materials"].tobytes()
    for k in a["camera"]:
        assert np.array_equal(np.asarray(a["camera"][k]).view(np.uint32), np.asarray(b["camera"][k]).view(np.uint32)), k


def test_loader_quirks(scene_dir):
    """matrix nodes, nested children composed in index order, ignored accessor.byteOffset on indices,
    default material type, JSON integers where floats are expected."""
    import raytracing_amd as rt
    from oracle import gltf_ref
    from raytracing_amd.scenes import GltfBuilder, DIFFUSER, quad
    g = GltfBuilder()
    m0 = g.material(DIFFUSER, (1, 0, 0))
    g.materials.append({})                                        # no pbr block -> METALLIC, white
    g.materials.append({"pbrMetallicRoughness": {"metallicFactor": 1, "baseColorFactor": [0, 1, 0, 0.25]},
                        "emissiveFactor": [1, 2, 3]})             # alpha < 1 -> DIELECTRIC; emissive without strength
    pos, idx = quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0))
    mesh = g.mesh([(pos, idx, m0), (pos + 2, idx, 1), (pos - 2, idx, 2)])
    g.node(children=[1], translation=(1, 2, 3), scale=(2, 2, 2))                       # 0
    g.node(children=[2], rotation=(0, 0.38268343, 0, 0.92387953))                       # 1
    g.node(mesh=mesh, matrix=[1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 5, 6, 7, 1])        # 2: column-major matrix
    g.node(mesh=mesh)                                                                   # 3: untransformed copy
    cam = g.camera(1)
    g.node(camera=cam, translation=(0, 0, 10), rotation=(0.1, 0.2, 0.3, 0.9273618), scale=(1, 2, 1))
    g.accessors[0]["byteOffset"] = 0
    path = g.write(os.path.join(scene_dir, "quirks.gltf"))
    a = rt.load_gltf(path, 320, 200).as_dict()
    b = gltf_ref.parse_gltf(path, 320, 200)
    assert a["tri_verts"].shape == (12, 3, 3)
    assert a["tri_verts"].tobytes() == b["tri_verts"].tobytes()
    assert a["materials"].tobytes() == b["materials"].tobytes()
    assert list(a["materials"]["type"]) == [0, 1, 2]
    assert np.array_equal(a["materials"]["emission"][2], np.array([1, 2, 3], np.float32))
    for k in a["camera"]:
        assert np.array_equal(np.asarray(a["camera"][k]), np.asarray(b["camera"][k])), k
    # the un-transformed instance is the raw quad; the nested one went through three matrices
    assert np.array_equal(a["tri_verts"][6], pos[[0, 1, 2]])
    assert not np.array_equal(a["tri_verts"][0], pos[[0, 1, 2]])


def test_loader_errors(scene_dir, tmp_path):
    import raytracing_amd as rt
    with pytest.raises(rt.RtError) as e:
        rt.load_gltf(os.path.join(scene_dir, "does_not_exist.gltf"), 64, 64)
    assert e.value.code == -2
    bad = tmp_path / "bad.gltf"
    bad.write_text("{ not json")
    with pytest.raises(rt.RtError) as e:
        rt.load_gltf(str(bad), 64, 64)
    assert e.value.code == -1
    path = make_scene(scene_dir, "cornell")
    doc = json.load(open(path))
    del doc["bufferViews"][0]["byteOffset"]            # undefined behaviour in the reference; an error here
    p2 = os.path.join(scene_dir, "cornell_nobo.gltf")
    json.dump(doc, open(p2, "w"))
    os.replace(os.path.join(scene_dir, "cornell.bin"), os.path.join(scene_dir, "cornell.bin"))
    with pytest.raises(rt.RtError) as e:
        rt.load_gltf(p2, 64, 64)
    assert "byteOffset" in str(e.value)
    doc = json.load(open(path))
    doc["accessors"][0]["componentType"] = 5121        # u8 indices are not supported (main.cpp:1077)
    p3 = os.path.join(scene_dir, "cornell_u8.gltf")
    json.dump(doc, open(p3, "w"))
    with pytest.raises(rt.RtError) as e:
        rt.load_gltf(p3, 64, 64)
    assert "5123 or 5125" in str(e.value)
    with pytest.raises(rt.RtError):
        rt.load_gltf(path, 0, 64)


def _check_flat(flat, tri_verts, tri_material, materials):
    nodes, tris = flat["nodes"], flat["tris"]
    n = tri_verts.shape[0]
    ids = tris[:, 12].view(np.uint32)
    pad = ids == 0xFFFFFFFF                                            # records no leaf names (RT_MISS_ID): never hit
    assert (tris[pad, 0:3] == np.float32(3.0e30)).all() and pad.sum() <= n // 2 + 1
    real = np.flatnonzero(~pad)
    assert sorted(ids[real].tolist()) == list(range(n))                # a permutation of the loader order
    assert np.array_equal(tris[real, 0:3], tri_verts[ids[real], 0])             # A
    assert np.array_equal(tris[real, 3:6], tri_verts[ids[real], 1] - tri_verts[ids[real], 0])   # AB
    assert np.array_equal(tris[real, 6:9], tri_verts[ids[real], 2] - tri_verts[ids[real], 0])   # AC
    assert np.array_equal(tris[real, 13].view(np.uint32), tri_material[ids[real]])
    import raytracing_amd as rt
    refs = nodes[:, 6:8].view(np.int32)
    boxes = rt.api.node_boxes(nodes, flat["grid"])                      # (nn, 2, 6): lo xyz, hi xyz
    absent = (nodes[:, :6].reshape(-1, 2, 3) == 0xFFFFFFFF).all(axis=2)
    cell = flat["grid"][3:].astype(np.float64)
    assert (cell > 0).all()
    tmin = tri_verts.min(axis=1)
    tmax = tri_verts.max(axis=1)
    seen = np.zeros(tris.shape[0], dtype=np.int32)
    visited = np.zeros(nodes.shape[0], dtype=np.int32)

    def walk(node, lo, hi, level):
        visited[node] += 1
        h = level
        for side in range(2):
            ref, b = refs[node, side], boxes[node, side]
            if absent[node, side]:
                continue
            assert ((b[3:] - b[:3]) >= 3.999 * cell).all()             # two guard cells per side: never flat
            if lo is not None:                                         # inside the parent up to the outward rounding
                assert (b[:3] >= lo - 4 * cell).all() and (b[3:] <= hi + 4 * cell).all()
            if ref < 0:
                r = ~int(ref)
                first, cnt = r >> 3, (r & 7) + 1
                assert cnt <= 2
                assert cnt == 1 or first % 2 == 0 or n <= 2            # a pair lies in one 128-byte line
                for k in range(first, first + cnt):
                    seen[k] += 1
                    assert (tmin[ids[k]] >= b[:3]).all() and (tmax[ids[k]] <= b[3:]).all()
            else:
                h = max(h, walk(int(ref), b[:3], b[3:], level + 1))
        return h

    height = walk(0, None, None, 1)
    assert (seen[real] == 1).all() and (seen[pad] == 0).all() and (visited == 1).all()
    assert height == flat["height"] <= 24
    em = np.linalg.norm(materials["emission"][tri_material], axis=1) > 1e-4
    assert np.array_equal(ids[flat["lights"]], np.flatnonzero(em))     # loader order, leaf positions

    # the tree over the light triangles: every light in exactly one leaf, inside its box, records = the triangles'
    lt = flat["light_tree"]
    lights = flat["lights"]
    nl = lights.size
    recs = lt["recs"]
    assert sorted(recs[:, 12].view(np.uint32).tolist()) == list(range(nl))
    assert np.array_equal(recs[:, 13].view(np.uint32), lights[recs[:, 12].view(np.uint32)])
    assert np.array_equal(recs[:, :12], tris[recs[:, 13].view(np.uint32), :12]) and np.array_equal(recs[:, 14], tris[recs[:, 13].view(np.uint32), 14])
    lrefs = lt["nodes"][:, 6:8].view(np.int32)
    lboxes = rt.api.node_boxes(lt["nodes"], lt["grid"])
    labsent = (lt["nodes"][:, :6].reshape(-1, 2, 3) == 0xFFFFFFFF).all(axis=2)
    lseen = np.zeros(max(nl, 1), dtype=np.int32)

    def lwalk(node, level):
        h = level
        for side in range(2):
            ref, b = lrefs[node, side], lboxes[node, side]
            if labsent[node, side]:
                continue
            if ref < 0:
                r = ~int(ref)
                for k in range(r >> 3, (r >> 3) + (r & 7) + 1):
                    lseen[k] += 1
                    tid = ids[int(recs[k, 13].view(np.uint32))]
                    assert (tmin[tid] >= b[:3]).all() and (tmax[tid] <= b[3:]).all()
            else:
                h = max(h, lwalk(int(ref), level + 1))
        return h

    if nl:
        assert lwalk(0, 1) == lt["height"] <= 24 and (lseen[:nl] == 1).all()

    # the four-child tree of the wavefront trace kernel: the same leaves, each exactly once; 2 to 4 children per
    # node; every child box (8 bits per plane in the node's frame) contains the two-child tree's box of the
    # same subtree -- checked against the box of every leaf below it, which the two-child tree was checked
    # to hold the triangles in -- and is at most one frame step per side larger than the union of those
    wide = flat["wide"]
    grid = flat["grid"].astype(np.float64)
    leafbox = {}
    for node in range(nodes.shape[0]):
        for side in range(2):
            if refs[node, side] < 0 and not absent[node, side]:
                leafbox[int(refs[node, side])] = boxes[node, side].astype(np.float64)
    wseen = {}

    def wwalk(node):
        w = wide[node]
        org = np.array([w[6] & 0xffff, w[6] >> 16, w[7] & 0xffff], dtype=np.float64)
        sh = np.array([(w[7] >> 16) & 15, (w[7] >> 20) & 15, (w[7] >> 24) & 15])
        nch = int(w[7] >> 28)
        assert 2 <= nch <= 4
        step = (2.0 ** sh) * grid[3:]
        lo_all, hi_all = np.full(3, np.inf), np.full(3, -np.inf)
        for c in range(4):
            ref = int(np.int32(w[8 + c]))
            if c >= nch:
                assert ref == 0x7FFFFFFF
                continue
            ql = np.array([(w[0] >> 8 * c) & 255, (w[2] >> 8 * c) & 255, (w[4] >> 8 * c) & 255], dtype=np.float64)
            qh = np.array([(w[1] >> 8 * c) & 255, (w[3] >> 8 * c) & 255, (w[5] >> 8 * c) & 255], dtype=np.float64)
            assert (ql <= qh).all()
            lo = grid[:3] + (org + ql * 2.0 ** sh) * grid[3:]
            hi = grid[:3] + (org + qh * 2.0 ** sh) * grid[3:]
            if ref < 0:
                wseen[ref] = wseen.get(ref, 0) + 1
                blo, bhi = leafbox[ref][:3], leafbox[ref][3:]
            else:
                blo, bhi = wwalk(ref)
            tol = 1e-5 * np.maximum(np.abs(blo), np.abs(bhi)) + 1e-9       # float32 grid constants, float64 arithmetic here
            assert (lo <= blo + tol).all() and (hi >= bhi - tol).all()
            assert (lo >= blo - step - tol).all() and (hi <= bhi + step + tol).all()
            lo_all, hi_all = np.minimum(lo_all, blo), np.maximum(hi_all, bhi)
        return lo_all, hi_all

    if n > 2:
        wwalk(0)
        assert sorted(wseen) == sorted(leafbox) and set(wseen.values()) == {1}
        assert wide.shape[0] <= nodes.shape[0]


@pytest.mark.parametrize("name,kw", [("cornell", {}), ("emissive", {"subdiv": 2, "lamps": 2}), ("heightfield100k", {"nx": 40, "nz": 40}),
                                     ("spheres1m", {"n_spheres": 10, "subdiv": 2, "ground": 8})])
def test_flat_bvh_is_valid(scene_dir, name, kw):
    import sys
    import raytracing_amd as rt
    sys.setrecursionlimit(10000)
    host = rt.load_gltf(make_scene(scene_dir, name, **kw), 64, 64)
    _check_flat(rt.build_flat(host), host.tri_verts, host.tri_material, host.materials)


def test_flat_bvh_degenerate_inputs():
    import raytracing_amd as rt
    mats = np.zeros(1, dtype=rt.api.MATERIAL_DTYPE)
    cam = dict(position=[0, 0, 0], right=[1, 0, 0], up=[0, 1, 0], forward=[0, 0, -1], tan_x=1, tan_y=1)
    empty = rt.HostScene(np.zeros((0, 3, 3), np.float32), np.zeros(0, np.uint32), mats, cam)
    f = rt.build_flat(empty)
    assert f["nodes"].shape == (1, 8) and f["tris"].shape[0] == 0
    # 3000 copies of the same triangle: no SAH split exists, the builder must still terminate and bound the height
    tri = np.tile(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32), (3000, 1, 1))
    same = rt.HostScene(tri, np.zeros(3000, np.uint32), mats, cam)
    _check_flat(rt.build_flat(same), same.tri_verts, same.tri_material, same.materials)
    bad = rt.HostScene(tri[:2], np.array([0, 7], np.uint32), mats, cam)
    with pytest.raises(rt.RtError):
        rt.build_flat(bad)
    nan = tri[:2].copy()
    nan[0, 0, 0] = np.nan
    with pytest.raises(rt.RtError):
        rt.build_flat(rt.HostScene(nan, np.zeros(2, np.uint32), mats, cam))


def test_ppm_writer(tmp_path):
    """printImage (main.cpp:856-864): "P6\\nW H\\n255\\n" + RGB bytes."""
    import raytracing_amd as rt
    rgb = (np.arange(5 * 3 * 3) % 256).astype(np.uint8).reshape(-1, 3)
    out = tmp_path / "x.ppm"
    rt.print_image(rt.SceneFloat(5, 3, None, rgb), str(out))
    data = out.read_bytes()
    assert data.startswith(b"P6\n5 3\n255\n") and data[len(b"P6\n5 3\n255\n"):] == rgb.tobytes()
    with pytest.raises(rt.RtError):
        rt.print_image(rt.SceneFloat(5, 3, None, rgb), str(tmp_path / "no_dir" / "x.ppm"))


def test_counter_seed_matches_oracle(oracle_mod):
    import raytracing_amd as rt
    rng = np.random.RandomState(1)
    for p, s, sd in rng.randint(0, 2 ** 31, size=(200, 3)):
        assert rt.counter_seed(int(p), int(s), int(sd)) == oracle_mod.counter_seed(int(p), int(s), int(sd))


def test_sample_range_partition():
    from raytracing_amd.dist import sample_range
    for total in (0, 1, 7, 128, 1024, 4097):
        for world in (1, 2, 3, 8):
            parts = [sample_range(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b0 + c0 == b1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        sample_range(8, 2, 2)


def test_cli_usage_and_gpu_requirement(scene_dir):
    exe = os.path.join(ROOT, "raytracing_amd", "bin", "raytracing")
    assert os.path.exists(exe), "build the host with make -C raytracing_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([exe, make_scene(scene_dir, "cornell"), "0", "10", "1", "/tmp/x.ppm"], capture_output=True, text=True)
    assert r.returncode == 2
    import raytracing_amd as rt
    if rt.device_count() == 0:  # no silent CPU fallback: without a GPU the host must fail loudly
        r = subprocess.run([exe, make_scene(scene_dir, "cornell"), "32", "24", "1", os.path.join(scene_dir, "o.ppm")],
                           capture_output=True, text=True)
        assert r.returncode != 0 and "no HIP device" in r.stderr
        with pytest.raises(rt.RtError) as e:
            rt.Scene(rt.load_gltf(make_scene(scene_dir, "cornell"), 32, 24))
        assert e.value.code == -3


def _write_two_triangle_scene(tmp_path, variant):
    """Two triangles under a parent/child node pair and a camera under a translated parent, laid out
    either the way the reference's loader reads files ("plain") or with features only the opt-in
    flags understand: interleaved positions (byteStride 24), two index accessors sharing a bufferView
    through accessor.byteOffset, a base64 buffer, children listed before their parents."""
    import base64
    import json
    import struct
    pos = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0.5)]
    idx_a, idx_b = (0, 1, 2), (1, 3, 2)
    if variant == "plain":
        blob = b"".join(struct.pack("<3f", *p) for p in pos) + struct.pack("<3H", *idx_a) + b"\0\0" + struct.pack("<3H", *idx_b) + b"\0\0"
        views = [dict(buffer=0, byteOffset=0, byteLength=48), dict(buffer=0, byteOffset=48, byteLength=6),
                 dict(buffer=0, byteOffset=56, byteLength=6)]
        accessors = [dict(bufferView=0, componentType=5126, count=4, type="VEC3"),
                     dict(bufferView=1, componentType=5123, count=3, type="SCALAR"),
                     dict(bufferView=2, componentType=5123, count=3, type="SCALAR")]
    else:
        blob = b"".join(struct.pack("<6f", *p, 0.0, 0.0, 1.0) for p in pos) + struct.pack("<3H", *idx_a) + b"\0\0" + struct.pack("<3H", *idx_b) + b"\0\0"
        views = [dict(buffer=0, byteLength=96, byteStride=24), dict(buffer=0, byteOffset=96, byteLength=16)]
        accessors = [dict(bufferView=0, componentType=5126, count=4, type="VEC3"),
                     dict(bufferView=1, componentType=5123, count=3, type="SCALAR"),
                     dict(bufferView=1, byteOffset=8, componentType=5123, count=3, type="SCALAR")]
    mat = dict(pbrMetallicRoughness=dict(metallicFactor=0.0, baseColorFactor=[0.5, 0.5, 0.5, 1.0]))
    meshes = [dict(primitives=[dict(attributes=dict(POSITION=0), indices=1, material=0)]),
              dict(primitives=[dict(attributes=dict(POSITION=0), indices=2, material=0)])]
    parent = dict(translation=[1.0, 2.0, 3.0], scale=[2.0, 2.0, 2.0])
    child_a = dict(mesh=0, translation=[0.5, 0.0, 0.0])
    child_b = dict(mesh=1, rotation=[0.0, 0.0, 0.70710678, 0.70710678])
    rig = dict(translation=[0.0, 1.0, 5.0])
    cam = dict(camera=0, translation=[0.0, 0.0, 1.0])
    if variant == "plain":      # parents before children, as the reference needs them
        nodes = [dict(parent, children=[1, 2]), child_a, child_b, dict(rig, children=[4]), cam]
    else:                       # children first
        nodes = [child_a, child_b, dict(parent, children=[0, 1]), cam, dict(rig, children=[3])]
    uri = "two.bin"
    if variant == "spec":
        uri = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    else:
        (tmp_path / "two.bin").write_bytes(blob)
    doc = dict(asset=dict(version="2.0"), buffers=[dict(uri=uri, byteLength=len(blob))], bufferViews=views,
               accessors=accessors, materials=[mat], meshes=meshes, nodes=nodes,
               cameras=[dict(type="perspective", perspective=dict(yfov=0.8, znear=0.1))])
    path = tmp_path / ("two_%s.gltf" % variant)
    path.write_text(json.dumps(doc))
    return str(path)


def test_gltf_spec_flags(tmp_path):
    """rt_gltf_load_ex: the opt-in glTF 2.0 rules (stride, index accessor offset, data URI, hierarchy in
    any node order, camera under a parent).  The same geometry written the reference's way and the
    specification's way must load to the same triangles; without the flags the second file is read
    by the reference's rules and comes out different (or is rejected)."""
    import raytracing_amd as rt
    plain = rt.load_gltf(_write_two_triangle_scene(tmp_path, "plain"), 64, 48)
    spec_path = _write_two_triangle_scene(tmp_path, "spec")
    spec = rt.load_gltf(spec_path, 64, 48, flags=rt.api.GLTF_SPEC)
    assert plain.tri_verts.shape == (2, 3, 3)
    order = np.argsort(spec.tri_verts[:, 0, 0] * 7 + spec.tri_verts[:, 1, 1])
    order_p = np.argsort(plain.tri_verts[:, 0, 0] * 7 + plain.tri_verts[:, 1, 1])
    assert np.array_equal(spec.tri_verts[order], plain.tri_verts[order_p])       # same transforms, same bits
    # parent (T, S=2) * child translation (0.5, 0, 0): vertex (0,0,0) of mesh 0 lands at (2, 2, 3)
    assert any(np.allclose(v, [2.0, 2.0, 3.0]) for v in plain.tri_verts.reshape(-1, 3))
    # the reference's rules put the camera at its local translation; the hierarchy flag adds the rig's
    assert np.allclose(plain.camera["position"], [0.0, 0.0, 1.0])
    assert np.allclose(spec.camera["position"], [0.0, 1.0, 6.0])
    same_rules = rt.load_gltf(_write_two_triangle_scene(tmp_path, "plain"), 64, 48, flags=rt.api.GLTF_SPEC)
    assert np.array_equal(same_rules.tri_verts, plain.tri_verts)                 # flags do not disturb a plain file
    assert np.allclose(same_rules.camera["position"], [0.0, 1.0, 6.0])
    with pytest.raises(rt.RtError):                                             # reference rules: byteOffset is required
        rt.load_gltf(spec_path, 64, 48)
    with pytest.raises(rt.RtError):
        rt.load_gltf(spec_path, 64, 48, flags=0x100)                             # unknown flag


def test_restated_libm_kernels_match_the_host_libm(tmp_path):
    """raytracing_amd/csrc/rt_libm.h (glibc's logf and powf kernels as the device evaluates them) against
    this host's libm, bit for bit, on every 499th float of [0, 1] (the exhaustive run, stride 1, takes
    half a minute: `tools/check_libm.cpp`).  logf decides the Gaussian samples' accept/reject
    branches (main.cpp:108-111), powf the gamma curve's image bytes (main.cpp:1228-1231)."""
    import subprocess
    exe = str(tmp_path / "check_libm")
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-ffp-contract=off", os.path.join(ROOT, "tools", "check_libm.cpp"),
                           "-o", exe])
    r = subprocess.run([exe, "499"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "logf mismatches 0" in r.stdout and "mismatches 0\n" in r.stdout


def test_multi_gpu_sample_split_matches_the_python_one():
    """rt_multi_sample_range (the C ABI's split of a frame's samples over the GPUs of one process) and
    raytracing_amd.dist.sample_range (one process per GPU) are the same partition: contiguous, complete,
    the first spp % n parts one sample longer."""
    import raytracing_amd as rt
    from raytracing_amd.dist import sample_range
    for spp in (0, 1, 3, 5, 128, 1024, 4097):
        for n in (1, 2, 3, 4, 8):
            parts = [rt.multi_sample_range(spp, n, g) for g in range(n)]
            assert parts == [sample_range(spp, n, g) for g in range(n)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == spp
            assert all(parts[g][0] + parts[g][1] == parts[g + 1][0] for g in range(n - 1))


def _to_glb(gltf_path, out_path, drop_uri=True):
    """Packs a .gltf + its single external .bin into a binary container (glTF 2.0, section 4.4)."""
    import struct
    doc = json.load(open(gltf_path))
    blob = open(os.path.join(os.path.dirname(gltf_path), doc["buffers"][0]["uri"]), "rb").read()
    if drop_uri:
        del doc["buffers"][0]["uri"]
    text = json.dumps(doc).encode()
    text += b" " * (-len(text) % 4)
    blob += b"\0" * (-len(blob) % 4)
    total = 12 + 8 + len(text) + 8 + len(blob)
    with open(out_path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<II", len(text), 0x4E4F534A) + text)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)
    return out_path


def test_gltf_glb_ior_background_flags(scene_dir, tmp_path):
    """The remaining opt-in loader flags: `.glb` containers, KHR_materials_ior, a background colour.  Each is
    off by default (the reference reads text files with external buffers, fixes ior = 1.5 and never sets its
    backgroundColor: main.cpp:869-882, 65, 694), and none disturbs a file that does not use it."""
    import raytracing_amd as rt
    A = rt.api
    path = make_scene(scene_dir, "cornell")
    plain = rt.load_gltf(path, 64, 48)
    assert (plain.materials["ior"] == np.float32(1.5)).all() and not plain.background.any()
    # .glb: same triangles, materials, camera
    glb = _to_glb(path, str(tmp_path / "cornell.glb"))
    packed = rt.load_gltf(glb, 64, 48, flags=A.GLTF_GLB)
    assert np.array_equal(packed.tri_verts, plain.tri_verts) and np.array_equal(packed.tri_material, plain.tri_material)
    assert packed.materials.tobytes() == plain.materials.tobytes()
    assert all(np.array_equal(packed.camera[k], plain.camera[k]) for k in plain.camera)
    with pytest.raises(rt.RtError):
        rt.load_gltf(glb, 64, 48)                                  # the reference's rules: not a JSON text file
    for cut in (10, 40, os.path.getsize(glb) - 7):                 # truncated containers are rejected, not read past
        open(str(tmp_path / "cut.glb"), "wb").write(open(glb, "rb").read()[:cut])
        with pytest.raises(rt.RtError):
            rt.load_gltf(str(tmp_path / "cut.glb"), 64, 48, flags=A.GLTF_GLB)
    assert np.array_equal(rt.load_gltf(path, 64, 48, flags=A.GLTF_ALL).tri_verts, plain.tri_verts)
    # KHR_materials_ior and extras.backgroundColor in a copy of the text file
    doc = json.load(open(path))
    glass = [i for i, m in enumerate(doc["materials"]) if m["pbrMetallicRoughness"].get("baseColorFactor", [1, 1, 1, 1])[3] < 1]
    assert glass
    doc["materials"][glass[0]].setdefault("extensions", {})["KHR_materials_ior"] = {"ior": 1.33}
    doc["materials"][0].setdefault("extensions", {})["KHR_materials_ior"] = {}          # the extension's default, 1.5
    doc["scenes"] = [{"nodes": [0], "extras": {"backgroundColor": [0.25, 0.5, 1.0]}}]
    doc["scene"] = 0
    doc["buffers"][0]["uri"] = os.path.basename(path)[:-5] + ".bin"
    mod = os.path.join(os.path.dirname(path), "cornell_ior_bg.gltf")
    json.dump(doc, open(mod, "w"))
    off = rt.load_gltf(mod, 64, 48)
    assert off.materials.tobytes() == plain.materials.tobytes() and not off.background.any()
    on = rt.load_gltf(mod, 64, 48, flags=A.GLTF_IOR | A.GLTF_BACKGROUND)
    assert on.materials["ior"][glass[0]] == np.float32(1.33) and on.materials["ior"][0] == np.float32(1.5)
    assert np.array_equal(on.background, np.array([0.25, 0.5, 1.0], np.float32))
    assert np.array_equal(on.tri_verts, plain.tri_verts)
    assert not rt.load_gltf(mod, 64, 48, flags=A.GLTF_SPEC).background.any()             # not part of the spec set
    doc["materials"][glass[0]]["extensions"]["KHR_materials_ior"] = {"ior": 0.5}
    json.dump(doc, open(mod, "w"))
    with pytest.raises(rt.RtError):
        rt.load_gltf(mod, 64, 48, flags=A.GLTF_IOR)


def test_analytic_primitive_validation(scene_dir):
    """rt_scene_desc.prims (Plane / Ellipsoid / Box, main.cpp:242-464) is checked on the host before any GPU
    work: type, unit quaternion, positive radii / extents, a non-zero plane normal, an existing material (an
    emitting one only for ellipsoids and boxes: a plane cannot be sampled, main.cpp:261-267), at most
    RT_MAX_PRIMITIVES.  An emitting ellipsoid joins the light list behind the light triangles."""
    import raytracing_amd as rt
    base = rt.load_gltf(make_scene(scene_dir, "cornell"), 64, 48)
    emits = np.linalg.norm(base.materials["emission"], axis=1) > 1e-4
    dark, bright = int(np.flatnonzero(~emits)[0]), int(np.flatnonzero(emits)[0])

    def build(**change):
        p = np.zeros(1, dtype=rt.PRIMITIVE_DTYPE)
        p["type"], p["rotation"], p["param"], p["material"] = rt.PRIM_ELLIPSOID, [0, 0, 0, 1], [1, 2, 3], dark
        for k, v in change.items():
            p[k] = v
        return rt.build_flat(rt.HostScene(base.tri_verts, base.tri_material, base.materials, base.camera, prims=p))

    assert build()["tris"].shape[0] == 36
    lit = build(material=bright)
    assert lit["lights"].tolist() == build()["lights"].tolist() + [lit["tris"].shape[0]]
    for bad in (dict(type=3), dict(rotation=[0, 0, 0, 2]), dict(param=[1, 0, 1]), dict(param=[1, -1, 1]),
                dict(type=rt.PRIM_PLANE, param=[0, 0, 0]), dict(material=99), dict(type=rt.PRIM_PLANE, material=bright),
                dict(position=[np.nan, 0, 0])):
        with pytest.raises(rt.RtError):
            build(**bad)
    many = np.zeros(257, dtype=rt.PRIMITIVE_DTYPE)
    many["rotation"], many["param"], many["material"] = [0, 0, 0, 1], [0, 1, 0], dark
    with pytest.raises(rt.RtError) as ei:
        rt.build_flat(rt.HostScene(base.tri_verts, base.tri_material, base.materials, base.camera, prims=many))
    assert ei.value.code == -4


@pytest.mark.parametrize("name,kw,with_prims", [
    ("cornell", {}, False),
    ("cornell_onfloor", {}, True),
    ("glass", {"grid": 3, "subdiv": 2}, False),
    ("heightfield100k", {"nx": 40, "nz": 40}, True),
    ("spheres1m", {"n_spheres": 30, "subdiv": 2, "ground": 12}, False),
])
def test_reference_tree_equals_the_oracles_build(scene_dir, oracle_mod, name, kw, with_prims):
    """The host builds the tree the reference itself would build (BVH::build, main.cpp:552-558, 608-686) for the
    kernels that answer in the reference's traversal order.  Against the oracle's restatement of that build: the
    same nodes in the same numbering, the same boxes bit for bit, the same primitive order inside every leaf --
    with analytic primitives too (planes at the root, ellipsoids and boxes under their calculateAABB boxes)."""
    import raytracing_amd as rt
    base = rt.load_gltf(make_scene(scene_dir, name, **kw), 64, 48)
    prims = None
    if with_prims:
        rng = np.random.RandomState(3)
        prims = np.zeros(9, dtype=rt.PRIMITIVE_DTYPE)
        prims["type"] = [rt.PRIM_ELLIPSOID, rt.PRIM_PLANE, rt.PRIM_BOX, rt.PRIM_BOX, rt.PRIM_ELLIPSOID, rt.PRIM_PLANE,
                         rt.PRIM_ELLIPSOID, rt.PRIM_BOX, rt.PRIM_ELLIPSOID]
        prims["position"] = rng.uniform(-0.8, 0.8, size=(9, 3))
        q = rng.normal(size=(9, 4))
        prims["rotation"] = q / np.linalg.norm(q, axis=1, keepdims=True)
        prims["param"] = rng.uniform(0.05, 0.4, size=(9, 3))
        emits = np.linalg.norm(base.materials["emission"], axis=1) > 1e-4
        prims["material"] = int(np.flatnonzero(~emits)[0])
    host = rt.HostScene(base.tri_verts, base.tri_material, base.materials, base.camera, prims=prims)
    d = host.as_dict()
    orc = oracle_mod.OracleScene(d["tri_verts"], d["tri_material"], d["materials"], d["camera"], prims=prims,
                                 prims_in_tree=True)
    assert orc.info()["unsplittable"] == 0
    ref = rt.build_flat(host)["ref_tree"]
    assert ref is not None
    nodes, boxes, items = orc.tree()
    assert ref["nodes"].shape[0] == nodes.shape[0] and ref["depth"] == orc.info()["max_depth"]
    assert np.array_equal(ref["nodes"][:, 12:], nodes)                      # left, right, first, count
    assert np.array_equal(ref["nodes"][:, :12], boxes.view(np.uint32))      # the four corner vectors of AABB
    assert np.array_equal(ref["order"], items)
    assert ref["nplanes"] == (2 if with_prims else 0)
    if with_prims:                                                          # planes lead the list; the root tests them
        ntri = host.tri_material.shape[0]
        assert set(ref["order"][:2]) == {ntri + 1, ntri + 5} and nodes[0, 2] == 0 and nodes[0, 3] >= 2


def test_reference_tree_on_random_soups(oracle_mod):
    """The same comparison on 60 random soups built to stress the ORDER of the build: many equal centroid
    coordinates (triangles on a lattice, duplicated triangles), so that what std::sort does with equal keys --
    it is not stable, and the reference sorts four times per node -- decides which primitive lands in which
    leaf.  Where the oracle's build meets a node that no split improves, the product must report that there is no
    reference tree instead of building one."""
    import raytracing_amd as rt
    mats = np.zeros(1, dtype=rt.api.MATERIAL_DTYPE)
    mats["color"] = 0.5
    cam = dict(position=[0, 0, 3], right=[1, 0, 0], up=[0, 1, 0], forward=[0, 0, -1], tan_x=0.5, tan_y=0.5)
    built = refused = 0
    for seed in range(60):
        rng = np.random.RandomState(1000 + seed)
        n = int(rng.choice([5, 9, 33, 200, 1000, 3000]))
        kind = seed % 4
        if kind == 0:      # a lattice: centroids share coordinates on every axis
            c = rng.randint(0, 6, size=(n, 1, 3)).astype(np.float32) * 0.25
            tri = c + np.array([[0, 0, 0], [0.2, 0, 0], [0, 0.2, 0]], np.float32)[None] * rng.choice([1.0, -1.0], size=(n, 1, 1))
        elif kind == 1:    # random, with a tenth of the triangles duplicated
            tri = rng.uniform(-1, 1, size=(n, 1, 3)) + rng.uniform(-0.1, 0.1, size=(n, 3, 3))
            k = max(n // 10, 1)
            tri[:k] = tri[k:2 * k]
        elif kind == 2:    # a flat sheet (one axis degenerate) of mixed sizes
            tri = rng.uniform(-1, 1, size=(n, 1, 3)) + rng.uniform(-0.5, 0.5, size=(n, 3, 3)) * 10.0 ** rng.uniform(-3, 0, size=(n, 1, 1))
            tri[:, :, 1] = 0.25
        else:              # clusters far apart
            tri = rng.choice([-50.0, 0.0, 70.0], size=(n, 1, 3)) + rng.uniform(-0.3, 0.3, size=(n, 3, 3))
        tri = np.ascontiguousarray(tri, dtype=np.float32)
        host = rt.HostScene(tri, np.zeros(n, np.uint32), mats, cam)
        orc = oracle_mod.OracleScene(tri, np.zeros(n, np.uint32), mats, cam)
        ref = rt.build_flat(host)["ref_tree"]
        if orc.info()["unsplittable"]:
            assert ref is None, "seed %d: a reference tree although the reference's build breaks on this scene" % seed
            refused += 1
        else:
            assert ref is not None, "seed %d" % seed
            nodes, boxes, items = orc.tree()
            assert np.array_equal(ref["nodes"][:, 12:], nodes), "seed %d: topology" % seed
            assert np.array_equal(ref["nodes"][:, :12], boxes.view(np.uint32)), "seed %d: boxes" % seed
            assert np.array_equal(ref["order"], items), "seed %d: primitive order" % seed
            built += 1
        orc.close()
    assert built >= 30 and refused >= 1, (built, refused)


def test_reference_tree_limits():
    """Beyond what the reference can hold (uint16_t ids, main.cpp:541-544) or build (a node of more than four
    primitives that no split improves, main.cpp:675-679) there is no reference tree: the build says why and the
    scene keeps the kernels' own hit order."""
    import raytracing_amd as rt
    mats = np.zeros(1, dtype=rt.api.MATERIAL_DTYPE)
    mats["color"] = 0.5
    cam = dict(position=[0, 0, 3], right=[1, 0, 0], up=[0, 1, 0], forward=[0, 0, -1], tan_x=0.5, tan_y=0.5)
    rng = np.random.RandomState(2)
    big = (rng.uniform(-1, 1, size=(65536, 1, 3)) + rng.uniform(-0.01, 0.01, size=(65536, 3, 3))).astype(np.float32)
    assert rt.build_flat(rt.HostScene(big, np.zeros(65536, np.uint32), mats, cam))["ref_tree"] is None
    assert rt.build_flat(rt.HostScene(big[:30000], np.zeros(30000, np.uint32), mats, cam))["ref_tree"] is not None
    same = np.repeat(big[:1], 6, axis=0)                                    # six copies of one triangle: no split helps
    assert rt.build_flat(rt.HostScene(same, np.zeros(6, np.uint32), mats, cam))["ref_tree"] is None
