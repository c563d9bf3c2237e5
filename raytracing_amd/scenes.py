"""Deterministic synthetic scenes, written as `.gltf` + external `.bin` files.

The reference ships no scenes, so every workload is generated here.  The files obey exactly
what the reference loader accepts (main.cpp:866-1111): external buffers only, every
bufferView carries `byteOffset`, POSITION is tightly packed float32 VEC3, indices are SCALAR
u16/u32 addressed from the bufferView's own offset (one bufferView per accessor), every mesh
primitive names `indices` and `material`, parents precede their children, and the material
rules of main.cpp:894-927 select the three shading models:

  * DIFFUSER   -- pbrMetallicRoughness.metallicFactor = 0
  * METALLIC   -- metallicFactor absent (the reference's default material type)
  * DIELECTRIC -- baseColorFactor alpha < 1
  * light      -- emissiveFactor x KHR_materials_emissive_strength.emissiveStrength

All geometry stays inside [-100, 100]^3 with features far above the reference's absolute
EPS = 1e-4 (main.cpp:22).
"""
import json
import math
import os

import numpy as np

DIFFUSER, METALLIC, DIELECTRIC = 0, 1, 2


class GltfBuilder:
    def __init__(self):
        self.bin = bytearray()
        self.views = []
        self.accessors = []
        self.materials = []
        self.meshes = []
        self.nodes = []
        self.cameras = []

    # -- buffers ---------------------------------------------------------------------------
    def _accessor(self, data, component_type, type_name, count):
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin += data
        self.views.append({"buffer": 0, "byteOffset": off, "byteLength": len(data)})
        self.accessors.append({"bufferView": len(self.views) - 1, "byteOffset": 0, "componentType": component_type,
                               "count": count, "type": type_name})
        return len(self.accessors) - 1

    # -- materials -------------------------------------------------------------------------
    def material(self, kind, color=(1.0, 1.0, 1.0), emission=None, strength=None):
        m = {}
        pbr = {}
        if kind == DIFFUSER:
            pbr["metallicFactor"] = 0.0
            pbr["baseColorFactor"] = [float(color[0]), float(color[1]), float(color[2]), 1.0]
        elif kind == METALLIC:
            pbr["baseColorFactor"] = [float(color[0]), float(color[1]), float(color[2]), 1.0]
        elif kind == DIELECTRIC:
            pbr["baseColorFactor"] = [float(color[0]), float(color[1]), float(color[2]), 0.5]
        else:
            raise ValueError(kind)
        m["pbrMetallicRoughness"] = pbr
        if emission is not None:
            m["emissiveFactor"] = [float(e) for e in emission]
            if strength is not None:
                m["extensions"] = {"KHR_materials_emissive_strength": {"emissiveStrength": float(strength)}}
        self.materials.append(m)
        return len(self.materials) - 1

    # -- meshes ----------------------------------------------------------------------------
    def mesh(self, parts):
        """parts: list of (positions (n,3) float32, indices (m,) int, material index)."""
        prims = []
        for pos, idx, mat in parts:
            pos = np.ascontiguousarray(pos, dtype="<f4")
            idx = np.asarray(idx)
            assert idx.size % 3 == 0 and idx.max() < pos.shape[0]
            if pos.shape[0] <= 65535:
                ia = self._accessor(idx.astype("<u2").tobytes(), 5123, "SCALAR", int(idx.size))
            else:
                ia = self._accessor(idx.astype("<u4").tobytes(), 5125, "SCALAR", int(idx.size))
            pa = self._accessor(pos.tobytes(), 5126, "VEC3", int(pos.shape[0]))
            prims.append({"attributes": {"POSITION": pa}, "indices": ia, "material": int(mat)})
        self.meshes.append({"primitives": prims})
        return len(self.meshes) - 1

    # -- nodes -----------------------------------------------------------------------------
    def node(self, mesh=None, translation=None, rotation=None, scale=None, matrix=None, children=None,
             camera=None):
        n = {}
        if mesh is not None:
            n["mesh"] = int(mesh)
        if translation is not None:
            n["translation"] = [float(x) for x in translation]
        if rotation is not None:
            n["rotation"] = [float(x) for x in rotation]  # x, y, z, w
        if scale is not None:
            n["scale"] = [float(x) for x in scale]
        if matrix is not None:
            n["matrix"] = [float(x) for x in matrix]  # column-major, 16 numbers
        if children:
            n["children"] = [int(c) for c in children]
        if camera is not None:
            n["camera"] = int(camera)
        self.nodes.append(n)
        return len(self.nodes) - 1

    def camera(self, yfov):
        self.cameras.append({"type": "perspective", "perspective": {"yfov": float(yfov), "znear": 0.01}})
        return len(self.cameras) - 1

    def write(self, path):
        base = os.path.splitext(os.path.basename(path))[0]
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        doc = {
            "asset": {"version": "2.0", "generator": "raytracing_amd.scenes"},
            "buffers": [{"uri": base + ".bin", "byteLength": len(self.bin)}],
            "bufferViews": self.views,
            "accessors": self.accessors,
            "materials": self.materials,
            "meshes": self.meshes,
            "nodes": self.nodes,
            "cameras": self.cameras,
            "scenes": [{"nodes": list(range(len(self.nodes)))}],
            "scene": 0,
        }
        with open(path, "w") as f:
            json.dump(doc, f)
        with open(os.path.join(os.path.dirname(os.path.abspath(path)), base + ".bin"), "wb") as f:
            f.write(bytes(self.bin))
        return path


# ------------------------------------------------------------------------------------------
# Geometry helpers (all outward-facing, counter-clockwise seen from outside)
# ------------------------------------------------------------------------------------------
def quad(p0, p1, p2, p3):
    pos = np.array([p0, p1, p2, p3], dtype=np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    return pos, idx


def unit_cube():
    """Cube [-0.5, 0.5]^3, 8 vertices, 12 outward-facing triangles."""
    v = np.array([[x, y, z] for z in (-0.5, 0.5) for y in (-0.5, 0.5) for x in (-0.5, 0.5)], dtype=np.float32)
    f = [
        (0, 2, 3), (0, 3, 1),  # z = -0.5, outward -z
        (4, 5, 7), (4, 7, 6),  # z = +0.5
        (0, 1, 5), (0, 5, 4),  # y = -0.5
        (2, 6, 7), (2, 7, 3),  # y = +0.5
        (0, 4, 6), (0, 6, 2),  # x = -0.5
        (1, 3, 7), (1, 7, 5),  # x = +0.5
    ]
    return v, np.array(f, dtype=np.uint32).reshape(-1)


def icosphere(subdiv):
    """Unit icosphere: 20 * 4^subdiv outward-facing triangles, shared vertices."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
             (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
             (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache = {}
        nf = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.array(verts, dtype=np.float32), np.array(faces, dtype=np.uint32).reshape(-1)


def heightfield(nx, nz, x0=-1.0, x1=1.0, z0=-1.0, z1=1.0, fn=None):
    """(nx x nz) quads = 2*nx*nz triangles, normals pointing +y."""
    if fn is None:
        fn = lambda x, z: -0.6 + 0.15 * np.sin(5.0 * x) * np.cos(4.0 * z)
    xs = np.linspace(x0, x1, nx + 1)
    zs = np.linspace(z0, z1, nz + 1)
    X, Z = np.meshgrid(xs, zs, indexing="xy")  # rows = z
    Y = fn(X, Z)
    pos = np.stack([X, Y, Z], axis=-1).reshape(-1, 3).astype(np.float32)
    i = np.arange(nx)[None, :] + (nx + 1) * np.arange(nz)[:, None]
    a, b, c, d = i, i + 1, i + nx + 1, i + nx + 2  # a--b / c--d (c,d at larger z)
    tri = np.stack([a, c, b, b, c, d], axis=-1).reshape(-1)
    return pos, tri.astype(np.uint32)


def _quat_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    s = math.sin(angle / 2.0)
    return [axis[0] * s, axis[1] * s, axis[2] * s, math.cos(angle / 2.0)]


# ------------------------------------------------------------------------------------------
# Scenes
# ------------------------------------------------------------------------------------------
def cornell_box(path, on_floor=False):
    """36 triangles: 5 diffuse wall quads, one emissive ceiling quad, a metallic box and a
    dielectric box (TRS nodes under one parent).  Camera at (0, 0, 3.4) looking down -z.
    on_floor: the boxes stand ON the floor (their bottom faces lie in the floor plane, as objects in
    scene files usually do) instead of floating 2 mm above it."""
    g = GltfBuilder()
    white = g.material(DIFFUSER, (0.73, 0.73, 0.73))
    red = g.material(DIFFUSER, (0.65, 0.05, 0.05))
    green = g.material(DIFFUSER, (0.12, 0.45, 0.15))
    light = g.material(DIFFUSER, (0.78, 0.78, 0.78), emission=(1.0, 0.8, 0.5), strength=15.0)
    mirror = g.material(METALLIC, (0.9, 0.9, 0.9))
    glass = g.material(DIELECTRIC, (0.95, 0.98, 1.0))

    walls = g.mesh([
        quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1)) + (white,),   # floor, +y
        quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)) + (white,),       # ceiling, -y
        quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)) + (white,),   # back, +z
        quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)) + (red,),     # left, +x
        quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)) + (green,),       # right, -x
    ])
    lamp = g.mesh([quad((-0.25, 0.998, -0.25), (0.25, 0.998, -0.25), (0.25, 0.998, 0.25), (-0.25, 0.998, 0.25))
                   + (light,)])
    cv, ci = unit_cube()
    cube_m = g.mesh([(cv, ci, mirror)])
    cube_g = g.mesh([(cv, ci, glass)])

    root = g.node(children=[1, 2, 3, 4])            # node 0: parent of everything below
    g.node(mesh=walls)                               # 1
    g.node(mesh=lamp)                                # 2
    # The boxes float 2 mm above the floor: a box face lying IN the floor plane gives every ray that leaves the
    # glass box downwards two hits at exactly the same distance, and which one wins depends on the order a tree
    # is walked in (the reference's strict minimum, main.cpp:563-571, keeps the first it meets).
    # With on_floor the two hits are there: only a traversal that resolves them in the reference's order renders
    # that scene as the reference does.
    lift = 0.0 if on_floor else 0.002
    g.node(mesh=cube_m, translation=(-0.38, -0.4 + lift, -0.35), rotation=_quat_axis_angle((0, 1, 0), 0.32),
           scale=(0.58, 1.2, 0.58))                  # 3: tall mirror box
    g.node(mesh=cube_g, translation=(0.4, -0.7 + lift, 0.3), rotation=_quat_axis_angle((0, 1, 0), -0.3),
           scale=(0.56, 0.6, 0.56))                  # 4: glass box
    cam = g.camera(0.6911)
    g.node(camera=cam, translation=(0.0, 0.0, 3.4))  # 5
    assert root == 0
    return g.write(path)


def heightfield_scene(path, nx=224, nz=224):
    """2*nx*nz + 2 triangles: a diffuse sinusoidal terrain lit by one emissive quad; the
    default 224 x 224 gives 100 354 triangles.  The camera looks down at the terrain, which
    fills the 16:9 frame."""
    g = GltfBuilder()
    ground = g.material(DIFFUSER, (0.6, 0.55, 0.4))
    light = g.material(DIFFUSER, (0.8, 0.8, 0.8), emission=(1.0, 0.95, 0.9), strength=20.0)
    pos, idx = heightfield(nx, nz, -3.0, 3.0, -3.4, 1.6)
    terrain = g.mesh([(pos, idx, ground)])
    lamp = g.mesh([quad((-0.4, 1.2, -0.4), (0.4, 1.2, -0.4), (0.4, 1.2, 0.4), (-0.4, 1.2, 0.4)) + (light,)])
    g.node(mesh=terrain)
    g.node(mesh=lamp)
    cam = g.camera(0.75)
    g.node(camera=cam, translation=(0.0, 0.9, 2.2), rotation=_quat_axis_angle((1, 0, 0), -0.68))
    return g.write(path)


def sphere_field_scene(path, n_spheres=780, subdiv=3, ground=64, seed=1234, glass_fraction=1.0 / 3.0,
                       metal_fraction=1.0 / 3.0, ground_kind=DIFFUSER):
    """A terrain, `n_spheres` instanced icospheres (20*4^subdiv triangles each) with diffuse /
    metallic / dielectric materials in the given proportions and four emissive quads.
    Defaults: 780 * 1280 + 2*64*64 + 8 = 1 006 600 triangles."""
    rng = np.random.RandomState(seed)
    g = GltfBuilder()
    m_ground = g.material(ground_kind, (0.55, 0.55, 0.5))
    m_light = g.material(DIFFUSER, (0.8, 0.8, 0.8), emission=(1.0, 0.9, 0.8), strength=25.0)
    diffuse = [g.material(DIFFUSER, c) for c in ((0.7, 0.2, 0.2), (0.2, 0.6, 0.25), (0.25, 0.3, 0.75), (0.7, 0.65, 0.3))]
    metals = [g.material(METALLIC, c) for c in ((0.95, 0.95, 0.95), (0.95, 0.8, 0.45), (0.75, 0.8, 0.95))]
    glasses = [g.material(DIELECTRIC, c) for c in ((0.97, 0.99, 1.0), (0.85, 1.0, 0.9))]

    sv, si = icosphere(subdiv)
    sphere_mesh = {m: g.mesh([(sv, si, m)]) for m in diffuse + metals + glasses}
    fn = lambda x, z: -0.8 + 0.08 * np.sin(2.5 * x) * np.cos(2.0 * z)
    gp, gi = heightfield(ground, ground, -3.0, 3.0, -3.0, 3.0, fn)
    g.node(mesh=g.mesh([(gp, gi, m_ground)]))
    lamps = []
    for cx, cz in ((-1.4, -1.4), (1.4, -1.4), (-1.4, 1.4), (1.4, 1.4)):
        lamps.append(quad((cx - 0.35, 2.2, cz - 0.35), (cx + 0.35, 2.2, cz - 0.35), (cx + 0.35, 2.2, cz + 0.35),
                          (cx - 0.35, 2.2, cz + 0.35)) + (m_light,))
    g.node(mesh=g.mesh(lamps))
    for _ in range(n_spheres):
        r = rng.uniform(0.05, 0.16)
        x, z = rng.uniform(-2.6, 2.6, size=2)
        y = float(fn(x, z)) + r + rng.uniform(0.0, 1.3)
        u = rng.uniform()
        if u < glass_fraction:
            m = glasses[rng.randint(len(glasses))]
        elif u < glass_fraction + metal_fraction:
            m = metals[rng.randint(len(metals))]
        else:
            m = diffuse[rng.randint(len(diffuse))]
        g.node(mesh=sphere_mesh[m], translation=(x, y, z), scale=(r, r, r))
    cam = g.camera(0.8)
    g.node(camera=cam, translation=(0.0, 1.1, 5.2), rotation=_quat_axis_angle((1, 0, 0), -0.25))
    return g.write(path)


def glass_scene(path, grid=4, subdiv=4):
    """Dielectric-heavy: a closed diffuse room, a grid of glass icospheres each holding a
    smaller metallic sphere, nested glass boxes in the middle, one ceiling light."""
    g = GltfBuilder()
    white = g.material(DIFFUSER, (0.75, 0.75, 0.75))
    blue = g.material(DIFFUSER, (0.2, 0.3, 0.7))
    light = g.material(DIFFUSER, (0.8, 0.8, 0.8), emission=(1.0, 1.0, 1.0), strength=18.0)
    glass = g.material(DIELECTRIC, (0.96, 0.99, 1.0))
    tint = g.material(DIELECTRIC, (0.8, 1.0, 0.85))
    metal = g.material(METALLIC, (0.95, 0.85, 0.6))
    room = g.mesh([
        quad((-2, -1, 2), (2, -1, 2), (2, -1, -2), (-2, -1, -2)) + (white,),
        quad((-2, 2, -2), (2, 2, -2), (2, 2, 2), (-2, 2, 2)) + (white,),
        quad((-2, -1, -2), (2, -1, -2), (2, 2, -2), (-2, 2, -2)) + (blue,),
        quad((-2, -1, 2), (-2, -1, -2), (-2, 2, -2), (-2, 2, 2)) + (white,),
        quad((2, -1, -2), (2, -1, 2), (2, 2, 2), (2, 2, -2)) + (white,),
    ])
    g.node(mesh=room)
    g.node(mesh=g.mesh([quad((-0.6, 1.995, -0.6), (0.6, 1.995, -0.6), (0.6, 1.995, 0.6), (-0.6, 1.995, 0.6)) + (light,)]))
    sv, si = icosphere(subdiv)
    s_glass = g.mesh([(sv, si, glass)])
    s_tint = g.mesh([(sv, si, tint)])
    sv2, si2 = icosphere(max(subdiv - 1, 0))
    s_metal = g.mesh([(sv2, si2, metal)])
    k = 0
    for i in range(grid):
        for j in range(grid):
            x = -1.5 + 3.0 * (i + 0.5) / grid
            z = -1.5 + 3.0 * (j + 0.5) / grid
            if abs(x) < 0.5 and abs(z) < 0.5:
                continue
            r = 0.3
            g.node(mesh=s_glass if (k % 2 == 0) else s_tint, translation=(x, -1 + r + 0.01, z), scale=(r, r, r))
            g.node(mesh=s_metal, translation=(x, -1 + r + 0.01, z), scale=(0.12, 0.12, 0.12))
            k += 1
    cv, ci = unit_cube()
    c_glass = g.mesh([(cv, ci, glass)])
    c_tint = g.mesh([(cv, ci, tint)])
    for n, s in enumerate((0.9, 0.6, 0.3)):
        g.node(mesh=c_glass if n % 2 == 0 else c_tint, translation=(0.0, -1 + 0.45 + 0.01, 0.0),
               rotation=_quat_axis_angle((0, 1, 0), 0.2 * (n + 1)), scale=(s, s, s))
    cam = g.camera(0.7)
    g.node(camera=cam, translation=(0.0, 0.4, 4.6), rotation=_quat_axis_angle((1, 0, 0), -0.12))
    return g.write(path)


def emissive_mesh_scene(path, subdiv=3, lamps=3):
    """Many light triangles: a closed diffuse room lit only by `lamps` emissive icospheres
    (20*4^subdiv triangles each, so 3840 lights by default) with a mirror ball and a glass ball
    between them.  Every light-pdf evaluation of the reference loops over all those triangles."""
    g = GltfBuilder()
    white = g.material(DIFFUSER, (0.75, 0.75, 0.75))
    red = g.material(DIFFUSER, (0.7, 0.25, 0.2))
    glow = [g.material(DIFFUSER, (0.8, 0.8, 0.8), emission=e, strength=6.0)
            for e in ((1.0, 0.85, 0.6), (0.6, 0.8, 1.0), (0.9, 1.0, 0.8))]
    mirror = g.material(METALLIC, (0.92, 0.92, 0.92))
    glass = g.material(DIELECTRIC, (0.97, 0.99, 1.0))
    room = g.mesh([
        quad((-2, -1, 2), (2, -1, 2), (2, -1, -2), (-2, -1, -2)) + (white,),
        quad((-2, 2, -2), (2, 2, -2), (2, 2, 2), (-2, 2, 2)) + (white,),
        quad((-2, -1, -2), (2, -1, -2), (2, 2, -2), (-2, 2, -2)) + (red,),
        quad((-2, -1, 2), (-2, -1, -2), (-2, 2, -2), (-2, 2, 2)) + (white,),
        quad((2, -1, -2), (2, -1, 2), (2, 2, 2), (2, 2, -2)) + (white,),
    ])
    g.node(mesh=room)
    sv, si = icosphere(subdiv)
    for k in range(lamps):
        a = 2.0 * np.pi * k / max(lamps, 1)
        g.node(mesh=g.mesh([(sv, si, glow[k % len(glow)])]), translation=(1.1 * np.cos(a), 0.9 + 0.3 * (k % 2), 1.1 * np.sin(a) - 0.3),
               scale=(0.28, 0.28, 0.28))
    sv2, si2 = icosphere(max(subdiv - 1, 0))
    g.node(mesh=g.mesh([(sv2, si2, mirror)]), translation=(-0.6, -0.6, 0.2), scale=(0.4, 0.4, 0.4))
    g.node(mesh=g.mesh([(sv2, si2, glass)]), translation=(0.7, -0.65, 0.5), scale=(0.35, 0.35, 0.35))
    cam = g.camera(0.75)
    g.node(camera=cam, translation=(0.0, 0.4, 4.4), rotation=_quat_axis_angle((1, 0, 0), -0.1))
    return g.write(path)


SCENES = {
    "emissive": emissive_mesh_scene,
    "cornell": cornell_box,
    "cornell_onfloor": lambda path, **kw: cornell_box(path, on_floor=True, **kw),
    "heightfield100k": heightfield_scene,
    "spheres1m": sphere_field_scene,
    "glass": glass_scene,
}


def generate(name, path, **kw):
    return SCENES[name](path, **kw)
