"""numpy restatement of the reference scene loader -- TEST INFRASTRUCTURE ONLY.

Follows parseGLTF (main.cpp:866-1111) step by step in float32 with glm 0.9.9.8's operation
order (mat4*mat4, mat4*vec4, mat3_cast, quat_cast, quat*vec3 restated from glm's published
implementation; glm itself is absent from this image, so this loader is "parity unpinned").
It exists to check the product's C++ host loader bit for bit on the same files.
"""
import ctypes
import json
import os

import numpy as np

f32 = np.float32
EPS = f32(0.0001)  # main.cpp:22

_libm = ctypes.CDLL("libm.so.6")
_libm.tanf.restype = ctypes.c_float
_libm.tanf.argtypes = [ctypes.c_float]

MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("emission", "<f4", 3), ("type", "<i4"), ("ior", "<f4")])
DIFFUSER, METALLIC, DIELECTRIC = 0, 1, 2


# mat4 is stored as M[col][row], like glm.
def _mat4_from_16(vals):
    """glm's 16-scalar constructor is column-major: the first four numbers are column 0."""
    return np.array(vals, dtype=f32).reshape(4, 4)


def _transpose(m):
    return np.ascontiguousarray(m.T)


def _mat4_mul(a, b):
    """glm: R[i] = ((A[0]*B[i][0] + A[1]*B[i][1]) + A[2]*B[i][2]) + A[3]*B[i][3]."""
    r = np.empty((4, 4), dtype=f32)
    for i in range(4):
        r[i] = ((a[0] * b[i][0] + a[1] * b[i][1]) + a[2] * b[i][2]) + a[3] * b[i][3]
    return r


def _mat3_cast(q):
    """glm::mat3_cast / toMat4; q = (w, x, y, z)."""
    w, x, y, z = (f32(v) for v in q)
    qxx, qyy, qzz = x * x, y * y, z * z
    qxz, qxy, qyz = x * z, x * y, y * z
    qwx, qwy, qwz = w * x, w * y, w * z
    one, two = f32(1), f32(2)
    m = np.zeros((4, 4), dtype=f32)
    m[0][0] = one - two * (qyy + qzz)
    m[0][1] = two * (qxy + qwz)
    m[0][2] = two * (qxz - qwy)
    m[1][0] = two * (qxy - qwz)
    m[1][1] = one - two * (qxx + qzz)
    m[1][2] = two * (qyz + qwx)
    m[2][0] = two * (qxz + qwy)
    m[2][1] = two * (qyz - qwx)
    m[2][2] = one - two * (qxx + qyy)
    m[3][3] = one
    return m


def _quat_cast(m):
    """glm::quat_cast on the upper 3x3; returns (w, x, y, z)."""
    fx = m[0][0] - m[1][1] - m[2][2]
    fy = m[1][1] - m[0][0] - m[2][2]
    fz = m[2][2] - m[0][0] - m[1][1]
    fw = m[0][0] + m[1][1] + m[2][2]
    idx, big = 0, fw
    if fx > big:
        big, idx = fx, 1
    if fy > big:
        big, idx = fy, 2
    if fz > big:
        big, idx = fz, 3
    val = np.sqrt(f32(big + f32(1))) * f32(0.5)
    mult = f32(0.25) / val
    if idx == 0:
        return (val, (m[1][2] - m[2][1]) * mult, (m[2][0] - m[0][2]) * mult, (m[0][1] - m[1][0]) * mult)
    if idx == 1:
        return ((m[1][2] - m[2][1]) * mult, val, (m[0][1] + m[1][0]) * mult, (m[2][0] + m[0][2]) * mult)
    if idx == 2:
        return ((m[2][0] - m[0][2]) * mult, (m[0][1] + m[1][0]) * mult, val, (m[1][2] + m[2][1]) * mult)
    return ((m[0][1] - m[1][0]) * mult, (m[2][0] + m[0][2]) * mult, (m[1][2] + m[2][1]) * mult, val)


def _cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], dtype=f32)


def _quat_rotate(q, v):
    """glm quat * vec3: v + ((uv * w) + uuv) * 2."""
    w = f32(q[0])
    qv = np.array([q[1], q[2], q[3]], dtype=f32)
    v = np.asarray(v, dtype=f32)
    uv = _cross(qv, v)
    uuv = _cross(qv, uv)
    return v + ((uv * w) + uuv) * f32(2)


def parse_gltf(path, width, height):
    with open(path) as fh:
        data = json.load(fh)
    base = os.path.dirname(os.path.abspath(path))

    buffers = []  # main.cpp:875-882
    for b in data["buffers"]:
        with open(os.path.join(base, b["uri"]), "rb") as fh:
            buffers.append(fh.read(int(b["byteLength"])))

    views = [(int(v["buffer"]), int(v["byteLength"]), int(v["byteOffset"])) for v in data["bufferViews"]]  # 885-891

    mats = np.zeros(len(data["materials"]), dtype=MATERIAL_DTYPE)  # 894-927
    for i, m in enumerate(data["materials"]):
        color = np.array([1, 1, 1], dtype=f32)
        emission = np.zeros(3, dtype=f32)
        kind = METALLIC
        if "pbrMetallicRoughness" in m:
            r = m["pbrMetallicRoughness"]
            if "metallicFactor" in r and float(r["metallicFactor"]) < float(EPS):
                kind = DIFFUSER
            if "baseColorFactor" in r:
                c = r["baseColorFactor"]
                color = np.array([c[0], c[1], c[2]], dtype=f32)
                if float(c[3]) < float(f32(1) - EPS):
                    kind = DIELECTRIC
        if "emissiveFactor" in m:
            e = m["emissiveFactor"]
            emission = np.array([e[0], e[1], e[2]], dtype=f32)
        ext = m.get("extensions", {})
        if "KHR_materials_emissive_strength" in ext:
            emission = emission * f32(ext["KHR_materials_emissive_strength"]["emissiveStrength"])
        mats[i] = (color, emission, kind, f32(1.5))

    meshes = [[(int(p["attributes"]["POSITION"]), int(p["indices"]), int(p["material"])) for p in m["primitives"]]
              for m in data["meshes"]]  # 930-940
    accessors = [(int(a["bufferView"]), int(a["count"]), int(a["componentType"]), a["type"],
                  int(a.get("byteOffset", 0))) for a in data["accessors"]]  # 943-953

    cam = dict(position=np.zeros(3, f32), right=np.array([1, 0, 0], f32), up=np.array([0, 1, 0], f32),
               forward=np.array([0, 0, -1], f32), tan_x=f32(0), tan_y=f32(0))
    transitions, children = [], []
    for node in data["nodes"]:  # 958-1020
        t = np.zeros(3, dtype=f32)
        if "translation" in node:
            t = np.array(node["translation"][:3], dtype=f32)
        tm = _mat4_from_16([1, 0, 0, t[0], 0, 1, 0, t[1], 0, 0, 1, t[2], 0, 0, 0, 1])
        q = (f32(1), f32(0), f32(0), f32(0))
        if "rotation" in node:
            r = node["rotation"]
            q = (f32(r[3]), f32(r[0]), f32(r[1]), f32(r[2]))
        rm = _transpose(_mat3_cast(q))
        s = np.ones(3, dtype=f32)
        if "scale" in node:
            s = np.array(node["scale"][:3], dtype=f32)
        sm = _mat4_from_16([s[0], 0, 0, 0, 0, s[1], 0, 0, 0, 0, s[2], 0, 0, 0, 0, 1])
        tr = _transpose(_mat4_mul(_mat4_mul(sm, rm), tm))
        if "matrix" in node:
            for i in range(16):
                tr[i // 4][i % 4] = f32(node["matrix"][i])
        translation = np.array([tr[3][0], tr[3][1], tr[3][2]], dtype=f32)
        rotation = _quat_cast(tr)
        transitions.append(tr)
        children.append([int(c) for c in node.get("children", [])])
        if "camera" in node:
            yfov = f32(data["cameras"][int(node["camera"])]["perspective"]["yfov"])
            cam["tan_y"] = f32(_libm.tanf(float(yfov / f32(2))))
            cam["position"] = translation
            cam["right"] = _quat_rotate(rotation, cam["right"])
            cam["up"] = _quat_rotate(rotation, cam["up"])
            cam["forward"] = _quat_rotate(rotation, cam["forward"])

    for i in range(len(transitions)):  # 1022-1026
        for c in children[i]:
            transitions[c] = _mat4_mul(transitions[i], transitions[c])

    tri_verts, tri_mat = [], []
    for i, node in enumerate(data["nodes"]):  # 1028-1101
        if "mesh" not in node:
            continue
        m = transitions[i]
        for (pa, ia, mat) in meshes[int(node["mesh"])]:
            view, count, _ctype, tname, aoff = accessors[pa]
            if tname != "VEC3":
                raise RuntimeError("Unsupported accessor type, expected VEC3")
            vb, _vl, voff = views[view]
            pts = np.frombuffer(buffers[vb], dtype="<f4", count=3 * count, offset=voff + aoff).reshape(count, 3)
            view, count, ctype, tname, _aoff = accessors[ia]
            if tname != "SCALAR":
                raise RuntimeError("Unsupported accessor type, expected SCALAR")
            vb, _vl, voff = views[view]
            ntri3 = ((count + 2) // 3) * 3  # the loop runs j = 0, 3, ... while j < count
            if ctype == 5123:
                idx = np.frombuffer(buffers[vb], dtype="<u2", count=ntri3, offset=voff).astype(np.int64)
            elif ctype == 5125:
                idx = np.frombuffer(buffers[vb], dtype="<u4", count=ntri3, offset=voff).astype(np.int64)
            else:
                raise RuntimeError("Unsupported accessor component type, expected 5123 or 5125")
            p = pts[idx]  # (3T, 3)
            # m * vec4(p, 1) = (m[0]*x + m[1]*y) + (m[2]*z + m[3]*w)
            w = (m[0][None, :] * p[:, 0:1] + m[1][None, :] * p[:, 1:2]) + \
                (m[2][None, :] * p[:, 2:3] + m[3][None, :] * f32(1))
            tri_verts.append(w[:, :3].reshape(-1, 3, 3))
            tri_mat.append(np.full(ntri3 // 3, mat, dtype=np.uint32))

    cam["tan_x"] = cam["tan_y"] * f32(width) / f32(height)  # 1104
    if tri_verts:
        tv = np.ascontiguousarray(np.concatenate(tri_verts), dtype=f32)
        tmat = np.concatenate(tri_mat)
    else:
        tv = np.zeros((0, 3, 3), dtype=f32)
        tmat = np.zeros(0, dtype=np.uint32)
    return dict(tri_verts=tv, tri_material=tmat, materials=mats, camera=cam)
