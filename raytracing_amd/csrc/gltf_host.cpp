// gltf_host.cpp -- host scene loading: .gltf JSON + external .bin buffers -> triangle soup,
// material table and camera.  Stands in for parseGLTF (main.cpp:866-1111) up to, but not
// including, the acceleration-structure build; it honours the reference loader's rules so the
// same file yields the same triangles, materials and camera:
//   * external buffers only, resolved against the scene file's directory (main.cpp:875-882)
//   * every bufferView needs buffer / byteLength / byteOffset; byteStride is ignored (885-891)
//   * material rules of main.cpp:894-927 (default METALLIC, metallicFactor < 1e-4 -> DIFFUSER,
//     baseColorFactor alpha < 1 - 1e-4 -> DIELECTRIC, emission x emissive_strength)
//   * POSITION is float32 VEC3 at stride 12 from view.byteOffset + accessor.byteOffset; indices
//     are u16/u32 SCALAR read from view.byteOffset only (1038-1078)
//   * node matrix = T*R*S or the column-major `matrix`; children are composed in one ascending
//     pass; the camera takes its own local transform only (958-1026)
//   * every node with a mesh is instantiated, `scenes` is ignored (1028-1031)
// Unlike the reference (undefined behaviour on a missing key, no bounds checks), malformed
// input is rejected with an error message.
// rt_gltf_load_ex adds opt-in flags that follow the glTF 2.0 specification where the reference
// does not (they change what some files mean, so they are off by default):
//   RT_GLTF_STRIDE      bufferView.byteStride for POSITION; bufferView.byteOffset defaults to 0
//   RT_GLTF_INDEX_OFFSET  accessor.byteOffset also applies to index accessors
//   RT_GLTF_DATA_URI    base64 `data:` buffers
//   RT_GLTF_HIERARCHY   world transform = product along the parent chain whatever the node order;
//                       the camera takes its world transform
//   RT_GLTF_GLB         binary containers: 12-byte header, JSON chunk, optional BIN chunk that backs the
//                       first buffer when it has no uri
//   RT_GLTF_IOR         KHR_materials_ior.ior instead of the constant 1.5 (main.cpp:65)
//   RT_GLTF_BACKGROUND  radiance of a ray that leaves the scene (InputData::backgroundColor,
//                       main.cpp:694, 1191, which the reference never sets) from
//                       scenes[scene].extras.backgroundColor = [r, g, b]
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "host_math.h"
#include "json_min.hpp"
#include "rt_internal.h"

namespace {

const float EPS = 0.0001f;  // main.cpp:22

struct BufferView {
    size_t buffer, byteLength, byteOffset, byteStride;
};
struct Accessor {
    size_t bufferView, count, componentType, byteOffset;
    std::string type;
};
struct MeshPrimitive {
    size_t position, indices, material;
};

std::string dir_of(const std::string &path) {
    size_t k = path.find_last_of('/');
    return k == std::string::npos ? std::string() : path.substr(0, k + 1);
}

std::vector<char> read_file(const std::string &path, size_t nbytes) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open buffer file " + path);
    std::vector<char> data(nbytes);
    f.read(data.data(), (std::streamsize)nbytes);
    if ((size_t)f.gcount() != nbytes) throw std::runtime_error("buffer file shorter than byteLength: " + path);
    return data;
}

float read_f32(const std::vector<char> &b, size_t off) {
    if (off + 4 > b.size()) throw std::runtime_error("POSITION accessor reads past its buffer");
    float v;
    std::memcpy(&v, b.data() + off, 4);
    return v;
}

size_t read_index(const std::vector<char> &b, size_t off, size_t width) {
    if (off + width > b.size()) throw std::runtime_error("index accessor reads past its buffer");
    if (width == 2) {
        uint16_t v;
        std::memcpy(&v, b.data() + off, 2);
        return v;
    }
    uint32_t v;
    std::memcpy(&v, b.data() + off, 4);
    return v;
}

std::vector<char> decode_base64(const std::string &text, size_t begin, size_t nbytes) {
    std::vector<char> outv;
    outv.reserve(nbytes);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = begin; i < text.size(); ++i) {
        const char c = text[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=') break;
        else throw std::runtime_error("data URI holds a character outside base64");
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            outv.push_back((char)((acc >> bits) & 0xFFu));
        }
    }
    if (outv.size() < nbytes) throw std::runtime_error("data URI shorter than byteLength");
    outv.resize(nbytes);
    return outv;
}

using jsonmin::Value;

// A node's own matrix by the reference's loading rules (main.cpp:958-1003).  The three factors are written as
// row-major arrays, multiplied as scale x rotation^T x translation and transposed at the end; a `matrix` property
// then overwrites the result element by element.  The float operations, zeros and ones included, are kept in
// exactly this sequence: the transformed vertices have to come out bit-identical to the reference's.
hm::mat4 local_matrix_of(const Value &node) {
    auto triple = [&node](const char *key, float fallback) {
        if (!node.contains(key)) return hm::V3(fallback, fallback, fallback);
        const Value &v = node.at(key);
        return hm::V3(v.idx(0).as_float(), v.idx(1).as_float(), v.idx(2).as_float());
    };
    const hm::vec3 t = triple("translation", 0.f), s = triple("scale", 1.f);
    hm::quat q{1.f, 0.f, 0.f, 0.f};
    if (node.contains("rotation")) {  // glTF order: x, y, z, w
        const Value &r = node.at("rotation");
        q = hm::quat{r.idx(3).as_float(), r.idx(0).as_float(), r.idx(1).as_float(), r.idx(2).as_float()};
    }
    const float tRows[16] = {1.f, 0.f, 0.f, t.x, 0.f, 1.f, 0.f, t.y, 0.f, 0.f, 1.f, t.z, 0.f, 0.f, 0.f, 1.f};
    const float sRows[16] = {s.x, 0.f, 0.f, 0.f, 0.f, s.y, 0.f, 0.f, 0.f, 0.f, s.z, 0.f, 0.f, 0.f, 0.f, 1.f};
    hm::mat4 m = hm::transpose(hm::mat4_from16(sRows) * hm::transpose(hm::to_mat4(q)) * hm::mat4_from16(tRows));
    if (node.contains("matrix")) {
        const Value &mv = node.at("matrix");
        for (int i = 0; i < 16; ++i) hm::elem(m.c[i / 4], i % 4) = mv.idx((size_t)i).as_float();
    }
    return m;
}

std::vector<size_t> child_indices_of(const Value &node, size_t nnodes) {
    std::vector<size_t> out;
    if (!node.contains("children")) return out;
    for (const Value &c : node.at("children").a) {
        const size_t k = (size_t)c.as_index();
        if (k >= nnodes) throw std::runtime_error("node child index out of range");
        out.push_back(k);
    }
    return out;
}


void load(const std::string &path, uint32_t width, uint32_t height, uint32_t flags, rt_host_scene &out) {
    using jsonmin::Value;
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open scene file " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    std::string text = ss.str();
    std::vector<char> glbBin;
    bool haveGlbBin = false;
    if ((flags & RT_GLTF_GLB) && text.size() >= 4 && text.compare(0, 4, "glTF") == 0) {
        // header: magic, version, total length; then chunks: length, type, payload (4-byte aligned)
        auto u32 = [&](size_t off) -> uint32_t {
            if (off + 4 > text.size()) throw std::runtime_error("binary glTF container is truncated");
            uint32_t v;
            std::memcpy(&v, text.data() + off, 4);
            return v;
        };
        if (u32(4) != 2u) throw std::runtime_error("binary glTF container version is not 2");
        const size_t total = u32(8);
        if (total > text.size() || total < 20) throw std::runtime_error("binary glTF container is truncated");
        std::string json;
        bool haveJson = false;
        for (size_t off = 12; off + 8 <= total;) {
            const size_t len = u32(off), type = u32(off + 4);
            if (off + 8 + len > total) throw std::runtime_error("binary glTF chunk runs past the container");
            if (type == 0x4E4F534Au && !haveJson) {  // "JSON"
                json.assign(text, off + 8, len);
                haveJson = true;
            } else if (type == 0x004E4942u && !haveGlbBin) {  // "BIN\0"
                glbBin.assign(text.begin() + (std::ptrdiff_t)(off + 8), text.begin() + (std::ptrdiff_t)(off + 8 + len));
                haveGlbBin = true;
            }
            off += 8 + ((len + 3) & ~(size_t)3);
        }
        if (!haveJson) throw std::runtime_error("binary glTF container has no JSON chunk");
        text.swap(json);
    }
    const Value data = jsonmin::parse(text);
    const std::string base = dir_of(path);

    std::vector<std::vector<char>> buffers;  // main.cpp:874-882
    for (const Value &b : data.at("buffers").a) {
        const size_t nbytes = (size_t)b.at("byteLength").as_index();
        if ((flags & RT_GLTF_GLB) && !b.contains("uri")) {  // the container's BIN chunk backs the first buffer
            if (!buffers.empty() || !haveGlbBin) throw std::runtime_error("buffer without uri and no BIN chunk to back it");
            if (glbBin.size() < nbytes) throw std::runtime_error("BIN chunk shorter than byteLength");
            buffers.push_back(std::vector<char>(glbBin.begin(), glbBin.begin() + (std::ptrdiff_t)nbytes));
            continue;
        }
        const std::string uri = b.at("uri").as_string();
        if ((flags & RT_GLTF_DATA_URI) && uri.rfind("data:", 0) == 0) {
            const size_t comma = uri.find(";base64,");
            if (comma == std::string::npos) throw std::runtime_error("data URI is not base64");
            buffers.push_back(decode_base64(uri, comma + 8, nbytes));
        } else {
            buffers.push_back(read_file(base + uri, nbytes));
        }
    }

    std::vector<BufferView> views;  // main.cpp:884-891
    for (const Value &v : data.at("bufferViews").a) {
        BufferView bv{(size_t)v.at("buffer").as_index(), (size_t)v.at("byteLength").as_index(), 0, 0};
        if (!(flags & RT_GLTF_STRIDE) || v.contains("byteOffset")) bv.byteOffset = (size_t)v.at("byteOffset").as_index();
        if ((flags & RT_GLTF_STRIDE) && v.contains("byteStride")) bv.byteStride = (size_t)v.at("byteStride").as_index();
        if (bv.buffer >= buffers.size()) throw std::runtime_error("bufferView names a missing buffer");
        views.push_back(bv);
    }

    out.materials.clear();  // main.cpp:893-927
    if (data.contains("materials")) {
        for (const Value &m : data.at("materials").a) {
            rt_material md;
            md.color[0] = md.color[1] = md.color[2] = 1.f;
            md.emission[0] = md.emission[1] = md.emission[2] = 0.f;
            md.type = RT_MAT_METALLIC;
            md.ior = 1.5f;
            if (m.contains("pbrMetallicRoughness")) {
                const Value &r = m.at("pbrMetallicRoughness");
                // The reference compares the JSON number (a double) with the float constant.
                if (r.contains("metallicFactor") && r.at("metallicFactor").as_double() < (double)EPS)
                    md.type = RT_MAT_DIFFUSER;
                if (r.contains("baseColorFactor")) {
                    const Value &c = r.at("baseColorFactor");
                    md.color[0] = c.idx(0).as_float();
                    md.color[1] = c.idx(1).as_float();
                    md.color[2] = c.idx(2).as_float();
                    if (c.idx(3).as_double() < (double)(1.f - EPS)) md.type = RT_MAT_DIELECTRIC;
                }
            }
            if (m.contains("emissiveFactor")) {
                const Value &e = m.at("emissiveFactor");
                md.emission[0] = e.idx(0).as_float();
                md.emission[1] = e.idx(1).as_float();
                md.emission[2] = e.idx(2).as_float();
            }
            if (m.contains("extensions") && m.at("extensions").contains("KHR_materials_emissive_strength")) {
                float strength =
                    m.at("extensions").at("KHR_materials_emissive_strength").at("emissiveStrength").as_float();
                md.emission[0] *= strength;
                md.emission[1] *= strength;
                md.emission[2] *= strength;
            }
            if ((flags & RT_GLTF_IOR) && m.contains("extensions") && m.at("extensions").contains("KHR_materials_ior")) {
                const Value &ext = m.at("extensions").at("KHR_materials_ior");
                md.ior = ext.contains("ior") ? ext.at("ior").as_float() : 1.5f;  // the extension's default
                if (!(md.ior >= 1.f)) throw std::runtime_error("KHR_materials_ior.ior must be at least 1");
            }
            out.materials.push_back(md);
        }
    }

    std::vector<std::vector<MeshPrimitive>> meshes;  // main.cpp:929-940
    if (data.contains("meshes")) {
        for (const Value &mesh : data.at("meshes").a) {
            std::vector<MeshPrimitive> prims;
            for (const Value &p : mesh.at("primitives").a)
                prims.push_back(MeshPrimitive{(size_t)p.at("attributes").at("POSITION").as_index(),
                                              (size_t)p.at("indices").as_index(),
                                              (size_t)p.at("material").as_index()});
            meshes.push_back(prims);
        }
    }

    std::vector<Accessor> accessors;  // main.cpp:942-953
    if (data.contains("accessors")) {
        for (const Value &a : data.at("accessors").a) {
            Accessor ad{(size_t)a.at("bufferView").as_index(), (size_t)a.at("count").as_index(),
                        (size_t)a.at("componentType").as_index(), 0, a.at("type").as_string()};
            if (a.contains("byteOffset")) ad.byteOffset = (size_t)a.at("byteOffset").as_index();
            if (ad.bufferView >= views.size()) throw std::runtime_error("accessor names a missing bufferView");
            accessors.push_back(ad);
        }
    }

    out.background[0] = out.background[1] = out.background[2] = 0.f;  // main.cpp:694
    if ((flags & RT_GLTF_BACKGROUND) && data.contains("scenes")) {
        const size_t which = data.contains("scene") ? (size_t)data.at("scene").as_index() : 0;
        const Value &sc = data.at("scenes").idx(which);
        if (sc.contains("extras") && sc.at("extras").contains("backgroundColor")) {
            const Value &c = sc.at("extras").at("backgroundColor");
            for (size_t k = 0; k < 3; ++k) {
                out.background[k] = c.idx(k).as_float();
                if (!(out.background[k] >= 0.f) || !std::isfinite(out.background[k]))
                    throw std::runtime_error("backgroundColor components must be finite and non-negative");
            }
        }
    }

    rt_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    hm::vec3 camRight = hm::V3(1.f, 0.f, 0.f), camUp = hm::V3(0.f, 1.f, 0.f), camFwd = hm::V3(0.f, 0.f, -1.f);
    hm::vec3 camPos = hm::V3(0.f, 0.f, 0.f);
    float tanY = 0.f;
    size_t cameraNode = (size_t)-1;

    const std::vector<Value> &nodes = data.at("nodes").a;
    std::vector<hm::mat4> world;               // per node: local matrix here, composed with its ancestors' below
    std::vector<std::vector<size_t>> kids;
    world.reserve(nodes.size());
    kids.reserve(nodes.size());
    for (const Value &node : nodes) {
        world.push_back(local_matrix_of(node));
        kids.push_back(child_indices_of(node, nodes.size()));
        if (node.contains("camera")) {
            // main.cpp:1012-1019: position and orientation from this node's OWN matrix; every camera node turns
            // the axes once more (they are not reset between nodes)
            const Value &cams = data.at("cameras");
            const float yfov = cams.idx((size_t)node.at("camera").as_index()).at("perspective").at("yfov").as_float();
            tanY = std::tan(yfov / 2.f);
            cameraNode = world.size() - 1;
            const hm::mat4 &m = world.back();
            const hm::quat q = hm::to_quat(m);
            camPos = hm::V3(m.c[3].x, m.c[3].y, m.c[3].z);
            camRight = hm::rotate(q, camRight);
            camUp = hm::rotate(q, camUp);
            camFwd = hm::rotate(q, camFwd);
        }
    }

    if (flags & RT_GLTF_HIERARCHY) {
        // world = parent's world * local, parents first whatever their indices
        std::vector<size_t> parent(world.size(), (size_t)-1);
        for (size_t i = 0; i < world.size(); ++i)
            for (size_t child : kids[i]) {
                if (parent[child] != (size_t)-1) throw std::runtime_error("node has two parents");
                parent[child] = i;
            }
        std::vector<size_t> order;
        std::vector<size_t> stack;
        for (size_t i = 0; i < world.size(); ++i)
            if (parent[i] == (size_t)-1) stack.push_back(i);
        while (!stack.empty()) {
            const size_t i = stack.back();
            stack.pop_back();
            order.push_back(i);
            for (size_t child : kids[i]) stack.push_back(child);
        }
        if (order.size() != world.size()) throw std::runtime_error("node hierarchy has a cycle");
        for (size_t i : order)
            if (parent[i] != (size_t)-1) world[i] = world[parent[i]] * world[i];
        if (cameraNode != (size_t)-1) {
            const hm::mat4 &w = world[cameraNode];
            const hm::quat q = hm::to_quat(w);
            camPos = hm::V3(w.c[3].x, w.c[3].y, w.c[3].z);
            camRight = hm::rotate(q, hm::V3(1.f, 0.f, 0.f));
            camUp = hm::rotate(q, hm::V3(0.f, 1.f, 0.f));
            camFwd = hm::rotate(q, hm::V3(0.f, 0.f, -1.f));
        }
    } else {
        for (size_t i = 0; i < world.size(); ++i)  // main.cpp:1022-1026
            for (size_t child : kids[i]) world[child] = world[i] * world[child];
    }

    out.tri_verts.clear();
    out.tri_material.clear();
    for (size_t i = 0; i < world.size(); ++i) {  // main.cpp:1028-1101
        if (!nodes[i].contains("mesh")) continue;
        size_t meshIdx = (size_t)nodes[i].at("mesh").as_index();
        if (meshIdx >= meshes.size()) throw std::runtime_error("node names a missing mesh");
        for (const MeshPrimitive &prim : meshes[meshIdx]) {
            if (prim.position >= accessors.size() || prim.indices >= accessors.size())
                throw std::runtime_error("mesh primitive names a missing accessor");
            if (prim.material >= out.materials.size())
                throw std::runtime_error("mesh primitive names a missing material");
            const Accessor &ap = accessors[prim.position];
            const BufferView &vp = views[ap.bufferView];
            const std::vector<char> &bp = buffers[vp.buffer];
            if (ap.type != "VEC3") throw std::runtime_error("Unsupported accessor type, expected VEC3");
            const size_t offset = vp.byteOffset + ap.byteOffset;
            const size_t stride = vp.byteStride ? vp.byteStride : 12;  // byteStride is 0 unless RT_GLTF_STRIDE
            std::vector<hm::vec3> points(ap.count);
            for (size_t j = 0; j < ap.count; ++j)
                points[j] = hm::V3(read_f32(bp, offset + stride * j), read_f32(bp, offset + stride * j + 4),
                                   read_f32(bp, offset + stride * j + 8));

            const Accessor &ai = accessors[prim.indices];
            const BufferView &vi = views[ai.bufferView];
            const std::vector<char> &bi = buffers[vi.buffer];
            if (ai.type != "SCALAR") throw std::runtime_error("Unsupported accessor type, expected SCALAR");
            size_t iw;
            if (ai.componentType == 5123) iw = 2;
            else if (ai.componentType == 5125) iw = 4;
            else throw std::runtime_error("Unsupported accessor component type, expected 5123 or 5125");

            const size_t indexBase = vi.byteOffset + ((flags & RT_GLTF_INDEX_OFFSET) ? ai.byteOffset : 0);
            const hm::mat4 &m = world[i];
            for (size_t j = 0; j + 2 < ai.count; j += 3) {
                size_t idx[3];
                for (size_t k = 0; k < 3; ++k) {
                    idx[k] = read_index(bi, indexBase + iw * (j + k), iw);
                    if (idx[k] >= points.size()) throw std::runtime_error("vertex index out of range");
                }
                for (size_t k = 0; k < 3; ++k) {
                    const hm::vec3 &p = points[idx[k]];
                    hm::vec4 w = m * hm::vec4{p.x, p.y, p.z, 1.f};
                    out.tri_verts.push_back(w.x);
                    out.tri_verts.push_back(w.y);
                    out.tri_verts.push_back(w.z);
                }
                out.tri_material.push_back((uint32_t)prim.material);
            }
        }
    }

    for (int k = 0; k < 3; ++k) {
        cam.position[k] = (&camPos.x)[k];
        cam.right[k] = (&camRight.x)[k];
        cam.up[k] = (&camUp.x)[k];
        cam.forward[k] = (&camFwd.x)[k];
    }
    cam.tan_y = tanY;
    cam.tan_x = tanY * float(width) / float(height);  // main.cpp:1104
    out.camera = cam;
}

}  // namespace

extern "C" {

int rt_gltf_load(const char *path, uint32_t width, uint32_t height, rt_host_scene **out) {
    return rt_gltf_load_ex(path, width, height, 0u, out);
}

int rt_gltf_load_ex(const char *path, uint32_t width, uint32_t height, uint32_t flags, rt_host_scene **out) {
    if (!path || !out || width == 0 || height == 0 || (flags & ~(uint32_t)RT_GLTF_ALL)) {
        rt::set_error("rt_gltf_load: null argument, zero image size or unknown flag");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    rt_host_scene *hs = nullptr;
    try {
        hs = new rt_host_scene();
        load(path, width, height, flags, *hs);
    } catch (const std::exception &e) {
        delete hs;
        rt::set_error(std::string("rt_gltf_load: ") + e.what());
        std::string msg = e.what();
        return (msg.rfind("cannot open", 0) == 0 || msg.rfind("buffer file", 0) == 0) ? RT_ERR_IO : RT_ERR_INVALID;
    }
    hs->parse_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out = hs;
    return RT_OK;
}

int rt_host_scene_get_desc(const rt_host_scene *hs, rt_scene_desc *out) {
    if (!hs || !out) {
        rt::set_error("rt_host_scene_get_desc: null argument");
        return RT_ERR_INVALID;
    }
    out->tri_verts = hs->tri_verts.data();
    out->tri_material = hs->tri_material.data();
    out->ntri = (uint32_t)hs->tri_material.size();
    out->materials = hs->materials.data();
    out->nmat = (uint32_t)hs->materials.size();
    out->camera = hs->camera;
    for (int k = 0; k < 3; ++k) out->background[k] = hs->background[k];
    out->prims = nullptr;  // scene files cannot describe analytic primitives
    out->nprim = 0;
    return RT_OK;
}

void rt_host_scene_free(rt_host_scene *hs) { delete hs; }

}  // extern "C"
