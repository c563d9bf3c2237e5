// rt_internal.h -- structures shared by the host-side translation units of librt_amd.so.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

// A parsed scene file (the host half of the reference's InputData, main.cpp:689-706).
struct rt_host_scene {
    std::vector<float> tri_verts;        // ntri * 9, loader emission order
    std::vector<uint32_t> tri_material;  // ntri
    std::vector<rt_material> materials;
    rt_camera camera;
    double parse_seconds = 0.0;
};

namespace rt {

// Traversal limits the kernels are built for; checked on the host before any launch.
constexpr uint32_t kMaxLeafTris = 2;     // reference: 4 (main.cpp:618); a lane slot of the triangle phase costs ~2.3 node slots, and 2 measured +2 %
constexpr uint32_t kStackEntries = 24;   // per-lane traversal stack held in LDS
constexpr uint32_t kMaxTreeHeight = 24;  // builder guarantee: pending stack entries <= height
constexpr uint32_t kMaxTriangles = 1u << 28;

// Flattened acceleration structure, exactly as it lies in HBM.
//
// Node (64 B = 4 x float4): the boxes of BOTH children plus the two child references, so one
// node fetch decides both descents.
//   q0 = (L.lo.x, L.lo.y, L.lo.z, L.hi.x)   q1 = (L.hi.y, L.hi.z, R.lo.x, R.lo.y)
//   q2 = (R.lo.z, R.hi.x, R.hi.y, R.hi.z)   q3 = (bits(left), bits(right), 0, 0)
// A child reference >= 0 is a node index; < 0 encodes a leaf: ~ref = (first << 3) | (count-1),
// `first` indexing the leaf-ordered triangle array.  An absent child has an inverted box.
//
// Triangle (64 B = 4 x float4), leaf order:
//   q0 = (A.x, A.y, A.z, AB.x)  q1 = (AB.y, AB.z, AC.x, AC.y)  q2 = (AC.z, N.x, N.y, N.z)
//   q3 = (bits(loader id), bits(material), pdfConst, 0)
// with AB = B-A, AC = C-A, N = cross(AB, AC) (unnormalised; its three components are exactly
// the third-row cofactors of the reference's 3x3 inverse, main.cpp:481) and
// pdfConst = 2 / |N| (main.cpp:532).
struct FlatScene {
    std::vector<float> nodes;       // nnodes * 16
    std::vector<float> tris;        // ntri * 16
    std::vector<uint32_t> lights;   // leaf-order positions of emissive triangles, loader order
    uint32_t nnodes = 0;
    uint32_t ntri = 0;
    uint32_t tree_height = 0;
    double build_seconds = 0.0;
};

// Builds the BVH (binned SAH, height-bounded) and flattens it.  Throws std::runtime_error.
void build_flat_scene(const rt_scene_desc &desc, FlatScene &out);

void set_error(const std::string &msg);

}  // namespace rt
