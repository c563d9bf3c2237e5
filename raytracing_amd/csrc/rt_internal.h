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
    float background[3] = {0.f, 0.f, 0.f};  // InputData::backgroundColor (main.cpp:694)
    double parse_seconds = 0.0;
};

namespace rt {

// Traversal limits the kernels are built for; checked on the host before any launch.
constexpr uint32_t kMaxLeafTris = 2;     // reference: 4 (main.cpp:618); a lane slot of the triangle phase costs ~2.3 node slots, and 2 measured +2 %
constexpr uint32_t kStackEntries = 24;   // per-lane traversal stack held in LDS
constexpr uint32_t kMaxTreeHeight = RT_MAX_TREE_HEIGHT;  // builder guarantee: pending stack entries <= height
constexpr uint32_t kMaxTriangles = RT_MAX_TRIANGLES;     // kMaxLeafTris leaves under kMaxTreeHeight inner levels
static_assert(kMaxTriangles == kMaxLeafTris << kMaxTreeHeight, "triangle limit follows from leaf size and height");
static_assert(kStackEntries >= kMaxTreeHeight, "the traversal stack must cover the tree height");

// Flattened acceleration structure, exactly as it lies in HBM.
//
// Node (32 B = 2 x 16 B): the boxes of BOTH children on a 16-bit grid plus the two child
// references, so one node fetch (two 16-byte loads per lane) decides both descents.
//   w0..w2 = left box, one word per axis x, y, z:  lo | hi << 16     w3..w5 = right box, same
//   w6 = left reference, w7 = right reference
//   plane coordinate = grid_origin[axis] + q * grid_cell[axis]; boxes are rounded outward and carry two
//   guard cells per side on top of the float padding (1e-5 + 4e-7 |v|) of the exact box.
// A child reference >= 0 is a node index; < 0 encodes a leaf: ~ref = (first << 3) | (count-1),
// `first` indexing the leaf-ordered triangle array.  An absent child (only in scenes of at most
// kMaxLeafTris triangles) has lo = hi = 65535 on every axis and refers to a valid leaf.
// Why 16 bits: the traversal kernels are bound by the per-CU vector-memory path, whose cost is per
// lane and per load instruction (about one cycle each), not per byte; the 64-byte float node
// (four loads) measured 8 % slower end to end although it needs 13 fewer VALU operations.
//
// Triangle (64 B = 4 x float4), leaf order:
//   q0 = (A.x, A.y, A.z, AB.x)  q1 = (AB.y, AB.z, AC.x, AC.y)  q2 = (AC.z, N.x, N.y, N.z)
//   q3 = (bits(loader id), bits(material), pdfConst, 0)
// with AB = B-A, AC = C-A, N = cross(AB, AC) (unnormalised; its three components are exactly
// the third-row cofactors of the reference's 3x3 inverse, main.cpp:481) and
// pdfConst = 2 / |N| (main.cpp:532).
struct FlatTree {
    std::vector<uint32_t> nodes;   // nnodes * 8
    std::vector<uint32_t> order;   // leaf position -> index of the triangle in the builder's input
    float grid_origin[3] = {0.f, 0.f, 0.f};
    float grid_cell[3] = {1.f, 1.f, 1.f};
    uint32_t nnodes = 0;
    uint32_t height = 0;
    uint32_t capped_splits = 0;  // splits where the height bound overrode the cheapest SAH candidate
    double sah_nodes = 0.0, sah_tris = 0.0;  // surface-area estimate of node visits / triangle tests per random ray
};

// Wide node (48 B = 3 x 16 B), the same tree with up to four children per node: a node's larger inner
// children are replaced by their own two children (largest box first) until four are reached.  The child
// boxes are re-quantised to 8 bits inside a frame of the node: per axis an origin on the 16-bit grid (the
// smallest lo of the children) and a shift s, child plane = origin + (q << s) grid cells, lo rounded down and
// hi up (so a wide box contains the two-child tree's box, at most 2^s - 1 cells looser per side).
//   w0 = lo.x of children 0..3 (one byte each), w1 = hi.x, w2 = lo.y, w3 = hi.y, w4 = lo.z, w5 = hi.z
//   w6 = origin.x | origin.y << 16      w7 = origin.z | s.x << 16 | s.y << 20 | s.z << 24 | children << 28
//   w8..w11 = child references: >= 0 index of a wide node, < 0 a leaf as above; unused ones 0x7FFFFFFF
// wf_extend walks this tree: half the node visits of the two-child one at three loads instead of two per
// visit, and the vector-memory path charges per lane and load instruction (four children with 16-bit boxes
// in 64 bytes need as many loads per ray as the two-child tree and measured 3 % faster; this one 8.5 %).
// Walking it can need more stack than the two-child tree guarantees (three pushes per visit); a ray that
// runs out is traced again on the two-child tree.
// The tree the reference itself would build for the scene (ref_tree.cpp), for the kernels that answer in the
// reference's own traversal order.  Node (64 B = 4 x 16 B): AABB::sizePlusCenter.xyz, negSizePlusCenter.xyz,
// minPoint.xyz, maxPoint.xyz (main.cpp:113-178), then left, right, firstPrimitiveId, primitiveCount
// (main.cpp:539-546; a leaf has left = right = 0).  items[i] = the i-th entry of the reference's re-ordered
// primitive list as a record index: a triangle's position in the leaf-ordered triangle array, or (number of
// triangle records) + k for analytic primitive k.
constexpr uint32_t kRefStackEntries = 64;  // frames of the reference-order walker = deepest tree it can follow
struct RefTree {
    bool available = false;
    std::string reason;            // why not, when not available
    std::vector<uint32_t> nodes;   // nnodes * 16
    // nnodes * 8: (lo.xyz, hi.x), (hi.yz, -, -): a box that really holds everything a ray can hit below the node.
    // Not the node's own box: the reference gives an analytic Box a tree box of HALF its size (main.cpp:457-463,
    // made whole again by the doubled slabs), so a walk that skips what a ray cannot hit must not trust it.
    std::vector<float> reach;
    std::vector<uint32_t> items;   // primitive list in the reference's order, as record indices
    std::vector<uint32_t> order;   // the same list as loader ids (triangle id, or ntri + k)
    uint32_t nnodes = 0;
    uint32_t depth = 0;            // deepest node
    uint32_t nplanes = 0;          // analytic planes at the front of the list, tested at the root
    double seconds = 0.0;
};
// recordOfTriangle: loader id -> position in the leaf-ordered triangle array; nrecords: records in that array.
void build_ref_tree(const rt_scene_desc &desc, const std::vector<uint32_t> &recordOfTriangle, uint32_t nrecords, RefTree &out);

struct FlatScene {
    RefTree ref;
    std::vector<uint32_t> nodes;    // nnodes * 8
    std::vector<uint32_t> wide;     // nwide * 12
    uint32_t nwide = 0;
    float grid_origin[3] = {0.f, 0.f, 0.f};
    float grid_cell[3] = {1.f, 1.f, 1.f};
    std::vector<float> tris;        // nslots * 16
    std::vector<uint32_t> lights;   // the light list (main.cpp:1096-1098) as record indices: leaf-order positions of the emissive
                                    // triangles in loader order, then nslots + k for every emissive analytic primitive k
    uint32_t ntrilights = 0;        // how many of them are triangles
    // tree over the light triangles only (same node format, its own grid) and their records in that
    // tree's leaf order: a triangle record whose word 12 = index in `lights`, word 13 = lights[index]
    std::vector<uint32_t> light_nodes;
    std::vector<float> light_recs;
    float light_grid_origin[3] = {0.f, 0.f, 0.f};
    float light_grid_cell[3] = {1.f, 1.f, 1.f};
    uint32_t light_nnodes = 0;
    uint32_t light_height = 0;
    // analytic primitives (main.cpp:242-464), 12 floats each: (position.xyz, bits(type)), rotation xyzw,
    // (param.xyz, bits(material)); tested one by one after the tree walk
    std::vector<float> prims;
    uint32_t nprims = 0;
    uint32_t nnodes = 0;
    uint32_t ntri = 0;
    uint32_t nslots = 0;  // records in `tris`: the ntri triangles in leaf order plus the pad records between them
    uint32_t tree_height = 0;
    uint32_t capped_splits = 0;
    double sah_nodes = 0.0, sah_tris = 0.0;
    double build_seconds = 0.0;
};

// Builds the BVH (binned SAH, height-bounded) and flattens it.  Throws std::runtime_error.
void build_flat_scene(const rt_scene_desc &desc, FlatScene &out);

void set_error(const std::string &msg);

}  // namespace rt
