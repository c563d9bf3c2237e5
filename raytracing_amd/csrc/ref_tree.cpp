// ref_tree.cpp -- the acceleration structure the reference itself would build for a scene (BVH::build,
// main.cpp:552-558 and 608-686), flattened for the device.
//
// The kernels trace on a tree of their own (bvh_build.cpp).  The closest hit does not depend on the tree --
// except where the reference's answer is a matter of ORDER: two primitives at exactly the same distance (the
// first one its walk meets wins, main.cpp:566-567, 589-590, 598-599), a NaN distance from a ray that lies in a
// triangle's plane (accepted when it is the first hit of a leaf, main.cpp:483, 566), a hit beyond INF, a
// subtree cut by `distance < minDistance` on boxes twice their size (main.cpp:174-177, 587, 596).  To answer
// those as the reference does, the same tree is needed: same nodes in the same pre-order numbering, same
// primitive order inside the leaves.  This file rebuilds it -- full-sweep SAH over the three axes, leaves of at
// most four, the primitive list sorted in place with std::sort four times per node, exactly the sequence
// of sorts the reference performs (the sort is not stable, so the order of equal keys depends on that sequence
// and on the library's algorithm: libstdc++'s, as for the reference built with the same compiler) -- and a
// walker on the device follows it in the reference's order (ref_closest_hit in rt_refwalk.h).
//
// Scope: scenes the reference itself can hold, i.e. at most 65 535 primitives and nodes (its ids are
// uint16_t, main.cpp:541-544), every node with more than four primitives splittable (otherwise the reference
// stores an out-of-range first index, main.cpp:675-679), and a depth the walker's stack covers.  Beyond that
// RefTree::available is false with the reason, and the kernels keep their own rule for equal distances.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "host_math.h"
#include "rt_internal.h"

namespace rt {
namespace {

const float INF = 10000.f, NEG_INF = -10000.f;  // main.cpp:23, 25

inline float gmin(float x, float y) { return (y < x) ? y : x; }  // glm::min
inline float gmax(float x, float y) { return (x < y) ? y : x; }  // glm::max

// AABB, main.cpp:113-178: the corner points only; centre and slabs are derived when a node is emitted
struct Bounds {
    float lo[3] = {INF, INF, INF};
    float hi[3] = {NEG_INF, NEG_INF, NEG_INF};
    void add(const float *p) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = gmin(lo[a], p[a]);
            hi[a] = gmax(hi[a], p[a]);
        }
    }
    void add(const Bounds &b) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = gmin(lo[a], b.lo[a]);
            hi[a] = gmax(hi[a], b.hi[a]);
        }
    }
    float area() const {  // main.cpp:120-130
        if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) return 0.f;
        const float a = hi[0] - lo[0], b = hi[1] - lo[1], c = hi[2] - lo[2];
        return a * b + a * c + b * c;
    }
    float centre(int a) const { return (hi[a] + lo[a]) / 2.f; }  // main.cpp:169-173
};

inline uint32_t fbits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// Ellipsoid / Box::calculateAABB (main.cpp:358-364, 457-463) through AABB::rotateAndTranslate (149-166)
// whole = true: the box the primitive really occupies (a Box intersects as +-param, main.cpp:370-373)
Bounds analytic_bounds(const rt_primitive &p, bool whole = false) {
    float h[3];
    for (int a = 0; a < 3; ++a) h[a] = (p.type == RT_PRIM_BOX && !whole) ? p.param[a] / 2.f : p.param[a];
    Bounds zero;
    zero.add(h);
    const float nh[3] = {-h[0], -h[1], -h[2]};
    zero.add(nh);
    const hm::quat q{p.rotation[3], p.rotation[0], p.rotation[1], p.rotation[2]};
    Bounds b;
    b.hi[0] = b.hi[1] = b.hi[2] = -INF;
    for (int c = 0; c < 8; ++c) {
        const hm::vec3 corner = hm::V3((c & 1) ? zero.lo[0] : zero.hi[0], (c & 2) ? zero.lo[1] : zero.hi[1],
                                       (c & 4) ? zero.lo[2] : zero.hi[2]);
        const hm::vec3 r = hm::rotate(q, corner);
        const float v[3] = {r.x, r.y, r.z};
        b.add(v);
    }
    for (int a = 0; a < 3; ++a) {
        b.lo[a] += p.position[a];
        b.hi[a] += p.position[a];
    }
    return b;
}

}  // namespace

void build_ref_tree(const rt_scene_desc &desc, const std::vector<uint32_t> &recordOfTriangle, uint32_t nrecords, RefTree &out) {
    const auto t0 = std::chrono::steady_clock::now();
    out = RefTree();
    const uint32_t ntri = desc.ntri, total = desc.ntri + desc.nprim;
    if (total > 65535u) {
        out.reason = "more than 65535 primitives (the reference's ids are 16 bits wide)";
        return;
    }
    // InputData::primitives: triangles in loader order, then the analytic primitives in list order; planes are
    // partitioned to the front (main.cpp:553-554) and stay outside the recursion
    std::vector<Bounds> box(total);
    std::vector<float> centre[3];
    for (int a = 0; a < 3; ++a) centre[a].resize(total);
    std::vector<uint8_t> isPlane(total, 0);
    for (uint32_t i = 0; i < ntri; ++i) {
        const float *v = desc.tri_verts + 9 * (size_t)i;
        box[i].add(v);
        box[i].add(v + 3);
        box[i].add(v + 6);
    }
    for (uint32_t k = 0; k < desc.nprim; ++k) {
        if (desc.prims[k].type == RT_PRIM_PLANE) isPlane[ntri + k] = 1;
        else box[ntri + k] = analytic_bounds(desc.prims[k]);
    }
    for (uint32_t i = 0; i < total; ++i)
        for (int a = 0; a < 3; ++a) centre[a][i] = box[i].centre(a);
    std::vector<uint32_t> order(total);
    for (uint32_t i = 0; i < total; ++i) order[i] = i;
    const uint32_t nplanes =
        (uint32_t)(std::partition(order.begin(), order.end(), [&](uint32_t i) { return isPlane[i] != 0; }) - order.begin());

    struct Node {
        Bounds b;
        uint32_t left = 0, right = 0, first = 0, count = 0;
    };
    std::vector<Node> nodes;
    struct Work {
        uint32_t begin, end, parent, depth;
    };
    const uint32_t kNone = 0xFFFFFFFFu;
    std::vector<Work> todo{Work{nplanes, total, kNone, 0u}};
    std::vector<Bounds> leftB, rightB;
    uint32_t depthMax = 0;
    // Pre-order: a node's left child is the next node, its right child the first node after the whole left
    // subtree (main.cpp:681-685) -- the right range waits on the stack until the left subtree is done.
    while (!todo.empty()) {
        const Work w = todo.back();
        todo.pop_back();
        const uint32_t cur = (uint32_t)nodes.size();
        if (cur >= 65535u) {
            out.reason = "more than 65535 tree nodes (the reference's ids are 16 bits wide)";
            return;
        }
        nodes.push_back(Node());
        if (w.parent != kNone) {
            if (nodes[w.parent].left == 0) nodes[w.parent].left = cur;
            else nodes[w.parent].right = cur;
        }
        depthMax = std::max(depthMax, w.depth);
        Bounds all;
        for (uint32_t i = w.begin; i < w.end; ++i) all.add(box[order[i]]);
        nodes[cur].b = all;
        const uint32_t n = w.end - w.begin;
        if (n <= 4) {  // main.cpp:618-622
            nodes[cur].first = w.begin;
            nodes[cur].count = n;
            continue;
        }
        float bestCost = (float)n * all.area();
        uint32_t middle = w.end;
        int bestAxis = 0;
        leftB.resize(n + 1);
        rightB.resize(n + 1);
        for (int axis = 0; axis < 3; ++axis) {  // main.cpp:627-663
            const std::vector<float> &c = centre[axis];
            std::sort(order.begin() + w.begin, order.begin() + w.end, [&c](uint32_t p1, uint32_t p2) { return c[p1] < c[p2]; });
            Bounds run;
            for (uint32_t j = 1; j <= n; ++j) {
                run.add(box[order[w.begin + j - 1]]);
                leftB[j] = run;
            }
            run = Bounds();
            for (uint32_t j = n - 1; j >= 1; --j) {
                run.add(box[order[w.begin + j]]);
                rightB[j] = run;
            }
            for (uint32_t i = 1; i < n; ++i) {
                const float cost = leftB[i].area() * (float)i + rightB[i].area() * (float)(n - i);
                if (cost < bestCost) {
                    bestCost = cost;
                    middle = w.begin + i;
                    bestAxis = axis;
                }
            }
        }
        {  // main.cpp:665-673: once more by the chosen axis, from the order the z sort left
            const std::vector<float> &c = centre[bestAxis];
            std::sort(order.begin() + w.begin, order.begin() + w.end, [&c](uint32_t p1, uint32_t p2) { return c[p1] < c[p2]; });
        }
        if (middle == w.end) {
            out.reason = "a node of more than four primitives that no split improves (the reference stores an out-of-range "
                         "index there, main.cpp:675-679)";
            return;
        }
        todo.push_back(Work{middle, w.end, cur, w.depth + 1});
        todo.push_back(Work{w.begin, middle, cur, w.depth + 1});
    }
    nodes[0].first = 0;  // main.cpp:556-557: the root also tests the planes
    nodes[0].count += nplanes;
    if (depthMax > kRefStackEntries) {
        out.reason = "tree deeper than the reference-order walker's stack (" + std::to_string(kRefStackEntries) + " levels)";
        return;
    }

    out.nnodes = (uint32_t)nodes.size();
    out.depth = depthMax;
    out.nplanes = nplanes;
    out.nodes.resize((size_t)out.nnodes * 16);
    for (uint32_t k = 0; k < out.nnodes; ++k) {
        const Node &nd = nodes[k];
        uint32_t *q = &out.nodes[(size_t)k * 16];
        for (int a = 0; a < 3; ++a) {  // AABB::calculateEverything, main.cpp:168-178
            const float c = nd.b.centre(a);
            const float size = nd.b.hi[a] - nd.b.lo[a];
            const float negSize = -size;
            q[a] = fbits(size + c);
            q[3 + a] = fbits(negSize + c);
            q[6 + a] = fbits(nd.b.lo[a]);
            q[9 + a] = fbits(nd.b.hi[a]);
        }
        q[12] = nd.left;
        q[13] = nd.right;
        q[14] = nd.first;
        q[15] = nd.count;
    }
    // what can really be hit below every node: children have larger numbers than their parent (pre-order)
    out.reach.assign((size_t)out.nnodes * 8, 0.f);
    {
        std::vector<Bounds> reach(out.nnodes);
        for (uint32_t k = out.nnodes; k-- > 0;) {
            const Node &nd = nodes[k];
            Bounds b;
            for (uint32_t i = nd.first; i < nd.first + nd.count; ++i) {
                const uint32_t p = order[i];
                if (isPlane[p]) continue;  // only at the root, which nothing skips
                b.add(p < ntri ? box[p] : analytic_bounds(desc.prims[p - ntri], true));
            }
            if (nd.left | nd.right) {
                b.add(reach[nd.left]);
                b.add(reach[nd.right]);
            }
            reach[k] = b;
            float *q = &out.reach[(size_t)k * 8];
            q[0] = b.lo[0], q[1] = b.lo[1], q[2] = b.lo[2], q[3] = b.hi[0], q[4] = b.hi[1], q[5] = b.hi[2];
        }
    }
    out.items.resize(std::max<uint32_t>(total, 1u), 0u);
    for (uint32_t i = 0; i < total; ++i) {
        const uint32_t p = order[i];
        out.items[i] = p < ntri ? recordOfTriangle[p] : nrecords + (p - ntri);
    }
    out.order.swap(order);
    out.available = true;
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace rt
