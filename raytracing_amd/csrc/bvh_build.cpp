// bvh_build.cpp -- host acceleration-structure build and flattening into the HBM layout.
//
// Replaces BVH::build (main.cpp:552-558, 608-686).  The reference sweeps every split position
// on three fully sorted axes per node, with sweep buffers sized by the whole scene (O(N^2)),
// and uses 16-bit ids.  The closest hit does not depend on the tree's shape (apart from exact
// distance ties), so this builder is free to differ: 32-bin SAH on triangle centroids,
// O(N log N), subtrees built as OpenMP tasks, 32-bit ids, leaves of at most 2 triangles (the
// reference uses 4, main.cpp:618), and a bounded height so that the kernels' fixed per-lane stack
// can never overflow.  Per-triangle derived data follows Triangle::update (main.cpp:527-536).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>

#include <omp.h>

#include "host_math.h"
#include "rt_internal.h"

namespace rt {
namespace {

const float EPS = 0.0001f;  // main.cpp:22
constexpr int kBins = 32;
constexpr uint32_t kTaskCutoff = 4096;  // ranges smaller than this are built inline
constexpr uint32_t kBfsTop = 2047;      // nodes laid out breadth-first at the front of the array

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = 3.0e38f;
            hi[k] = -3.0e38f;
        }
    }
    void grow(const Box &b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    void grow(const float *p) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    float half_area() const {
        float a = hi[0] - lo[0], b = hi[1] - lo[1], c = hi[2] - lo[2];
        if (a < 0.f || b < 0.f || c < 0.f) return 0.f;
        return a * b + a * c + b * c;
    }
};

struct Prim {
    Box box;
    float c[3];
    uint32_t id;
};

struct BNode {
    Box box[2];
    int32_t child[2];  // >= 0: BNode index; < 0: ~((first << 3) | (count - 1))
};

constexpr uint32_t kPadRecord = 0xFFFFFFFFu;  // FlatTree::order entry of a record that holds no triangle

inline int32_t leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

// Inner levels a range of n triangles needs at least (leaves hold <= kMaxLeafTris).
inline uint32_t min_levels(uint32_t n) {
    uint32_t leaves = (n + kMaxLeafTris - 1) / kMaxLeafTris, lv = 0;
    while ((1u << lv) < leaves) ++lv;
    return lv;
}

struct Builder {
    std::vector<Prim> prims;
    std::vector<BNode> nodes;
    std::atomic<uint32_t> next{0};
    std::atomic<uint32_t> capped{0};  // splits where the height bound changed the outcome

    uint32_t alloc() { return next.fetch_add(1, std::memory_order_relaxed); }

    Box bounds(uint32_t b, uint32_t e) const {
        Box r;
        r.reset();
        for (uint32_t i = b; i < e; ++i) r.grow(prims[i].box);
        return r;
    }

    // Splits [b, e) (more than kMaxLeafTris triangles) at inner level `level` (root = 1) and
    // returns the split position.
    uint32_t split(uint32_t b, uint32_t e, uint32_t level) {
        const uint32_t n = e - b;
        Box cb;
        cb.reset();
        for (uint32_t i = b; i < e; ++i) cb.grow(prims[i].c);

        float bestCost = 3.0e38f, freeCost = 3.0e38f;  // cheapest admissible candidate / cheapest of all
        int bestAxis = -1, bestBin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            float ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.f)) continue;
            Box bb[kBins];
            uint32_t cnt[kBins];
            for (int k = 0; k < kBins; ++k) {
                bb[k].reset();
                cnt[k] = 0;
            }
            const float scale = (float)kBins / ext;
            for (uint32_t i = b; i < e; ++i) {
                int k = (int)((prims[i].c[axis] - cb.lo[axis]) * scale);
                k = std::min(std::max(k, 0), kBins - 1);
                bb[k].grow(prims[i].box);
                ++cnt[k];
            }
            float rightArea[kBins];
            uint32_t rightCnt[kBins];
            Box acc;
            acc.reset();
            uint32_t c = 0;
            for (int k = kBins - 1; k > 0; --k) {
                acc.grow(bb[k]);
                c += cnt[k];
                rightArea[k] = acc.half_area();
                rightCnt[k] = c;
            }
            acc.reset();
            c = 0;
            for (int k = 0; k < kBins - 1; ++k) {
                acc.grow(bb[k]);
                c += cnt[k];
                uint32_t nl = c, nr = rightCnt[k + 1];
                if (nl == 0 || nr == 0) continue;
                float cost = acc.half_area() * (float)nl + rightArea[k + 1] * (float)nr;
                freeCost = std::min(freeCost, cost);
                // both sides must still fit under the height bound
                if (level + min_levels(nl) > kMaxTreeHeight || level + min_levels(nr) > kMaxTreeHeight) continue;
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = axis;
                    bestBin = k;
                }
            }
        }

        if (freeCost < bestCost) capped.fetch_add(1, std::memory_order_relaxed);
        if (bestAxis >= 0) {
            const float lo = cb.lo[bestAxis];
            const float scale = (float)kBins / (cb.hi[bestAxis] - lo);
            const int axis = bestAxis, bin = bestBin;
            auto mid = std::partition(prims.begin() + b, prims.begin() + e, [=](const Prim &p) {
                int k = (int)((p.c[axis] - lo) * scale);
                k = std::min(std::max(k, 0), kBins - 1);
                return k <= bin;
            });
            uint32_t m = (uint32_t)(mid - prims.begin());
            if (m > b && m < e) return m;
        }
        // No usable SAH split (coincident centroids, or the height bound forbids every
        // candidate): halve the range along the widest centroid axis.
        int axis = 0;
        for (int k = 1; k < 3; ++k)
            if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        uint32_t m = b + n / 2;
        std::nth_element(prims.begin() + b, prims.begin() + m, prims.begin() + e,
                         [axis](const Prim &x, const Prim &y) { return x.c[axis] < y.c[axis]; });
        return m;
    }

    int32_t build_child(uint32_t b, uint32_t e, uint32_t level) {
        if (e - b <= kMaxLeafTris) return leaf_ref(b, e - b);
        uint32_t id = alloc();
        build_node(id, b, e, level);
        return (int32_t)id;
    }

    void build_node(uint32_t id, uint32_t b, uint32_t e, uint32_t level) {
        uint32_t m = split(b, e, level);
        BNode &nd = nodes[id];
        nd.box[0] = bounds(b, m);
        nd.box[1] = bounds(m, e);
        if (e - b >= kTaskCutoff) {
            int32_t c0 = 0, c1 = 0;
#pragma omp task shared(c0) firstprivate(b, m, level)
            c0 = build_child(b, m, level + 1);
#pragma omp task shared(c1) firstprivate(m, e, level)
            c1 = build_child(m, e, level + 1);
#pragma omp taskwait
            nodes[id].child[0] = c0;
            nodes[id].child[1] = c1;
        } else {
            int32_t c0 = build_child(b, m, level + 1);
            int32_t c1 = build_child(m, e, level + 1);
            nodes[id].child[0] = c0;
            nodes[id].child[1] = c1;
        }
    }
};

// Conservative padding: the device slab test must never reject a box that contains a hit the
// exact-arithmetic triangle test accepts, including zero-thickness boxes of axis-aligned walls.
inline float pad_lo(float v) { return v - (1.0e-5f + std::fabs(v) * 4.0e-7f); }
inline float pad_hi(float v) { return v + (1.0e-5f + std::fabs(v) * 4.0e-7f); }

inline float bits_to_float(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

uint32_t height_of(const std::vector<BNode> &nodes, int32_t ref) {
    if (ref < 0) return 0;
    return 1 + std::max(height_of(nodes, nodes[ref].child[0]), height_of(nodes, nodes[ref].child[1]));
}


// Builds the tree over triangles which[0..n) of `verts` (9 floats each; which == nullptr: triangles
// 0..n-1) and flattens it.  out.order[k] = position in `which` of the triangle at leaf position k.
void build_tree(const float *verts, const uint32_t *which, uint32_t n, FlatTree &out) {
    Builder B;
    B.prims.resize(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const float *v = verts + 9 * (size_t)(which ? which[i] : (uint32_t)i);
        Prim &p = B.prims[i];
        p.box.reset();
        p.box.grow(v);
        p.box.grow(v + 3);
        p.box.grow(v + 6);
        for (int k = 0; k < 3; ++k) p.c[k] = 0.5f * (p.box.lo[k] + p.box.hi[k]);
        p.id = (uint32_t)i;
    }

    // Node 0 is always an inner node, so the kernels never special-case tiny scenes.
    B.nodes.resize(std::max<uint32_t>(n, 1));
    uint32_t root = B.alloc();
    Box never;
    for (int k = 0; k < 3; ++k) never.lo[k] = never.hi[k] = 3.0e30f;  // marks an absent child
    if (n <= kMaxLeafTris) {
        BNode &nd = B.nodes[root];
        nd.box[0] = n ? B.bounds(0, n) : never;
        nd.box[1] = never;
        nd.child[0] = nd.child[1] = leaf_ref(0, 1);
        if (n) nd.child[0] = leaf_ref(0, n);
    } else {
#pragma omp parallel
#pragma omp single
        B.build_node(root, 0, n, 1);
    }
    const uint32_t nnodes = B.next.load();
    const uint32_t height = height_of(B.nodes, (int32_t)root);
    if (height > kMaxTreeHeight) throw std::runtime_error("internal: BVH height bound violated");

    // Layout: the top of the tree breadth-first (those nodes are the hot, shared ones and sit
    // contiguously at the front), every subtree below it depth-first.
    std::vector<uint32_t> order;  // new index -> old index
    order.reserve(nnodes);
    std::vector<uint32_t> newIndex(nnodes, 0);
    {
        std::vector<uint32_t> frontier{root}, nextFrontier;
        while (!frontier.empty() && order.size() + frontier.size() <= kBfsTop) {
            nextFrontier.clear();
            for (uint32_t id : frontier) {
                order.push_back(id);
                for (int c = 0; c < 2; ++c)
                    if (B.nodes[id].child[c] >= 0) nextFrontier.push_back((uint32_t)B.nodes[id].child[c]);
            }
            frontier.swap(nextFrontier);
        }
        std::vector<uint32_t> stack;
        for (uint32_t sub : frontier) {
            stack.push_back(sub);
            while (!stack.empty()) {
                uint32_t id = stack.back();
                stack.pop_back();
                order.push_back(id);
                if (B.nodes[id].child[1] >= 0) stack.push_back((uint32_t)B.nodes[id].child[1]);
                if (B.nodes[id].child[0] >= 0) stack.push_back((uint32_t)B.nodes[id].child[0]);
            }
        }
    }
    if (order.size() != nnodes) throw std::runtime_error("internal: BVH layout lost nodes");
    for (uint32_t k = 0; k < nnodes; ++k) newIndex[order[k]] = k;

    out.nnodes = nnodes;
    out.height = height;
    out.capped_splits = B.capped.load();
    {   // surface-area estimate of the work per random ray: a child is entered with probability
        // area(child box) / area(root box); an inner child costs a node visit, a leaf its triangle tests
        Box rootBox;
        rootBox.reset();
        rootBox.grow(B.nodes[root].box[0]);
        if (B.nodes[root].box[1].lo[0] < 1.0e30f) rootBox.grow(B.nodes[root].box[1]);
        const double ra = std::max((double)rootBox.half_area(), 1e-30);
        double en = 1.0, et = 0.0;
        for (uint32_t k = 0; k < nnodes; ++k)
            for (int c = 0; c < 2; ++c) {
                const Box &b = B.nodes[k].box[c];
                if (!(b.lo[0] < 1.0e30f)) continue;
                const double p = (double)b.half_area() / ra;
                const int32_t ref = B.nodes[k].child[c];
                if (ref >= 0) en += p;
                else et += p * (double)(((uint32_t)~ref & 7u) + 1u);
            }
        out.sah_nodes = n ? en : 0.0;
        out.sah_tris = n ? et : 0.0;
    }
    if (!which && n > kMaxLeafTris) {
        // Main tree: a two-triangle leaf starts on an even record, so that its 2 x 64 bytes are ONE 128-byte
        // cache line (the trace kernel fetches both triangles in one pass; nearly every triangle fetch of a
        // large scene goes to memory, and a pair across two lines cost two of them: -2.8 % of the trace
        // kernel on the million-triangle scene).  A record skipped for that is a pad (kPadRecord in `order`):
        // no leaf names it and it can never be hit.  The order of the triangles does not change.
        std::vector<uint8_t> cntAt(n, 0);
        for (uint32_t k = 0; k < nnodes; ++k)
            for (int c = 0; c < 2; ++c) {
                const int32_t ref = B.nodes[k].child[c];
                if (ref < 0) cntAt[(uint32_t)~ref >> 3] = (uint8_t)(((uint32_t)~ref & 7u) + 1u);
            }
        std::vector<uint32_t> newPos(n);
        uint32_t pos = 0;
        for (uint32_t p = 0; p < n;) {
            const uint32_t cnt = cntAt[p];
            if (cnt == 0) throw std::runtime_error("internal: leaves do not tile the triangle range");
            if (cnt == 2 && (pos & 1u)) ++pos;
            for (uint32_t j = 0; j < cnt; ++j) newPos[p + j] = pos + j;
            pos += cnt;
            p += cnt;
        }
        for (uint32_t k = 0; k < nnodes; ++k)
            for (int c = 0; c < 2; ++c) {
                const int32_t ref = B.nodes[k].child[c];
                if (ref < 0) B.nodes[k].child[c] = leaf_ref(newPos[(uint32_t)~ref >> 3], ((uint32_t)~ref & 7u) + 1u);
            }
        out.order.assign(pos, kPadRecord);
        for (uint32_t k = 0; k < n; ++k) out.order[newPos[k]] = B.prims[k].id;
    } else {
        out.order.resize(n);
        for (uint32_t k = 0; k < n; ++k) out.order[k] = B.prims[k].id;
    }

    // Node boxes go to the device on a 16-bit grid spanning the (padded) bounds of the triangles:
    //   plane = grid_origin + q * grid_cell,  q in [0, 65535],
    // rounded outward, plus two guard cells per side.  The kernels evaluate a plane distance as one FMA
    // with per-ray constants cell/d and (origin - o)/d and widen each slab by the rounding bound of that
    // arithmetic; the guard cells are on top of it.
    auto real = [](const Box &b) { return b.lo[0] <= b.hi[0] && b.lo[0] < 1.0e30f; };
    float glo[3] = {3.0e30f, 3.0e30f, 3.0e30f}, ghi[3] = {-3.0e30f, -3.0e30f, -3.0e30f};
    for (uint32_t k = 0; k < nnodes; ++k)
        for (int c = 0; c < 2; ++c) {
            const Box &b = B.nodes[k].box[c];
            if (!real(b)) continue;
            for (int a = 0; a < 3; ++a) {
                glo[a] = std::min(glo[a], pad_lo(b.lo[a]));
                ghi[a] = std::max(ghi[a], pad_hi(b.hi[a]));
            }
        }
    for (int a = 0; a < 3; ++a) {
        if (glo[a] > ghi[a]) glo[a] = 0.f, ghi[a] = 1.f;  // no triangles at all
        const float mag = std::max(1.f, std::max(std::fabs(glo[a]), std::fabs(ghi[a])));
        const float ext = std::max(ghi[a] - glo[a], 1.0e-3f * mag);  // flat scenes still get a usable cell
        out.grid_cell[a] = ext / 65000.f;
        out.grid_origin[a] = glo[a] - 16.f * out.grid_cell[a];
    }
    auto quantise = [&](float v, int a, bool up) -> uint32_t {
        const double o = out.grid_origin[a], c = out.grid_cell[a];
        long q = (long)(up ? std::ceil(((double)v - o) / c) : std::floor(((double)v - o) / c));
        if (up) {
            while (o + (double)q * c < (double)v) ++q;
            q += 2;
        } else {
            while (o + (double)q * c > (double)v) --q;
            q -= 2;
        }
        if (q < 0 || q > 65535) throw std::runtime_error("internal: box outside the node grid");
        return (uint32_t)q;
    };
    out.nodes.assign((size_t)nnodes * 8, 0u);
    for (uint32_t k = 0; k < nnodes; ++k) {
        const BNode &nd = B.nodes[order[k]];
        uint32_t *w = &out.nodes[(size_t)k * 8];
        for (int c = 0; c < 2; ++c) {
            const Box &b = nd.box[c];
            for (int a = 0; a < 3; ++a) {
                // an absent child: a point in the far corner of the grid, outside every real box
                const uint32_t lo = real(b) ? quantise(pad_lo(b.lo[a]), a, false) : 65535u;
                const uint32_t hi = real(b) ? quantise(pad_hi(b.hi[a]), a, true) : 65535u;
                w[3 * c + a] = lo | (hi << 16);
            }
            w[6 + c] = (uint32_t)(nd.child[c] >= 0 ? (int32_t)newIndex[nd.child[c]] : nd.child[c]);
        }
    }
}

// The flattened two-child tree with up to four children per node (layout: rt_internal.h).  Depth-first
// order: a node is followed by the subtrees of its children.
struct WideBuilder {
    const std::vector<uint32_t> &n2;
    const float *cell;
    std::vector<uint32_t> out;

    float area(const uint32_t *w) const {
        float e[3];
        for (int a = 0; a < 3; ++a) e[a] = (float)((w[a] >> 16) - (w[a] & 0xffffu)) * cell[a];
        return e[0] * e[1] + e[0] * e[2] + e[1] * e[2];
    }
    // returns the wide index of two-child node `id`
    uint32_t emit(uint32_t id) {
        uint32_t box[4][3];
        int32_t ref[4];
        int n = 2;
        for (int c = 0; c < 2; ++c) {
            std::memcpy(box[c], &n2[(size_t)id * 8 + 3 * c], 12);
            ref[c] = (int32_t)n2[(size_t)id * 8 + 6 + c];
        }
        while (n < 4) {
            int best = -1;
            float bestA = -1.f;
            for (int c = 0; c < n; ++c) {
                if (ref[c] < 0) continue;
                const float a = area(box[c]);
                if (a > bestA) bestA = a, best = c;
            }
            if (best < 0) break;
            const uint32_t y = (uint32_t)ref[best];
            std::memcpy(box[best], &n2[(size_t)y * 8], 12);
            ref[best] = (int32_t)n2[(size_t)y * 8 + 6];
            std::memcpy(box[n], &n2[(size_t)y * 8 + 3], 12);
            ref[n] = (int32_t)n2[(size_t)y * 8 + 7];
            ++n;
        }
        const uint32_t me = (uint32_t)(out.size() / 12);
        out.resize(out.size() + 12);
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t org[3], shift[3];
        for (int a = 0; a < 3; ++a) {
            uint32_t lo = 65535u, hi = 0u;
            for (int c = 0; c < n; ++c) {
                lo = std::min(lo, box[c][a] & 0xffffu);
                hi = std::max(hi, box[c][a] >> 16);
            }
            uint32_t s = 0;
            while (((hi - lo + (1u << s) - 1u) >> s) > 255u) ++s;
            org[a] = lo;
            shift[a] = s;
            for (int c = 0; c < 4; ++c) {
                uint32_t ql = 255u, qh = 0u;  // an unused child: inverted (and masked by the child count)
                if (c < n) {
                    ql = ((box[c][a] & 0xffffu) - lo) >> s;
                    qh = ((box[c][a] >> 16) - lo + (1u << s) - 1u) >> s;
                    if (qh > 255u || ql > qh) throw std::runtime_error("internal: wide node quantisation");
                }
                w[2 * a] |= ql << (8 * c);
                w[2 * a + 1] |= qh << (8 * c);
            }
        }
        w[6] = org[0] | (org[1] << 16);
        w[7] = org[2] | (shift[0] << 16) | (shift[1] << 20) | (shift[2] << 24) | ((uint32_t)n << 28);
        for (int k = 0; k < 8; ++k) out[(size_t)me * 12 + k] = w[k];
        for (int c = 0; c < 4; ++c) {
            uint32_t r = 0x7FFFFFFFu;
            if (c < n) r = ref[c] < 0 ? (uint32_t)ref[c] : emit((uint32_t)ref[c]);
            out[(size_t)me * 12 + 8 + c] = r;
        }
        return me;
    }
};

// Triangle record (16 floats) of loader triangle `id`: Triangle::update (main.cpp:527-536).
void triangle_record(const rt_scene_desc &desc, uint32_t id, float *q) {
    const float *v = desc.tri_verts + 9 * (size_t)id;
    hm::vec3 A = hm::V3(v[0], v[1], v[2]), Bv = hm::V3(v[3], v[4], v[5]), C = hm::V3(v[6], v[7], v[8]);
    hm::vec3 AB = Bv - A, AC = C - A;
    hm::vec3 N = hm::cross(AB, AC);
    float pdfConst = 2.f / hm::length(N);
    q[0] = A.x; q[1] = A.y; q[2] = A.z; q[3] = AB.x;
    q[4] = AB.y; q[5] = AB.z; q[6] = AC.x; q[7] = AC.y;
    q[8] = AC.z; q[9] = N.x; q[10] = N.y; q[11] = N.z;
    q[12] = bits_to_float(id);
    q[13] = bits_to_float(desc.tri_material[id]);
    q[14] = pdfConst;
    q[15] = 0.f;
}

}  // namespace

void build_flat_scene(const rt_scene_desc &desc, FlatScene &out) {
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = desc.ntri;
    if (n > kMaxTriangles)
        throw std::runtime_error("too many triangles (limit " + std::to_string(kMaxTriangles) + " = " +
                                 std::to_string(kMaxLeafTris) + " per leaf under " + std::to_string(kMaxTreeHeight) +
                                 " inner levels, the kernels' traversal stack)");
    if (n && (!desc.tri_verts || !desc.tri_material)) throw std::runtime_error("null triangle arrays");
    if (desc.nmat && !desc.materials) throw std::runtime_error("null material table");
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(desc.background[k]) || desc.background[k] < 0.f)
            throw std::runtime_error("background colour components must be finite and non-negative");
    for (uint32_t i = 0; i < n; ++i)
        if (desc.tri_material[i] >= desc.nmat) throw std::runtime_error("triangle names a missing material");
    for (size_t i = 0; i < (size_t)n * 9; ++i)
        if (!std::isfinite(desc.tri_verts[i])) throw std::runtime_error("non-finite vertex coordinate");

    // Analytic primitives (main.cpp:242-464): checked and packed; they stay outside the tree.
    if (desc.nprim > RT_MAX_PRIMITIVES) throw std::runtime_error("too many analytic primitives (limit " + std::to_string(RT_MAX_PRIMITIVES) + ")");
    if (desc.nprim && !desc.prims) throw std::runtime_error("null primitive array");
    out.nprims = desc.nprim;
    out.prims.assign((size_t)std::max<uint32_t>(desc.nprim, 1) * 12, 0.f);
    for (uint32_t k = 0; k < desc.nprim; ++k) {
        const rt_primitive &p = desc.prims[k];
        if (p.type != RT_PRIM_PLANE && p.type != RT_PRIM_ELLIPSOID && p.type != RT_PRIM_BOX)
            throw std::runtime_error("analytic primitive type must be RT_PRIM_PLANE, _ELLIPSOID or _BOX");
        if (p.material >= desc.nmat) throw std::runtime_error("analytic primitive names a missing material");
        const rt_material &m = desc.materials[p.material];
        if (p.type == RT_PRIM_PLANE && hm::length(hm::V3(m.emission[0], m.emission[1], m.emission[2])) > EPS)
            throw std::runtime_error("a plane must not emit: it cannot be sampled (Plane::samplePoint throws, main.cpp:261-267)");
        float q2 = 0.f, n2 = 0.f;
        for (int a = 0; a < 4; ++a) {
            if (!std::isfinite(p.rotation[a])) throw std::runtime_error("non-finite primitive rotation");
            q2 += p.rotation[a] * p.rotation[a];
        }
        if (std::fabs(q2 - 1.f) > 1.0e-3f) throw std::runtime_error("primitive rotation must be a unit quaternion");
        for (int a = 0; a < 3; ++a) {
            if (!std::isfinite(p.position[a]) || !std::isfinite(p.param[a])) throw std::runtime_error("non-finite primitive coordinate");
            if (p.type != RT_PRIM_PLANE && !(p.param[a] > 0.f)) throw std::runtime_error("ellipsoid radii and box extents must be positive");
            n2 += p.param[a] * p.param[a];
        }
        if (p.type == RT_PRIM_PLANE && !(n2 > 0.f)) throw std::runtime_error("plane normal must not be zero");
        float *q = &out.prims[(size_t)k * 12];
        q[0] = p.position[0]; q[1] = p.position[1]; q[2] = p.position[2]; q[3] = bits_to_float((uint32_t)p.type);
        q[4] = p.rotation[0]; q[5] = p.rotation[1]; q[6] = p.rotation[2]; q[7] = p.rotation[3];
        q[8] = p.param[0]; q[9] = p.param[1]; q[10] = p.param[2]; q[11] = bits_to_float(p.material);
    }

    FlatTree tree;
    build_tree(desc.tri_verts, nullptr, n, tree);
    out.nnodes = tree.nnodes;
    out.ntri = n;
    out.tree_height = tree.height;
    out.capped_splits = tree.capped_splits;
    out.sah_nodes = tree.sah_nodes;
    out.sah_tris = tree.sah_tris;
    out.nodes.swap(tree.nodes);
    for (int a = 0; a < 3; ++a) {
        out.grid_origin[a] = tree.grid_origin[a];
        out.grid_cell[a] = tree.grid_cell[a];
    }
    {
        WideBuilder wb{out.nodes, out.grid_cell, {}};
        wb.out.reserve((size_t)out.nnodes * 10);
        wb.emit(0);
        out.wide.swap(wb.out);
        out.nwide = (uint32_t)(out.wide.size() / 12);
    }

    // Triangles in leaf order, with the derived data of Triangle::update (main.cpp:527-536).
    const uint32_t nslots = (uint32_t)tree.order.size();
    out.nslots = nslots;
    out.tris.assign((size_t)std::max<uint32_t>(nslots, 1) * 16, 0.f);
    std::vector<uint32_t> posOfId(n);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < (int64_t)nslots; ++k) {
        const uint32_t id = tree.order[k];
        float *q = &out.tris[(size_t)k * 16];
        if (id == kPadRecord) {  // never hit, like the dummy of an empty scene below
            q[0] = q[1] = q[2] = 3.0e30f;
            q[12] = bits_to_float(RT_MISS_ID);
            continue;
        }
        posOfId[id] = (uint32_t)k;
        triangle_record(desc, id, q);
    }
    if (n == 0) {  // one never-hit dummy so leaf_ref(0, 1) stays in bounds
        float *q = &out.tris[0];
        q[0] = q[1] = q[2] = 3.0e30f;
        q[12] = bits_to_float(RT_MISS_ID);
    }

    // Light list in loader order (main.cpp:1096-1098): length(emission) > EPS.
    out.lights.clear();
    std::vector<uint32_t> lightIds;
    for (uint32_t id = 0; id < n; ++id) {
        const rt_material &m = desc.materials[desc.tri_material[id]];
        if (hm::length(hm::V3(m.emission[0], m.emission[1], m.emission[2])) > EPS) {
            out.lights.push_back(posOfId[id]);
            lightIds.push_back(id);
        }
    }
    out.ntrilights = (uint32_t)out.lights.size();
    // ... then the emissive analytic primitives in list order, as record indices behind the triangle records
    for (uint32_t k = 0; k < desc.nprim; ++k) {
        const rt_material &m = desc.materials[desc.prims[k].material];
        if (hm::length(hm::V3(m.emission[0], m.emission[1], m.emission[2])) > EPS) out.lights.push_back(nslots + k);
    }

    // A second tree over the light triangles alone, for LightSurface::pdf (main.cpp:771-783) when the
    // lights are too many to loop over.  Its records repeat the triangle data in that tree's leaf order;
    // record word 12 = index in the light list, word 13 = leaf position in the main triangle array.
    const uint32_t nl = (uint32_t)lightIds.size();
    FlatTree ltree;
    build_tree(desc.tri_verts, lightIds.data(), nl, ltree);
    out.light_nodes.swap(ltree.nodes);
    out.light_nnodes = ltree.nnodes;
    out.light_height = ltree.height;
    for (int a = 0; a < 3; ++a) {
        out.light_grid_origin[a] = ltree.grid_origin[a];
        out.light_grid_cell[a] = ltree.grid_cell[a];
    }
    out.light_recs.assign((size_t)std::max<uint32_t>(nl, 1) * 16, 0.f);
    for (uint32_t k = 0; k < nl; ++k) {
        const uint32_t li = ltree.order[k];
        float *q = &out.light_recs[(size_t)k * 16];
        triangle_record(desc, lightIds[li], q);
        q[12] = bits_to_float(li);
        q[13] = bits_to_float(out.lights[li]);
    }
    if (nl == 0) out.light_recs[0] = out.light_recs[1] = out.light_recs[2] = 3.0e30f;  // never hit
    // The reference's own tree, for the kernels that resolve order-dependent hits as it does (scenes within its limits)
    build_ref_tree(desc, posOfId, nslots, out.ref);
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace rt

// ---- host-only C entry points over the builder (declared in include/rt_amd.h) -----------------
struct rt_flat {
    rt::FlatScene flat;
};

extern "C" {

int rt_flat_build(const rt_scene_desc *desc, rt_flat **out) {
    if (!desc || !out) {
        rt::set_error("rt_flat_build: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    rt_flat *f = new rt_flat();
    try {
        rt::build_flat_scene(*desc, f->flat);
    } catch (const std::exception &e) {
        delete f;
        std::string msg = e.what();
        rt::set_error("rt_flat_build: " + msg);
        return msg.find("too many") != std::string::npos ? RT_ERR_LIMIT : RT_ERR_INVALID;
    }
    *out = f;
    return RT_OK;
}

int rt_flat_get(const rt_flat *f, uint32_t info[4], float grid[6], uint32_t *nodes, float *tris, uint32_t *lights) {
    if (!f) {
        rt::set_error("rt_flat_get: null argument");
        return RT_ERR_INVALID;
    }
    const rt::FlatScene &s = f->flat;
    if (info) {
        info[0] = s.nnodes;
        info[1] = s.nslots;
        info[2] = (uint32_t)s.lights.size();
        info[3] = s.tree_height;
    }
    if (grid)
        for (int a = 0; a < 3; ++a) {
            grid[a] = s.grid_origin[a];
            grid[3 + a] = s.grid_cell[a];
        }
    if (nodes) std::memcpy(nodes, s.nodes.data(), (size_t)s.nnodes * 8 * sizeof(uint32_t));
    if (tris) std::memcpy(tris, s.tris.data(), (size_t)s.nslots * 16 * sizeof(float));
    if (lights && !s.lights.empty()) std::memcpy(lights, s.lights.data(), s.lights.size() * sizeof(uint32_t));
    return RT_OK;
}

int rt_flat_get_light_tree(const rt_flat *f, uint32_t info[2], float grid[6], uint32_t *nodes, float *recs) {
    if (!f) {
        rt::set_error("rt_flat_get_light_tree: null argument");
        return RT_ERR_INVALID;
    }
    const rt::FlatScene &s = f->flat;
    if (info) {
        info[0] = s.light_nnodes;
        info[1] = s.light_height;
    }
    if (grid)
        for (int a = 0; a < 3; ++a) {
            grid[a] = s.light_grid_origin[a];
            grid[3 + a] = s.light_grid_cell[a];
        }
    if (nodes) std::memcpy(nodes, s.light_nodes.data(), (size_t)s.light_nnodes * 8 * sizeof(uint32_t));
    if (recs) std::memcpy(recs, s.light_recs.data(), (size_t)s.ntrilights * 16 * sizeof(float));
    return RT_OK;
}

int rt_flat_get_wide(const rt_flat *f, uint32_t *nwide, uint32_t *nodes) {
    if (!f) {
        rt::set_error("rt_flat_get_wide: null argument");
        return RT_ERR_INVALID;
    }
    if (nwide) *nwide = f->flat.nwide;
    if (nodes) std::memcpy(nodes, f->flat.wide.data(), f->flat.wide.size() * sizeof(uint32_t));
    return RT_OK;
}

int rt_flat_get_ref_tree(const rt_flat *f, uint32_t info[4], uint32_t *nodes, uint32_t *order) {
    if (!f) {
        rt::set_error("rt_flat_get_ref_tree: null argument");
        return RT_ERR_INVALID;
    }
    const rt::RefTree &r = f->flat.ref;
    if (info) {
        info[0] = r.available ? 1u : 0u;
        info[1] = r.available ? r.nnodes : 0u;
        info[2] = r.available ? r.depth : 0u;
        info[3] = r.available ? r.nplanes : 0u;
    }
    if (!r.available) {
        rt::set_error("rt_flat_get_ref_tree: " + r.reason);
        return (nodes || order) ? RT_ERR_LIMIT : RT_OK;
    }
    if (nodes) std::memcpy(nodes, r.nodes.data(), r.nodes.size() * sizeof(uint32_t));
    if (order && !r.order.empty()) std::memcpy(order, r.order.data(), r.order.size() * sizeof(uint32_t));
    return RT_OK;
}

int rt_flat_get_quality(const rt_flat *f, uint32_t *height_capped_splits, double *sah_nodes, double *sah_tris) {
    if (!f) {
        rt::set_error("rt_flat_get_quality: null argument");
        return RT_ERR_INVALID;
    }
    if (height_capped_splits) *height_capped_splits = f->flat.capped_splits;
    if (sah_nodes) *sah_nodes = f->flat.sah_nodes;
    if (sah_tris) *sah_tris = f->flat.sah_tris;
    return RT_OK;
}

void rt_flat_free(rt_flat *f) { delete f; }

}  // extern "C"
