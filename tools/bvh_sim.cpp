// bvh_sim.cpp -- CPU counting model of the traversal on the flattened tree the library builds: how many node
// visits and ray/triangle tests a closest-hit walk costs per ray on the two-child tree and on four-child
// trees collapsed from it under different stack budgets.  Host only (no GPU call).
//
//   g++ -O2 -fopenmp -std=gnu++17 -I include tools/bvh_sim.cpp -o /tmp/bvh_sim -L raytracing_amd/lib -lrt_amd -Wl,-rpath,$PWD/raytracing_amd/lib
//   /tmp/bvh_sim scene.gltf [rays per bounce, default 200000] [summary.json]
// summary.json: the per-bounce counts of the two-child tree and of the four-child tree with 8-bit boxes (the two
// the kernels walk), for bench.py to print beside the device's own counts.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "rt_amd.h"

struct Box { float lo[3], hi[3]; };
struct Ray { float o[3], d[3]; };

static std::vector<uint32_t> g_nodes;  // 8 words per node
static std::vector<float> g_tris;      // 16 floats per triangle
static float g_grid[6];

static Box child_box(uint32_t node, int c) {
    Box b;
    for (int a = 0; a < 3; ++a) {
        uint32_t w = g_nodes[(size_t)node * 8 + 3 * c + a];
        b.lo[a] = g_grid[a] + (float)(w & 0xffffu) * g_grid[3 + a];
        b.hi[a] = g_grid[a] + (float)(w >> 16) * g_grid[3 + a];
    }
    return b;
}
static int32_t child_ref(uint32_t node, int c) { return (int32_t)g_nodes[(size_t)node * 8 + 6 + c]; }

static bool slab(const Box &b, const Ray &r, const float *inv, float tBest, float &tn) {
    float t0 = 0.f, t1 = tBest;
    for (int a = 0; a < 3; ++a) {
        float x0 = (b.lo[a] - r.o[a]) * inv[a], x1 = (b.hi[a] - r.o[a]) * inv[a];
        if (x0 > x1) std::swap(x0, x1);
        t0 = std::max(t0, x0);
        t1 = std::min(t1, x1);
    }
    tn = t0;
    return t0 <= t1;
}

static bool tri_hit(uint32_t tri, const Ray &r, float &t) {
    const float *q = &g_tris[(size_t)tri * 16];
    const float A[3] = {q[0], q[1], q[2]}, AB[3] = {q[3], q[4], q[5]}, AC[3] = {q[6], q[7], q[8]};
    float p[3] = {r.d[1] * AC[2] - r.d[2] * AC[1], r.d[2] * AC[0] - r.d[0] * AC[2], r.d[0] * AC[1] - r.d[1] * AC[0]};
    float det = AB[0] * p[0] + AB[1] * p[1] + AB[2] * p[2];
    if (std::fabs(det) < 1e-20f) return false;
    float inv = 1.f / det;
    float s[3] = {r.o[0] - A[0], r.o[1] - A[1], r.o[2] - A[2]};
    float u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
    if (u < 0.f || u > 1.f) return false;
    float qv[3] = {s[1] * AB[2] - s[2] * AB[1], s[2] * AB[0] - s[0] * AB[2], s[0] * AB[1] - s[1] * AB[0]};
    float v = (r.d[0] * qv[0] + r.d[1] * qv[1] + r.d[2] * qv[2]) * inv;
    if (v < 0.f || u + v > 1.f) return false;
    t = (AC[0] * qv[0] + AC[1] * qv[1] + AC[2] * qv[2]) * inv;
    return t > 1e-4f;
}

// ---- wide tree collapsed from the two-child one ---------------------------------------------------
struct WNode {
    int n;           // children in use
    Box box[4];
    int32_t ref[4];  // >= 0: WNode index, < 0: leaf reference
};
static std::vector<WNode> g_wide;
static std::vector<int> g_h2;  // two-child height (inner levels) of every two-child node

static int h2_of(int32_t ref) { return ref < 0 ? 0 : g_h2[ref]; }
static int32_t maybe_merge(int32_t ref);
static int calc_h2(uint32_t node) {
    int h = 1 + std::max(child_ref(node, 0) >= 0 ? calc_h2((uint32_t)child_ref(node, 0)) : 0,
                         child_ref(node, 1) >= 0 ? calc_h2((uint32_t)child_ref(node, 1)) : 0);
    g_h2[node] = h;
    return h;
}
static float area(const Box &b) {
    float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return x * y + y * z + x * z;
}

// mode 0: top-down greedy, largest child first, within the budget; mode 1: the same but a child is opened only
// if its own two-child height is at most `low` (spend the budget near the leaves)
static int g_mode = 0, g_low = 0, g_maxArity = 4, g_order = 0;
static int g_qbits = 0;  // > 0: child boxes re-quantised to this many bits in a frame of the node (origin + shift per axis)

static void requantise(WNode &w) {
    const uint32_t top = (1u << g_qbits) - 1u;
    for (int a = 0; a < 3; ++a) {
        uint32_t lo[4], hi[4], mn = 65535u, mx = 0u;
        for (int c = 0; c < w.n; ++c) {
            lo[c] = (uint32_t)std::lround((w.box[c].lo[a] - g_grid[a]) / g_grid[3 + a]);
            hi[c] = (uint32_t)std::lround((w.box[c].hi[a] - g_grid[a]) / g_grid[3 + a]);
            mn = std::min(mn, lo[c]);
            mx = std::max(mx, hi[c]);
        }
        uint32_t s = 0;
        while (((mx - mn + (1u << s) - 1u) >> s) > top) ++s;
        for (int c = 0; c < w.n; ++c) {
            uint32_t ql = (lo[c] - mn) >> s, qh = (hi[c] - mn + (1u << s) - 1u) >> s;
            w.box[c].lo[a] = g_grid[a] + (float)(mn + (ql << s)) * g_grid[3 + a];
            w.box[c].hi[a] = g_grid[a] + (float)(mn + (qh << s)) * g_grid[3 + a];
        }
    }
}
static long g_stackNeedMax = 0;

static int32_t collapse(uint32_t node2, int budget, int used) {
    WNode w;
    w.n = 2;
    for (int c = 0; c < 2; ++c) {
        w.box[c] = child_box(node2, c);
        w.ref[c] = maybe_merge(child_ref(node2, c));
    }
    for (;;) {
        if (w.n >= g_maxArity) break;
        int best = -1;
        float bestA = -1.f;
        for (int c = 0; c < w.n; ++c) {
            if (w.ref[c] < 0) continue;
            if (g_mode == 1 && g_h2[w.ref[c]] > g_low) continue;
            // after opening child c: arity n+1, every child needs budget - n >= its two-child height
            int maxh = 0;
            for (int k = 0; k < w.n; ++k)
                if (k != c) maxh = std::max(maxh, h2_of(w.ref[k]));
            maxh = std::max(maxh, std::max(h2_of(child_ref((uint32_t)w.ref[c], 0)), h2_of(child_ref((uint32_t)w.ref[c], 1))));
            if (budget - w.n < maxh) continue;
            float a = area(w.box[c]);
            if (a > bestA) bestA = a, best = c;
        }
        if (best < 0) break;
        uint32_t y = (uint32_t)w.ref[best];
        w.box[best] = child_box(y, 0);
        w.ref[best] = maybe_merge(child_ref(y, 0));
        w.box[w.n] = child_box(y, 1);
        w.ref[w.n] = maybe_merge(child_ref(y, 1));
        ++w.n;
    }
    g_stackNeedMax = std::max<long>(g_stackNeedMax, used + w.n - 1);
    if (g_qbits) requantise(w);
    int32_t id = (int32_t)g_wide.size();
    g_wide.push_back(w);
    for (int c = 0; c < w.n; ++c)
        if (w.ref[c] >= 0) {
            int32_t r = collapse((uint32_t)w.ref[c], budget - (w.n - 1), used + w.n - 1);
            g_wide[id].ref[c] = r;
        }
    return id;
}

struct Counts { double nodes = 0, tris = 0, boxes = 0, leaves = 0; long maxsp = 0; };

// Leaves merged for the "up to g_leafMax triangles per leaf" model: a two-child subtree that holds no more
// than that becomes one leaf (one fetch of a line of packed geometry records); its reference is kMergedBase - index.
static int g_leafMax = 0;
static const int32_t kMergedBase = -(1 << 30);
static std::vector<std::vector<int32_t>> g_merged;
static std::vector<int> g_ntri2;  // triangles below every two-child node
static int ntri_of(int32_t ref) { return ref < 0 ? (int)((~(uint32_t)ref & 7u) + 1u) : g_ntri2[ref]; }
static int calc_ntri(uint32_t node) {
    int n = 0;
    for (int c = 0; c < 2; ++c) {
        const int32_t r = child_ref(node, c);
        n += r >= 0 ? calc_ntri((uint32_t)r) : ntri_of(r);
    }
    g_ntri2[node] = n;
    return n;
}
static void gather_leaves(int32_t ref, std::vector<int32_t> &out) {
    if (ref < 0) { out.push_back(ref); return; }
    gather_leaves(child_ref((uint32_t)ref, 0), out);
    gather_leaves(child_ref((uint32_t)ref, 1), out);
}
static int32_t maybe_merge(int32_t ref) {
    if (ref < 0 || !g_leafMax || g_ntri2[ref] > g_leafMax) return ref;
    g_merged.emplace_back();
    gather_leaves(ref, g_merged.back());
    return kMergedBase - (int32_t)(g_merged.size() - 1);
}

static void leaf_tests(int32_t ref, const Ray &r, float &tBest, int &best, Counts &c) {
    if (ref <= kMergedBase) {
        c.leaves -= (double)g_merged[(size_t)(kMergedBase - ref)].size() - 1.0;  // one line, not one per part
        for (int32_t part : g_merged[(size_t)(kMergedBase - ref)]) leaf_tests(part, r, tBest, best, c);
        return;
    }
    c.leaves += 1;
    uint32_t u = ~(uint32_t)ref, first = u >> 3, n = (u & 7u) + 1u;
    for (uint32_t k = 0; k < n; ++k) {
        float t;
        c.tris += 1;
        if (tri_hit(first + k, r, t) && t < tBest) tBest = t, best = (int)(first + k);
    }
}

static int trace2(const Ray &r, float &tBest, Counts &c) {
    float inv[3];
    for (int a = 0; a < 3; ++a) inv[a] = 1.f / (std::fabs(r.d[a]) < 1e-30f ? 1e-30f : r.d[a]);
    int32_t stack[64];
    int sp = 0, best = -1;
    int32_t cur = 0;
    tBest = 1e4f;
    for (;;) {
        if (cur < 0) {
            leaf_tests(cur, r, tBest, best, c);
        } else {
            c.nodes += 1;
            c.boxes += 2;
            float tn[2];
            bool h[2];
            for (int k = 0; k < 2; ++k) h[k] = slab(child_box((uint32_t)cur, k), r, inv, tBest, tn[k]);
            if (h[0] && h[1]) {
                int f = tn[0] <= tn[1] ? 0 : 1;
                stack[sp++] = child_ref((uint32_t)cur, 1 - f);
                c.maxsp = std::max<long>(c.maxsp, sp);
                cur = child_ref((uint32_t)cur, f);
                continue;
            } else if (h[0] || h[1]) {
                cur = child_ref((uint32_t)cur, h[0] ? 0 : 1);
                continue;
            }
        }
        if (!sp) break;
        cur = stack[--sp];
    }
    return best;
}

static int trace4(const Ray &r, float &tBest, Counts &c) {
    float inv[3];
    for (int a = 0; a < 3; ++a) inv[a] = 1.f / (std::fabs(r.d[a]) < 1e-30f ? 1e-30f : r.d[a]);
    int32_t stack[128];
    int sp = 0, best = -1;
    int32_t cur = 0;
    tBest = 1e4f;
    for (;;) {
        if (cur < 0) {
            leaf_tests(cur, r, tBest, best, c);
        } else {
            const WNode &w = g_wide[cur];
            c.nodes += 1;
            c.boxes += w.n;
            float tn[4];
            int idx[4], n = 0;
            for (int k = 0; k < w.n; ++k) {
                float t;
                if (slab(w.box[k], r, inv, tBest, t)) tn[n] = t, idx[n++] = k;
            }
            if (n) {
                if (g_order == 0) {
                    for (int i = 1; i < n; ++i)  // insertion sort, nearest first
                        for (int j = i; j > 0 && tn[j] < tn[j - 1]; --j) std::swap(tn[j], tn[j - 1]), std::swap(idx[j], idx[j - 1]);
                } else {  // only the nearest is found; the others keep their storage order
                    int m = 0;
                    for (int i = 1; i < n; ++i)
                        if (tn[i] < tn[m]) m = i;
                    const float tm = tn[m];
                    const int im = idx[m];
                    for (int i = m; i > 0; --i) tn[i] = tn[i - 1], idx[i] = idx[i - 1];
                    tn[0] = tm, idx[0] = im;
                }
                for (int i = n - 1; i >= 1; --i) stack[sp++] = w.ref[idx[i]];
                c.maxsp = std::max<long>(c.maxsp, sp);
                cur = w.ref[idx[0]];
                continue;
            }
        }
        if (!sp) break;
        cur = stack[--sp];
    }
    return best;
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const int nrays = argc > 2 ? atoi(argv[2]) : 200000;
    rt_host_scene *hs = nullptr;
    if (rt_gltf_load(argv[1], 1920, 1080, &hs) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    rt_scene_desc desc;
    memset(&desc, 0, sizeof desc);
    rt_host_scene_get_desc(hs, &desc);
    rt_flat *flat = nullptr;
    if (rt_flat_build(&desc, &flat) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    uint32_t info[4];
    rt_flat_get(flat, info, g_grid, nullptr, nullptr, nullptr);
    g_nodes.resize((size_t)info[0] * 8);
    g_tris.resize((size_t)info[1] * 16);
    rt_flat_get(flat, nullptr, nullptr, g_nodes.data(), g_tris.data(), nullptr);
    printf("nodes %u tris %u height %u\n", info[0], info[1], info[3]);
    g_h2.assign(info[0], 0);
    calc_h2(0);
    g_ntri2.assign(info[0], 0);
    calc_ntri(0);

    // rays: camera rays, then three bounces off the hits in cosine-distributed directions
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<std::vector<Ray>> gen(4);
    const rt_camera &cam = desc.camera;
    for (int i = 0; i < nrays; ++i) {
        float x = (2.f * U(rng) - 1.f) * cam.tan_x, y = -(2.f * U(rng) - 1.f) * cam.tan_y;
        Ray r;
        float l = 0.f;
        for (int a = 0; a < 3; ++a) {
            r.o[a] = cam.position[a];
            r.d[a] = x * cam.right[a] + y * cam.up[a] + cam.forward[a];
            l += r.d[a] * r.d[a];
        }
        for (int a = 0; a < 3; ++a) r.d[a] /= std::sqrt(l);
        gen[0].push_back(r);
    }
    for (int b = 0; b < 3; ++b) {
        Counts dummy;
        for (const Ray &r : gen[b]) {
            float t;
            int tri = trace2(r, t, dummy);
            if (tri < 0) continue;
            const float *q = &g_tris[(size_t)tri * 16];
            float n[3] = {q[9], q[10], q[11]};
            float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            float dn = 0.f;
            for (int a = 0; a < 3; ++a) n[a] /= l, dn += n[a] * r.d[a];
            if (dn > 0.f) for (int a = 0; a < 3; ++a) n[a] = -n[a];
            float g[3], gl = 0.f;
            do {
                gl = 0.f;
                for (int a = 0; a < 3; ++a) g[a] = 2.f * U(rng) - 1.f, gl += g[a] * g[a];
            } while (gl > 1.f || gl < 1e-6f);
            Ray nr;
            float dl = 0.f;
            for (int a = 0; a < 3; ++a) {
                nr.o[a] = r.o[a] + t * r.d[a] + n[a] * 1e-4f;
                nr.d[a] = g[a] / std::sqrt(gl) + n[a] * 1.0001f;
                dl += nr.d[a] * nr.d[a];
            }
            for (int a = 0; a < 3; ++a) nr.d[a] /= std::sqrt(dl);
            gen[b + 1].push_back(nr);
        }
    }

    std::string json;
    auto report = [&](const char *name, bool wide, const char *key = nullptr) {
        printf("%-34s", name);
        if (key) json += std::string(json.empty() ? "" : ", ") + "\"" + key + "\": [";
        for (int b = 0; b < 4; ++b) {
            Counts c;
            long mism = 0;
            for (const Ray &r : gen[b]) {
                float t;
                int tri = wide ? trace4(r, t, c) : trace2(r, t, c);
                (void)tri;
                (void)mism;
            }
            double n = (double)gen[b].size();
            printf(" | b%d: %5.1f nodes %5.1f boxes %4.1f tris %4.2f leaves sp %ld", b, c.nodes / n, c.boxes / n, c.tris / n, c.leaves / n, c.maxsp);
            if (key) {
                char buf[160];
                snprintf(buf, sizeof buf, "%s{\"bounce\": %d, \"rays\": %zu, \"nodes_per_ray\": %.3f, \"tri_tests_per_ray\": %.3f}", b ? ", " : "", b,
                         gen[b].size(), c.nodes / n, c.tris / n);
                json += buf;
            }
        }
        if (key) json += "]";
        printf("\n");
    };
    report("two-child tree", false, "two_child_tree");
    struct Cfg { const char *name; int mode, low, budget, arity, order, qbits, leafMax; };
    const Cfg cfgs[] = {
        {"4-wide, no stack bound", 0, 0, 1000, 4},
        {"4-wide, 8-bit boxes in a node frame", 0, 0, 1000, 4, 0, 8},
        {"4-wide, 8-bit boxes, leaves of up to 3", 0, 0, 1000, 4, 0, 8, 3},
        {"4-wide, 8-bit boxes, leaves of up to 4", 0, 0, 1000, 4, 0, 8, 4},
        {"4-wide, 7-bit boxes in a node frame", 0, 0, 1000, 4, 0, 7},
        {"4-wide, 6-bit boxes in a node frame", 0, 0, 1000, 4, 0, 6},
        {"4-wide, 5-bit boxes in a node frame", 0, 0, 1000, 4, 0, 5},
        {"4-wide, nearest first, rest in storage order", 0, 0, 1000, 4, 1},
        {"4-wide, budget 24 top-down", 0, 0, 24, 4},
        {"4-wide, budget 24, open h2<=6", 1, 6, 24, 4},
        {"4-wide, budget 24, open h2<=8", 1, 8, 24, 4},
        {"4-wide, budget 24, open h2<=10", 1, 10, 24, 4},
        {"4-wide, budget 28 top-down", 0, 0, 28, 4},
        {"4-wide, budget 32 top-down", 0, 0, 32, 4},
        {"3-wide, budget 24 top-down", 0, 0, 24, 3},
        {"3-wide, no stack bound", 0, 0, 1000, 3},
    };
    for (const Cfg &cf : cfgs) {
        g_wide.clear();
        g_mode = cf.mode, g_low = cf.low, g_maxArity = cf.arity, g_order = cf.order, g_qbits = cf.qbits;
        g_leafMax = cf.leafMax;
        g_merged.clear();
        g_stackNeedMax = 0;
        collapse(0, cf.budget, 0);
        double fill = 0;
        for (const WNode &w : g_wide) fill += w.n;
        char nm[128];
        snprintf(nm, sizeof nm, "%s (%zu nodes, fill %.2f, need %ld)", cf.name, g_wide.size(), fill / g_wide.size(), g_stackNeedMax);
        report(nm, true, cf.qbits == 8 && !cf.leafMax ? "four_child_tree_8bit_boxes" : nullptr);
    }
    if (argc > 3) {
        FILE *f = fopen(argv[3], "w");
        if (f) {
            fprintf(f, "{\"tool\": \"tools/bvh_sim.cpp: CPU model of the walk over the flattened tree (camera rays, then cosine-distributed bounces; leaves tested when reached)\", "
                       "\"nodes\": %u, \"triangle_records\": %u, %s}\n", info[0], info[1], json.c_str());
            fclose(f);
        }
    }
    rt_flat_free(flat);
    rt_host_scene_free(hs);
    return 0;
}
