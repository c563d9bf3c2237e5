/*
 * rt_amd.h -- C ABI of librt_amd.so, the MI355X (gfx950) implementation of the
 * per-pixel Monte Carlo path-tracing loop of sergeyrid/raytracing.
 *
 * The reference has no plugin / FFI interface: its only seams are the process command
 * line (main.cpp:1244-1259) and three C++ functions called from main():
 *
 *     InputData  parseGLTF(string&, uint32_t w, uint32_t h)      main.cpp:866
 *     SceneFloat generateScene(const InputData&)                 main.cpp:1235
 *     void       printImage(SceneFloat&, string&)                main.cpp:856
 *
 * plus the inner seam  Intersection intersectScene(o, d, const InputData&)  main.cpp:1113.
 * Each entry point below names the reference function it stands in for.  Plain C types
 * only; every call returns RT_OK (0) or a negative code and never throws; the message of
 * the last failure on the calling thread is available from rt_last_error().
 *
 * Ownership: the caller owns every buffer it passes for the duration of the call.  A
 * rt_scene owns its device memory until rt_scene_destroy.  Output buffers are caller
 * allocated.  Entry points ending in _dev take DEVICE pointers valid on the scene's GPU
 * and a HIP stream (hipStream_t passed as void*, NULL = default stream); they enqueue
 * work and return without synchronising.  The others take HOST pointers and return when
 * the result is in place.
 *
 * Threading: calls on one rt_scene must not overlap (the scene holds scratch buffers and
 * counters that every launch reuses).  Different scenes may be used from different threads.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)  /* bad argument / unsupported scene file content */
#define RT_ERR_IO (-2)       /* file could not be read or written */
#define RT_ERR_HIP (-3)      /* a HIP runtime call failed */
#define RT_ERR_LIMIT (-4)    /* a size exceeds what the kernels were built for */

/* Material types, values of main.cpp:55-59 (enum class Material). */
#define RT_MAT_DIFFUSER 0
#define RT_MAT_METALLIC 1
#define RT_MAT_DIELECTRIC 2

/* RNG modes of rt_render*.
 * RT_RNG_REFERENCE reproduces main.cpp:1221-1225: one minstd_rand per pixel seeded with
 * py*W+px, carried across that pixel's samples (so spp_begin must be 0).  It also evaluates each
 * sample in the reference's recursion order (innermost bounce first), so the radiance sums equal the
 * CPU path's bit for bit; it is the validation mode (max_depth <= 64, allocates its record buffer on
 * first use), not the fast one.
 * RT_RNG_COUNTER seeds a fresh minstd_rand per (pixel, sample) from a hash of
 * (pixel, sample, seed); any sample range can then be rendered on any GPU. */
#define RT_RNG_REFERENCE 0
#define RT_RNG_COUNTER 1

/* Pipelines of rt_render*: both produce the same estimator.
 * RT_PIPE_MEGAKERNEL: one persistent thread per pixel, a lane starts its next sample as soon
 * as its path ends.  RT_PIPE_WAVEFRONT: queue-per-bounce kernels (generate / trace / shade; optionally, from
 * the second bounce on, a regrouping of the ray queue by the cell of the ray origin) over 48-byte path
 * records and index queues; needs RT_RNG_COUNTER. */
#define RT_PIPE_MEGAKERNEL 0
#define RT_PIPE_WAVEFRONT 1

/* Limits of the acceleration structure: leaves hold at most 2 triangles and the tree has at most
 * RT_MAX_TREE_HEIGHT inner levels (the per-lane traversal stack of the kernels), so a scene holds at
 * most 2 * 2^24 triangles; rt_scene_create / rt_flat_build return RT_ERR_LIMIT beyond that. */
#define RT_MAX_TREE_HEIGHT 24
#define RT_MAX_TRIANGLES (2u << RT_MAX_TREE_HEIGHT)

/* Where the reference's answer depends on the ORDER in which its own tree (BVH::build, main.cpp:552-558,
 * 608-686) is walked (BVH::intersect, main.cpp:560-605):
 *   - two primitives at exactly the same distance -- coincident faces, a box standing on a floor, a ray through
 *     an edge shared by two triangles: the first one met wins (strict `<`, main.cpp:566-567, 589-590, 598-599);
 *   - a ray lying in the plane of a triangle: a "hit" at distance NaN, kept when it is the first of its leaf
 *     (main.cpp:483, 566);  a hit at t >= INF, likewise;
 *   - subtrees cut by `distance < minDistance` on boxes of twice their size (main.cpp:174-177, 571, 587, 596).
 * RT_HITS_REFERENCE_ORDER: the library builds that very tree on the host (the reference's full-sweep SAH with
 * its sequence of std::sort calls, leaves of four, pre-order numbering) next to its own and resolves these
 * cases as the reference does: the closest-hit calls and the RT_RNG_REFERENCE render walk the reference's
 * tree in the reference's order for EVERY ray (all of the above reproduced); the RT_RNG_COUNTER pipelines trace
 * on the library's own tree and hand to that walk the rays that met two primitives within 2e-5 of each other
 * (coincident faces with different vertices land an ulp apart, and the reference's box arithmetic then decides),
 * that met an analytic primitive at a distance not below INF (a plane at a grazing angle), or that start more
 * than four scene sizes outside the scene (equal and nearly equal distances and such plane hits reproduced; not
 * the NaN and beyond-INF "hits" on triangles that only the reference's own walk meets).  Available for scenes the reference itself can hold: at most 65535 primitives and
 * tree nodes (its ids are uint16_t, main.cpp:541-544), no unsplittable node of more than four primitives
 * (main.cpp:675-679 stores an out-of-range index there) and a tree of at most 64 levels; it is the default for
 * those (environment RT_HIT_ORDER=leaf at rt_scene_create overrides).
 * RT_HITS_REFERENCE_WALK: as RT_HITS_REFERENCE_ORDER, and the RT_RNG_COUNTER pipelines too send every ray through
 * the reference's walk (slow; a validation aid: their ray-segment counts then equal the CPU path's even where a
 * path meets a NaN hit).
 * RT_HITS_LEAF_ORDER: the library's own tree only; of two primitives at exactly the same distance the one
 * earlier in the library's leaf order wins; NaN distances and t >= INF are misses.  The only mode for larger
 * scenes. */
#define RT_HITS_LEAF_ORDER 0
#define RT_HITS_REFERENCE_ORDER 1
#define RT_HITS_REFERENCE_WALK 2

#define RT_MISS_ID 0xFFFFFFFFu /* prim_id of a ray that hits nothing */
#define RT_INF 10000.0f        /* t of a ray that hits nothing (INF, main.cpp:23) */

/* InputData's camera fields, main.cpp:696-700. */
typedef struct rt_camera {
    float position[3];
    float right[3];
    float up[3];
    float forward[3];
    float tan_x; /* cameraFovTan.x = tan_y * W / H, main.cpp:1104 */
    float tan_y; /* tan(yfov / 2), main.cpp:1013 */
} rt_camera;

/* MaterialData, main.cpp:61-66. */
typedef struct rt_material {
    float color[3];
    float emission[3];
    int32_t type; /* RT_MAT_* */
    float ior;    /* always 1.5 from the loader (main.cpp:65) */
} rt_material;

/* Analytic primitives: Primitive + Plane / Ellipsoid / Box (main.cpp:199-464).  The reference's loader
 * never constructs them (it emits triangles only, main.cpp:1089), so scene files cannot describe them;
 * they enter through rt_scene_desc.  Their intersectPrimitive arithmetic is reproduced; they are shaded
 * like triangles with the material they name.  An ellipsoid or box whose material emits joins the light list
 * (main.cpp:1096-1098) behind the light triangles, in list order: LightSurface samples it (Ellipsoid /
 * Box::samplePoint, main.cpp:320-330, 425-441) and adds its density (Ellipsoid / Box::pdf through
 * Primitive::intersectFull / pdfFull, main.cpp:218-239, 332-356, 443-455), as written in the reference, its
 * oddities included.  A plane must not emit (Plane::samplePoint throws, main.cpp:261-267).  In the closest-hit
 * results primitive k has id ntri + k.  At most RT_MAX_PRIMITIVES per scene.  Under RT_HITS_LEAF_ORDER they are
 * tested one by one after the walk through the triangle tree; under RT_HITS_REFERENCE_ORDER they sit where
 * BVH::build puts them: planes at the root, the others in the tree under their calculateAABB boxes (552-558). */
#define RT_PRIM_PLANE 0     /* param = normal (local space; need not be unit length), main.cpp:242-276 */
#define RT_PRIM_ELLIPSOID 1 /* param = radius per axis, main.cpp:278-365 */
#define RT_PRIM_BOX 2       /* param = half extent per axis ("size" as intersectPrimitive uses it, 374-375) */
#define RT_MAX_PRIMITIVES 256u
typedef struct rt_primitive {
    int32_t type;      /* RT_PRIM_* */
    float position[3]; /* Primitive::position */
    float rotation[4]; /* Primitive::rotation, unit quaternion x, y, z, w; conjRotation is derived */
    float param[3];
    uint32_t material;
} rt_primitive;

/* The flattened InputData (main.cpp:689-706) handed across the boundary: a triangle soup in
 * loader emission order (that order defines the primitive ids returned by the trace calls),
 * one material index per triangle, the material table and the camera.  The light list
 * (main.cpp:1096-1098) and the BVH are derived by rt_scene_create. */
typedef struct rt_scene_desc {
    const float *tri_verts;       /* ntri * 9 floats: A.xyz, B.xyz, C.xyz, world space */
    const uint32_t *tri_material; /* ntri */
    uint32_t ntri;
    const rt_material *materials;
    uint32_t nmat;
    rt_camera camera;
    /* InputData::backgroundColor (main.cpp:694): what applyLight returns for a ray that hits nothing
     * (main.cpp:1190-1192).  The reference never sets it: (0, 0, 0) reproduces it. */
    float background[3];
    const rt_primitive *prims; /* nprim analytic primitives, may be NULL when nprim == 0 */
    uint32_t nprim;
} rt_scene_desc;

typedef struct rt_scene rt_scene;           /* device-resident scene */
typedef struct rt_host_scene rt_host_scene; /* parsed scene file, host memory only */

typedef struct rt_scene_stats {
    uint32_t ntri;
    uint32_t nlights;     /* light triangles + emissive analytic primitives */
    uint32_t nnodes;      /* 32-byte two-child nodes (the four-child tree wf_extend walks has about half as many) */
    uint32_t tree_height; /* inner-node levels on the longest root-to-leaf path */
    double build_seconds; /* host BVH build + flatten */
    uint64_t device_bytes; /* device memory the scene holds now (grows with rt_scene_reserve / first renders) */
    /* Splits of the main tree where the height bound (RT_MAX_TREE_HEIGHT inner levels = the kernels'
     * traversal stack) ruled out the cheapest surface-area-heuristic candidate, or every candidate
     * (median split).  0 = the bound cost nothing; large values mean a deeper, worse tree than the
     * heuristic wanted, i.e. the scene is close to the triangle limit. */
    uint32_t height_capped_splits;
    uint32_t nprims; /* analytic primitives */
    /* triangle records of the flattened structure: ntri plus the pad records that keep every two-triangle
     * leaf inside one 128-byte line (rt_flat_get) */
    uint32_t ntri_records;
    uint32_t hit_order;      /* RT_HITS_*: how order-dependent hits are resolved now */
    uint32_t ref_tree_nodes; /* nodes / deepest level of the reference's own tree; 0 when the scene is beyond */
    uint32_t ref_tree_depth; /* the reference's limits (rt_scene_set_hit_order then reports why) */
} rt_scene_stats;

/* Counters accumulated on the device by the launches since the last rt_counters_reset:
 * segments = rays traced (primary + every continuation; the light-pdf triangle tests of
 * main.cpp:771-783 are not rays); nodes / tri_tests are only counted by the *_counted
 * launches (rt_set_counting). */
typedef struct rt_counters {
    uint64_t segments;
    uint64_t nodes;
    uint64_t tri_tests;
    uint64_t light_tests;
    /* counted launches only: lane slots the wave64s spent in the node phase / triangle phase of the
     * traversal (iterations of the wave, summed over its 64 lanes); nodes / node_lane_slots is the
     * fraction of those slots that did useful work */
    uint64_t node_lane_slots;
    uint64_t tri_lane_slots;
    /* wavefront pipeline: rays the trace kernel handed to its second kernel -- their walk of the four-child tree ran
     * out of stack, or (RT_HITS_REFERENCE_ORDER) they met two primitives at the best distance */
    uint64_t retraced;
} rt_counters;

const char *rt_last_error(void);
int rt_device_count(void);

/* ---- scene files: stands in for parseGLTF, main.cpp:866-1111 (host only, no GPU call) ---- */
int rt_gltf_load(const char *path, uint32_t width, uint32_t height, rt_host_scene **out);
/* The same loader with opt-in departures from the reference's rules towards the glTF 2.0
 * specification (each changes what some files mean, so none is on by default):
 * STRIDE: bufferView.byteStride is honoured for POSITION and bufferView.byteOffset may be absent
 * (reference: stride 12 always, byteOffset required, main.cpp:885-891, 1044-1050);
 * INDEX_OFFSET: accessor.byteOffset applies to index accessors too (reference ignores it,
 * main.cpp:1060-1078); DATA_URI: base64 `data:` buffers (reference: external files only, 875-882);
 * HIERARCHY: a node's world transform is the product along its parent chain whatever the node
 * order, and the camera uses its world transform (reference: one ascending pass, camera local
 * only, main.cpp:1015-1026);
 * GLB: `.glb` binary containers (header, JSON chunk, BIN chunk backing the first buffer when it has no
 * uri; reference: text files with external buffers only, main.cpp:869-882); IOR: KHR_materials_ior.ior per
 * material (reference: 1.5 always, main.cpp:65).  RT_GLTF_SPEC = all of these.
 * BACKGROUND: InputData::backgroundColor (main.cpp:694, never set by the reference) from
 * scenes[scene].extras.backgroundColor = [r, g, b] -- not a glTF feature, hence outside RT_GLTF_SPEC. */
#define RT_GLTF_STRIDE 1u
#define RT_GLTF_INDEX_OFFSET 2u
#define RT_GLTF_DATA_URI 4u
#define RT_GLTF_HIERARCHY 8u
#define RT_GLTF_GLB 16u
#define RT_GLTF_IOR 32u
#define RT_GLTF_BACKGROUND 64u
#define RT_GLTF_SPEC 63u
#define RT_GLTF_ALL 127u
int rt_gltf_load_ex(const char *path, uint32_t width, uint32_t height, uint32_t flags, rt_host_scene **out);
int rt_host_scene_get_desc(const rt_host_scene *hs, rt_scene_desc *out); /* pointers stay owned by hs */
void rt_host_scene_free(rt_host_scene *hs);

/* ---- scene upload: Triangle::update (main.cpp:527-536), the light list (1096-1098) and the
 * acceleration structure (BVH::build, main.cpp:552-558 / 608-686, here a binned-SAH builder
 * flattened to 32-byte two-child nodes and, for the wavefront pipeline's trace kernel, to 48-byte
 * four-child nodes of the same tree) ---- */
int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out);
void rt_scene_destroy(rt_scene *s);
int rt_scene_get_stats(const rt_scene *s, rt_scene_stats *out);
/* Selects RT_HITS_LEAF_ORDER / RT_HITS_REFERENCE_ORDER / RT_HITS_REFERENCE_WALK for the later calls on this scene.  RT_ERR_LIMIT with the
 * reason in rt_last_error() when the reference's own tree is not available for the scene. */
int rt_scene_set_hit_order(rt_scene *s, int order);
/* Copies of the flattened structure for checkers: the node grid (origin xyz, cell xyz), nodes
 * (nnodes * 8 words, layout under rt_flat_get), triangle records in leaf order (ntri_records * 16 floats).  Any
 * pointer may be NULL. */
int rt_scene_get_flat(const rt_scene *s, float grid[6], uint32_t *nodes, float *tris);

/* Host-only view of the same build (no GPU call), for tools and checkers: builds the
 * acceleration structure of `desc` and hands out the flattened arrays exactly as rt_scene_create
 * would upload them.  info[0] = node count, [1] = triangle record count, [2] = light count,
 * [3] = tree height (inner levels).  grid: origin xyz then cell xyz of the 16-bit node grid.
 * nodes: nnodes*8 words; words 0-2 = left child's box (one word per axis: lo | hi << 16, plane =
 * origin + q * cell, rounded outward), 3-5 = right child's box, 6-7 = child references (>= 0 node
 * index, < 0 leaf: ~ref = first << 3 | count-1).  tris: info[1]*16 floats, leaf order (A, AB, AC,
 * cross(AB,AC), loader id, material, 2/|N|, 0).  A leaf of two triangles starts on an even record (both
 * in one 128-byte line); the record skipped for that is a pad no leaf names (A = 3e30, loader id =
 * RT_MISS_ID), so info[1] >= ntri.  lights: leaf-order positions of the emissive triangles in loader
 * order.  Any pointer may be NULL. */
typedef struct rt_flat rt_flat;
int rt_flat_build(const rt_scene_desc *desc, rt_flat **out);
int rt_flat_get(const rt_flat *f, uint32_t info[4], float grid[6], uint32_t *nodes, float *tris,
                uint32_t *lights);
/* The second tree of the same build, over the light triangles only; LightSurface::pdf
 * (main.cpp:771-783) walks it instead of looping over the light list when a scene has 48
 * lights or more (environment RT_LIGHT_TREE_MIN overrides that count at rt_scene_create).
 * info[0] = node count, [1] = tree height; grid and nodes as in rt_flat_get; recs: one triangle
 * record (16 floats) per light in this tree's leaf order, word 12 = index in the light list,
 * word 13 = the light's leaf position in the main triangle array.  Any pointer may be NULL. */
int rt_flat_get_light_tree(const rt_flat *f, uint32_t info[2], float grid[6], uint32_t *nodes, float *recs);
/* The main tree once more with up to four children per node, as the wavefront pipeline's trace kernel walks
 * it (the other kernels walk the two-child nodes).  *nwide = node count; nodes: nwide * 12 words.  Child boxes
 * are 8-bit in a frame of the node: per axis an origin on the node grid and a shift s, plane = grid_origin +
 * (origin + (q << s)) * grid_cell.  Words 0-5 = lo.x, hi.x, lo.y, hi.y, lo.z, hi.z of children 0..3 (one byte
 * each); word 6 = origin.x | origin.y << 16; word 7 = origin.z | s.x << 16 | s.y << 20 | s.z << 24 |
 * children << 28; words 8-11 = child references (>= 0 index of a wide node, < 0 a leaf as in rt_flat_get,
 * 0x7FFFFFFF unused).  Every child box contains the box the two-child tree holds for the same subtree. */
int rt_flat_get_wide(const rt_flat *f, uint32_t *nwide, uint32_t *nodes);
/* The reference's own tree of the same scene (BVH::build, main.cpp:552-558, 608-686), when the scene is within
 * the reference's limits.  info[0] = 1 if available, [1] = node count, [2] = deepest level, [3] = analytic planes
 * at the front of the primitive list.  nodes: info[1] * 16 words: AABB::sizePlusCenter.xyz,
 * negSizePlusCenter.xyz, minPoint.xyz, maxPoint.xyz as floats, then left, right, firstPrimitiveId, primitiveCount
 * (a leaf has left = right = 0).  order: the reference's re-ordered primitive list, ntri + nprim entries: loader
 * triangle id, or ntri + k for analytic primitive k.  Any pointer may be NULL. */
int rt_flat_get_ref_tree(const rt_flat *f, uint32_t info[4], uint32_t *nodes, uint32_t *order);
/* Quality of the main tree: the rt_scene_stats counter of splits the height bound overrode, and the
 * surface-area estimate of the work per random ray through the scene's bounds (expected node visits,
 * expected ray/triangle tests).  Any pointer may be NULL. */
int rt_flat_get_quality(const rt_flat *f, uint32_t *height_capped_splits, double *sah_nodes, double *sah_tris);
void rt_flat_free(rt_flat *f);

/* ---- closest hit: stands in for intersectScene / BVH::intersect, main.cpp:1113, 560-605 ---- */
/* n rays, origins and directions as n*3 floats; prim_id[i] = loader-order triangle id or
 * RT_MISS_ID, t[i] = hit distance or RT_INF. */
int rt_intersect_rays(rt_scene *s, uint64_t n, const float *origins, const float *directions,
                      uint32_t *prim_id, float *t);
int rt_intersect_rays_dev(rt_scene *s, uint64_t n, const float *d_origins, const float *d_directions,
                          uint32_t *d_prim_id, float *d_t, void *stream);
/* The primary ray of sample 0 of every pixel (generateSample, main.cpp:1212-1216, with the
 * engine seeded as in generatePixel, main.cpp:1221), row-major W*H outputs. */
int rt_trace_primary(rt_scene *s, uint32_t width, uint32_t height, uint32_t *prim_id, float *t);
int rt_trace_primary_dev(rt_scene *s, uint32_t width, uint32_t height, uint32_t *d_prim_id, float *d_t,
                         void *stream);

/* ---- render: stands in for generateScene / generatePixel's sample loop, main.cpp:1220-1242 ----
 * radiance_sum[3*(py*W+px) + c] receives the SUM over samples [spp_begin, spp_begin+spp_count)
 * of generateSample (linear radiance, before the mean / tonemap), so that partial sums of
 * disjoint sample ranges from several GPUs add up to the full frame.  max_depth is
 * InputData::rayDepth (6 in the reference, main.cpp:692). */
int rt_render(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_count,
              uint32_t max_depth, int rng_mode, uint32_t seed, int pipeline, float *radiance_sum);
int rt_render_dev(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_count,
                  uint32_t max_depth, int rng_mode, uint32_t seed, int pipeline, float *d_radiance_sum,
                  void *stream);
/* Scratch of the render calls (no reference counterpart: the CPU path keeps its state on the stack).
 * The wavefront pipeline keeps 90 bytes per path in flight (up to 128 frames of samples or 256 Mi
 * paths, never more than spp_count frames), the reference-RNG mode 12 bytes per pixel and bounce.
 * rt_scene_reserve allocates them for the given call shape; buffers only grow and live until
 * rt_scene_destroy.  rt_render_dev on a scene whose scratch already covers the call only enqueues;
 * otherwise it first does what rt_scene_reserve does (waits for the device, allocates). */
int rt_scene_reserve(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp_count, uint32_t max_depth,
                     int rng_mode, int pipeline);

/* The wavefront render for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * Same estimator, kernels and results as rt_render_dev(..., RT_RNG_COUNTER, seed, RT_PIPE_WAVEFRONT, ...)
 * (bit-identical sums), but the path records, queues and cursors live in a fixed array inside the
 * library's code object (206 MB per GPU: 189 MB of path state for 2^21 paths and 17 MB of frames for the
 * reference-order walker; held from the library's first launch on that GPU until the process ends)
 * instead of scene-owned allocations, and the call makes no hipMalloc / hipFree.
 * Limit: width * height <= RT_NOALLOC_MAX_PIXELS (one 1920x1080 frame fits), RT_ERR_LIMIT beyond;
 * smaller images keep up to 64 frames of samples in flight.  Every call on a GPU shares that GPU's
 * array, whatever the scene: calls that overlap on one GPU (two streams, two threads) race.  The ray-sorting
 * tuner's books live in the array too and are kept from call to call (per GPU, not per scene: after a change of
 * scene it re-learns over a few passes; results do not depend on it).  The call can be captured into a HIP graph
 * (hipStreamBeginCapture on `stream` before it, hipStreamEndCapture after): it enqueues one memset and kernels,
 * and nothing else. */
#define RT_NOALLOC_MAX_PIXELS (1u << 21)
int rt_render_noalloc_dev(rt_scene *s, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_count,
                          uint32_t max_depth, uint32_t seed, float *d_radiance_sum, void *stream);

/* ---- several GPUs of one node, one process (no reference counterpart: its only parallel loop is the
 * OpenMP loop over pixels, main.cpp:1237) ----
 * rt_multi_create builds the acceleration structure ONCE on the host, uploads it to every listed GPU
 * (devices == NULL: GPUs 0..ndev-1) and opens an RCCL communicator over them (ncclCommInitAll; librccl is
 * loaded on the first call).  rt_multi_render is generateScene's frame with the samples split: GPU g of n
 * renders the g-th contiguous part of [spp_begin, spp_begin + spp_count) (rt_multi_sample_range) of every
 * pixel with RT_RNG_COUNTER, all GPUs at once on their own streams; one ncclReduce (sum, fp32, root = the
 * first GPU) brings the radiance sums together and the result is copied to the host buffer
 * (width * height * 3 floats).  Because every (pixel, sample) has its own engine the image does not depend
 * on n, except for the order of the float additions: the reduce adds n partial sums per pixel where one GPU
 * adds the samples one after the other, a relative difference of at most about n * 2^-24 of a pixel's sum
 * (bit-identical for n = 1).  rt_multi_scene hands out GPU i's scene, e.g. for rt_tonemap on the result. */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc *desc, const int *devices, int ndev, rt_multi **out);
void rt_multi_destroy(rt_multi *m);
int rt_multi_device_count(const rt_multi *m);
rt_scene *rt_multi_scene(rt_multi *m, int i);
/* GPU g's part of spp_count samples split over ndev GPUs (offset relative to spp_begin, and length); an empty
 * range (0, 0) when ndev <= 0 or g is outside [0, ndev). */
void rt_multi_sample_range(uint32_t spp_count, int ndev, int g, uint32_t *begin, uint32_t *count);
int rt_multi_render(rt_multi *m, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_count,
                    uint32_t max_depth, uint32_t seed, int pipeline, float *radiance_sum);

/* ---- image: mean, ACES, gamma (main.cpp:1226-1231, 827-838), truncating byte conversion
 * (colorFloatToInt, main.cpp:840-846), PPM writer (printImage, main.cpp:856-864) ---- */
/* out_float (npixels*3, may be NULL) = SceneFloat contents; out_rgb (npixels*3, may be NULL). */
int rt_tonemap(rt_scene *s, const float *radiance_sum, uint64_t npixels, uint32_t spp, float *out_float,
               uint8_t *out_rgb);
int rt_tonemap_dev(rt_scene *s, const float *d_radiance_sum, uint64_t npixels, uint32_t spp,
                   float *d_out_float, uint8_t *d_out_rgb, void *stream);
int rt_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb);

/* ---- counters ---- */
int rt_set_counting(rt_scene *s, int enable); /* 1: launches also count nodes / triangle tests (slower) */
int rt_counters_reset(rt_scene *s, void *stream);
/* Synchronises the device.  Counting launches also watch every ray's walk: one that takes more rounds or node
 * visits than four times the tree's nodes and leaves is cut short, and this call then fills `out` and returns
 * RT_ERR_INVALID (a broken tree or traversal state machine would otherwise spin forever; only counting launches
 * carry the check, the timed kernels do not pay for it). */
int rt_counters_read(rt_scene *s, rt_counters *out);

/* ---- per-kernel device time (HIP events on the launch stream) ----
 * rt_set_timing(s, 1) synchronises, clears the totals and makes every later launch record an event
 * pair around each kernel; rt_timing_read synchronises and returns, per kernel kind, the summed
 * device milliseconds and the number of launches since then. */
#define RT_KERNEL_EXTEND 0      /* wavefront: closest hit of a ray queue */
#define RT_KERNEL_SHADE 1       /* wavefront: shading + queue compaction */
#define RT_KERNEL_GENERATE 2    /* wavefront: camera rays */
#define RT_KERNEL_ACCUMULATE 3  /* wavefront: per-pixel sum of the finished samples */
#define RT_KERNEL_MEGA 4        /* megakernel render */
#define RT_KERNEL_CLOSEST_HIT 5 /* rt_trace_primary* / rt_intersect_rays* */
#define RT_KERNEL_TONEMAP 6
#define RT_KERNEL_SORT 7        /* wavefront: regrouping a ray queue by the cell of the ray origin (three small kernels) */
#define RT_KERNEL_KINDS 8
typedef struct rt_kernel_times {
    double ms[RT_KERNEL_KINDS];
    uint64_t launches[RT_KERNEL_KINDS];
} rt_kernel_times;
int rt_set_timing(rt_scene *s, int enable);
int rt_timing_read(rt_scene *s, rt_kernel_times *out);

/* Self-test of the device's restatements of the RNG helpers (sampleUInt / sampleUniform /
 * sampleNormal over std::minstd_rand, main.cpp:98-111): engine k is seeded with seeds[k] as
 * generatePixel seeds its engine (main.cpp:1221) and produces n values per output, stored at
 * [k * n + i].  what = 0: per i one sampleUniform(), one sampleNormal(), one sampleUInt(0, nlights-1)
 * in that order from the one engine; 1 / 2 / 3: only uniforms / normals / uints; 4: raw engine
 * outputs (in ui).  Host pointers, each nseeds * n long; arrays a mode does not produce are zeroed. */
int rt_selftest_rng(int device, const uint32_t *seeds, uint32_t nseeds, uint32_t n, uint32_t nlights, int what,
                    float *uni, float *nrm, uint32_t *ui);

/* The seed of the per-(pixel, sample) stream of RT_RNG_COUNTER (host helper for checkers). */
uint32_t rt_counter_seed(uint32_t pixel, uint32_t sample, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
