// rt_multi.hip -- several GPUs of one node behind the C ABI (rt_multi_* in include/rt_amd.h).
//
// The reference's only parallel loop is the OpenMP loop over pixels (main.cpp:1237).  Here the frame
// is split by SAMPLES: GPU g of n renders samples [begin_g, begin_g + count_g) of every pixel
// (RT_RNG_COUNTER: one engine per pixel and sample, so the image does not depend on the split) into its
// own W*H*3 radiance-sum buffer, and ONE ncclReduce (sum, fp32, root = the first GPU) over xGMI brings
// the buffers together; the mean / tone curve (main.cpp:1226-1231) must run after that sum.
// One process drives all GPUs (ncclCommInitAll), one stream per GPU; the scene's BVH is built once on
// the host and uploaded to every GPU.
//
// RCCL is loaded with dlopen when the first rt_multi is created: single-GPU users of the library
// neither link nor initialise it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "rt_scene.h"

namespace {

int fail(int code, const std::string &msg) {
    rt::set_error(msg);
    return code;
}

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("cannot load librccl.so: ") + dlerror();
            return;
        }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n;
            return p;
        };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

}  // namespace

struct rt_multi {
    std::vector<int> devices;
    std::vector<rt_scene *> scenes;
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<float *> bufs;  // per GPU: W*H*3 floats, grown on demand
    size_t buf_floats = 0;
};

extern "C" {

void rt_multi_destroy(rt_multi *m) {
    if (!m) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (size_t i = 0; i < m->devices.size(); ++i) {
        if (hipSetDevice(m->devices[i]) != hipSuccess) continue;
        if (i < m->streams.size() && m->streams[i]) {
            (void)hipStreamSynchronize(m->streams[i]);
            (void)hipStreamDestroy(m->streams[i]);
        }
        if (i < m->bufs.size()) (void)hipFree(m->bufs[i]);
    }
    for (ncclComm_t c : m->comms)
        if (c) (void)rccl().CommDestroy(c);
    for (rt_scene *s : m->scenes) rt_scene_destroy(s);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete m;
}

int rt_multi_create(const rt_scene_desc *desc, const int *devices, int ndev, rt_multi **out) {
    if (!desc || !out) return fail(RT_ERR_INVALID, "rt_multi_create: null argument");
    *out = nullptr;
    const int visible = rt_device_count();
    if (visible <= 0) return fail(RT_ERR_HIP, "rt_multi_create: no HIP device is visible");
    if (ndev <= 0 || ndev > visible) return fail(RT_ERR_INVALID, "rt_multi_create: ndev must be between 1 and the visible GPU count");
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= visible) return fail(RT_ERR_INVALID, "rt_multi_create: device index out of range");
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return fail(RT_ERR_INVALID, "rt_multi_create: a device is listed twice");
    }
    Rccl &R = rccl();
    if (!R.error.empty()) return fail(RT_ERR_HIP, "rt_multi_create: " + R.error);

    std::shared_ptr<const rt::FlatScene> flat;  // ONE host BVH build for all GPUs
    int rc = rt::build_flat_checked(desc, &flat, "rt_multi_create");
    if (rc != RT_OK) return rc;

    int prev = -1;
    (void)hipGetDevice(&prev);
    rt_multi *m = new rt_multi();
    m->devices = devs;
    m->streams.assign(ndev, nullptr);
    m->bufs.assign(ndev, nullptr);
    for (int i = 0; i < ndev && rc == RT_OK; ++i) {
        rt_scene *s = nullptr;
        rc = rt::scene_create_from_flat(flat, *desc, devs[i], &s);
        if (rc != RT_OK) break;
        m->scenes.push_back(s);
        hipError_t e = hipSetDevice(devs[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, std::string("rt_multi_create: stream: ") + hipGetErrorString(e));
    }
    if (rc == RT_OK) {
        m->comms.assign(ndev, nullptr);
        ncclResult_t r = R.CommInitAll(m->comms.data(), ndev, devs.data());
        if (r != ncclSuccess) {
            m->comms.clear();
            rc = fail(RT_ERR_HIP, std::string("rt_multi_create: ncclCommInitAll: ") + R.GetErrorString(r));
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc != RT_OK) {
        const std::string keep = rt_last_error();
        rt_multi_destroy(m);
        rt::set_error(keep);
        return rc;
    }
    *out = m;
    return RT_OK;
}

int rt_multi_device_count(const rt_multi *m) { return m ? (int)m->devices.size() : 0; }

rt_scene *rt_multi_scene(rt_multi *m, int i) {
    if (!m || i < 0 || i >= (int)m->scenes.size()) return nullptr;
    return m->scenes[i];
}

void rt_multi_sample_range(uint32_t spp_count, int ndev, int g, uint32_t *begin, uint32_t *count) {
    if (ndev <= 0 || g < 0 || g >= ndev) {  // no such GPU: an empty range
        if (begin) *begin = 0u;
        if (count) *count = 0u;
        return;
    }
    // contiguous split; the first spp_count % ndev GPUs take one sample more
    const uint32_t base = spp_count / (uint32_t)ndev, extra = spp_count % (uint32_t)ndev, ug = (uint32_t)g;
    if (begin) *begin = ug * base + (ug < extra ? ug : extra);
    if (count) *count = base + (ug < extra ? 1u : 0u);
}

int rt_multi_render(rt_multi *m, uint32_t width, uint32_t height, uint32_t spp_begin, uint32_t spp_count, uint32_t max_depth,
                    uint32_t seed, int pipeline, float *radiance_sum) {
    if (!m || !radiance_sum) return fail(RT_ERR_INVALID, "rt_multi_render: null argument");
    if (width == 0 || height == 0) return fail(RT_ERR_INVALID, "image width and height must be positive");
    if ((uint64_t)width * height > (1ull << 28)) return fail(RT_ERR_LIMIT, "image has more than 2^28 pixels");
    const int n = (int)m->devices.size();
    const size_t floats = (size_t)width * height * 3;
    Rccl &R = rccl();
    int prev = -1;
    (void)hipGetDevice(&prev);
    int rc = RT_OK;
    hipError_t e = hipSuccess;
    // buffers and path-record scratch first, so that the launches below only enqueue
    for (int g = 0; g < n && rc == RT_OK; ++g) {
        uint32_t cnt = 0;
        rt_multi_sample_range(spp_count, n, g, nullptr, &cnt);
        if ((e = hipSetDevice(m->devices[g])) != hipSuccess) break;
        if (floats > m->buf_floats) {
            (void)hipStreamSynchronize(m->streams[g]);
            (void)hipFree(m->bufs[g]);
            m->bufs[g] = nullptr;
            if ((e = hipMalloc((void **)&m->bufs[g], floats * sizeof(float))) != hipSuccess) break;
        }
        rc = rt_scene_reserve(m->scenes[g], width, height, cnt, max_depth, RT_RNG_COUNTER, pipeline);
    }
    if (e == hipSuccess && rc == RT_OK && floats > m->buf_floats) m->buf_floats = floats;
    // every GPU renders its sample range on its own stream, all at once
    for (int g = 0; g < n && rc == RT_OK && e == hipSuccess; ++g) {
        uint32_t begin = 0, cnt = 0;
        rt_multi_sample_range(spp_count, n, g, &begin, &cnt);
        rc = rt_render_dev(m->scenes[g], width, height, spp_begin + begin, cnt, max_depth, RT_RNG_COUNTER, seed, pipeline,
                           m->bufs[g], m->streams[g]);
    }
    // one sum-reduce onto the first GPU, in place, stream-ordered after each GPU's render
    if (rc == RT_OK && e == hipSuccess) {
        ncclResult_t r = R.GroupStart();
        for (int g = 0; g < n && r == ncclSuccess; ++g)
            r = R.Reduce(m->bufs[g], m->bufs[g], floats, ncclFloat, ncclSum, 0, m->comms[g], m->streams[g]);
        const ncclResult_t r2 = R.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) rc = fail(RT_ERR_HIP, std::string("rt_multi_render: ncclReduce: ") + R.GetErrorString(r));
    }
    if (rc == RT_OK && e == hipSuccess) {
        for (int g = n - 1; g >= 0 && e == hipSuccess; --g) {
            if ((e = hipSetDevice(m->devices[g])) != hipSuccess) break;
            e = hipStreamSynchronize(m->streams[g]);
        }
        if (e == hipSuccess) e = hipMemcpy(radiance_sum, m->bufs[0], floats * sizeof(float), hipMemcpyDeviceToHost);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (rc == RT_OK && e != hipSuccess) rc = fail(RT_ERR_HIP, std::string("rt_multi_render: ") + hipGetErrorString(e));
    return rc;
}

}  // extern "C"
