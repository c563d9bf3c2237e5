// Microbenchmark: reading a 64-byte per-path record through a compacted index list.
//   soa   16 dword loads per lane from 16 arrays (lanes of a wave read neighbouring words, with gaps)
//   aos   4 dwordx4 loads per lane from one array of 64-byte records
// `gap` = average distance between the path indices of neighbouring lanes (1 = dense, 2-4 = a queue after
// some paths died).  Build: hipcc -O3 --offload-arch=gfx950 soa_aos.hip -o soa_aos
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_soa(const float* __restrict__ rec, size_t stride, const unsigned* __restrict__ idx, unsigned n,
                                              float* out) {
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned p = idx[i];
        float acc = 0.f;
#pragma unroll
        for (int f = 0; f < 16; ++f) acc += rec[f * stride + p];
        out[p] = acc;
    }
}
__global__ __launch_bounds__(256) void k_aos(const float4* __restrict__ rec, const unsigned* __restrict__ idx, unsigned n, float* out) {
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned p = idx[i];
        const float4 a = rec[4 * (size_t)p], b = rec[4 * (size_t)p + 1], c = rec[4 * (size_t)p + 2], d = rec[4 * (size_t)p + 3];
        out[p] = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
    }
}

int main() {
    const unsigned npaths = 32u << 20;   // 2 GB of records
    float* rec; CHECK(hipMalloc(&rec, (size_t)npaths * 64));
    CHECK(hipMemset(rec, 0, (size_t)npaths * 64));
    float* out; CHECK(hipMalloc(&out, (size_t)npaths * 4));
    unsigned* idx; CHECK(hipMalloc(&idx, (size_t)npaths * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int gap = 1; gap <= 4; ++gap) {
        std::vector<unsigned> h;
        unsigned x = 777, p = 0;
        while (p < npaths) {
            h.push_back(p);
            x = x * 1664525u + 1013904223u;
            p += gap == 1 ? 1 : 1 + (x >> 16) % (2 * gap - 1);   // mean step = gap
        }
        const unsigned n = (unsigned)h.size();
        CHECK(hipMemcpy(idx, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        for (int which = 0; which < 2; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (which == 0) k_soa<<<256 * 8, 256>>>(rec, npaths, idx, n, out);
                else k_aos<<<256 * 8, 256>>>((const float4*)rec, idx, n, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("gap %d  %s  %8.3f ms  %6.2f G records/s  %5.2f TB/s useful\n", gap, which ? "aos" : "soa", best, n / best * 1e-6,
                   n * 64.0 / best * 1e-9);
        }
    }
    return 0;
}
