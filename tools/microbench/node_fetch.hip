// Microbenchmark: how fast can a wave gather 64-byte records at per-lane random indices?
//   A  every lane issues four global_load_dwordx4 on its own record (what a per-lane BVH walk does)
//   B0 the four lanes of a quad read one record's 64 contiguous bytes; four instructions cover the wave's
//      64 records (data left where it lands: the upper bound of what quad-shaped requests can give)
//   B  as B0 but the loads write LDS directly (lane-linear image) and each lane reads its own record back
//      with four conflict-free ds_read_b128
// The next index depends on the loaded data, as in a tree walk.
// Build: hipcc -O3 --offload-arch=gfx950 node_fetch.hip -o node_fetch ; run: ./node_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 1024;

__device__ inline unsigned mix(unsigned s, const float4& a, const float4& b, const float4& c, const float4& d) {
    return s * 1664525u + 1013904223u + (__float_as_uint(a.x) ^ __float_as_uint(b.y) ^ __float_as_uint(c.z) ^ __float_as_uint(d.w));
}

__global__ __launch_bounds__(256) void k_a(const float4* __restrict__ recs, unsigned mask, unsigned* out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int it = 0; it < kIters; ++it) {
        const float4* p = recs + (size_t)((s >> 8) & mask) * 4;
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        s = mix(s, a, b, c, d);
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_b0(const float4* __restrict__ recs, unsigned mask, unsigned* out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < kIters; ++it) {
        unsigned idx = (s >> 8) & mask;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int src = 16 * i + (lane >> 2);
            unsigned n = __shfl(idx, src, 64);
            v[i] = recs[(size_t)n * 4 + (lane & 3)];
        }
        s = mix(s, v[0], v[1], v[2], v[3]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_b(const float4* __restrict__ recs, unsigned mask, unsigned* out) {
    __shared__ float4 stage[4][256];
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* mine = stage[wave];
    for (int it = 0; it < kIters; ++it) {
        unsigned idx = (s >> 8) & mask;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int src = 16 * i + (lane >> 2);
            unsigned n = __shfl(idx, src, 64);
            int chunk = ((lane & 3) - (src >> 2)) & 3;
            const float4* g = recs + (size_t)n * 4 + chunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(mine + i * 64), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = mine[lane * 4 + ((r + (lane >> 2)) & 3)];
        s = mix(s, v[0], v[1], v[2], v[3]);
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 32-byte records: two dwordx4 per lane
__global__ __launch_bounds__(256) void k_a32(const float4* __restrict__ recs, unsigned mask, unsigned* out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int it = 0; it < kIters; ++it) {
        const float4* p = recs + (size_t)((s >> 8) & mask) * 2;
        float4 a = p[0], b = p[1];
        s = mix(s, a, b, a, b);
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 32-byte records through LDS: a pair of lanes reads one record, two instructions cover the wave
__global__ __launch_bounds__(256) void k_b32(const float4* __restrict__ recs, unsigned mask, unsigned* out) {
    __shared__ float4 stage[4][128];
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* mine = stage[wave];
    for (int it = 0; it < kIters; ++it) {
        unsigned idx = (s >> 8) & mask;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int src = 32 * i + (lane >> 1);
            unsigned n = __shfl(idx, src, 64);
            int chunk = ((lane & 1) - (src >> 3)) & 1;
            const float4* g = recs + (size_t)n * 2 + chunk;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(mine + i * 64), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float4 v[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) v[r] = mine[lane * 2 + ((r + (lane >> 3)) & 1)];
        s = mix(s, v[0], v[1], v[0], v[1]);
        asm volatile("" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// check B delivers each lane its own record in order
__global__ __launch_bounds__(256) void k_b_check(const float4* __restrict__ recs, unsigned mask, unsigned* bad) {
    __shared__ float4 stage[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* mine = stage[wave];
    unsigned idx = ((blockIdx.x * 256 + threadIdx.x) * 2654435761u >> 8) & mask;
    for (int i = 0; i < 4; ++i) {
        int src = 16 * i + (lane >> 2);
        unsigned n = __shfl(idx, src, 64);
        int chunk = ((lane & 3) - (src >> 2)) & 3;
        const float4* g = recs + (size_t)n * 4 + chunk;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(mine + i * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int r = 0; r < 4; ++r) {
        float4 v = mine[lane * 4 + ((r + (lane >> 2)) & 3)];
        float4 w = recs[(size_t)idx * 4 + r];
        if (v.x != w.x || v.y != w.y || v.z != w.z || v.w != w.w) atomicAdd(bad, 1u);
    }
}

int main() {
    const size_t sizes[] = {256, 32768, 1u << 20};   // records: 16 KB, 2 MB, 64 MB
    const int blocks = 256 * 6;
    unsigned* out; CHECK(hipMalloc(&out, blocks * 256 * 4));
    unsigned* bad; CHECK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t n : sizes) {
        std::vector<unsigned> h(n * 16);
        unsigned x = 12345;
        for (auto& w : h) { x = x * 1103515245u + 12345u; w = (x >> 9) | 0x3f800000u; }
        float4* recs; CHECK(hipMalloc(&recs, n * 64));
        CHECK(hipMemcpy(recs, h.data(), n * 64, hipMemcpyHostToDevice));
        CHECK(hipMemset(bad, 0, 4));
        k_b_check<<<64, 256>>>(recs, (unsigned)n - 1, bad);
        unsigned hb = 1; CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("records %zu (%.1f MB): B check mismatches %u\n", n, n * 64 / 1e6, hb);
        for (int which = 0; which < 5; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (which == 0) k_a<<<blocks, 256>>>(recs, (unsigned)n - 1, out);
                if (which == 1) k_b0<<<blocks, 256>>>(recs, (unsigned)n - 1, out);
                if (which == 2) k_b<<<blocks, 256>>>(recs, (unsigned)n - 1, out);
                if (which == 3) k_a32<<<blocks, 256>>>(recs, (unsigned)n * 2 - 1, out);
                if (which == 4) k_b32<<<blocks, 256>>>(recs, (unsigned)n * 2 - 1, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double recs_s = (double)blocks * 256 * kIters / (best * 1e-3);
            printf("  %-3s %8.3f ms  %7.1f G records/s  %6.2f TB/s\n", which == 0 ? "A" : which == 1 ? "B0" : which == 2 ? "B" : which == 3 ? "A32" : "B32", best, recs_s / 1e9,
                   recs_s * (which >= 3 ? 32 : 64) / 1e12);
        }
        CHECK(hipFree(recs));
    }
    return 0;
}
