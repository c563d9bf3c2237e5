// Microbenchmark: cost of per-lane random gathers as a function of (loads per record, distinct lines per lane).
//   k<L, D>: every lane issues L global_load_dwordx4 per iteration, spread over D distinct random 64-byte records.
// Build: hipcc -O3 --offload-arch=gfx950 fetch_cost.hip -o fetch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kIters = 2048;

template <int L, int D>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ recs, unsigned mask, unsigned* out) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int it = 0; it < kIters; ++it) {
        unsigned acc = 0;
        float4 v[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            unsigned r = ((s >> 8) + (l % D) * 0x9E3779B1u) & mask;   // D distinct records
            v[l] = recs[(size_t)r * 4 + (l / D) % 4];
        }
#pragma unroll
        for (int l = 0; l < L; ++l) acc ^= __float_as_uint(v[l].x) ^ __float_as_uint(v[l].w);
        s = s * 1664525u + 1013904223u + acc;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Same gather with only `active` of the 64 lanes enabled (the others sit the iteration out, as
// lanes of a diverged wave do): does the cost follow the active lanes or the instruction?
template <int L>
__global__ __launch_bounds__(256) void k_masked(const float4* __restrict__ recs, unsigned mask, unsigned* out, int active) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < kIters; ++it) {
        // a different subset every iteration, the same size every time
        const bool on = ((lane * 37 + it * 11) & 63) < active;
        if (on) {
            float4 v[L];
            unsigned r = (s >> 8) & mask, acc = 0;
#pragma unroll
            for (int l = 0; l < L; ++l) v[l] = recs[(size_t)r * 4 + l % 4];
#pragma unroll
            for (int l = 0; l < L; ++l) acc ^= __float_as_uint(v[l].x) ^ __float_as_uint(v[l].w);
            s = s * 1664525u + 1013904223u + acc;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int L>
void run_masked(const float4* recs, unsigned mask, unsigned* out, hipEvent_t e0, hipEvent_t e1, int active) {
    const int blocks = 256 * 6;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        k_masked<L><<<blocks, 256>>>(recs, mask, out, active);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double cyc = best * 1e-3 * 2.4e9 / kIters / 24.0;
    printf("  loads %d, %2d of 64 lanes active: %8.3f ms  %6.1f CU-cycles per wave-iteration  %5.2f per active lane-load\n", L, active,
           best, cyc, cyc / active / L);
}

template <int L, int D>
void run(const float4* recs, unsigned mask, unsigned* out, hipEvent_t e0, hipEvent_t e1) {
    const int blocks = 256 * 6;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        k<L, D><<<blocks, 256>>>(recs, mask, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double cyc = best * 1e-3 * 2.4e9 / kIters / 24.0;   // CU cycles per wave-iteration (24 waves per CU)
    printf("  loads %d over %d records: %8.3f ms  %6.1f CU-cycles per wave-iteration  %5.2f per lane-load\n", L, D, best, cyc,
           cyc / 64 / L);
}

int main() {
    unsigned* out; CHECK(hipMalloc(&out, 256 * 6 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t n : {256ul, 32768ul}) {
        std::vector<unsigned> h(n * 16);
        unsigned x = 12345;
        for (auto& w : h) { x = x * 1103515245u + 12345u; w = (x >> 9) | 0x3f800000u; }
        float4* recs; CHECK(hipMalloc(&recs, n * 64));
        CHECK(hipMemcpy(recs, h.data(), n * 64, hipMemcpyHostToDevice));
        printf("records %zu (%.2f MB)\n", n, n * 64 / 1e6);
        unsigned m = (unsigned)n - 1;
        run<1, 1>(recs, m, out, e0, e1);
        run<2, 1>(recs, m, out, e0, e1);
        run<3, 1>(recs, m, out, e0, e1);
        run<4, 1>(recs, m, out, e0, e1);
        run<2, 2>(recs, m, out, e0, e1);
        run<4, 2>(recs, m, out, e0, e1);
        run<4, 4>(recs, m, out, e0, e1);
        run<8, 2>(recs, m, out, e0, e1);
        run<8, 8>(recs, m, out, e0, e1);
        for (int act : {64, 48, 40, 32, 16, 8}) run_masked<2>(recs, m, out, e0, e1, act);
        for (int act : {64, 40, 16}) run_masked<4>(recs, m, out, e0, e1, act);
        CHECK(hipFree(recs));
    }
    return 0;
}
