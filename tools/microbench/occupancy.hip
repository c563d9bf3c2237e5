// How many 256-thread workgroups with a given LDS footprint really run together on a CU?
// Each workgroup records when it starts and spins ~200 us; late starters show up in the histogram.
// Build: hipcc -O3 --offload-arch=gfx950 occupancy.hip -o occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k(unsigned long long* start, int* sink) {
    __shared__ int lds[LDS_BYTES / 4];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    int acc = 0;
    while (wall_clock64() - t0 < 20000ull) acc += lds[(threadIdx.x * 7 + acc) & 255];   // 200 us at 100 MHz
    if (acc == 123456789) sink[0] = acc;
}

template <int LDS_BYTES>
void run(int per_cu) {
    const int blocks = 256 * per_cu;
    unsigned long long* d; CHECK(hipMalloc(&d, blocks * 8));
    int* sink; CHECK(hipMalloc(&sink, 4));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<LDS_BYTES>, 256, 0));
    k<LDS_BYTES><<<blocks, 256>>>(d, sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int late = 0;
    for (auto t : h) late += (t - t0 > 10000ull);
    printf("LDS %6d B, %d workgroups per CU launched: API says %d per CU fit; %d of %d started late\n", LDS_BYTES, per_cu, occ, late, blocks);
    CHECK(hipFree(d)); CHECK(hipFree(sink));
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d, LDS per block max %zu, per CU %zu, regs per CU %d\n", p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerMultiprocessor);
    run<24576>(5); run<24576>(6); run<24576>(7);
    run<20480>(6); run<20480>(7); run<20480>(8);
    run<16384>(8); run<16384>(10);
    run<32768>(4); run<32768>(5);
    return 0;
}
