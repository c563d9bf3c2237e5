// check_libm.cpp -- compares the restated glibc kernels of raytracing_amd/csrc/rt_libm.h with the
// host's libm, bit for bit, over the domains the renderer uses:
//   logf:  every normal float in (0, 1]            (argument of the polar method's log)
//   powf:  every float in [0, 1], exponent 1.f / 2.2f                     (the gamma curve)
// Build twice to cover both ways the host's libm may have been compiled (glibc selects an FMA
// build of these kernels at run time on CPUs that have it):
//   g++ -O2 -fopenmp -ffp-contract=off            tools/check_libm.cpp -o /tmp/check_libm
//   g++ -O2 -fopenmp -mfma -ffp-contract=off      tools/check_libm.cpp -o /tmp/check_libm_fma   (hardware fma())
// Usage: check_libm [stride]   (stride 1 = exhaustive, about a minute on 8 threads; the unit test
// uses a stride of a few hundred).  Exit status 0 when nothing differs.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../raytracing_amd/csrc/rt_libm.h"

int main(int argc, char **argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)std::strtoul(argv[1], nullptr, 0) : 1u;
    const float gamma = 1.f / 2.2f;
    const uint32_t lo = 0x00800000u, hi = 0x3f800000u;  // smallest normal .. 1.0f
    unsigned long long badLog = 0, badPow = 0, n = 0;
    uint32_t firstLog = 0, firstPow = 0;
#pragma omp parallel for schedule(static) reduction(+ : badLog, badPow, n)
    for (long long b = lo; b <= (long long)hi; b += stride) {
        const float v = rtd::libm_u2f((uint32_t)b);
        const float l0 = logf(v), l1 = rtd::rt_logf(v);
        const float p0 = powf(v, gamma), p1 = rtd::rt_powf(v, gamma);
        ++n;
        if (rtd::libm_f2u(l0) != rtd::libm_f2u(l1)) {
            ++badLog;
#pragma omp critical
            if (!firstLog) firstLog = (uint32_t)b;
        }
        if (rtd::libm_f2u(p0) != rtd::libm_f2u(p1)) {
            ++badPow;
#pragma omp critical
            if (!firstPow) firstPow = (uint32_t)b;
        }
    }
#pragma omp parallel for schedule(static) reduction(+ : badPow, n)
    for (long long b = 0; b < (long long)lo; b += stride) {  // +0 and the subnormals: powf only
        const float v = rtd::libm_u2f((uint32_t)b);
        ++n;
        if (rtd::libm_f2u(powf(v, gamma)) != rtd::libm_f2u(rtd::rt_powf(v, gamma))) {
            ++badPow;
#pragma omp critical
            if (!firstPow) firstPow = (uint32_t)b;
        }
    }
    std::printf("checked %llu floats in [0, 1] (stride %u; logf on the normal ones): logf mismatches %llu", n, stride, badLog);
    if (badLog) std::printf(" (first at 0x%08x)", firstLog);
    std::printf(", powf(x, 1/2.2f) mismatches %llu", badPow);
    if (badPow) std::printf(" (first at 0x%08x)", firstPow);
    std::printf("\n");
    return (badLog || badPow) ? 1 : 0;
}
