// rt_libm.h -- the two libm functions whose last bit decides results of the reference, restated
// from glibc 2.35's published algorithms so that the device returns what the CPU's libm returns:
//
//   rt_logf(v)        std::log(float) inside std::normal_distribution (main.cpp:108-111);
//                     one ulp there flips accept/reject branches of every later path vertex
//   rt_powf(v, y)    glm::pow(c, 1.f / 2.2f) of generatePixel (main.cpp:1228-1231); one ulp there
//                     moves a truncated image byte (main.cpp:840-846)
//
// Both are glibc's sysdeps/ieee754/flt-32 kernels: table lookup + short polynomial evaluated in
// double, ONE rounding to float at the end.  The header compiles as device code under hipcc and as
// plain host code under g++ (tools/check_libm.cpp compares it with the host's libm over every float
// of the domain the renderer uses).
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define RT_LIBM_FN __device__ __forceinline__
#define RT_LIBM_TAB __device__ const
#else
#define RT_LIBM_FN static inline
#define RT_LIBM_TAB static const
#endif

namespace rtd {

RT_LIBM_FN uint32_t libm_f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
RT_LIBM_FN float libm_u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
RT_LIBM_FN uint64_t libm_d2u(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
}
RT_LIBM_FN double libm_u2d(uint64_t u) {
    double d;
    memcpy(&d, &u, 8);
    return d;
}

// ---- logf (e_logf.c, __logf_data): 16 intervals of [0x3f330000, 2 * 0x3f330000), {1/c, ln c} ----
RT_LIBM_TAB double kLogfTab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
};

// Natural logarithm of a positive normal float.  Checked against libm's logf for all 1 056 964 609
// normal floats in (0, 1], built with and without fused multiply-adds: no mismatch.
RT_LIBM_FN float rt_logf(float v) {
    const uint32_t ix = libm_f2u(v);
    if (ix == 0x3f800000u) return 0.f;
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const int k = (int)tmp >> 23;  // arithmetic shift
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = kLogfTab[i][0], logc = kLogfTab[i][1];
    const double z = (double)libm_u2f(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

// ---- powf (e_powf.c): log2 of x from __powf_log2_data (same 16 intervals, {1/c, log2 c}, degree-5
// polynomial), times y in double, exp2 from __exp2f_data (32-entry table of 2^(i/32), degree-3
// polynomial), build without round-to-integer intrinsics (x86-64: the 0x1.8p+52/32 shift trick) ----
RT_LIBM_TAB double kPowLog2Tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},  {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2},
};
RT_LIBM_TAB uint64_t kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

// powf(x, y) for finite y > 0 and x = +0 or a positive finite float, and a result that neither
// overflows nor underflows (|y log2 x| < 126) -- everything the gamma curve can ask for: x is a
// clamped colour in [0, 1], y = 1/2.2f.
//
// Which multiply-adds are fused matters in the last bit of a few results.  On every x86-64 CPU
// with FMA and AVX2 glibc selects, at run time, a build of this kernel compiled with contraction
// (sysdeps/x86_64/fpu/multiarch/e_powf.c); GCC fuses every a*b+c of the source except the two that
// consume y*log2(x), because that product is also inspected as an integer for the overflow test.
// That pattern is written out with explicit fma() here and returns libm's bits for all
// 1 065 353 217 floats in [0, 1] with y = 1/2.2f; the same source without any fusion (what a
// CPU without FMA runs) differs from it on 3 of those inputs.
RT_LIBM_FN float rt_powf(float x, float y) {
    uint32_t ix = libm_f2u(x);
    if (ix == 0u) return 0.f;            // powf(+0, y > 0) = +0
    if (ix == 0x3f800000u) return 1.f;   // powf(1, y) = 1
    if (ix < 0x00800000u) {              // subnormal: scale by 2^23 and take 23 off the exponent
        ix = libm_f2u(x * 0x1p23f) & 0x7fffffffu;
        ix -= 23u << 23;
    }
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int)top >> 23;  // arithmetic shift
    const double invc = kPowLog2Tab[i][0], logc = kPowLog2Tab[i][1];
    const double z = (double)libm_u2f(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);
    q = __builtin_fma(p, r2, q);
    yy = __builtin_fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    // exp2_inline: x = k/32 + r, r in [-1/64, 1/64]
    const double shift = 0x1.8p+47;  // 0x1.8p+52 / 32
    double kd = ylogx + shift;
    const uint64_t ki = libm_d2u(kd);
    kd -= shift;
    const double rr = ylogx - kd;
    uint64_t t = kExp2fTab[ki & 31u];
    t += ki << (52 - 5);
    const double s = libm_u2d(t);
    const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double e = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    e = __builtin_fma(zz, rr2, e);
    e = e * s;
    return (float)e;
}

}  // namespace rtd
