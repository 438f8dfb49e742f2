// Single-precision sin, cos, log and x^5 with the exact results of glibc 2.35's sinf / cosf / logf /
// powf(x, 5.0f).
//
// The reference's host-compiled code calls libm for its transcendentals (include/ptMath.h:34-45),
// and ROCm's device sinf/cosf differ from glibc's in the last bit for a few percent of arguments.
// A last-bit difference in a sampled direction moves every later hit of that path, so images stop
// matching bit for bit.  These functions restate glibc's algorithm (the ARM optimized-routines
// sincosf: double-precision range reduction and minimax polynomials, evaluated with fused
// multiply-adds as the FMA build that glibc selects on current x86-64 CPUs does), so device results
// equal the host's.  logf and powf follow the same library's table-driven log / log2 / exp2 kernels.
// Checked against this image's libm: sinf and cosf on every finite float of both signs, logf on every
// positive finite float, powf(x, 5) on every normal x in (0, 2] -- no mismatch for any of them.
// asinf and atan2f (MovingSphere's uv, include/ptSphere.h:22-28) follow the same library's
// single-precision fdlibm code, which its x86-64 build compiles without fused multiply-adds:
// asinf equal on all 2^32 floats, atanf on all 2^32, atan2f on 2^31 pseudo-random pairs (an eighth
// of them with exponents within 4 of each other) and a grid of special values.
//
// Provenance of the algorithms and constants restated here: sinf/cosf/sincosf/logf/powf are glibc
// 2.35's sysdeps/ieee754/flt-32/{s_sincosf.h,e_logf.c,e_powf.c,e_exp2f_data.c}, contributed by Arm
// (Optimized Routines, Copyright (c) Arm Limited, MIT / LGPL-2.1-or-later as distributed in glibc);
// asinf/atanf/atan2f are glibc 2.35's e_asinf.c, s_atanf.c, e_atan2f.c, derived from FreeBSD /
// fdlibm: "Copyright (C) 1993 by Sun Microsystems, Inc.  All rights reserved.  Developed at SunPro,
// a Sun Microsystems, Inc. business.  Permission to use, copy, modify, and distribute this software
// is freely granted, provided that this notice is preserved."  The code below is a re-expression
// for host and device, not a copy of those files.
#ifndef PT_LIBM_H
#define PT_LIBM_H

#include <cstdint>
#include <cmath>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace ptm {

PT_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PT_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ff; }

// Polynomial for quadrant n (odd: cosine), sign-folded coefficient set `neg`.
PT_HD float sincos_poly(double x, double x2, bool neg, int n)
{
    const double c0 = neg ? -0x1p0 : 0x1p0;
    const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
    const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double t1 = __builtin_fma(x2, s3, s2);
        double x7 = x3 * x2;
        double s = __builtin_fma(x3, s1, x);
        return (float)__builtin_fma(x7, t1, s);
    } else {
        double x4 = x2 * x2;
        double t2 = __builtin_fma(x2, c4, c3);
        double t1 = __builtin_fma(x2, c1, c0);
        double x6 = x4 * x2;
        double c = __builtin_fma(x4, c2, t1);
        return (float)__builtin_fma(x6, t2, c);
    }
}

// |x| < 120: one multiply-subtract by pi/2.
PT_HD double reduce_fast(double x, int* np)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double r = x * hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return __builtin_fma(-(double)n, hpi, x);
}

// 4/pi, 192 bits, in 8-bit steps (word i holds bits starting 8*i bits into the expansion).
PT_HD uint32_t inv_pio4_word(int i)
{
    // the 24 overlapping words are windows of these bytes
    const unsigned char b[28] = { 0x00, 0x00, 0x00, 0xa2, 0xf9, 0x83, 0x6e, 0x4e, 0x44, 0x15, 0x29, 0xfc, 0x27, 0x57,
                                  0xd1, 0xf5, 0x34, 0xdd, 0xc0, 0xdb, 0x62, 0x95, 0x99, 0x3c, 0x43, 0x90, 0x41, 0x00 };
    return ((uint32_t)b[i] << 24) | ((uint32_t)b[i + 1] << 16) | ((uint32_t)b[i + 2] << 8) | (uint32_t)b[i + 3];
}

// |x| >= 120: integer Payne-Hanek step against 4/pi.
PT_HD double reduce_large(uint32_t xi, int* np)
{
    const int base = (xi >> 26) & 15;
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * inv_pio4_word(base));
    res1 = (uint64_t)xi * inv_pio4_word(base + 4);
    res2 = (uint64_t)xi * inv_pio4_word(base + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}

PT_HD float sinf_glibc(float y)
{
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sincos_poly(x, x * x, false, 0);
    } else if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return sincos_poly(x * s, x * x, (n & 2) != 0, n);
    } else if (abstop12(y) < 0x7f8) {
        const uint32_t xi = f2u(y);
        const int sign = (int)(xi >> 31);
        x = reduce_large(xi, &n);
        const int q = (n + sign) & 3;
        const double s = (q == 1 || q == 2) ? -1.0 : 1.0;
        return sincos_poly(x * s, x * x, ((n + sign) & 2) != 0, n);
    }
    return y - y;
}

PT_HD float cosf_glibc(float y)
{
    double x = y;
    int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sincos_poly(x, x * x, false, 1);
    } else if (abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, &n);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        return sincos_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
    } else if (abstop12(y) < 0x7f8) {
        const uint32_t xi = f2u(y);
        const int sign = (int)(xi >> 31);
        x = reduce_large(xi, &n);
        const int q = (n + sign) & 3;
        const double s = (q == 1 || q == 2) ? -1.0 : 1.0;
        return sincos_poly(x * s, x * x, ((n + sign) & 2) != 0, n ^ 1);
    }
    return y - y;
}


// sinf_glibc(y) and cosf_glibc(y) with one shared range reduction; the two results are the same
// bits as the separate calls.  Both polynomials are evaluated once on the reduced argument and the
// quadrant only picks which result is the sine and which signs they carry: negating x negates the
// odd polynomial exactly, and the sign-folded coefficient set negates the even one exactly
// (round-to-nearest is symmetric), so no lane evaluates a polynomial twice and a wave whose lanes
// sit in different quadrants does not diverge.  Below pi/4 the reduction yields n = 0 and returns
// x itself, so that range needs no branch of its own.
// BELOW_120: the caller knows 0 <= y < 120 (an angle of 2 pi times a number below 1), so the large-argument
// reduction and the non-finite case are not compiled in; the result is the same.
template <bool BELOW_120 = false>
PT_HD void sincosf_glibc(float y, float* sp, float* cp)
{
    double x = y;
    int n, q;
    if (abstop12(y) < abstop12(0x1p-12f)) {     // also keeps the sign of a zero argument
        *sp = y; *cp = 1.0f;
        return;
    }
    if (BELOW_120 || abstop12(y) < abstop12(120.0f)) {
        x = reduce_fast(x, &n);
        q = n;
    } else if (abstop12(y) < 0x7f8) {
        const uint32_t xi = f2u(y);
        x = reduce_large(xi, &n);
        q = n + (int)(xi >> 31);
    } else {
        *sp = y - y; *cp = y - y;
        return;
    }
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
                 c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x2 = x * x;
    const double x3 = x * x2, x4 = x2 * x2;
    const double ts = __builtin_fma(x2, s3, s2);
    const double tc2 = __builtin_fma(x2, c4, c3);
    const double tc1 = __builtin_fma(x2, c1, c0);
    const double x7 = x3 * x2, x6 = x4 * x2;
    const double s = __builtin_fma(x3, s1, x);
    const double c = __builtin_fma(x4, c2, tc1);
    const uint32_t odd = f2u((float)__builtin_fma(x7, ts, s));     // sine polynomial of +x
    const uint32_t even = f2u((float)__builtin_fma(x6, tc2, c));   // cosine polynomial, positive set
    // sign of the odd polynomial: x was multiplied by -1 in quadrants 1 and 2 of q;
    // sign of the even one: the negated coefficient set is used when q & 2.
    const uint32_t odd_sign = (uint32_t)(((q & 3) == 1 || (q & 3) == 2) ? 1 : 0) << 31;
    const uint32_t even_sign = (uint32_t)((q >> 1) & 1) << 31;
    const uint32_t o = odd ^ odd_sign, e = even ^ even_sign;
    uint32_t sb = (n & 1) ? e : o, cb = (n & 1) ? o : e;
    memcpy(sp, &sb, 4);
    memcpy(cp, &cb, 4);
}

PT_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
PT_HD uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
PT_HD double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

// ---- logf: 16-interval table + degree-3 polynomial in double ----------------------------------
// One 16-byte row per interval, read with the lane's own index: on the device the array lives in constant memory and a
// lookup is one load.  (Written as a switch over the index, the sixteen rows became thirty-two double constants that the
// compiler kept in registers across the kernels' bounce loop -- that is, in scratch memory, reloaded through a tree of
// branches at every lookup.)
PT_HD void logf_entry(int i, double* invc, double* logc)
{
    alignas(16) const double t[16][2] = {
        { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 },
        { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
        { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },
        { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
        { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 },
        { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
        { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 },
        { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
        { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 },
        { 0x1p+0, 0x0p+0 },
        { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 },
        { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
        { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 },
        { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 },
        { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },
        { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 },
    };
    *invc = t[i][0]; *logc = t[i][1];
}
PT_HD float logf_glibc(float x)
{
    uint32_t ix = f2u(x);
    if (ix == 0x3f800000) return 0;
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
        if (ix * 2 == 0) return -__builtin_huge_valf();
        if (ix == 0x7f800000) return x;
        if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return (x - x) / (x - x);
        ix = f2u(x * 0x1p23f);
        ix -= 23 << 23;
    }
    const uint32_t tmp = ix - 0x3f330000;
    const int i = (tmp >> 19) % 16;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0x1ffu << 23);
    double invc, logc;
    logf_entry(i, &invc, &logc);
    const double z = (double)u2f(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = __builtin_fma((double)k, 0x1.62e42fefa39efp-1, logc);
    const double r2 = r * r;
    double y = __builtin_fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = __builtin_fma(-0x1.00ea348b88334p-2, r2, y);
    y = __builtin_fma(y, r2, y0 + r);
    return (float)y;
}

// ---- powf(x, 5.0f): log2 table kernel, then exp2 table kernel, both scaled by 32 ------------------
// Rows as in logf_entry: one load per lookup.
PT_HD void powf_log2_entry(int i, double* invc, double* logc)
{
    alignas(16) const double t[16][2] = {
        { 0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2 * 32 },
        { 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2 * 32 },
        { 0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2 * 32 },
        { 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2 * 32 },
        { 0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2 * 32 },
        { 0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3 * 32 },
        { 0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3 * 32 },
        { 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4 * 32 },
        { 0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5 * 32 },
        { 0x1p+0, 0x0p+0 },
        { 0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4 * 32 },
        { 0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3 * 32 },
        { 0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3 * 32 },
        { 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2 * 32 },
        { 0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2 * 32 },
        { 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 * 32 },
    };
    *invc = t[i][0]; *logc = t[i][1];
}
// bits of 2^(i/32) with i << 47 taken off the exponent field
PT_HD uint64_t exp2_entry(int i)
{
    const uint64_t t[32] = {
        0x3ff0000000000000ULL, 0x3fefd9b0d3158574ULL, 0x3fefb5586cf9890fULL, 0x3fef9301d0125b51ULL,
        0x3fef72b83c7d517bULL, 0x3fef54873168b9aaULL, 0x3fef387a6e756238ULL, 0x3fef1e9df51fdee1ULL,
        0x3fef06fe0a31b715ULL, 0x3feef1a7373aa9cbULL, 0x3feedea64c123422ULL, 0x3feece086061892dULL,
        0x3feebfdad5362a27ULL, 0x3feeb42b569d4f82ULL, 0x3feeab07dd485429ULL, 0x3feea47eb03a5585ULL,
        0x3feea09e667f3bcdULL, 0x3fee9f75e8ec5f74ULL, 0x3feea11473eb0187ULL, 0x3feea589994cce13ULL,
        0x3feeace5422aa0dbULL, 0x3feeb737b0cdc5e5ULL, 0x3feec49182a3f090ULL, 0x3feed503b23e255dULL,
        0x3feee89f995ad3adULL, 0x3feeff76f2fb5e47ULL, 0x3fef199bdd85529cULL, 0x3fef3720dcef9069ULL,
        0x3fef5818dcfba487ULL, 0x3fef7c97337b9b5fULL, 0x3fefa4afa2a490daULL, 0x3fefd0765b6e4540ULL };
    return t[i];
}
PT_HD float pow5f_glibc(float x)
{
    uint32_t sign_bias = 0;
    uint32_t ix = f2u(x);
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
        if (2 * ix == 0 || 2 * ix >= 2u * 0x7f800000) {       // zero, inf or nan
            float x2 = x * x;
            if (ix & 0x80000000) x2 = -x2;                      // 5 is an odd integer
            return x2;
        }
        if (ix & 0x80000000) { sign_bias = 1u << 16; ix &= 0x7fffffff; }
        if (ix < 0x00800000) {
            ix = f2u(x * 0x1p23f);
            ix &= 0x7fffffff;
            ix -= 23 << 23;
        }
    }
    // log2(x) * 32
    const uint32_t tmp = ix - 0x3f330000;
    const int i = (tmp >> 19) % 16;
    const uint32_t top = tmp & 0xff800000;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 18;
    double invc, logc;
    powf_log2_entry(i, &invc, &logc);
    const double z = (double)u2f(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = __builtin_fma(0x1.27616c9496e0bp-2 * 32, r, -0x1.71969a075c67ap-2 * 32);
    const double p = __builtin_fma(0x1.ec70a6ca7baddp-2 * 32, r, -0x1.7154748bef6c8p-1 * 32);
    const double r4 = r2 * r2;
    double q = __builtin_fma(0x1.71547652ab82bp0 * 32, r, y0);
    q = __builtin_fma(p, r2, q);
    y = __builtin_fma(y, r4, q);
    const double ylogx = 5.0 * y;
    if (((d2u(ylogx) >> 47) & 0xffff) >= (d2u(126.0 * 32) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6 * 32) return sign_bias ? -__builtin_huge_valf() : __builtin_huge_valf();
        if (ylogx <= -150.0 * 32) return sign_bias ? -0.0f : 0.0f;
    }
    // exp2(ylogx / 32)
    double kd = ylogx + 0x1.8p+52;
    const uint64_t ki = d2u(kd);
    kd -= 0x1.8p+52;
    const double rr = ylogx - kd;
    uint64_t t = exp2_entry((int)(ki % 32));
    t += (ki + sign_bias) << 47;
    const double s = u2d(t);
    const double zz = __builtin_fma(0x1.c6af84b912394p-5 / 32 / 32 / 32, rr, 0x1.ebfce50fac4f3p-3 / 32 / 32);
    const double rr2 = rr * rr;
    double yy = __builtin_fma(0x1.62e42ff0c52d6p-1 / 32, rr, 1.0);
    yy = __builtin_fma(zz, rr2, yy);
    yy = yy * s;
    return (float)yy;
}

// ---- asinf, atanf, atan2f: plain single-precision evaluation, one rounding per operation -----------
PT_HD float asinf_glibc(float x)
{
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f;
    const float p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;               // asin(+-1) = +-pi/2
    if (ix > 0x3f800000) return (x - x) / (x - x);                        // |x| > 1 or NaN
    if (ix < 0x3f000000) {                                                // |x| < 0.5
        if (ix < 0x32000000) return x;                                    // |x| < 2^-27
        const float t = x * x;
        const float w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
        return x + x * w;
    }
    const float w = 1.0f - fabsf(x);
    float t = w * 0.5f;
    float p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    const float s = sqrtf(t);
    if (ix >= 0x3F79999A) {                                               // |x| > 0.975
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        const float sh = u2f(f2u(s) & 0xfffff000u);                       // s with its low 12 bits cleared
        const float c = (t - sh * sh) / (s + sh);
        const float r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        const float q = pio4_hi - 2.0f * sh;
        t = pio4_hi - (p - q);
    }
    return hx > 0 ? t : -t;
}

PT_HD float atanf_glibc(float x)
{
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    float hi = 0.0f, lo = 0.0f;                                           // atan of the interval's pivot, in two parts
    bool reduced = true;
    if (ix >= 0x4c000000) {                                               // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        const float h3 = u2f(0x3fc90fda), l3 = u2f(0x33a22168);
        return hx > 0 ? h3 + l3 : -h3 - l3;
    }
    if (ix < 0x3ee00000) {                                                // |x| < 7/16
        if (ix < 0x31000000) return x;                                    // |x| < 2^-29
        reduced = false;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {                                            // |x| < 19/16
            if (ix < 0x3f300000) { hi = u2f(0x3eed6338); lo = u2f(0x31ac3769); x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { hi = u2f(0x3f490fda); lo = u2f(0x33222168); x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { hi = u2f(0x3f7b985e); lo = u2f(0x33140fb4); x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { hi = u2f(0x3fc90fda); lo = u2f(0x33a22168); x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (u2f(0x3eaaaaab) + w * (u2f(0x3e124925) + w * (u2f(0x3dba2e6e) + w * (u2f(0x3d886b35) + w * (u2f(0x3d4bda59) + w * u2f(0x3c8569d7))))));
    const float s2 = w * (u2f(0xbe4ccccd) + w * (u2f(0xbde38e38) + w * (u2f(0xbd9d8795) + w * (u2f(0xbd6ef16b) + w * u2f(0xbd15a221)))));
    if (!reduced) return x - x * (s1 + s2);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

PT_HD float atan2f_glibc(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = u2f(0x3f490fdb), pi_o_2 = u2f(0x3fc90fdb), pi = u2f(0x40490fdb), pi_lo = u2f(0xb3bbbd2e);
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;                 // NaN
    if (hx == 0x3f800000) return atanf_glibc(y);                          // x = 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                    // 2 * sign(x) + sign(y)
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                                // |y / x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;                                 // |y| / x < -2^60
    else z = atanf_glibc(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return u2f(f2u(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

}  // namespace ptm
#endif
