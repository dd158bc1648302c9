// powf, acosf and atan2f with the bits the reference gets from its C library.
//
// The reference calls powf in the Phong term (src/Scene.cpp:78) and acos/atan2 -- both binding to
// the float overloads -- in the sky lookup (src/SceneObjects.cpp:441-442).  A device math library
// that is merely accurate differs from glibc in the last bit often enough to flip a colour level
// or a texel here and there, so these three are restated from the algorithms glibc 2.35 uses on
// x86-64 (the platform the reference frames were produced on) and checked against it on the CPU
// over 10^9 inputs each (all 2.1e9 floats in [-1, 1] for acosf):
//   - powf: the table-driven log2/exp2 scheme in double precision (16-entry log2 table, degree-4
//     polynomial; 32-entry 2^(k/32) table, degree-3 polynomial).  glibc selects its FMA build of
//     this routine on every CPU with FMA, so the polynomial steps below are explicit double fma();
//   - acosf, atanf/atan2f: the classic fdlibm single-precision rational/polynomial kernels,
//     evaluated in float without fusing.
// Inputs outside what the trace loop produces (negative bases, NaN, infinities) take simplified
// paths; the loop never passes them.
#pragma once
#include "rtx_device.hpp"

namespace rtx {

// x >= 0 (the caller tests specIntDot >= 0), y finite
RTX_DEV float refPowf(const RefMathTables &T, float x, float y) {
#ifdef RTX_FAST_MATH
    // tolerance build: 2^(y log2 x) on the hardware's exp / log units (relative error about 1e-5 at y = 1425)
    (void)T;
    return y == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
#endif
    unsigned ix = __float_as_uint(x);
    if (y == 0.0f) return 1.0f;
    if (ix == 0x3f800000u) return 1.0f;
    if ((ix << 1) == 0u) return 0.0f;  // +-0 to a positive power
    if (ix < 0x00800000u) {            // subnormal: scale up, fix the exponent
        ix = __float_as_uint(x * 0x1p23f) & 0x7fffffffu;
        ix -= 23u << 23;
    }
    // log2(x) = log2(z/c) + log2(c) + k with z in [0x1.66p-1, 0x1.66p0), c the sub-interval centre
    const unsigned tmp = ix - 0x3f330000u;
    const int i = (int)(tmp >> 19) & 15;
    const unsigned top = tmp & 0xff800000u;
    const int k = (int)top >> 23;
    const double invc = T.log2_tab[i][0], logc = T.log2_tab[i][1];
    const double z = (double)__uint_as_float(ix - top);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double p0 = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p1 = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);
    q = __builtin_fma(p1, r2, q);
    const double logx = __builtin_fma(p0, r4, q);
    const double ylogx = (double)y * logx;
    if (((unsigned)((unsigned long long)__double_as_longlong(ylogx) >> 47) & 0xffffu) >= (unsigned)(0x405f800000000000ull >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return __uint_as_float(0x7f800000u);
        if (ylogx <= -150.0) return 0.0f;
    }
    // 2^(ylogx) = 2^(k/32) * 2^r, k/32 obtained by the shift trick
    const double shift = 0x1.8p+47;
    double kd = ylogx + shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= shift;
    const double rr = ylogx - kd;
    const unsigned long long t = T.exp2_tab[ki & 31u] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double yy = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    yy = __builtin_fma(zz, rr2, yy);
    return (float)(yy * s);
}

// The three routines below follow fdlibm's case analysis, but a wave executes every case some lane
// takes, so whatever the cases have in common -- the rational approximation and its division in
// acosf, the division of the argument reduction in atanf -- is written once, on operands selected
// per case.  Each case still performs exactly the operations fdlibm performs, in its order.

// acosf / atan2f decide a texel of the environment map: unfused in the tolerance build as well (the pragma holds to the end of the file)
#pragma clang fp contract(off)

// fdlibm e_acosf
RTX_DEV float refAcosf(float x) {
    const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return __uint_as_float(0x7fc00000u);  // |x| > 1: NaN
    const bool small = ix < 0x3f000000;                        // |x| < 0.5
    if (small && ix <= 0x23000000) return pio2_hi + pio2_lo;
    // z = x*x, (1+x)/2 or (1-x)/2; for x < -0.5, 1 + x is the same operation as 1 - |x|
    const float z = small ? x * x : (1.0f - fabsf(x)) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    if (small) return pio2_hi - (x - (pio2_lo - x * r));
    const float s = sqrtf(z);
    if (hx < 0) {  // x < -0.5
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);  // x > 0.5
    const float c = (z - df * df) / (s + df);
    const float w = r * s + c;
    return 2.0f * (df + w);
}

// fdlibm s_atanf
RTX_DEV float refAtanf(float x) {
    const float aT[11] = {3.3333334327e-01f,  -2.0000000298e-01f, 1.4285714924e-01f,  -1.1111110449e-01f,
                          9.0908870101e-02f,  -7.6918758452e-02f, 6.6610731184e-02f,  -5.8335702866e-02f,
                          4.9768779427e-02f,  -3.6531571299e-02f, 1.6285819933e-02f};
    const int hx = (int)__float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix >= 0x50800000) {  // |x| >= 2^34
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? 1.5707962513e+00f + 7.5497894159e-08f : -1.5707962513e+00f - 7.5497894159e-08f;
    }
    const bool small = ix < 0x3ee00000;  // |x| < 0.4375: no reduction
    if (small && ix < 0x31000000) return x;
    // reduced argument num / den: (2x-1)/(2+x), (x-1)/(x+1), (x-1.5)/(1+1.5x), -1/x
    const float ax = fabsf(x);
    float num, den, hi, lo;
    if (ix < 0x3f300000) num = 2.0f * ax - 1.0f, den = 2.0f + ax, hi = 4.6364760399e-01f, lo = 5.0121582440e-09f;
    else if (ix < 0x3f980000) num = ax - 1.0f, den = ax + 1.0f, hi = 7.8539812565e-01f, lo = 3.7748947079e-08f;
    else if (ix < 0x401c0000) num = ax - 1.5f, den = 1.0f + 1.5f * ax, hi = 9.8279368877e-01f, lo = 3.4473217170e-08f;
    else num = -1.0f, den = ax, hi = 1.5707962513e+00f, lo = 7.5497894159e-08f;
    const float t = small ? x : num / den;
    const float z = t * t, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (small) return t - t * (s1 + s2);
    const float r = hi - ((t * (s1 + s2) - lo) - t);
    return hx < 0 ? -r : r;
}

// fdlibm e_atan2f for finite arguments
RTX_DEV float refAtan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int hx = (int)__float_as_uint(x), hy = (int)__float_as_uint(y);
    const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    const bool unit = hx == 0x3f800000;  // x == 1: atanf(y)
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (!unit) {
        if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
        if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    }
    const int k = (iy - ix) >> 23;
    const float t = refAtanf(unit ? y : fabsf(y / x));  // the one evaluation both cases share
    if (unit) return t;
    float z = t;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    switch (m) {
        case 0: return z;
        case 1: return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

#ifdef RTX_FAST_MATH
#pragma clang fp contract(fast)
#endif

}  // namespace rtx
