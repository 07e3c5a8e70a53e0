// libm_float.hh -- select-form (branch-free) restatement of the host libm's single-precision
// tanhf, for use inside gfx950 kernels.  No HIP dependency: the same text compiles for the host
// (tests/test_libm_float.py builds it with g++ and sweeps it against libm.so.6 bit for bit).
//
// Why: the Range module calls libm tanhf (src/domains/core/range/module_impl_native_cpu.cc:67-82).
// On the reference's CPU target (x86-64 glibc; this image ships 2.35) that is the FDLIBM float
// pair tanhf -> expm1f (Sun Microsystems' s_tanhf.c / s_expm1f.c as carried in glibc
// sysdeps/ieee754/flt-32), which is accurate to ~1 ulp but NOT correctly rounded, so only a
// restatement of the same operation sequence reproduces its bits.  The published algorithm is a
// ladder of data-dependent branches; on a 64-lane wavefront every lane would walk every taken
// branch, so all alternatives are computed and selected (v_cndmask), with identical arithmetic in
// the selected lane: one rounding per written operation, no FMA (-ffp-contract=off).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define JST_FN __host__ __device__ __forceinline__
#else
#define JST_FN static inline
#endif

namespace jst::dev {

JST_FN uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
JST_FN float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// expm1f(x) for the arguments tanhf feeds it: x = 2|a| with 1 <= |a| < 22, or x = -2|a| with
// 2^-55 <= |a| < 1.  (Other x produce an unspecified value that the caller discards.)
JST_FN float libm_expm1f_for_tanh(float x) {
    constexpr float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f,
                    invln2 = 1.4426950216e+00f, Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f,
                    Q3 = -7.9365076090e-05f, Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
    const uint32_t bits = f2u(x);
    const uint32_t hx = bits & 0x7fffffffu;
    const bool neg = (bits >> 31) != 0;
    const bool red = hx > 0x3eb17218u;  // |x| > 0.5 ln2: argument reduction
    const bool mid = hx < 0x3F851592u;  // ... and |x| < 1.5 ln2: k = +-1

    const int32_t kg = (int32_t)(invln2 * x + (neg ? -0.5f : 0.5f));
    const float tg = (float)kg;
    const float hi_g = x - tg * ln2_hi, lo_g = tg * ln2_lo;
    const float hi_m = neg ? x + ln2_hi : x - ln2_hi;
    const float lo_m = neg ? -ln2_lo : ln2_lo;
    const float hi = mid ? hi_m : hi_g, lo = mid ? lo_m : lo_g;
    const int32_t k = red ? (mid ? (neg ? -1 : 1) : kg) : 0;
    const float xr_red = hi - lo;
    const float c_red = (hi - xr_red) - lo;
    const float xr = red ? xr_red : x;
    const float c = red ? c_red : 0.0f;

    const float hfx = 0.5f * xr;
    const float hxs = xr * hfx;
    const float r1 = 1.0f + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    const float t = 3.0f - r1 * hfx;
    const float e = hxs * ((r1 - t) / (6.0f - xr * t));

    const float res_0 = xr - (xr * e - hxs);  // k == 0 (c is 0)
    float e2 = (xr * (e - c) - c);
    e2 -= hxs;
    const float res_m1 = 0.5f * (xr - e2) - 0.5f;
    const float res_p1 = (xr < -0.25f) ? -2.0f * (e2 - (xr + 0.5f)) : 1.0f + 2.0f * (xr - e2);
    const uint32_t kshift = (uint32_t)k << 23;  // "add k to y's exponent"
    const float d = e2 - xr;
    const float y_far = u2f(f2u(1.0f - d) + kshift) - 1.0f;  // k <= -2 or k > 56
    const float t_lo = u2f(0x3f800000u - (0x1000000u >> ((uint32_t)k & 31u)));  // 1 - 2^-k
    const float y_lo = u2f(f2u(t_lo - d) + kshift);                              // 2 <= k < 23
    const float t_hi = u2f((uint32_t)(0x7f - k) << 23);                          // 2^-k
    float y_hi = xr - (e2 + t_hi);                                               // 23 <= k <= 56
    y_hi += 1.0f;
    y_hi = u2f(f2u(y_hi) + kshift);

    float r = (k < 23) ? y_lo : y_hi;
    r = (k <= -2 || k > 56) ? y_far : r;
    r = (k == 1) ? res_p1 : r;
    r = (k == -1) ? res_m1 : r;
    r = (k == 0) ? res_0 : r;
    r = (hx < 0x33000000u) ? x : r;  // |x| < 2^-25 (only reached unreduced)
    return r;
}

JST_FN float libm_tanhf(float x) {
    const uint32_t jx = f2u(x);
    const uint32_t ix = jx & 0x7fffffffu;
    const float ax = u2f(ix);
    const bool ge1 = ix >= 0x3f800000u;
    const float two_ax = 2.0f * ax;
    const float t = libm_expm1f_for_tanh(ge1 ? two_ax : -two_ax);
    const float q = (ge1 ? 2.0f : t) / (t + 2.0f);
    float z = ge1 ? 1.0f - q : -q;             // one - two/(t+two)   |   -t/(t+two)
    z = (ix >= 0x41b00000u) ? 1.0f : z;        // |x| >= 22 (and +-inf): one - tiny == 1.0f
    float r = ((jx >> 31) != 0) ? -z : z;
    r = (ix < 0x24000000u) ? x * (1.0f + x) : r;  // |x| < 2^-55, including +-0
    r = (ix > 0x7f800000u) ? x + x : r;           // NaN
    return r;
}

// Correctly rounded a / b for the operand ranges tanhf produces (|b| in [1, 2^64], quotient and
// residuals far from the subnormal range).  On the device this is the compiler's own FDIV32
// expansion (rcp, two Newton steps on the reciprocal, two on the quotient, final fma) WITHOUT the
// v_div_scale / v_div_fixup range handling and its VCC hazards; on the host it is the IEEE divide.
// Both are correctly rounded, hence equal.
#if defined(__HIP_DEVICE_COMPILE__)
JST_FN float div_rn_midrange(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = a * r;
    const float e1 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e2, r, q);
}
#else
JST_FN float div_rn_midrange(float a, float b) { return a / b; }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// Candidate shorter divides (see the hooks below): v1 drops the last residual correction, v2 the Newton
// step on the reciprocal, v3 both.
JST_FN float div_v1(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = a * r;
    const float e1 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e1, r, q);
}
JST_FN float div_v2(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    float q = a * r;
    const float e1 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e2, r, q);
}
JST_FN float div_v3(float a, float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    const float e1 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e1, r, q);
}
JST_FN float div_v4(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }  // NOT correctly rounded: sweep sensitivity check
#else
JST_FN float div_v4(float a, float b) { return a / b; }
JST_FN float div_v1(float a, float b) { return a / b; }
JST_FN float div_v2(float a, float b) { return a / b; }
JST_FN float div_v3(float a, float b) { return a / b; }
#endif
// copysign(|magnitude|, sign_of): one v_bfi_b32 instead of a compare + select pair
JST_FN float with_sign_of(float magnitude, float sign_of) {
    return u2f((f2u(magnitude) & 0x7fffffffu) | (f2u(sign_of) & 0x80000000u));
}

// Branch-structured variant of the same two functions, following the published control flow
// (s_tanhf.c / s_expm1f.c) instead of computing every alternative: on the GPU a divergent `if`
// costs an exec-mask update and is skipped outright when no lane of the wavefront takes it, which
// is cheaper than the select form as soon as a reconstruction class is absent from a wavefront,
// and it drops the cmp+cndmask pair per alternative.  Arithmetic identical, operation by operation.
JST_FN float libm_expm1f_for_tanh_branchy(float x) {
    constexpr float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f,
                    invln2 = 1.4426950216e+00f, Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f,
                    Q3 = -7.9365076090e-05f, Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
    const uint32_t bits = f2u(x);
    const uint32_t hx = bits & 0x7fffffffu;
    int32_t k = 0;
    float c = 0.0f;
    if (hx > 0x3eb17218u) {          // |x| > 0.5 ln2
        float hi, lo;
        if (hx < 0x3F851592u) {      // and |x| < 1.5 ln2: k = +-1 by the sign of x
            hi = x - with_sign_of(ln2_hi, x);   // x + ln2_hi == x - (-ln2_hi), bit for bit
            lo = with_sign_of(ln2_lo, x);
            k = 1 | ((int32_t)bits >> 31);      // neg ? -1 : 1
        } else {
            k = (int32_t)(invln2 * x + with_sign_of(0.5f, x));
            const float t = (float)k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (hx < 0x33000000u) {   // |x| < 2^-25
        return x;
    }
    const float hfx = 0.5f * x;
    const float hxs = x * hfx;
    const float r1 = 1.0f + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    float t = 3.0f - r1 * hfx;
    float e = hxs * div_rn_midrange(r1 - t, 6.0f - x * t);
    if (k == 0) return x - (x * e - hxs);
    e = (x * (e - c) - c);
    e -= hxs;
    if (k == -1) return 0.5f * (x - e) - 0.5f;
    if (k == 1) return (x < -0.25f) ? -2.0f * (e - (x + 0.5f)) : 1.0f + 2.0f * (x - e);
    const uint32_t kshift = (uint32_t)k << 23;
    if (k <= -2 || k > 56) return u2f(f2u(1.0f - (e - x)) + kshift) - 1.0f;
    if (k < 23) {
        t = u2f(0x3f800000u - (0x1000000u >> ((uint32_t)k & 31u)));  // 1 - 2^-k
        return u2f(f2u(t - (e - x)) + kshift);
    }
    t = u2f((uint32_t)(0x7f - k) << 23);  // 2^-k
    float y = x - (e + t);
    y += 1.0f;
    return u2f(f2u(y) + kshift);
}

JST_FN float libm_tanhf_branchy(float x) {
    const uint32_t jx = f2u(x);
    const uint32_t ix = jx & 0x7fffffffu;
    const float ax = u2f(ix);
    const bool ge1 = ix >= 0x3f800000u;
    const float two_ax = 2.0f * ax;
    // one expm1f evaluation serves both |x| >= 1 (argument 2|x|) and |x| < 1 (argument -2|x|)
    const float t = libm_expm1f_for_tanh_branchy(ge1 ? two_ax : -two_ax);
    const float q = div_rn_midrange(ge1 ? 2.0f : t, t + 2.0f);
    float z = ge1 ? 1.0f - q : -q;             // one - two/(t+two)   |   -t/(t+two)
    if (ix >= 0x41b00000u) z = 1.0f;           // |x| >= 22 (and +-inf): one - tiny == 1.0f
    float r = u2f(f2u(z) ^ (jx & 0x80000000u));  // (jx < 0) ? -z : z
    if (ix < 0x24000000u) r = x * (1.0f + x);  // |x| < 2^-55, including +-0
    if (ix > 0x7f800000u) r = x + x;           // NaN
    return r;
}

// Division hooks of the main-path form.  tanhf is a function of ONE float, so is every operand pair its
// two divisions ever see: a cheaper sequence than the fully general correctly rounded divide is admissible
// as soon as an exhaustive device sweep (tools/ubench/exact_sweep.hip, tests/test_gpu_exact_sweep.py) shows
// it returns the correctly rounded quotient on all of them.
// Result of that sweep on MI355X (all 2^32 arguments, 0 mismatches, round 2): the shortest candidate, div_v3
// = v_rcp_f32, one product, one exact residual, one fused correction, is correctly rounded on both operand
// families; the uncorrected product div_v4 is not (93 720 wrong tanhf values), which is how the sweep shows
// that it can tell.
#ifndef JST_DIV_EXPM1
#define JST_DIV_EXPM1(a, b) div_v3(a, b)
#endif
#ifndef JST_DIV_TANH
#define JST_DIV_TANH(a, b) div_v3(a, b)
#endif

// ---- main-path form ---------------------------------------------------------------------------
// One straight-line evaluation, free of compare/select pairs, for 2^-26 <= |x| < 7.5 -- every
// argument the Range module produces for a value inside (or within 1.4 spans of) its display range.
// On gfx950 a v_cmp -> v_cndmask pair costs two half-rate instructions plus the SGPR-hazard nops
// between them, and the published ladder of classes turns into a dozen exec-mask regions per
// element, so the classes are merged arithmetically instead:
//   * the expm1f argument a = +-2|x| gets its sign by a bit operation;
//   * ONE argument reduction serves all of them: glibc's |a| <= 0.5 ln2 (k = 0, no reduction) and
//     0.5 ln2 < |a| < 1.5 ln2 (k = +-1 by the sign) shortcuts compute exactly what the general
//     formula k = (int)(invln2*a +- 0.5), hi = a - k*ln2_hi, lo = k*ln2_lo computes whenever the
//     general k agrees (k*ln2_hi, k*ln2_lo are exact for |k| <= 1; k = 0 gives hi = a, lo = c = 0);
//     it does agree on every float of both intervals (tests/test_libm_float.py sweeps all of them);
//   * the reconstructions for k <= -2 and 2 <= k < 23 are one formula: 0x1000000 >> (k & 31) is 0
//     for k in {-7..-1}, i.e. t = 1 - 2^-k degenerates to the `one` of the k <= -2 branch, and the
//     trailing `- one` of that branch becomes `- (|x| >= 1 ? 0 : 1)` (y - 0 == y bit for bit);
//   * k == 0 and k == -1 (|x| < 0.52, the middle quarter of the display range) keep their own
//     three-operation reconstructions, in one exec-masked block that wavefronts without such a
//     lane skip;
//   * z = (|x| >= 1 ? 1 : 0) - (|x| >= 1 ? 2 : t) / (t + 2): `0 - q` is `-q` bit for bit (q != 0).
// Everything else (|x| < 2^-26 incl. 0, |x| >= 7.5 incl. inf, NaN) is reported through `rare` and
// left to libm_tanhf_branchy.  Same operations in the same order as s_tanhf.c / s_expm1f.c on the
// taken path, so the bits are glibc's.
JST_FN float libm_tanhf_main(float x, bool& rare) {
    constexpr float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f,
                    invln2 = 1.4426950216e+00f, Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f,
                    Q3 = -7.9365076090e-05f, Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
    const uint32_t jx = f2u(x);
    const uint32_t ix = jx & 0x7fffffffu;
    rare = (ix - 0x32800000u) >= (0x40f00000u - 0x32800000u);
    const uint32_t mA = (uint32_t)((int32_t)(0x3f7fffffu - ix) >> 31);  // ~0 when |x| >= 1
    const float a = u2f((ix + 0x00800000u) | (~mA & 0x80000000u));       // +-2|x|
    const int32_t k = (int32_t)(invln2 * a + with_sign_of(0.5f, a));
    const float t = (float)k;
    const float hi = a - t * ln2_hi, lo = t * ln2_lo;
    const float xr = hi - lo;
    const float c = (hi - xr) - lo;
    const float hfx = 0.5f * xr;
    const float hxs = xr * hfx;
    const float r1 = 1.0f + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    const float t3 = 3.0f - r1 * hfx;
    const float e = hxs * JST_DIV_EXPM1(r1 - t3, 6.0f - xr * t3);
    float e2 = (xr * (e - c) - c);
    e2 -= hxs;
    const float tl = u2f(0x3f800000u - (0x1000000u >> ((uint32_t)k & 31u)));  // 1 - 2^-k (1 for k < 0)
    const float y = u2f(f2u(tl - (e2 - xr)) + ((uint32_t)k << 23));
    float em1 = y - u2f(~mA & 0x3f800000u);
    if ((uint32_t)(k + 1) < 2u) {  // k == 0 or k == -1
#if defined(__HIP_DEVICE_COMPILE__) && !defined(JST_TANH_KBLOCK_SELECT)
        __asm__ volatile("");  // keep it a branch (a wavefront without such a lane skips it); if-converted it is
                               // six operations and two selects for everyone
#endif
        const float r0 = xr - (xr * e - hxs);
        const float rm = 0.5f * (xr - e2) - 0.5f;
        em1 = (k == 0) ? r0 : rm;
    }
    const float num = u2f((mA & 0x40000000u) | (~mA & f2u(em1)));
    const float q = JST_DIV_TANH(num, em1 + 2.0f);
    const float z = u2f(mA & 0x3f800000u) - q;
    return u2f(f2u(z) ^ (jx & 0x80000000u));
}


// =================================================================================================
// sinf / cosf / atanf / atan2f of the host libm the reference's FM module calls
// (src/domains/dsp/fm/module_impl_native_cpu.cc:93,123-139), glibc 2.35 on x86-64, restated from the
// published sources AND checked against the shipped binary (objdump of libm.so.6: constants read back from
// its tables, the placement of every fused multiply-add read off the instruction stream):
//   * sinf / cosf are sysdeps/ieee754/flt-32/s_sinf.c / s_cosf.c (ARM's 2018 routines): the argument is
//     widened to double, reduced by n = round(x * 2/pi) with ONE fused x - n*(pi/2), and a degree-7 / degree-8
//     polynomial in double gives the result, rounded to float once.  libm resolves them through IFUNC to the
//     `_fma` build on every CPU with FMA + AVX2 (all current x86 servers), whose polynomial steps are contracted
//     exactly as written below (fma_d = one rounding).  |x| >= 120 goes through the 192-bit 4/pi table
//     (reduce_large), integer arithmetic, restated literally.
//   * atanf / atan2f are FDLIBM's s_atanf.c / e_atan2f.c in float arithmetic; libm ships ONE build of them
//     (baseline x86-64, no FMA): every operation below is one rounding (-ffp-contract=off), the divisions are
//     IEEE divisions on both sides.
// tests/test_libm_float.py sweeps all four against libm.so.6 on the host (sinf/cosf/atanf: every float).
// =================================================================================================
JST_FN double fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }

namespace sincosf_data {
constexpr double kHpiInv = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
constexpr double kHpi = 0x1.921FB54442D18p0;       // pi/2
constexpr double kPi63 = 0x1.921FB54442D18p-62;    // 2 pi / 2^64
constexpr double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
constexpr double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
}  // namespace sincosf_data

// sinf_poly, (n & 1) == 0 branch: s + x7*s1 with s1 = S2 + x2*S3, s = x + x3*S1 (three fused steps)
JST_FN double sincosf_sin_poly(double x, double x2) {
    using namespace sincosf_data;
    const double s1 = fma_d(x2, S3, S2);
    const double x3 = x * x2;
    const double x7 = x3 * x2;
    const double s = fma_d(x3, S1, x);
    return fma_d(s1, x7, s);
}
// sinf_poly, (n & 1) == 1 branch; `negate` selects __sincosf_table[1], whose cosine coefficients are the negated
// ones: every step is odd in the coefficients, so the result is exactly the negated value.
JST_FN double sincosf_cos_poly(double x2, bool negate) {
    using namespace sincosf_data;
    const double x4 = x2 * x2;
    const double c1 = fma_d(x2, C1, C0);
    const double c2 = fma_d(x2, C4, C3);
    const double x6 = x2 * x4;
    const double c = fma_d(x4, C2, c1);
    const double r = fma_d(c2, x6, c);
    return negate ? -r : r;
}
// reduce_large (sincosf.h): x mod pi/2 from the 4/pi bits, |x| >= 120.  Returns the reduced argument, *np = quadrant.
JST_FN double sincosf_reduce_large(uint32_t xi, int32_t* np) {
    const uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                                   0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                   0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                                   0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (int)((xi >> 23) & 7);
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    uint64_t res0 = (uint32_t)(xi * arr[0]);
    const uint64_t res1 = (uint64_t)xi * arr[4];
    const uint64_t res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = (int32_t)n;
    return (double)(int64_t)res0 * sincosf_data::kPi63;
}
// the shared body: COS = false gives sinf, true gives cosf (the two sources differ in the tiny-argument result and
// in the parity handed to sinf_poly: n for sinf, n ^ 1 for cosf)
template <bool COS>
JST_FN float libm_sincosf(float y) {
    using namespace sincosf_data;
    const uint32_t bits = f2u(y);
    const uint32_t top = (bits >> 20) & 0x7ffu;  // abstop12
    const double x = (double)y;
    if (top <= 0x3f3u) {  // |y| < pi/4
        const double x2 = x * x;
        if (top <= 0x397u) return COS ? 1.0f : y;  // |y| < 2^-12
        return (float)(COS ? sincosf_cos_poly(x2, false) : sincosf_sin_poly(x, x2));
    }
    double xr;
    int32_t n, q;  // n: quadrant whose parity picks the polynomial; q: quadrant for sign / table
    if (top <= 0x42eu) {  // |y| < 120: reduce_fast
        const double r = x * kHpiInv;
        n = ((int32_t)r + 0x800000) >> 24;
        xr = fma_d(-(double)n, kHpi, x);
        q = n;
    } else if (top <= 0x7f7u) {
        xr = sincosf_reduce_large(bits, &n);
        q = n + (int32_t)(bits >> 31);
    } else {
        return y - y;  // inf, NaN -> NaN (__math_invalidf)
    }
    const double x2 = xr * xr;
    const bool odd = ((n & 1) != 0) != COS;  // sinf: cosine polynomial when n is odd; cosf: when n is even
    if (odd) return (float)sincosf_cos_poly(x2, (q & 2) != 0);
    const double sgn = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0;  // sign[] = {1, -1, -1, 1}
    return (float)sincosf_sin_poly(xr * sgn, x2);
}
JST_FN float libm_sinf(float y) { return libm_sincosf<false>(y); }
JST_FN float libm_cosf(float y) { return libm_sincosf<true>(y); }

// FDLIBM s_atanf.c (float arithmetic; constants as stored in libm.so.6)
JST_FN float libm_atanf(float x) {
    const float atanhi[4] = {u2f(0x3eed6338u), u2f(0x3f490fdau), u2f(0x3f7b985eu), u2f(0x3fc90fdau)};
    const float atanlo[4] = {u2f(0x31ac3769u), u2f(0x33222168u), u2f(0x33140fb4u), u2f(0x33a22168u)};
    const float aT0 = u2f(0x3eaaaaabu), aT1 = u2f(0xbe4ccccdu), aT2 = u2f(0x3e124925u), aT3 = u2f(0xbde38e38u),
                aT4 = u2f(0x3dba2e6eu), aT5 = u2f(0xbd9d8795u), aT6 = u2f(0x3d886b35u), aT7 = u2f(0xbd6ef16bu),
                aT8 = u2f(0x3d4bda59u), aT9 = u2f(0xbd15a221u), aT10 = u2f(0x3c8569d7u);
    const uint32_t hx = f2u(x);
    const uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x4c000000u) {  // |x| >= 2^25
        if (ix > 0x7f800000u) return x + x;  // NaN
        return ((int32_t)hx > 0) ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    int id;
    if (ix < 0x3ee00000u) {  // |x| < 0.4375
        if (ix < 0x31000000u) return x;  // |x| < 2^-29 (huge + x > one always holds)
        id = -1;
    } else {
        x = u2f(ix);  // fabsf
        if (ix < 0x3f980000u) {      // |x| < 1.1875
            if (ix < 0x3f300000u) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {                 // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000u) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {                 // 2.4375 <= |x| < 2^25
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return ((int32_t)hx < 0) ? -r : r;
}

// FDLIBM e_atan2f.c
JST_FN float libm_atan2f(float y, float x) {
    const float tiny = u2f(0x0da24260u), pi_o_4 = u2f(0x3f490fdbu), pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu),
                pi_lo = u2f(0xb3bbbd2eu);
    const uint32_t hx = f2u(x), hy = f2u(y);
    const uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
    if (ix > 0x7f800000u || iy > 0x7f800000u) return x + y;  // NaN
    if (hx == 0x3f800000u) return libm_atanf(y);             // x = 1.0
    const uint32_t m = (hy >> 31) | ((hx >> 30) & 2u);       // 2*sign(x) + sign(y)
    if (iy == 0) {                                           // y = +-0
        if (m < 2) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000u) {
        if (iy == 0x7f800000u) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000u) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = ((int32_t)iy - (int32_t)ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                   // |y/x| > 2^60
    else if ((int32_t)hx < 0 && k < -60) z = 0.0f;           // |y|/x < -2^60
    else z = libm_atanf(u2f(f2u(y / x) & 0x7fffffffu));      // atanf(fabsf(y / x))
    switch (m) {
        case 0: return z;
        case 1: return u2f(f2u(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

}  // namespace jst::dev
