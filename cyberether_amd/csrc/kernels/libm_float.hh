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

#if defined(__HIPCC__) || defined(__CUDACC__)
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


}  // namespace jst::dev
