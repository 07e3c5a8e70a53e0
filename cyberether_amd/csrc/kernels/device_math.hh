// device_math.hh -- gfx950 device-side scalar math shared by every Jetstream HIP kernel.
//
// Every routine here computes, operation by operation, what the reference's CPU module
// computes (IEEE-754 binary32, round-to-nearest, no fused multiply-add: the whole library is
// built with -ffp-contract=off), so results are bit-identical to the reference CPU path, not
// merely close.  Citations are relative to the CyberEther 1.9.1 tree.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "libm_float.hh"

namespace jst::dev {

using f2 = float2;

__device__ __forceinline__ f2 mk(float r, float i) { return make_float2(r, i); }
__device__ __forceinline__ f2 cadd(f2 a, f2 b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ f2 csub(f2 a, f2 b) { return mk(a.x - b.x, a.y - b.y); }

// std::complex<float>::operator* for finite operands (multiply/module_impl_native_cpu.cc:94-100):
// four individually rounded products, re = ac - bd, im = ad + bc.  The C99 Annex G recovery
// branch (only taken when both parts come out NaN) is in cmul_full().
__device__ __forceinline__ f2 cmul(f2 a, f2 b) {
    const float ac = a.x * b.x, bd = a.y * b.y, ad = a.x * b.y, bc = a.y * b.x;
    return mk(ac - bd, ad + bc);
}

__device__ __forceinline__ f2 cmul_recover_body(f2 p, f2 q) {
    float a = p.x, b = p.y, c = q.x, d = q.y;
    const float ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    float x = ac - bd, y = ad + bc;
    if (__builtin_isnan(x) && __builtin_isnan(y)) {
        bool recalc = false;
        if (__builtin_isinf(a) || __builtin_isinf(b)) {
            a = __builtin_copysignf(__builtin_isinf(a) ? 1.0f : 0.0f, a);
            b = __builtin_copysignf(__builtin_isinf(b) ? 1.0f : 0.0f, b);
            if (__builtin_isnan(c)) c = __builtin_copysignf(0.0f, c);
            if (__builtin_isnan(d)) d = __builtin_copysignf(0.0f, d);
            recalc = true;
        }
        if (__builtin_isinf(c) || __builtin_isinf(d)) {
            c = __builtin_copysignf(__builtin_isinf(c) ? 1.0f : 0.0f, c);
            d = __builtin_copysignf(__builtin_isinf(d) ? 1.0f : 0.0f, d);
            if (__builtin_isnan(a)) a = __builtin_copysignf(0.0f, a);
            if (__builtin_isnan(b)) b = __builtin_copysignf(0.0f, b);
            recalc = true;
        }
        if (!recalc && (__builtin_isinf(ac) || __builtin_isinf(bd) || __builtin_isinf(ad) ||
                        __builtin_isinf(bc))) {
            if (__builtin_isnan(a)) a = __builtin_copysignf(0.0f, a);
            if (__builtin_isnan(b)) b = __builtin_copysignf(0.0f, b);
            if (__builtin_isnan(c)) c = __builtin_copysignf(0.0f, c);
            if (__builtin_isnan(d)) d = __builtin_copysignf(0.0f, d);
            recalc = true;
        }
        if (recalc) {
            x = __builtin_inff() * (a * c - b * d);
            y = __builtin_inff() * (a * d + b * c);
        }
    }
    return mk(x, y);
}

__device__ __attribute__((noinline)) f2 cmul_recover(f2 p, f2 q) { return cmul_recover_body(p, q); }

// Full std::complex<float> product: the plain formula, and -- only when a part came out NaN,
// which finite data never does -- the Annex G recovery path, kept out of line (cold).
#ifndef JST_COLD_INLINE  // A/B switch: 1 = the cold paths of cmul_full / the exact epilogue are inlined (no calls in the kernel)
#define JST_COLD_INLINE 0
#endif
__device__ __forceinline__ f2 cmul_recover_inl(f2 p, f2 q) { return cmul_recover_body(p, q); }
__device__ __forceinline__ f2 cmul_full(f2 p, f2 q) {
    const f2 r = cmul(p, q);
#if JST_COLD_INLINE
    if (__builtin_expect(__builtin_isunordered(r.x, r.y), 0)) return cmul_recover_inl(p, q);
#else
    if (__builtin_expect(__builtin_isunordered(r.x, r.y), 0)) return cmul_recover(p, q);
#endif
    return r;
}

// ---- pocketfft butterfly helpers (fft/pocketfft.hh:266-272, :290-291, :1124-1139) -----------
template <bool FWD>
__device__ __forceinline__ f2 special_mul(f2 v, f2 w) {
    if constexpr (FWD) return mk(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);
    else return mk(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
}
template <bool FWD>
__device__ __forceinline__ f2 rotx90(f2 a) {
    if constexpr (FWD) return mk(a.y, -a.x);
    else return mk(-a.y, a.x);
}
template <bool FWD>
__device__ __forceinline__ f2 rotx45(f2 a) {
    constexpr float h = 0.707106781186547524400844362104849f;
    if constexpr (FWD) return mk(h * (a.x + a.y), h * (a.y - a.x));
    else return mk(h * (a.x - a.y), h * (a.y + a.x));
}
template <bool FWD>
__device__ __forceinline__ f2 rotx135(f2 a) {
    constexpr float h = 0.707106781186547524400844362104849f;
    if constexpr (FWD) return mk(h * (a.y - a.x), h * (-a.x - a.y));
    else return mk(h * (-a.x - a.y), h * (a.x - a.y));
}

// Radix-2/4/8 butterflies WITHOUT the output twiddles: x[b] = CC(i,b,k) in, x[c] = the value
// pocketfft multiplies by WA(c-1,i) (or stores directly when i == 0) out.
// pass2 :843-872, pass4 :929-975, pass8 :1141-1223.
template <bool FWD>
__device__ __forceinline__ void butterfly2(f2 (&x)[2]) {
    const f2 a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}
template <bool FWD>
__device__ __forceinline__ void butterfly4(f2 (&x)[4]) {
    const f2 t2 = cadd(x[0], x[2]), t1 = csub(x[0], x[2]);
    const f2 t3 = cadd(x[1], x[3]);
    const f2 t4 = rotx90<FWD>(csub(x[1], x[3]));
    x[0] = cadd(t2, t3);
    x[2] = csub(t2, t3);
    x[1] = cadd(t1, t4);
    x[3] = csub(t1, t4);
}
template <bool FWD>
__device__ __forceinline__ void butterfly8(f2 (&x)[8]) {
    f2 a1 = cadd(x[1], x[5]), a5 = csub(x[1], x[5]);
    f2 a3 = cadd(x[3], x[7]), a7 = csub(x[3], x[7]);
    a7 = rotx90<FWD>(a7);
    f2 t = a1;
    a1 = cadd(a1, a3);
    a3 = rotx90<FWD>(csub(t, a3));
    t = a5;
    a5 = cadd(a5, a7);
    a7 = csub(t, a7);
    a5 = rotx45<FWD>(a5);
    a7 = rotx135<FWD>(a7);
    f2 a0 = cadd(x[0], x[4]), a4 = csub(x[0], x[4]);
    f2 a2 = cadd(x[2], x[6]), a6 = csub(x[2], x[6]);
    t = a0;
    a0 = cadd(a0, a2);
    a2 = csub(t, a2);
    x[0] = cadd(a0, a1);
    x[4] = csub(a0, a1);
    x[2] = cadd(a2, a3);
    x[6] = csub(a2, a3);
    a6 = rotx90<FWD>(a6);
    t = a4;
    a4 = cadd(a4, a6);
    a6 = csub(t, a6);
    x[1] = cadd(a4, a5);
    x[5] = csub(a4, a5);
    x[3] = cadd(a6, a7);
    x[7] = csub(a6, a7);
}
template <int IP, bool FWD>
__device__ __forceinline__ void butterfly(f2 (&x)[IP]) {
    if constexpr (IP == 8) butterfly8<FWD>(x);
    else if constexpr (IP == 4) butterfly4<FWD>(x);
    else butterfly2<FWD>(x);
}

// ---- Amplitude -----------------------------------------------------------------------------
// Backend::ApproxLog10 (include/jetstream/backend/devices/cpu/helpers.hh:59-74): frexpf + cubic,
// separate multiplies and adds.  v_frexp_mant_f32 / v_frexp_exp_i32_f32 give exactly C frexpf
// (mantissa in [0.5,1), subnormals handled, inf/NaN -> exponent 0 like glibc).
__device__ __forceinline__ float approx_log10(float x) {
    int e;
    const float f = __builtin_frexpf(__builtin_fabsf(x), &e);
    float y = 1.23149591368684f;
    y *= f;
    y += -4.11852516267426f;
    y *= f;
    y += 6.02197014179219f;
    y *= f;
    y += -3.13396450166353f;
    y += (float)e;
    return y * 0.3010299956639812f;
}

// amplitude/module_impl_native_cpu.cc:73-86 (CF32) and :88-99 (F32).  sqrtf is the correctly
// rounded device sqrt (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), == libm sqrtf.
__device__ __forceinline__ float amplitude_cf32(f2 v, float coeff) {
    const float mag = __builtin_sqrtf((v.x * v.x) + (v.y * v.y));
    return (mag == 0.0f) ? -__builtin_inff() : 20.0f * approx_log10(mag) + coeff;
}
__device__ __forceinline__ float amplitude_f32(float v, float coeff) {
    const float mag = __builtin_fabsf(v);
    return (mag == 0.0f) ? -__builtin_inff() : 20.0f * approx_log10(mag) + coeff;
}

// ---- Range ---------------------------------------------------------------------------------
// tanhf: see libm_float.hh (select-form restatement of the host libm's FDLIBM tanhf/expm1f).
// range/module_impl_native_cpu.cc:67-82.
__device__ __forceinline__ float range_f32_general(float v, float scale, float offset) {
    if (scale == 0.0f) return 0.5f;
    const float normalized = v * scale + offset;
#ifdef JST_TANH_SELECT_FORM
    return 0.5f + 0.5f * libm_tanhf(4.0f * (normalized - 0.5f));
#else
    return 0.5f + 0.5f * libm_tanhf_branchy(4.0f * (normalized - 0.5f));
#endif
}
__device__ __attribute__((noinline)) float range_f32_cold(float v, float scale, float offset) {
    return range_f32_general(v, scale, offset);
}
// The Range module's element function: the main-path tanhf (libm_float.hh) with one bail-out to the general
// ladder for arguments outside [2^-26, 7.5).
__device__ __forceinline__ float range_f32(float v, float scale, float offset) {
    if (scale == 0.0f) return 0.5f;
    const float normalized = v * scale + offset;
    bool rare;
    float r = 0.5f + 0.5f * libm_tanhf_main(4.0f * (normalized - 0.5f), rare);
    if (__builtin_expect(rare, 0)) r = range_f32_cold(v, scale, offset);
    return r;
}

// ---- main-path form of the exact Amplitude -> Range epilogue ----------------------------------
// The general forms above are ladders of classes (zero / subnormal / inf magnitude in the compiler's
// sqrt expansion and in frexpf, the tanhf classes); in a fused FFT epilogue every element walks all of
// them and each costs compare + select pairs at half rate.  The main path below is ONE straight-line
// sequence that is bit-identical to the general form on the operands a spectrum actually produces --
// power in [2^-100, 2^100], tanh argument in [2^-26, 7.5) -- and a single wave-uniform bail-out to the
// general form (one out-of-line copy, exec-masked) for wavefronts holding anything else (exact zeros,
// inf/NaN, values 1.4 display ranges outside the display range).

// Correctly rounded sqrt for 2^-100 <= p <= 2^100 without the input scaling and class handling of the
// compiler's general expansion: Markstein's coupled iteration from v_rsq_f32 (1 ulp), g -> sqrt(p),
// h -> 1/(2 sqrt(p)), one exact residual and one fused correction.  Swept against the compiler's
// correctly rounded expansion for EVERY float of the interval on the device (tests/test_gpu_exact_sweep.py).
__device__ __forceinline__ float sqrt_markstein(float p) {
    const float y = __builtin_amdgcn_rsqf(p);
    float g = p * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, p);
    return __builtin_fmaf(d, h, g);
}
// shorter candidates (see tools/ubench/exact_sweep.hip): hardware sqrt + one fused correction
__device__ __forceinline__ float sqrt_v2(float p) {
    const float s = __builtin_amdgcn_sqrtf(p), y = __builtin_amdgcn_rsqf(p);
    const float d = __builtin_fmaf(-s, s, p);
    return __builtin_fmaf(d, 0.5f * y, s);
}
__device__ __forceinline__ float sqrt_v4(float p) {
    const float y = __builtin_amdgcn_rsqf(p);
    const float g = p * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, p);
    return __builtin_fmaf(d, h, g);
}
// All three are correctly rounded on every float of [2^-100, 2^100] (0 mismatches of 2^31-ish arguments each,
// MI355X, round 2; the raw v_sqrt_f32 differs on 253 545 200 of them): the shortest one ships.
#ifndef JST_SQRT_MAIN
#define JST_SQRT_MAIN(p) sqrt_v4(p)
#endif

constexpr uint32_t kPowerLo = 0x0d800000u, kPowerHi = 0x71800000u;  // 2^-100, 2^100

// Amplitude from the power re^2 + im^2 in [2^-100, 2^100]: the magnitude is a normal float, so frexpf is
// two bit operations and the zero test is void.  Same operations as approx_log10 / amplitude_cf32.
__device__ __forceinline__ float amplitude_from_power_main(float p, float coeff) {
    const uint32_t mb = f2u(JST_SQRT_MAIN(p));
    const float f = u2f((mb & 0x007fffffu) | 0x3f000000u);
    const float e = (float)((int32_t)(mb >> 23) - 126);
    float y = 1.23149591368684f;
    y *= f;
    y += -4.11852516267426f;
    y *= f;
    y += 6.02197014179219f;
    y *= f;
    y += -3.13396450166353f;
    y += e;
    return 20.0f * (y * 0.3010299956639812f) + coeff;
}
__device__ __forceinline__ float amplitude_from_power(float p, float coeff) {  // general form
    const float mag = __builtin_sqrtf(p);
    return (mag == 0.0f) ? -__builtin_inff() : 20.0f * approx_log10(mag) + coeff;
}
__device__ __attribute__((noinline)) float amplitude_from_power_cold(float p, float coeff) {
    return amplitude_from_power(p, coeff);
}
__device__ __attribute__((noinline)) float amplitude_range_from_power_cold(float p, float coeff, float scale,
                                                                           float offset) {
    return range_f32_general(amplitude_from_power(p, coeff), scale, offset);
}
__device__ __forceinline__ float amplitude_exact(f2 v, float coeff) {
    const float p = (v.x * v.x) + (v.y * v.y);
    float r = amplitude_from_power_main(p, coeff);
    if (__builtin_expect((f2u(p) - kPowerLo) > (kPowerHi - kPowerLo), 0)) r = amplitude_from_power_cold(p, coeff);
    return r;
}
__device__ __forceinline__ float amplitude_range_from_power(float p, float coeff, float scale, float offset) {
    if (scale == 0.0f) return 0.5f;  // wave-uniform
    const float normalized = amplitude_from_power_main(p, coeff) * scale + offset;
    bool rare;
    const float t = libm_tanhf_main(4.0f * (normalized - 0.5f), rare);
    float r = 0.5f + 0.5f * t;
    rare |= (f2u(p) - kPowerLo) > (kPowerHi - kPowerLo);
#if JST_COLD_INLINE
    if (__builtin_expect(rare, 0)) r = range_f32_general(amplitude_from_power(p, coeff), scale, offset);
#else
    if (__builtin_expect(rare, 0)) r = amplitude_range_from_power_cold(p, coeff, scale, offset);
#endif
    return r;
}
__device__ __forceinline__ float amplitude_range_exact(f2 v, float coeff, float scale, float offset) {
    return amplitude_range_from_power((v.x * v.x) + (v.y * v.y), coeff, scale, offset);
}

// ---- "fast" provider variants ----------------------------------------------------------------
// Registered as provider "fast" (the registry's 4th key exists to select alternative
// implementations, src/registry.cc:608-613).  They use the gfx950 transcendental unit directly
// (v_sqrt_f32, v_exp_f32, v_rcp_f32: ~1 ulp each) instead of restating libm bit for bit:
// amplitude stays within 2e-6 dB and range within 3e-7 absolute of the reference CPU path
// (tests/test_gpu_fast_provider.py), well inside BASELINE.json's 1e-5 tolerance for float
// spectra, for ~1/8 of the instructions.
__device__ __forceinline__ float amplitude_from_power_fast(float p, float coeff) {
    const float mag = __builtin_amdgcn_sqrtf(p);
    return (mag == 0.0f) ? -__builtin_inff() : 20.0f * approx_log10(mag) + coeff;
}
__device__ __forceinline__ float amplitude_cf32_fast(f2 v, float coeff) {
    return amplitude_from_power_fast((v.x * v.x) + (v.y * v.y), coeff);
}
__device__ __forceinline__ float tanhf_fast(float a) {
    const float ax = __builtin_fabsf(a);
    const float e = __builtin_amdgcn_exp2f(ax * -2.885390081777927f);  // exp(-2|a|)
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return __builtin_copysignf(t, a);  // NaN in -> NaN out; +-inf -> +-1
}
__device__ __forceinline__ float range_f32_fast(float v, float scale, float offset) {
    if (scale == 0.0f) return 0.5f;
    const float normalized = v * scale + offset;
    return 0.5f + 0.5f * tanhf_fast(4.0f * (normalized - 0.5f));
}

// Provider "fast" with exact bins.  A Spectrogram consumer quantises the range value r to
// bin = trunc(r * height) (hit <=> 1 <= r * height < height); the fast value may differ from the
// libm-exact one by a few 1e-7, which flips a bin only when r * height lies that close to an integer.
// So: whenever the fast r * height is within `guard` of an integer k >= 1, the element is recomputed
// with the exact amplitude + range arithmetic (and stored exactly); every other element keeps the fast
// value, whose bin equals the exact one.  guard = height * 7.5e-7: the two F32 roundings of the
// products (<= height * 1.2e-7) plus twice the largest fast-vs-exact difference over dense sweeps
// (< 3e-7: one-ulp v_exp/v_rcp/v_sqrt against the libm restatement; tests/test_gpu_fast_provider.py
// asserts the bound).  About 4e-4 of the elements take the exact path at height 256.
struct BinGuard {
    float h0 = 0.0f, h1 = 0.0f;  // consumer heights (0 = none)
    float t0 = 0.0f, t1 = 0.0f;  // their guard widths h * 7.5e-7 once a kernel has pinned them in VGPRs (0 = form them on use)
};
// f = r * h, already formed.  v_fract_f32 (for finite f >= 0 it IS f - floor(f): that difference is exact and below 1, so the
// instruction's clamp never acts; written as `f - floorf(f)` hipcc emits v_floor + v_sub because it cannot know the sign).
__device__ __forceinline__ bool near_bin_edge_of(float f, float h) {
    const float fr = __builtin_amdgcn_fractf(f);
    return f >= 0.5f && __builtin_fabsf(fr - 0.5f) > 0.5f - h * 7.5e-7f;
}
__device__ __forceinline__ bool near_bin_edge(float r, float h) { return near_bin_edge_of(r * h, h); }
// One out-of-line copy of the exact arithmetic serves the rare guarded elements (amplitude_range_from_power_cold,
// above).  Everything from the power p = re^2 + im^2 on is a function of ONE float, identical in both providers up to
// p, so the guarantee "guarded fast value and exact value fall into the same Spectrogram bin" is checked on
// every float p, per height, by the exhaustive device sweep (exact_sweep.hip, tests/test_gpu_exact_sweep.py).
//
// Round 3: the fast value itself is the LEAN form.  Amplitude -> Range is affine in (P(F) + E) up to the tanh --
//   arg = 4 * ((20 * L * (P(F) + E) + coeff) * scale + offset - 0.5),  0.5 + 0.5 * tanh(arg) = 1 / (1 + 2^z),
//   z = -2 log2(e) * arg  --
// so every constant folds into the four coefficients of ApproxLog10's cubic and one exponent weight (computed once per
// launch in double, FastRangePoly), the chain runs as four fused multiply-adds and the logistic form needs one v_exp_f32
// and one v_rcp_f32: 17 VALU instructions from the power on, against ~50 for the round-2 fast form and ~105 for the
// exact one.  Its deviation from the exact provider stays below 4e-7 (every float p, WHICH = 6 of the sweep: 2.4e-7 .. 3.6e-7).
#ifndef JST_FAST_COLD_INLINE
#define JST_FAST_COLD_INLINE 1
#endif
struct FastRangePoly {
    float k3 = 0.0f, k2 = 0.0f, k1 = 0.0f, k0 = 0.0f, ke = 0.0f;
};
__host__ __device__ inline FastRangePoly make_fast_range_poly(float coeff, float scale, float offset) {
    const double L = 0.3010299956639812, M = -2.885390081777926814719849362003784;  // log10(2), -2 log2(e)
    const double A = 80.0 * L * (double)scale;                                      // d arg / d (P + E)
    const double B = 4.0 * ((double)coeff * (double)scale + (double)offset - 0.5);
    FastRangePoly q;
    q.k3 = (float)(M * A * 1.23149591368684);
    q.k2 = (float)(M * A * -4.11852516267426);
    q.k1 = (float)(M * A * 6.02197014179219);
    q.k0 = (float)(M * (A * -3.13396450166353 + B));
    q.ke = (float)(M * A);
    return q;
}
// Domain: power in [2^-100, 2^100] (the caller bails out to the exact arithmetic for anything else).
__device__ __forceinline__ float amplitude_range_lean(float p, const FastRangePoly& q) {
    const float mag = __builtin_amdgcn_sqrtf(p);
    const float f = __builtin_amdgcn_frexp_mantf(mag);
    const float e = (float)__builtin_amdgcn_frexp_expf(mag);
    const float tail = __builtin_fmaf(q.ke, e, q.k0);
    const float z = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(q.k3, f, q.k2), f, q.k1), f, tail);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));  // 2^z = inf -> 0, 2^z = 0 -> 1
}
// Round 4: the same lean value by instruction CLASS.  Measured on MI355X (tools/ubench/valu_forms.hip, ns per wave64
// instruction and SIMD at >= 2 wavefronts per SIMD): add / mul / fma / and-or / shifts 1.1; v_cmp_*, v_cndmask, v_fract,
// v_frexp_*, v_cvt_*, v_min / v_med3, DPP moves 2.0; v_sqrt / v_exp / v_rcp / v_log 3.5; a v_cmp + v_cndmask pair through
// VCC 7.  The round-3 epilogue spent ~50 ns per output, more than half of it on the half-rate classes around three
// unavoidable transcendentals; here
//   * the mantissa is ONE v_and_or_b32 on the magnitude's bits (instead of v_frexp_mant), the exponent two integer
//     operations and the conversion -- the same numbers frexpf gives for a normal float;
//   * the domain test is one v_cmp_class on the magnitude (anything but a positive normal float goes to the exact
//     arithmetic) instead of an integer add + compare on the power;
//   * "within thr of a bin edge" is |f - rint(f)| < thr with rint(f) = (f + 1.5 * 2^23) - 1.5 * 2^23 (two full-rate adds,
//     exact for 0 <= f < 2^22) and ONE compare with an |.| modifier, instead of v_fract + add + two compares; it also
//     flags f < thr (values below thr / height: the exact path answers them, nothing else changes);
//   * k2 and k0 live in VGPRs (pinned once per thread: StoreAmplitudeRangeT::pin_constants) -- a VOP3 fma takes one
//     scalar operand, and hipcc re-materialised the second one with a v_mov per fma.
// ~36 ns per output.  The VALUE is the round-3 lean value bit for bit on its old domain (same operations on the same
// numbers); what changed is how mantissa / exponent / guard flags are formed and the wider domain, and the bin guarantee
// is re-proved by the exhaustive sweep (exact_sweep.hip WHICH 4 calls this very function).
#ifndef JST_EPI_V2  // A/B switch: 0 = the round-3 form
#define JST_EPI_V2 1
#endif
__device__ __forceinline__ float amplitude_range_lean2(float p, const FastRangePoly& q, bool& special) {
    const float mag = __builtin_amdgcn_sqrtf(p);
    special = __builtin_amdgcn_classf(mag, 0x2ff);  // everything but +normal: zero, denormal, inf, NaN (negative cannot occur)
    const uint32_t mb = f2u(mag);
    const float f = u2f((mb & 0x007fffffu) | 0x3f000000u);  // frexpf mantissa of a normal float, in [0.5, 1)
    // frexpf's exponent as a SMALL number (one integer subtract in front of the conversion): with the biased field and
    // the bias folded into k0 the rounding errors of ke and of the folded constant are multiplied by ~127 instead of
    // |e| <= 50 -- 33 powers per 2^32 then changed their Spectrogram bin at height 256 (exhaustive sweep, first build)
    const float e = (float)((int32_t)(mb >> 23) - 126);
    const float tail = __builtin_fmaf(q.ke, e, q.k0);
    const float z = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(q.k3, f, q.k2), f, q.k1), f, tail);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));  // 2^z = inf -> 0, 2^z = 0 -> 1
}
// |f - rint(f)| < thr for 0 <= f < 2^22 (f = value * height, height <= 2^16 in every caller)
__device__ __forceinline__ bool near_integer(float f, float thr) {
    const float n = (f + 12582912.0f) - 12582912.0f;
    return __builtin_fabsf(f - n) < thr;
}

// f0 receives value * g.h0 (the product the first consumer's quantiser forms: the side-output epilogue of fft_lds.hh takes
// its row index from it instead of multiplying again); undefined when g.h0 == 0.
__device__ __forceinline__ float amplitude_range_fast_guarded_from_power(float p, float coeff, float scale,
                                                                         float offset, const BinGuard& g,
                                                                         const FastRangePoly& q, float& f0) {
    if (scale == 0.0f) {  // wave-uniform
        f0 = 0.5f * g.h0;
        return 0.5f;
    }
#if JST_EPI_V2
    bool cold;
    float r = amplitude_range_lean2(p, q, cold);
    f0 = r * g.h0;
    if (g.h0 > 0.0f) {  // wave-uniform
        cold |= near_integer(f0, g.t0 > 0.0f ? g.t0 : g.h0 * 7.5e-7f);
        if (g.h1 > 0.0f) cold |= near_integer(r * g.h1, g.t1 > 0.0f ? g.t1 : g.h1 * 7.5e-7f);
    }
#else
    float r = amplitude_range_lean(p, q);
    bool cold = (f2u(p) - kPowerLo) > (kPowerHi - kPowerLo);  // zero, subnormal, huge, inf, NaN: the exact ladder
    f0 = r * g.h0;
    if (g.h0 > 0.0f) {  // wave-uniform
        cold |= near_bin_edge_of(f0, g.h0);
        if (g.h1 > 0.0f) cold |= near_bin_edge(r, g.h1);
    }
#endif
    // A guard hit (~2.5 % of the wavefront-elements at height 256) goes through the INLINED exact main path, not through a
    // call: with the out-of-line copy every epilogue carried a call site whose ABI (caller-saved v0-v31, live values
    // parked in callee-saved registers around it) cost the whole kernel 1.5 us per launch although the call is rare --
    // 17.6 -> 16.1 us, same box, same checksum (profiles/r03_experiments/g_fast_guard_inline.log).  The exact kernel's
    // own bail-out (taken never on real spectra) showed no such effect (h_no_calls.log) and stays out of line.
#if JST_FAST_COLD_INLINE  // A/B switch
    if (__builtin_expect(cold, 0)) {
        r = amplitude_range_from_power(p, coeff, scale, offset);
        f0 = r * g.h0;
#if JST_EPI_V2  // f0 == h0 (value 1.0) is no hit: only an element that came through here can sit ON an integer, so the
                // test lives here and the side output's index is the plain conversion of f0 (fft_lds.hh)
        if (!(f0 < g.h0)) f0 = 0.0f;
#endif
    }
#else
    if (__builtin_expect(cold, 0)) {
        r = amplitude_range_from_power_cold(p, coeff, scale, offset);
        f0 = r * g.h0;
#if JST_EPI_V2
        if (!(f0 < g.h0)) f0 = 0.0f;
#endif
    }
#endif
    return r;
}
__device__ __forceinline__ float amplitude_range_fast_guarded_from_power(float p, float coeff, float scale,
                                                                         float offset, const BinGuard& g,
                                                                         const FastRangePoly& q) {
    float f0;
    return amplitude_range_fast_guarded_from_power(p, coeff, scale, offset, g, q, f0);
}
__device__ __forceinline__ float amplitude_range_fast_guarded_from_power(float p, float coeff, float scale,
                                                                         float offset, const BinGuard& g) {
    return amplitude_range_fast_guarded_from_power(p, coeff, scale, offset, g, make_fast_range_poly(coeff, scale, offset));
}
__device__ __forceinline__ float amplitude_range_fast_guarded(f2 v, float coeff, float scale, float offset,
                                                              const BinGuard& g) {
    return amplitude_range_fast_guarded_from_power((v.x * v.x) + (v.y * v.y), coeff, scale, offset, g);
}
__device__ __forceinline__ float amplitude_range_fast_guarded(f2 v, float coeff, float scale, float offset,
                                                              const BinGuard& g, const FastRangePoly& q) {
    return amplitude_range_fast_guarded_from_power((v.x * v.x) + (v.y * v.y), coeff, scale, offset, g, q);
}
// the round-2 fast form (v_sqrt + restated cubic + tanh through v_exp / v_rcp), kept for the A/B in the sweep
__device__ __forceinline__ float amplitude_range_fast_r02_from_power(float p, float coeff, float scale, float offset) {
    return range_f32_fast(amplitude_from_power_fast(p, coeff), scale, offset);
}

}  // namespace jst::dev
