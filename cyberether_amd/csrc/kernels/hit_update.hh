// hit_update.hh -- the Spectrogram's saturating hit update applied k times: std::min(val + 0.02f, 1.0f) once per hit
// (spectrogram/module_impl_native_cpu.cc:70-77), for w in [0, 1] and k <= 64.  No HIP dependency: the same text compiles
// for the host (tests/test_hit_update.py builds it with g++ and checks the fast form against the additions written out).
//
// Without the clamp the additions form a non-decreasing sequence s_n = fl(s_(n-1) + c), c = 0.02f; the clamped sequence
// equals it until it first reaches 1 and is 1.0f from there on, so the result is min(s_k, 1).
//
// apply_hits (round 3): the k dependent additions and one clamp.  A cell of the noise floor takes ~30 hits per cycle, and
// the wavefronts that own those rows walk 30-64 dependent additions per cycle while the rest of the workgroup waits at the
// next barrier: ~40 % of the cycle-batched kernel's life.
//
// apply_hits_binade (round 4): the same floats in at most a dozen steps.  c = 0xA3D70A * 2^-29.  While s stays inside one
// binade [2^E, 2^(E+1)) its bit pattern is an integer count of ulps (2^(E-23)), and fl(s + c) adds a FIXED number of
// ulps q_E -- c / ulp = 0xA3D70A / 2^(E+6), rounded to nearest: E = -1: 335544.31 -> 335544; E = -2: 671088.63 -> 671089;
// E = -3: 1342177.25 -> 1342177; E = -4: 2684354.5, a TIE, to even: an even pattern gains 2684354 and stays even (an odd
// one gains 2684355 once and is even from then on); E = -5, -6: exact.  So n additions inside a binade are ONE integer
// multiply-add on the pattern.  Every step of the loop below performs one REAL addition (which handles whatever the
// integer view does not: a start below 2^-6, the crossing into the next binade, the odd pattern in the tie binade) and then
// jumps by as many additions as keep the pattern inside the binade it reached.
#pragma once

#include "libm_float.hh"

namespace jst::dev {

JST_FN float apply_hits(float w, uint32_t k) {
    if (w + 0.02f * (float)k >= 1.001f) return 1.0f;
    uint32_t n = 0;
    for (; n + 4u <= k; n += 4u) {
        w += 0.02f;
        w += 0.02f;
        w += 0.02f;
        w += 0.02f;
    }
    for (; n < k; ++n) w += 0.02f;
    return w < 1.0f ? w : 1.0f;  // fminf(w, 1.0f) for the non-NaN values that reach this point
}

JST_FN float apply_hits_binade(float w, uint32_t k) {
    if (w + 0.02f * (float)k >= 1.001f) return 1.0f;  // far past 1: exactly 1.0f (as in apply_hits)
    while (k != 0u) {
        w += 0.02f;  // one real addition
        --k;
        if (!(w < 1.0f)) return 1.0f;  // reached 1: a fixed point of the clamped sequence
        const uint32_t b = f2u(w);     // 2^-6 <= 0.02 <= w < 1: biased exponent 121..126
        const uint32_t shift = (b >> 23) - 121u;                       // E + 6: 0..5
        const uint32_t q = (0xA3D70Au >> shift) + (shift == 4u ? 1u : 0u);   // ulps one addition gains in this binade
        if (shift == 2u && (b & 1u)) continue;                         // tie binade, odd pattern: the next addition is real
        const uint32_t room = ((b | 0x007fffffu) - b);                 // ulps up to the last pattern of the binade
        // n = floor(room / q), room < 2^23, q >= 335544: n <= 25; an approximate reciprocal (2^shift / 0xA3D70A, within
        // 1e-6 of 1 / q), then the estimate is corrected by at most one either way in exact integer arithmetic
        const float rq = u2f(0x33c80000u + (shift << 23));
        uint32_t n = (uint32_t)((float)room * rq);
        if (n * q > room) --n;
        if ((n + 1u) * q <= room) ++n;
        n = n < k ? n : k;
        w = u2f(b + n * q);
        k -= n;
    }
    return w;
}

}  // namespace jst::dev
