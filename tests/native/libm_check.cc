// Host-side checker for cyberether_amd/csrc/kernels/libm_float.hh (the branch-free tanhf used by
// the Range epilogue on the GPU): counts bit mismatches against this host's libm tanhf.
#include <math.h>
#include <stdint.h>

#include "../../cyberether_amd/csrc/kernels/libm_float.hh"

// variant 0: select form (libm_tanhf); variant 1: branch-structured form (libm_tanhf_branchy);
// variant 2: main-path form (libm_tanhf_main) with the branch-structured form for what it reports as rare
extern "C" uint64_t jst_tanhf_mismatches(uint32_t start, uint32_t stride, uint64_t count,
                                         uint32_t* first_bad, int variant) {
    uint64_t bad = 0;
    uint32_t u = start;
    for (uint64_t i = 0; i < count; ++i, u += stride) {
        const float x = jst::dev::u2f(u);
        const float a = tanhf(x);
        float b;
        if (variant == 2) {
            bool rare;
            b = jst::dev::libm_tanhf_main(x, rare);
            if (rare) b = jst::dev::libm_tanhf_branchy(x);
        } else {
            b = variant ? jst::dev::libm_tanhf_branchy(x) : jst::dev::libm_tanhf(x);
        }
        if (isnan(a) && isnan(b)) continue;
        if (jst::dev::f2u(a) != jst::dev::f2u(b)) {
            if (bad == 0 && first_bad) *first_bad = u;
            ++bad;
        }
    }
    return bad;
}

extern "C" float jst_tanhf_select(float x) { return jst::dev::libm_tanhf(x); }
extern "C" float jst_tanhf_branchy(float x) { return jst::dev::libm_tanhf_branchy(x); }
extern "C" float jst_tanhf_main(float x) {
    bool rare;
    const float r = jst::dev::libm_tanhf_main(x, rare);
    return rare ? jst::dev::libm_tanhf_branchy(x) : r;
}

// sinf / cosf / atanf (one argument, which = 0 / 1 / 2) and atan2f against this host's libm
extern "C" uint64_t jst_trig_mismatches(uint32_t start, uint32_t stride, uint64_t count, uint32_t* first_bad, int which) {
    uint64_t bad = 0;
    uint32_t u = start;
    for (uint64_t i = 0; i < count; ++i, u += stride) {
        const float x = jst::dev::u2f(u);
        const float a = which == 0 ? sinf(x) : which == 1 ? cosf(x) : atanf(x);
        const float b = which == 0 ? jst::dev::libm_sinf(x) : which == 1 ? jst::dev::libm_cosf(x) : jst::dev::libm_atanf(x);
        if (isnan(a) && isnan(b)) continue;
        if (jst::dev::f2u(a) != jst::dev::f2u(b)) {
            if (bad == 0 && first_bad) *first_bad = u;
            ++bad;
        }
    }
    return bad;
}
extern "C" uint64_t jst_atan2f_mismatches(const float* ys, const float* xs, uint64_t count, uint64_t* first_bad) {
    uint64_t bad = 0;
    for (uint64_t i = 0; i < count; ++i) {
        const float a = atan2f(ys[i], xs[i]), b = jst::dev::libm_atan2f(ys[i], xs[i]);
        if (isnan(a) && isnan(b)) continue;
        if (jst::dev::f2u(a) != jst::dev::f2u(b)) {
            if (bad == 0 && first_bad) *first_bad = i;
            ++bad;
        }
    }
    return bad;
}
