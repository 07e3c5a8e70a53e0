// hits_check.cc -- host build of kernels/hit_update.hh for tests/test_hit_update.py: the binade form of the Spectrogram's
// hit update against the additions written out, on a sweep of starting values (every `step`-th float pattern in
// [first, last]) x every count 0..64.
#include <cstdint>

#include "../../cyberether_amd/csrc/kernels/hit_update.hh"

extern "C" {

uint64_t jst_hits_mismatches(uint32_t first, uint32_t last, uint32_t step, uint32_t* first_bad, uint32_t* first_bad_k) {
    uint64_t bad = 0;
    for (uint64_t bits = first; bits <= last; bits += step) {
        const float w = jst::dev::u2f((uint32_t)bits);
        for (uint32_t k = 0; k <= 64; ++k) {
            const float a = jst::dev::apply_hits(w, k), b = jst::dev::apply_hits_binade(w, k);
            if (jst::dev::f2u(a) != jst::dev::f2u(b)) {
                if (bad++ == 0) {
                    *first_bad = (uint32_t)bits;
                    *first_bad_k = k;
                }
            }
        }
    }
    return bad;
}

// the reference's own loop, one clamp per hit (spectrogram/module_impl_native_cpu.cc:70-77), for the cross-check
float jst_hits_reference(float w, uint32_t k) {
    for (uint32_t i = 0; i < k; ++i) {
        w += 0.02f;
        if (w > 1.0f) w = 1.0f;
    }
    return w;
}
float jst_hits_binade(float w, uint32_t k) { return jst::dev::apply_hits_binade(w, k); }

}  // extern "C"
