// exact_sweep.hip -- exhaustive device sweeps behind the main-path exact epilogue (device_math.hh,
// libm_float.hh) and behind the fast provider's Spectrogram bin guard, reachable through the C ABI
// (jst_probe_exact_sweep) so that tests/test_gpu_exact_sweep.py checks the code that ships.
//
// Everything from the power p = re^2 + im^2 on is a function of ONE float, so a candidate instruction
// sequence (a sqrt without the compiler's range handling, a divide without the last correction steps, a
// merged class ladder) is judged on EVERY float of its domain against the general form, which is itself
// swept against glibc on the host (tests/test_libm_float.py) and pinned by the oracle (tests/test_gpu_*).
#include "device_math.hh"
#include "kernels.hh"

namespace jst::kernels {

using namespace jst::dev;

namespace {

struct SweepResult {
    unsigned long long bad;
    unsigned long long visited;  // arguments the main path (not its bail-out) answered
    unsigned int first;
    unsigned int pad;
};

__device__ __forceinline__ bool same_bits(float a, float b) {
    return f2u(a) == f2u(b) || (__builtin_isnan(a) && __builtin_isnan(b));
}
__device__ __forceinline__ void note(SweepResult* r, uint32_t bits) {
    if (atomicAdd(&r->bad, 1ull) == 0) r->first = bits;
}
// Spectrogram hit rule (spectrogram/module_impl_native_cpu.cc:61-87; oracle/jst_oracle.c): hit <=> 1 <= r*H < H,
// bin = trunc(r*H); -1 = no hit.
__device__ __forceinline__ int bin_of(float r, float h) {
    const float f = r * h;
    return (f >= 1.0f && f < h) ? (int)f : -1;
}

template <int WHICH>
__global__ __launch_bounds__(256) void sweep_kernel(float coeff, float scale, float offset, float height,
                                                    SweepResult* res) {
    unsigned long long main_hits = 0;
    for (uint64_t u = blockIdx.x * 256ull + threadIdx.x; u < (1ull << 32); u += (uint64_t)gridDim.x * 256ull) {
        const uint32_t bits = (uint32_t)u;
        const float a = u2f(bits);
        if constexpr (WHICH == 0) {  // sqrt of the main path vs the compiler's correctly rounded expansion
            if ((bits - kPowerLo) > (kPowerHi - kPowerLo)) continue;
            ++main_hits;
            if (!same_bits(__builtin_sqrtf(a), JST_SQRT_MAIN(a))) note(res, bits);
        } else if constexpr (WHICH == 1) {  // tanhf main path vs the class-ladder form
            bool rare;
            const float got = libm_tanhf_main(a, rare);
            if (rare) continue;
            ++main_hits;
            if (!same_bits(libm_tanhf_branchy(a), got)) note(res, bits);
        } else if constexpr (WHICH == 2) {  // amplitude -> range from the power, main path + bail-out vs general
            const float ref = range_f32_general(amplitude_from_power(a, coeff), scale, offset);
            if (!same_bits(ref, amplitude_range_from_power(a, coeff, scale, offset))) note(res, bits);
        } else if constexpr (WHICH == 3) {  // amplitude alone
            const float ref = amplitude_from_power(a, coeff);
            float got = amplitude_from_power_main(a, coeff);
            if ((bits - kPowerLo) > (kPowerHi - kPowerLo)) got = ref;
            else ++main_hits;
            if (!same_bits(ref, got)) note(res, bits);
        } else if constexpr (WHICH == 4) {  // fast provider WITH the bin guard: same Spectrogram bin as exact
            const float ref = range_f32_general(amplitude_from_power(a, coeff), scale, offset);
            const float got = amplitude_range_fast_guarded_from_power(a, coeff, scale, offset, BinGuard{height, 0.0f});
            if (f2u(got) != f2u(ref)) ++main_hits;  // elements that kept the fast value
            if (bin_of(ref, height) != bin_of(got, height)) note(res, bits);
        } else if constexpr (WHICH == 5) {  // fast provider WITHOUT the guard (shows the sweep can tell: bins do move)
            const float ref = range_f32_general(amplitude_from_power(a, coeff), scale, offset);
#if JST_EPI_V2
            bool special;
            float got = amplitude_range_lean2(a, make_fast_range_poly(coeff, scale, offset), special);
            if (special) got = ref;
#else
            float got = amplitude_range_lean(a, make_fast_range_poly(coeff, scale, offset));
            if ((bits - kPowerLo) > (kPowerHi - kPowerLo)) got = ref;
#endif
            if (bin_of(ref, height) != bin_of(got, height)) note(res, bits);
        } else {  // WHICH == 6: largest |lean fast value - exact value| over the lean form's domain (bits of it in `first`)
            const float ref = range_f32_general(amplitude_from_power(a, coeff), scale, offset);
#if JST_EPI_V2  // domain: every power whose magnitude is a positive normal float
            bool special;
            const float got = amplitude_range_lean2(a, make_fast_range_poly(coeff, scale, offset), special);
            if (special) continue;
            ++main_hits;
#else
            if ((bits - kPowerLo) > (kPowerHi - kPowerLo)) continue;
            ++main_hits;
            const float got = amplitude_range_lean(a, make_fast_range_poly(coeff, scale, offset));
#endif
            const float d = __builtin_fabsf(got - ref);
            if (!(d <= 1.0f)) note(res, bits);              // NaN or nonsense: counted as bad
            else atomicMax(&res->pad, f2u(d));              // non-negative floats order like their bits
        }
    }
    if (main_hits) atomicAdd(&res->visited, main_hits);
}

}  // namespace

hipError_t launch_exact_sweep(int which, float coeff, float scale, float offset, float height,
                              uint64_t* mismatches, uint64_t* visited, uint32_t* first_bad) {
    SweepResult* d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(SweepResult));
    if (e != hipSuccess) return e;
    (void)hipMemset(d, 0, sizeof(SweepResult));
    (void)hipGetLastError();
    const dim3 grid(8192), block(256);
    switch (which) {
        case 0: hipLaunchKernelGGL(sweep_kernel<0>, grid, block, 0, nullptr, coeff, scale, offset, height, d); break;
        case 1: hipLaunchKernelGGL(sweep_kernel<1>, grid, block, 0, nullptr, coeff, scale, offset, height, d); break;
        case 2: hipLaunchKernelGGL(sweep_kernel<2>, grid, block, 0, nullptr, coeff, scale, offset, height, d); break;
        case 3: hipLaunchKernelGGL(sweep_kernel<3>, grid, block, 0, nullptr, coeff, scale, offset, height, d); break;
        case 4: hipLaunchKernelGGL(sweep_kernel<4>, grid, block, 0, nullptr, coeff, scale, offset, height, d); break;
        case 5: hipLaunchKernelGGL(sweep_kernel<5>, grid, block, 0, nullptr, coeff, scale, offset, height, d); break;
        case 6: hipLaunchKernelGGL(sweep_kernel<6>, grid, block, 0, nullptr, coeff, scale, offset, height, d); break;
        default: (void)hipFree(d); return hipErrorInvalidValue;
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    SweepResult h{};
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    if (mismatches) *mismatches = h.bad;
    if (visited) *visited = h.visited;
    if (first_bad) *first_bad = which == 6 ? h.pad : h.first;  // WHICH 6 reports the largest deviation's float bits
    return hipSuccess;
}

}  // namespace jst::kernels
