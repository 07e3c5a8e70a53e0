// fft_wave.hip -- instantiations and launcher of the one-wavefront-per-transform 4096-point spectrum kernel
// (fft_wave.hh) with the Spectrogram's row-index side output.  Its own translation unit because it is built with
// -fno-slp-vectorize (Makefile): the SLP vectorizer merges loads of neighbouring elements of the kernel's 64-element
// register arrays into <3 x float> accesses BEFORE the arrays are promoted to registers, the promotion then fails and the
// arrays live in scratch (976 B per lane, 132 spilled VGPRs; without the vectorizer: 254 VGPRs, no scratch).
#ifndef JST_LOAD_AUX  // the input stream is read once: `nt` (see fft_side.hip)
#define JST_LOAD_AUX 2
#endif
#include "fft_wave.hh"
#include "kernels.hh"

#include <cstdlib>

namespace jst::kernels {

using namespace jst::dev;

#ifndef JST_WAVE_DEFAULT  // which 4096-point side kernel runs when JST_FFT_KERNEL does not say (decided by measurement)
#define JST_WAVE_DEFAULT 0
#endif

// JST_FFT_KERNEL=wave / pipe selects (read once)
bool spectrum_wave_selected() {
    static const bool on = [] {
        const int k = jst::switch_value(jst::SW_FFT_KERNEL);
        if (k == 'w') return true;
        if (k == 'p') return false;
        return JST_WAVE_DEFAULT != 0;
    }();
    return on;
}

namespace {

int wave_compute_units() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

template <class Pro, class Epi>
hipError_t launch_wave(const FftLayout& L, const float2* W, const Pro& pro, const Epi& epi, hipStream_t stream) {
    constexpr size_t lds = fft_wave_lds_bytes();
    auto kernel = fft_wave4096_kernel<true, Pro, Epi>;
    const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
    if (e != hipSuccess) return e;
    if (L.transforms == 0) return hipSuccess;
    const uint64_t cus = (uint64_t)wave_compute_units();
    const uint64_t blocks = L.transforms < cus ? L.transforms : cus;  // one workgroup (8 wavefronts) per CU
    (void)hipGetLastError();
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kWaveWaves * 64), lds, stream, L, W, pro, epi);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_spectrum_wave_side(const FftLayout& L, const float2* W, const float2* in, const float2* window, float* out,
                                     float amp_coeff, float range_scale, float range_offset, bool fast, float guard_h0,
                                     float guard_h1, uint8_t* side, float side_height, uint32_t side_batches,
                                     uint32_t side_pitch, bool real_window, hipStream_t stream) {
    if (L.outer_rank != 1 || L.in_axis_stride != 1 || L.out_axis_stride != 1) return hipErrorInvalidValue;
    const float other = guard_h0 != side_height ? guard_h0 : (guard_h1 != side_height ? guard_h1 : 0.0f);
    const StoreAmplitudeRangeSideT<true> ef{{out, amp_coeff, range_scale, range_offset, dev::BinGuard{side_height, other}}, side, side_height, side_batches, side_pitch};
    const StoreAmplitudeRangeSideT<false> ee{{out, amp_coeff, range_scale, range_offset, dev::BinGuard{}}, side, side_height, side_batches, side_pitch};
    const LoadCF32TimesWindow pro{in, window, 1};
    if (fast) {
        if (real_window) return launch_wave(L, W, RealOperand<LoadCF32TimesWindow>{pro}, ef, stream);
        return launch_wave(L, W, pro, ef, stream);
    }
    return launch_wave(L, W, pro, ee, stream);
}

}  // namespace jst::kernels
