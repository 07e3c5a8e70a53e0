// ref_pocketfft.cc -- thin extern "C" shim around the REFERENCE's own vendored pocketfft.hh.
//
// TEST INFRASTRUCTURE ONLY (see oracle/jst_oracle.c header).  This file contains no reference
// code: it #includes /root/reference/src/domains/dsp/fft/pocketfft.hh where it lies (never copied
// into this repo) and exposes exactly the three calls the reference FFT module makes
// (src/domains/dsp/fft/module_impl_native_cpu.cc:125-167), with the same fct = 1.0f and with
// POCKETFFT_NO_MULTITHREADING as at :1-2.  Built by oracle/Makefile into oracle/_ref/ only when
// /root/reference exists (i.e. in the build container); the .so then travels to the GPU box.
#define POCKETFFT_NO_MULTITHREADING
#include "pocketfft.hh"

#include <complex>
#include <cstddef>
#include <cstdint>

namespace {
pocketfft::shape_t to_shape(uint32_t rank, const uint64_t* v) {
    pocketfft::shape_t s;
    for (uint32_t i = 0; i < rank; ++i) s.push_back(static_cast<std::size_t>(v[i]));
    return s;
}
pocketfft::stride_t to_stride(uint32_t rank, const int64_t* v) {
    pocketfft::stride_t s;
    for (uint32_t i = 0; i < rank; ++i) s.push_back(static_cast<std::ptrdiff_t>(v[i]));
    return s;
}
}  // namespace

extern "C" {

// Strides in BYTES (as the reference passes them, module_impl_native_cpu.cc:102-108).
int ref_fft_c2c(uint32_t rank, const uint64_t* shape, const int64_t* stride_in,
                const int64_t* stride_out, uint64_t axis, int forward, const float* in,
                float* out) {
    try {
        pocketfft::c2c(to_shape(rank, shape), to_stride(rank, stride_in),
                       to_stride(rank, stride_out), pocketfft::shape_t{static_cast<std::size_t>(axis)},
                       forward != 0, reinterpret_cast<const std::complex<float>*>(in),
                       reinterpret_cast<std::complex<float>*>(out), 1.0f);
    } catch (...) {
        return -1;
    }
    return 0;
}

int ref_fft_r2c(uint32_t rank, const uint64_t* shape, const int64_t* stride_in,
                const int64_t* stride_out, uint64_t axis, int forward, const float* in,
                float* out) {
    try {
        pocketfft::r2c(to_shape(rank, shape), to_stride(rank, stride_in),
                       to_stride(rank, stride_out), pocketfft::shape_t{static_cast<std::size_t>(axis)},
                       forward != 0, in, reinterpret_cast<std::complex<float>*>(out), 1.0f);
    } catch (...) {
        return -1;
    }
    return 0;
}

int ref_fft_r2r_fftpack(uint32_t rank, const uint64_t* shape, const int64_t* stride_in,
                        const int64_t* stride_out, uint64_t axis, int forward, const float* in,
                        float* out) {
    try {
        pocketfft::r2r_fftpack(to_shape(rank, shape), to_stride(rank, stride_in),
                               to_stride(rank, stride_out),
                               pocketfft::shape_t{static_cast<std::size_t>(axis)}, forward != 0,
                               forward != 0, in, out, 1.0f);
    } catch (...) {
        return -1;
    }
    return 0;
}

}  // extern "C"
