// fft_kernels.hip -- instantiations and host launchers of the LDS Stockham FFT (fft_lds.hh).
#include "fft_lds.hh"
#include "kernels.hh"

#include <cstdlib>

namespace jst::kernels {

using namespace jst::dev;

namespace {

// JST_FFT_KERNEL=slot selects the non-pipelined kernel (A/B comparisons, tests run both).
inline bool use_pipe_kernel() {
    const char* e = getenv("JST_FFT_KERNEL");
    return !(e && e[0] == 's');
}
inline bool window_contig(const LoadCF32&) { return true; }
inline bool window_contig(const LoadCF32TimesWindow& p) { return p.wstride == 1; }

int compute_units() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

template <int N, bool FWD, class Pro, class Epi>
hipError_t launch_pipe(const FftLayout& L, const float2* W, const Pro& pro, const Epi& epi,
                       hipStream_t stream) {
    constexpr size_t lds = fft_pipe_lds_bytes(N);
    const bool contig = L.in_axis_stride == 1 && L.out_axis_stride == 1 && window_contig(pro);
    auto kernel = contig ? fft_pipe_kernel<N, FWD, true, Pro, Epi> : fft_pipe_kernel<N, FWD, false, Pro, Epi>;
    if (lds > 64 * 1024) {
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
        if (e != hipSuccess) return e;
    }
    if (L.transforms == 0) return hipSuccess;
    // Resident workgroups: LDS allows floor(160 KiB / lds) per CU.  Give every resident
    // workgroup at least two transforms when there are enough (so the prefetch pipeline fills).
    uint64_t per_cu = (160 * 1024) / lds > 0 ? (160 * 1024) / lds : 1;
    if (per_cu > 2048 / (N / 8)) per_cu = 2048 / (N / 8);  // 32 waves per CU
    const uint64_t resident = per_cu * (uint64_t)compute_units();
    const uint64_t blocks = L.transforms < resident ? L.transforms : resident;
    (void)hipGetLastError();  // drop any stale error: only this launch is judged
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(N / 8), lds, stream, L, W, pro, epi);
    return hipGetLastError();
}

template <int N, bool FWD, class Pro, class Epi>
hipError_t launch_one(const FftLayout& L, const float2* W, const Pro& pro, const Epi& epi,
                      hipStream_t stream) {
    if constexpr (fft_pipe_supported(N)) {
        if (use_pipe_kernel()) return launch_pipe<N, FWD, Pro, Epi>(L, W, pro, epi, stream);
    }
    constexpr int TPB = fft_transforms_per_block(N);
    constexpr size_t lds = fft_lds_bytes(N);
    const bool contig = L.in_axis_stride == 1 && L.out_axis_stride == 1 && window_contig(pro);
    auto kernel = contig ? fft_lds_kernel<N, FWD, true, Pro, Epi> : fft_lds_kernel<N, FWD, false, Pro, Epi>;
    if constexpr (lds > 64 * 1024) {
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint64_t blocks_needed = (L.transforms + TPB - 1) / TPB;
    // Enough workgroups to occupy every CU several times over, grid-stride beyond that.
    const uint64_t blocks = blocks_needed < 8192 ? blocks_needed : 8192;
    if (blocks == 0) return hipSuccess;
    (void)hipGetLastError();  // drop any stale error: only this launch is judged
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(fft_block_threads(N)), lds, stream, L,
                       W, pro, epi);
    return hipGetLastError();
}

template <bool FWD, class Pro, class Epi>
hipError_t dispatch_n(uint64_t n, const FftLayout& L, const float2* W, const Pro& pro,
                      const Epi& epi, hipStream_t stream) {
    switch (n) {
#define JST_FFT_CASE(NN) \
    case NN:             \
        return launch_one<NN, FWD, Pro, Epi>(L, W, pro, epi, stream);
        JST_FFT_CASE(1)
        JST_FFT_CASE(2)
        JST_FFT_CASE(4)
        JST_FFT_CASE(8)
        JST_FFT_CASE(16)
        JST_FFT_CASE(32)
        JST_FFT_CASE(64)
        JST_FFT_CASE(128)
        JST_FFT_CASE(256)
        JST_FFT_CASE(512)
        JST_FFT_CASE(1024)
        JST_FFT_CASE(2048)
        JST_FFT_CASE(4096)
        JST_FFT_CASE(8192)
        JST_FFT_CASE(16384)
#undef JST_FFT_CASE
        default:
            return hipErrorInvalidValue;
    }
}

template <class Pro, class Epi>
hipError_t dispatch_fused_n(uint64_t n, const FftLayout& L, const float2* W, const Pro& pro,
                            const Epi& epi, hipStream_t stream) {
    switch (n) {
#define JST_FFT_CASE(NN) \
    case NN:             \
        return launch_one<NN, true, Pro, Epi>(L, W, pro, epi, stream);
        JST_FFT_CASE(256)
        JST_FFT_CASE(512)
        JST_FFT_CASE(1024)
        JST_FFT_CASE(2048)
        JST_FFT_CASE(4096)
        JST_FFT_CASE(8192)
        JST_FFT_CASE(16384)
#undef JST_FFT_CASE
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace

bool fft_lds_supported(uint64_t n) { return n >= 1 && n <= 16384 && (n & (n - 1)) == 0; }
bool fft_fused_supported(uint64_t n) { return n >= 256 && fft_lds_supported(n); }

hipError_t launch_fft_c2c(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                          const float2* in, float2* out, hipStream_t stream) {
    const LoadCF32 pro{in};
    const StoreCF32 epi{out};
    return forward ? dispatch_n<true>(n, L, W, pro, epi, stream)
                   : dispatch_n<false>(n, L, W, pro, epi, stream);
}

hipError_t launch_spectrum_fused(uint64_t n, const FftLayout& L, const float2* W,
                                 const float2* in, const float2* window, int64_t window_stride,
                                 float* out, float amp_coeff, bool with_range, float range_scale,
                                 float range_offset, bool fast, float guard_h0, float guard_h1,
                                 hipStream_t stream) {
    const LoadCF32TimesWindow pro{in, window, window_stride};
    if (with_range) {
        if (fast)
            return dispatch_fused_n(n, L, W, pro,
                                    StoreAmplitudeRangeT<true>{out, amp_coeff, range_scale, range_offset, dev::BinGuard{guard_h0, guard_h1}},
                                    stream);
        return dispatch_fused_n(n, L, W, pro,
                                StoreAmplitudeRangeT<false>{out, amp_coeff, range_scale, range_offset, dev::BinGuard{}},
                                stream);
    }
    if (fast) return dispatch_fused_n(n, L, W, pro, StoreAmplitudeT<true>{out, amp_coeff}, stream);
    return dispatch_fused_n(n, L, W, pro, StoreAmplitudeT<false>{out, amp_coeff}, stream);
}

}  // namespace jst::kernels
