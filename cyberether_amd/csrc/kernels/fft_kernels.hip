// fft_kernels.hip -- instantiations and host launchers of the LDS Stockham FFT (fft_lds.hh).
#include "fft_lds.hh"
#include "kernels.hh"
#include "spectrogram_body.hh"

#include <cstdlib>

namespace jst::kernels {

using namespace jst::dev;

namespace {

// JST_FFT_KERNEL=slot selects the non-pipelined kernel (A/B comparisons, tests run both).
inline bool use_pipe_kernel() { return jst::switch_value(jst::SW_FFT_KERNEL) != 's'; }
inline bool window_contig(const LoadCF32&) { return true; }
inline bool window_contig(const LoadCF32TimesWindow& p) { return p.wstride == 1; }
template <class RAW, bool SIGNED>
inline bool window_contig(const LoadCITimesWindow<RAW, SIGNED>& p) { return p.wstride == 1; }

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

namespace {
template <class Pro>
hipError_t spectrum_fused_with(uint64_t n, const FftLayout& L, const float2* W, const Pro& pro, float* out,
                               float amp_coeff, bool with_range, float range_scale, float range_offset, bool fast,
                               float guard_h0, float guard_h1, hipStream_t stream) {
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
}  // namespace

hipError_t launch_spectrum_fused(uint64_t n, const FftLayout& L, const float2* W,
                                 const float2* in, const float2* window, int64_t window_stride,
                                 float* out, float amp_coeff, bool with_range, float range_scale,
                                 float range_offset, bool fast, float guard_h0, float guard_h1,
                                 hipStream_t stream) {
    return spectrum_fused_with(n, L, W, LoadCF32TimesWindow{in, window, window_stride}, out, amp_coeff, with_range,
                               range_scale, range_offset, fast, guard_h0, guard_h1, stream);
}

// The same chain fed with raw SDR samples: Cast folded into the transform's first load (LoadCITimesWindow).
// in_format: 1 = CI16, 2 = CI8, 3 = CU8; scaler = the Cast module's divisor (32768 / 128).
hipError_t launch_spectrum_fused_cast(uint64_t n, const FftLayout& L, const float2* W, const void* in, int in_format,
                                      float scaler, const float2* window, int64_t window_stride, float* out,
                                      float amp_coeff, bool with_range, float range_scale, float range_offset, bool fast,
                                      float guard_h0, float guard_h1, hipStream_t stream) {
    const float inv = 1.0f / scaler;  // a power of two: x / scaler == x * inv, exactly
    switch (in_format) {
        case 1:
            return spectrum_fused_with(n, L, W, LoadCI16TimesWindow{static_cast<const uint32_t*>(in), window, window_stride, inv},
                                       out, amp_coeff, with_range, range_scale, range_offset, fast, guard_h0, guard_h1, stream);
        case 2:
            return spectrum_fused_with(n, L, W, LoadCI8TimesWindow{static_cast<const uint16_t*>(in), window, window_stride, inv},
                                       out, amp_coeff, with_range, range_scale, range_offset, fast, guard_h0, guard_h1, stream);
        case 3:
            return spectrum_fused_with(n, L, W, LoadCU8TimesWindow{static_cast<const uint16_t*>(in), window, window_stride, inv},
                                       out, amp_coeff, with_range, range_scale, range_offset, fast, guard_h0, guard_h1, stream);
        default:
            return hipErrorInvalidValue;
    }
}

// ---- spectrum of cycle k + spectrogram of cycle k - 1 in ONE launch ------------------------------------------------
// A Spectrogram that is the only consumer of the fused spectrum kernel's output needs, per cycle, one more kernel
// whose life is a chain of latencies (dispatch, one Infinity-Cache round trip, LDS atomics, state update) and, like
// every kernel, ~2 us of begin / end processing.  Carried by the NEXT cycle's spectrum launch as 512-thread workgroups
// in front of the transforms (blockIdx < tiles: they take the first slot of every CU), those latencies run under the
// spectrum kernel's own start-up (its first ~4 us are dispatch and the first HBM round trip) and a launch disappears:
// 26.25 -> 24.92 us per cycle under graph replay, same box (tools/ubench/combined_bench.hip).  The output tensor is a
// ring of two: launch k writes half k & 1 while the spectrogram part reads half (k - 1) & 1.
//   ctrl[0]: 1 when the other half holds a spectrum whose spectrogram has not run yet (set by the last workgroup of
//            every launch through the ticket ctrl[1], cleared by the flush at the end of a compute call)
template <class Epi>
__global__ __launch_bounds__(512, 4) void spectrum_spectrogram_kernel(
    const FftLayout L, const float2* __restrict__ W, const LoadCF32TimesWindow pro, const Epi epi, float* __restrict__ bins,
    const float* __restrict__ spec_in, uint32_t batches, uint32_t width, uint32_t height, float decay, uint32_t tiles,
    uint32_t copies, uint32_t* __restrict__ ctrl) {
    if (blockIdx.x >= tiles) {
        fft_pipe_body<4096, true, true, LoadCF32TimesWindow, Epi>(L, W, pro, epi, blockIdx.x - tiles, gridDim.x - tiles);
    } else if (__hip_atomic_load(ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        using specdev::spectrogram_body;
        if (copies == 4u)
            spectrogram_body<16, 4, 512, 32, true>(bins, spec_in, 0, batches, width, height, (int64_t)width, 1, decay, blockIdx.x, tiles);
        else if (copies == 2u)
            spectrogram_body<16, 2, 512, 32, true>(bins, spec_in, 0, batches, width, height, (int64_t)width, 1, decay, blockIdx.x, tiles);
        else
            spectrogram_body<16, 1, 512, 32, true>(bins, spec_in, 0, batches, width, height, (int64_t)width, 1, decay, blockIdx.x, tiles);
    }
    // Ticket: the last workgroup to finish marks this launch's spectrum as pending.  Relaxed atomics only: the flag is
    // read by the NEXT launch (the kernel boundary orders it), and a release fence here would make every workgroup
    // write back its XCD's L2 (measured: 69 instead of 25 us per launch).
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1u) {
            __hip_atomic_store(ctrl + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

bool spectrum_spectrogram_supported(uint64_t n, const FftLayout& L, int64_t window_stride, uint64_t height) {
    return n == 4096 && use_pipe_kernel() && L.in_axis_stride == 1 && L.out_axis_stride == 1 && window_stride == 1 &&
           L.outer_rank == 1 && L.in_outer_stride[0] == (int64_t)n && L.out_outer_stride[0] == (int64_t)n &&
           L.transforms >= 2 && L.transforms < (1ull << 17) && height >= 2 && height <= 1024;
}

template <class Epi>
hipError_t launch_ss(const FftLayout& L, const float2* W, const LoadCF32TimesWindow& pro, const Epi& epi, float* bins,
                     const float* spec_in, uint64_t height, float decay, uint32_t* ctrl, hipStream_t stream) {
    constexpr uint64_t n = 4096;
    const size_t lds_f = fft_pipe_lds_bytes(4096), lds_s = spectrogram_lds_bytes(height);
    const size_t lds = lds_f > lds_s ? lds_f : lds_s;
    auto kernel = spectrum_spectrogram_kernel<Epi>;
    {
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)(80 * 1024));
        if (e != hipSuccess) return e;
    }
    const uint64_t resident = 2ull * (uint64_t)compute_units();
    const uint64_t fft_blocks = L.transforms < resident ? L.transforms : resident;
    const uint32_t tiles = (uint32_t)(n / 16);
    const uint32_t copies = height <= 256 ? 4u : (height <= 512 ? 2u : 1u);
    (void)hipGetLastError();
    hipLaunchKernelGGL(kernel, dim3((unsigned)(tiles + fft_blocks)), dim3(512), lds, stream, L, W, pro, epi, bins, spec_in,
                       (uint32_t)L.transforms, (uint32_t)n, (uint32_t)height, decay, tiles, copies, ctrl);
    return hipGetLastError();
}

hipError_t launch_spectrum_spectrogram_fused(const FftLayout& L, const float2* W, const float2* in, const float2* window,
                                             float* out, float amp_coeff, bool with_range, float range_scale,
                                             float range_offset, bool fast, float guard_h0, float guard_h1, float* bins,
                                             const float* spec_in, uint64_t height, float decay, uint32_t* ctrl,
                                             hipStream_t stream) {
    const LoadCF32TimesWindow pro{in, window, 1};
    if (with_range) {
        if (fast)
            return launch_ss(L, W, pro, StoreAmplitudeRangeT<true>{out, amp_coeff, range_scale, range_offset, dev::BinGuard{guard_h0, guard_h1}},
                             bins, spec_in, height, decay, ctrl, stream);
        return launch_ss(L, W, pro, StoreAmplitudeRangeT<false>{out, amp_coeff, range_scale, range_offset, dev::BinGuard{}}, bins,
                         spec_in, height, decay, ctrl, stream);
    }
    if (fast) return launch_ss(L, W, pro, StoreAmplitudeT<true>{out, amp_coeff}, bins, spec_in, height, decay, ctrl, stream);
    return launch_ss(L, W, pro, StoreAmplitudeT<false>{out, amp_coeff}, bins, spec_in, height, decay, ctrl, stream);
}

}  // namespace jst::kernels
