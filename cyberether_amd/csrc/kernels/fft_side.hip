// fft_side.hip -- the fused spectrum kernel (fft_lds.hh: fft_pipe_kernel) with the Spectrogram's row index as a side
// output (StoreAmplitudeRangeSideT): Multiply(window) -> FFT -> Amplitude -> Range writes its F32 rows as before and,
// beside every value, the one-byte index `(u32)(value * height)` (0 = no hit; tile-major, see the functor) the
// Spectrogram consumer would derive from it (spectrogram/module_impl_native_cpu.cc:70-77).  The consumer (spectrogram.hip: spectrogram_index_kernel)
// then reads 1 byte per sample instead of 4.  Its own translation unit: the instantiations compile beside
// fft_kernels.hip, not behind it.
// The input stream is read ONCE: `nt` on its 8-byte loads keeps it from displacing what the consumer comes back for -- under
// cycle batching the Spectrogram walks 64 MiB of indices behind a launch that streamed 512 MiB in and 256 MiB out, and with
// plain loads little of them was left in the Infinity Cache (same box, tools/ubench/run_r03u.sh: spectrogram span 40.8 ->
// 36.1 us, fused span 188-193 -> 187 us, step 14.46-14.73 -> 14.07 us; per-cycle launches within the noise: 19.3 vs 19.4-20.1
// us).  `nt` on the F32 stores as well was no better (35.5 / 190.0 us); fft_kernels.hip keeps plain loads.
#ifndef JST_LOAD_AUX
#define JST_LOAD_AUX 2
#endif
// Round 4: the F32 VALUE stores are `sc1 nt` (written through at agent scope AND streaming), the one-byte index stores stay
// `sc1`: nobody in this chain comes back for the values (the Spectrogram reads the indices), and without `nt` the 16 MiB of
// values per cycle push the 4 MiB of indices out of the Infinity Cache before the span Spectrogram reads them.  Same box,
// bench.py (tools/ubench/run_r04s.sh): ring period 16: step 12.94 -> 12.78 us; ring period 32 (one launch = 32 cycles):
// 12.92 -> 12.36 us, fused kernel 161.8 -> 165.9 us per 16 cycles but the span Spectrogram no longer reads cold indices
// (`nt` alone: 12.81 / 12.51 us).
#ifndef JST_STORE_AUX
#define JST_STORE_AUX 18
#endif
#ifndef JST_SIDE_STORE_AUX
#define JST_SIDE_STORE_AUX 16
#endif
// The window operand stays RESIDENT in 16 VGPRs instead of being re-requested from L2 behind every retired output
// (fft_lds.hh: JST_OPND_RESIDENT).  Per 1024-transform launch that was a wash (the launch is bound by its ramp and tail,
// DESIGN.md section 4); in the steady state of a cycle-batched launch it is not: 188.6 -> 178.0 us per 16384 transforms
// (fast; exact 253.3 -> 250.6), same box, bit-identical (tools/ubench/run_r03v.sh).  Re-checked there and left as they
// were: 16-byte loads (195.9), no wave priorities (201.1), two other priority sets (193.5), the real-operand form (193.0).
#ifndef JST_OPND_RESIDENT
#define JST_OPND_RESIDENT 1
#endif
#include "fft_quad.hh"
#include "kernels.hh"

#include <cstdlib>

namespace jst::kernels {

using namespace jst::dev;

namespace {

int side_compute_units() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

template <int N, class Pro, class Epi>
hipError_t launch_side(const FftLayout& L, const float2* W, const Pro& pro, const Epi& epi, hipStream_t stream) {
    constexpr size_t lds = fft_pipe_lds_bytes(N);
    auto kernel = fft_pipe_kernel<N, true, true, Pro, Epi>;
    if (lds > 64 * 1024) {
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
        if (e != hipSuccess) return e;
    }
    if (L.transforms == 0) return hipSuccess;
    uint64_t per_cu = (160 * 1024) / lds > 0 ? (160 * 1024) / lds : 1;  // as launch_pipe (fft_kernels.hip)
    if (per_cu > 2048 / (N / 8)) per_cu = 2048 / (N / 8);
    const uint64_t resident = per_cu * (uint64_t)side_compute_units();
    const uint64_t blocks = L.transforms < resident ? L.transforms : resident;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(N / 8), lds, stream, L, W, pro, epi);
    return hipGetLastError();
}

// Round 5: 4096 points, CF32 rows, provider fast with a real window -- fft_quad_kernel (fft_quad.hh): 256 threads per transform,
// in-place exchange, rows by LDS-DMA, four workgroups per CU.  Same box, alternating launches of 16384 transforms
// (tools/ubench/run_r05e.sh): 166-168 us against 180-183 for fft_pipe_kernel (0.92), every value and index byte identical.
// SHORT launches keep fft_pipe_kernel: its 512 workgroups start in half the time of the quad kernel's 1024 (the dispatcher
// hands out ~85 workgroups per us), and a launch of one round has nothing to pipeline in either -- 1024 transforms 17.5 us
// against 20.3, 4096 transforms 52.7 = 52.9, 8192 transforms 93.0 against 92.0 (r05_experiments/q_small_launches.log); the
// quad kernel takes launches of eight rounds and more (where its claimed rounds start, too).
// JST_FFT_KERNEL=pipe keeps the pipelined kernel throughout, =quad the quad kernel at every size (A/B, tests run both).
bool quad_selected(uint64_t transforms) {  // a switch, not the environment: the tests switch kernels inside one process (jst_debug_set)
    const int k = jst::switch_value(jst::SW_FFT_KERNEL);
    if (k == 'p' || k == 'w' || k == 's') return false;
    if (k == 'q') return true;
    return transforms >= 8ull * 4ull * (uint64_t)side_compute_units();
}

template <class Pro, class Epi>
hipError_t launch_side_quad(const FftLayout& L, const float2* W, const Pro& pro, const Epi& epi, uint32_t* sched, hipStream_t stream) {
    constexpr size_t lds = fft_quad_lds_bytes();
    static_assert(lds <= 64 * 1024, "no opt-in needed for the dynamic LDS size");
    if (L.transforms == 0) return hipSuccess;
    const uint64_t resident = 4ull * (uint64_t)side_compute_units();
    const uint64_t blocks = L.transforms < resident ? L.transforms : resident;
    // Claimed rounds only for launches of eight rounds and more (a cycle-batched span): with fewer, the claims of a whole
    // round arrive at the counters at once -- a 1024-transform launch took 32 us instead of 20 (i_hybrid_eight_counters.log).
    // JST_QUAD_STATIC=1: the static round robin throughout (A/B).
    const bool all_static = jst::switch_value(jst::SW_QUAD_STATIC) != 0;
    if (all_static || L.transforms < 8 * resident) sched = nullptr;
    (void)hipGetLastError();
    hipLaunchKernelGGL((fft_quad_kernel<true, Pro, Epi>), dim3((unsigned)blocks), dim3(kQuadT), lds, stream, L, W, pro, epi, sched);
    return hipGetLastError();
}

template <class Pro>
hipError_t side_with(uint64_t n, const FftLayout& L, const float2* W, const Pro& pro, float* out, float amp_coeff,
                     float range_scale, float range_offset, bool fast, float guard_h0, float guard_h1, uint8_t* side,
                     float side_height, uint32_t side_batches, uint32_t side_pitch, uint32_t* sched, hipStream_t stream) {
    // the fed Spectrogram's height first in the guard (the epilogue takes the row index from the guard's own product
    // value * h0), the other consumer height -- if any -- second
    const float other = guard_h0 != side_height ? guard_h0 : (guard_h1 != side_height ? guard_h1 : 0.0f);
    const StoreAmplitudeRangeSideT<true> ef{{out, amp_coeff, range_scale, range_offset, dev::BinGuard{side_height, other}}, side, side_height, side_batches, side_pitch};
    const StoreAmplitudeRangeSideT<false> ee{{out, amp_coeff, range_scale, range_offset, dev::BinGuard{}}, side, side_height, side_batches, side_pitch};
    if constexpr (Pro::kRawBytes == 8 && requires { Pro::kRealOperand; }) {
        if (n == 4096 && fast && quad_selected(L.transforms)) return launch_side_quad(L, W, pro, ef, sched, stream);
    }
    switch (n) {
#define JST_SIDE_CASE(NN)                                                  \
    case NN:                                                               \
        return fast ? launch_side<NN, Pro>(L, W, pro, ef, stream) : launch_side<NN, Pro>(L, W, pro, ee, stream);
        JST_SIDE_CASE(1024)
        JST_SIDE_CASE(2048)
        JST_SIDE_CASE(4096)
        JST_SIDE_CASE(8192)
#undef JST_SIDE_CASE
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace

// Rows a 128-column group of the tile-major side tensor occupies: the batches plus two pad rows (256 bytes of skew per
// group) that take the group stride off the powers of two.
uint64_t spectrum_side_pitch(uint64_t batches) { return batches + 2; }
uint64_t spectrum_sched_words() { return kQuadSchedWords; }

bool spectrum_side_supported(uint64_t n, const FftLayout& L, int64_t window_stride, uint64_t height) {
    if (jst::switch_value(jst::SW_FFT_KERNEL) == 's') return false;  // the non-pipelined kernel has no side store
    if (n != 1024 && n != 2048 && n != 4096 && n != 8192) return false;
    // dense rows on both sides, transform t's output row at element t * n (the side tensor follows the same numbering)
    if (L.in_axis_stride != 1 || L.out_axis_stride != 1 || window_stride != 1 || L.outer_rank != 1) return false;
    if (L.out_outer_stride[0] != (int64_t)n) return false;
    if (L.ring_transforms != 0 && (L.ring_first >= L.ring_transforms || L.ring_transforms * n >= (1ull << 28) ||
                                   L.transforms >= (1ull << 31)))
        return false;  // a ring's rows are addressed with 32-bit byte offsets: < 2^31 bytes of cf32
    return height >= 2 && height <= 256 && (L.ring_transforms != 0 || L.transforms * n < (1ull << 31));
}

hipError_t launch_spectrum_fused_side(uint64_t n, const FftLayout& L, const float2* W, const void* in, int in_format,
                                      float scaler, const float2* window, float* out, float amp_coeff,
                                      float range_scale, float range_offset, bool fast, float guard_h0, float guard_h1,
                                      uint8_t* side, uint64_t height, uint64_t side_batches, uint64_t side_pitch, bool real_window,
                                      hipStream_t stream, uint32_t* sched) {
    // side_batches: the rows of ONE index tensor (a compute cycle's batches); L.transforms is a whole number of them
    if (!spectrum_side_supported(n, L, 1, height) || !side || side_batches == 0 || L.transforms % side_batches != 0 ||
        side_pitch < side_batches ||
        ((L.ring_transforms ? L.ring_transforms : L.transforms) / side_batches) * side_pitch * n >= (1ull << 31) ||
        (L.ring_transforms % side_batches) != 0 || (L.ring_first % side_batches) != 0)
        return hipErrorInvalidValue;
    const float h = (float)height;
    // CF32 input, 4096 points: the one-wavefront-per-transform kernel (fft_wave.hip) when it is selected
    if (n == 4096 && in_format == 0 && spectrum_wave_selected())
        return launch_spectrum_wave_side(L, W, static_cast<const float2*>(in), window, out, amp_coeff, range_scale,
                                         range_offset, fast, guard_h0, guard_h1, side, h, (uint32_t)side_batches,
                                         (uint32_t)side_pitch, real_window && fast, stream);
    const float inv = in_format ? 1.0f / scaler : 1.0f;  // a power of two: x / scaler == x * inv, exactly
    // real_window (host knowledge: every imaginary part of the window is +-0) with provider "fast": the RealOperand
    // instantiations (fft_lds.hh) -- two products per sample instead of std::complex's full product
    const bool real = real_window && fast;
#define JST_SIDE_WITH(PRO)                                                                                             \
    (real ? side_with(n, L, W, RealOperand<decltype(PRO)>{PRO}, out, amp_coeff, range_scale, range_offset, fast, guard_h0, \
                      guard_h1, side, h, (uint32_t)side_batches, (uint32_t)side_pitch, sched, stream)                   \
          : side_with(n, L, W, PRO, out, amp_coeff, range_scale, range_offset, fast, guard_h0, guard_h1, side, h,       \
                      (uint32_t)side_batches, (uint32_t)side_pitch, sched, stream))
    switch (in_format) {
        case 0:
            return JST_SIDE_WITH((LoadCF32TimesWindow{static_cast<const float2*>(in), window, 1}));
        case 1:
            return JST_SIDE_WITH((LoadCI16TimesWindow{static_cast<const uint32_t*>(in), window, 1, inv}));
        case 2:
            return JST_SIDE_WITH((LoadCI8TimesWindow{static_cast<const uint16_t*>(in), window, 1, inv}));
        case 3:
            return JST_SIDE_WITH((LoadCU8TimesWindow{static_cast<const uint16_t*>(in), window, 1, inv}));
        default:
            return hipErrorInvalidValue;
    }
#undef JST_SIDE_WITH
}

}  // namespace jst::kernels
