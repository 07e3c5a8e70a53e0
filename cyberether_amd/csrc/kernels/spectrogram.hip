// spectrogram.hip -- the Spectrogram module's compute (decaying 2-D persistence histogram) as a
// column-tiled LDS histogram on gfx950.  The reference has a CPU implementation only
// (src/domains/visualization/spectrogram/module_impl_native_cpu.cc:61-87):
//
//     for all bins: bins *= decay
//     for b, x:  index = (U64)(in[b,x] * height);  if 0 < index < height:
//                bins[x + index*width] = min(bins[...] + 0.02f, 1.0f)
//
// The float update is the SAME operation repeated once per hit, so the result depends only on
// the hit COUNT per bin, not on the order: we count hits with integer LDS atomics (exact, order
// free) and then apply min(v + 0.02f, 1.0f) count-times to the decayed value -- bit-identical
// to the sequential CPU loop.  The bin index rule is stated without the reference's undefined
// float->U64 cast: hit <=> 1.0f <= f < (float)height, index = (u32)f ("hard
// parts"; on x86-64 every other input fails 0 < index < height).
//
// Mapping: one workgroup owns a tile of TW adjacent columns for ALL batches (the only place a
// bin's hits can come from), so no inter-workgroup communication exists.  Rows of the F32[B,N]
// input are read TW*4 bytes at a time (64 B at TW = 16) -- they were just written by the
// spectrum kernel and sit in L2 / Infinity Cache.  The state tile is read-modified-written once.
#include <cstdlib>

#include "device_math.hh"
#include "kernels.hh"
#include "spectrogram_body.hh"

namespace jst::kernels {

namespace {
using namespace specdev;

template <int TW, int COPIES, int kThreads = kThreadsDefault, int kDepthT = 16, bool BUF = false>
__global__ __launch_bounds__(kThreads) void spectrogram_kernel(
    float* __restrict__ bins, const float* __restrict__ in, uint64_t in_offset, uint32_t batches,
    uint32_t width, uint32_t height, int64_t batch_stride, int64_t elem_stride, float decay) {
    spectrogram_body<TW, COPIES, kThreads, kDepthT, BUF>(bins, in, in_offset, batches, width, height, batch_stride,
                                                        elem_stride, decay, blockIdx.x, gridDim.x);
}

// The same histogram with the COUNTS as the result (U32[height][width], written over `counts`): the device half of
// the exact multi-GPU spectrogram merge -- spectrogram/module_impl_native_cpu.cc:61-87 applies min(v + 0.02f, 1.0f)
// once per hit, so the update depends on the hit count per bin only, and integer counts add exactly across ranks.
template <int TW, int COPIES>
__global__ __launch_bounds__(kThreadsDefault) void spectrogram_counts_kernel(
    uint32_t* __restrict__ counts, const float* __restrict__ in, uint64_t in_offset, uint32_t batches,
    uint32_t width, uint32_t height, int64_t batch_stride, int64_t elem_stride) {
    spectrogram_body<TW, COPIES, kThreadsDefault, 16, false, true>(reinterpret_cast<float*>(counts), in, in_offset, batches,
                                                                   width, height, batch_stride, elem_stride, 1.0f,
                                                                   blockIdx.x, gridDim.x);
}

// bins = decay * bins, then min(v + 0.02f, 1.0f) applied counts[i] times: what spectrogram_body does per tile, on
// counts that were summed over the ranks first.  decay = 0.999^(total batches of all ranks) (module_impl.cc:104).
__global__ __launch_bounds__(256) void spectrogram_apply_counts_kernel(float* __restrict__ bins,
                                                                       const uint32_t* __restrict__ counts,
                                                                       uint64_t cells, float decay) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (uint64_t)gridDim.x * 256) {
        float w = bins[i] * decay;
        uint32_t k = counts[i];
        k = k < 64u ? k : 64u;  // 0.02 * 51 > 1: pinned at 1.0f long before 64 hits
        for (uint32_t n = 0; n < k && w < 1.0f; ++n) w = fminf(w + 0.02f, 1.0f);
        store_state(bins + i, w);
    }
}

}  // namespace

namespace {
int spectrogram_copies(uint64_t height) { return height <= 256 ? 4 : (height <= 512 ? 2 : 1); }
}  // namespace

size_t spectrogram_lds_bytes(uint64_t height) {
    const int tw = height <= 1024 ? 16 : 8;
    return ((size_t)height * tw + 8) * spectrogram_copies(height) * sizeof(uint32_t);
}

hipError_t launch_spectrogram(float* bins, const float* in, uint64_t in_offset, uint64_t batches,
                              uint64_t width, uint64_t height, int64_t batch_stride,
                              int64_t elem_stride, float decay, hipStream_t stream) {
    if (width == 0 || height == 0) return hipSuccess;
    if (height > 2048 || batches > 0xffffffffull || width > 0xffffffffull)
        return hipErrorInvalidValue;
    const size_t lds = spectrogram_lds_bytes(height);
    const unsigned tiles16 = (unsigned)((width + 15) / 16), tiles8 = (unsigned)((width + 7) / 8);
    (void)hipGetLastError();  // drop any stale error: only this launch is judged
#define JST_SPEC_LAUNCH_B(TW, COPIES, THREADS, DEPTH, TILES, BUFV)                                  \
    do {                                                                                          \
        { /* the padded copies can exceed the 64 KiB default by a few words */                    \
            const hipError_t e = raise_dynamic_lds(                                               \
                reinterpret_cast<const void*>(spectrogram_kernel<TW, COPIES, THREADS, DEPTH, BUFV>), 80 * 1024); \
            if (e != hipSuccess) return e;                                                        \
        }                                                                                         \
        hipLaunchKernelGGL((spectrogram_kernel<TW, COPIES, THREADS, DEPTH, BUFV>), dim3(TILES),   \
                           dim3(THREADS), lds, stream, bins, in, in_offset, (uint32_t)batches,    \
                           (uint32_t)width, (uint32_t)height, batch_stride, elem_stride, decay);  \
    } while (0)
    // dense-enough input for the one-descriptor form: non-negative strides, rows that do not interleave, < 2 GiB
    const int64_t span = (int64_t)(batches ? batches - 1 : 0) * batch_stride + (int64_t)(width - 1) * elem_stride + 1;
    const bool buf_ok = elem_stride >= 0 && batch_stride >= (int64_t)(width - 1) * elem_stride + 1 && batches > 0 &&
                        span * 4 < (int64_t)0x7fff0000 && getenv("JST_SPEC_FLAT") == nullptr;
#define JST_SPEC_LAUNCH(TW, COPIES, THREADS, DEPTH, TILES)                          \
    do {                                                                            \
        if (buf_ok) JST_SPEC_LAUNCH_B(TW, COPIES, THREADS, DEPTH, TILES, true);     \
        else JST_SPEC_LAUNCH_B(TW, COPIES, THREADS, DEPTH, TILES, false);           \
    } while (0)
    if (height <= 256) {
        static const int threads = [] {  // A/B switch: JST_SPEC_THREADS=512|256 (fewer wavefronts to dispatch)
            const char* e = getenv("JST_SPEC_THREADS");
            return e ? atoi(e) : 1024;
        }();
        if (threads == 512) JST_SPEC_LAUNCH(16, 4, 512, 32, tiles16);
        else if (threads == 256) JST_SPEC_LAUNCH(16, 4, 256, 32, tiles16);
        else JST_SPEC_LAUNCH(16, 4, 1024, 16, tiles16);
    }
    else if (height <= 512) JST_SPEC_LAUNCH(16, 2, 1024, 16, tiles16);
    else if (height <= 1024) JST_SPEC_LAUNCH(16, 1, 1024, 16, tiles16);
    else JST_SPEC_LAUNCH(8, 1, 1024, 16, tiles8);
#undef JST_SPEC_LAUNCH
#undef JST_SPEC_LAUNCH_B
    return hipGetLastError();
}

hipError_t launch_spectrogram_counts(uint32_t* counts, const float* in, uint64_t in_offset, uint64_t batches,
                                     uint64_t width, uint64_t height, int64_t batch_stride, int64_t elem_stride,
                                     hipStream_t stream) {
    if (width == 0 || height == 0) return hipSuccess;
    if (height > 2048 || batches > 0xffffffffull || width > 0xffffffffull) return hipErrorInvalidValue;
    const size_t lds = spectrogram_lds_bytes(height);
    const unsigned tiles16 = (unsigned)((width + 15) / 16), tiles8 = (unsigned)((width + 7) / 8);
    (void)hipGetLastError();
#define JST_SPEC_COUNTS(TW, COPIES, TILES)                                                                          \
    do {                                                                                                            \
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(spectrogram_counts_kernel<TW, COPIES>), \
                                               80 * 1024);                                                          \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL((spectrogram_counts_kernel<TW, COPIES>), dim3(TILES), dim3(kThreadsDefault), lds, stream, \
                           counts, in, in_offset, (uint32_t)batches, (uint32_t)width, (uint32_t)height,             \
                           batch_stride, elem_stride);                                                              \
    } while (0)
    if (height <= 256) JST_SPEC_COUNTS(16, 4, tiles16);
    else if (height <= 512) JST_SPEC_COUNTS(16, 2, tiles16);
    else if (height <= 1024) JST_SPEC_COUNTS(16, 1, tiles16);
    else JST_SPEC_COUNTS(8, 1, tiles8);
#undef JST_SPEC_COUNTS
    return hipGetLastError();
}

hipError_t launch_spectrogram_apply_counts(float* bins, const uint32_t* counts, uint64_t cells, float decay,
                                           hipStream_t stream) {
    if (cells == 0) return hipSuccess;
    (void)hipGetLastError();
    const unsigned grid = (unsigned)((cells + 255) / 256 < 4096 ? (cells + 255) / 256 : 4096);
    hipLaunchKernelGGL(spectrogram_apply_counts_kernel, dim3(grid), dim3(256), 0, stream, bins, counts, cells, decay);
    return hipGetLastError();
}

}  // namespace jst::kernels
