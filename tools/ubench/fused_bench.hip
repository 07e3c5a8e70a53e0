// A/B harness for the fused 4096-point spectrum kernel (fft_lds.hh) outside the runtime: the bench workload
// (1024 x 4096 cf32, tone + AWGN per row, Blackman x (-1)^n window, range -100..0 dB) over a ring of 16 input
// slots (512 MiB > Infinity Cache), median/mean launch time by hipEvents and an order-independent checksum of
// the output bits so that variants compiled with different -D switches can be compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off [-DJST_...] -I cyberether_amd/csrc/kernels \
//         -I cyberether_amd/csrc tools/ubench/fused_bench.hip -o fused_bench_<variant>
// Diagnostic only; the product path is cyberether_amd/lib/libjetstream_hip.so.
#include "fft_lds_r03_variants.hh"  // the round-3 header with its A/B switches (the product header dropped them)
#ifdef FB_SPLIT
#include "fft_split_experiment.hh"
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
using namespace jst::dev;

#ifndef FB_FAST
#define FB_FAST false
#endif

#ifdef FB_TRIVIAL_EPI  // decomposition experiment: the FFT with a two-multiply epilogue
struct StorePower {
    float* out;
    static constexpr uint32_t kElemBytes = 4;
    __device__ __forceinline__ const void* row(int64_t base) const { return out + base; }
    __device__ __forceinline__ float value(float2 v) const { return v.x * v.x + v.y * v.y; }
    __device__ __forceinline__ void store_buf(rsrc_t r, uint32_t voff, uint32_t soff, float2 v) const {
        buf_store_f1(r, voff, soff, value(v));
    }
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t axis_stride, int pos, float2 v) const {
        out[base + (int64_t)pos * axis_stride] = v.x * v.x + v.y * v.y;
    }
};
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void checksum_kernel(const uint32_t* v, uint64_t n, unsigned long long* acc) {
    unsigned long long s = 0, x = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = v[i];
        s += (unsigned long long)b * (2654435761ull + (i & 0xffff));
        x ^= (unsigned long long)b << (i & 31);
    }
    atomicAdd(&acc[0], s);
    atomicXor(&acc[1], x);
}

static uint32_t lcg_state = 12345u;
static inline float urand() { lcg_state = lcg_state * 1664525u + 1013904223u; return ((lcg_state >> 8) + 0.5f) * (1.0f / 16777216.0f); }
static inline float gauss() { const float u1 = urand(), u2 = urand(); return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2); }

int main(int argc, char** argv) {
    constexpr int N = 4096;
    const uint64_t B = 1024;
    const int SLOTS = 16;
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    const char* name = argc > 2 ? argv[2] : "variant";
    const int mode = argc > 3 ? atoi(argv[3]) : 0;  // 0: bench data, 1: mid-range data (every tanh class), 2: zeros
#ifdef FB_WG
    const int grid = 1024;
#elif defined(FB_SPLIT)
    const int grid = argc > 4 ? atoi(argv[4]) : 256;
#else
    const int grid = argc > 4 ? atoi(argv[4]) : 512;
#endif
    const int guard = argc > 5 ? atoi(argv[5]) : 1;
    (void)guard;

    float2 *in, *win, *W; float* out; unsigned long long* acc;
    CK(hipMalloc(&in, (size_t)SLOTS * B * N * 8)); CK(hipMalloc(&win, N * 8)); CK(hipMalloc(&W, N * 8));
    CK(hipMalloc(&out, B * N * 4)); CK(hipMalloc(&acc, 16)); CK(hipMemset(acc, 0, 16));
    std::vector<float2> h(N);
    for (int k = 0; k < N; ++k) {  // pocketfft-style table is built by the product; here plain double cos/sin (timing + A/B only)
        const double a = 6.283185307179586476925286766559 * k / N;
        h[k] = make_float2((float)cos(a), (float)sin(a));
    }
    CK(hipMemcpy(W, h.data(), N * 8, hipMemcpyHostToDevice));
    for (int i = 0; i < N; ++i) {
        const double w = 0.42 - 0.5 * cos(6.283185307179586 * i / (N - 1)) + 0.08 * cos(2 * 6.283185307179586 * i / (N - 1));
        h[i] = make_float2((float)((i & 1) ? -w : w), 0.0f);
    }
    CK(hipMemcpy(win, h.data(), N * 8, hipMemcpyHostToDevice));
    {
        std::vector<float2> hin(B * N);
        for (int s = 0; s < SLOTS; ++s) {
            const float sigma = mode == 1 ? 0.3f : 1e-3f;
            for (uint64_t b = 0; b < B; ++b) {
                const double f = fmod(100.25 + (double)b + s, (double)N) / N;
                for (int n = 0; n < N; ++n) {
                    const double ph = 6.283185307179586 * f * n;
                    float re = (float)cos(ph) + sigma * gauss(), im = (float)sin(ph) + sigma * gauss();
                    if (mode == 2) re = im = 0.0f;
                    hin[b * N + n] = make_float2(re, im);
                }
            }
            CK(hipMemcpy(in + (size_t)s * B * N, hin.data(), B * N * 8, hipMemcpyHostToDevice));
        }
    }
    FftLayout L{};
    L.transforms = B; L.outer_rank = 1; L.outer_shape[0] = B; L.in_outer_stride[0] = N; L.out_outer_stride[0] = N;
    L.in_axis_stride = 1; L.out_axis_stride = 1;
    [[maybe_unused]] const float coeff = 20.0f * log10f(1.0f / (float)N);
    [[maybe_unused]] const float scale = 1.0f / (0.0f - (-100.0f)), offset = 100.0f * scale;
#ifdef FB_TRIVIAL_EPI
    using Epi = StorePower;
#else
    using Epi = StoreAmplitudeRangeT<FB_FAST>;
#endif
#ifdef FB_WG  // one transform per workgroup, 8 wavefronts per SIMD (fft_wg_kernel)
    auto k = fft_wg_kernel<N, true, true, LoadCF32TimesWindow, Epi>;
    const size_t lds = (size_t)lds_elems(N) * sizeof(float2);
#elif defined(FB_SPLIT)  // role-split kernel: 16 wavefronts per workgroup, FFT role + epilogue role
    auto k = fft_split_kernel<N, true, LoadCF32TimesWindow, Epi>;
    const size_t lds = fft_split_lds_bytes(N);
#else
    auto k = fft_pipe_kernel<N, true, true, LoadCF32TimesWindow, Epi>;
    const size_t lds = fft_pipe_lds_bytes(N);
#endif
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<hipEvent_t> ev(2 * reps);
    for (auto& e : ev) CK(hipEventCreate(&e));
    auto launch = [&](int slot) {
        LoadCF32TimesWindow pro{in + (size_t)slot * B * N, win, 1};
#ifdef FB_TRIVIAL_EPI
        Epi epi{out};
#else
        Epi epi{out, coeff, scale, offset, BinGuard{(FB_FAST && guard) ? 256.0f : 0.0f, 0.0f}};
#endif
#ifdef FB_SPLIT
        k<<<grid, N / 4, lds, st>>>(L, W, pro, epi);
#else
        k<<<grid, N / 8, lds, st>>>(L, W, pro, epi);
#endif
    };
    for (int i = 0; i < 20; ++i) launch(i % SLOTS);
    CK(hipStreamSynchronize(st));
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(ev[2 * i], st));
        launch(i % SLOTS);
        CK(hipEventRecord(ev[2 * i + 1], st));
    }
    CK(hipStreamSynchronize(st));
    std::vector<float> us(reps);
    double mean = 0;
    for (int i = 0; i < reps; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); us[i] = ms * 1e3f; mean += us[i]; }
    mean /= reps;
    std::sort(us.begin(), us.end());
    // back-to-back launches without events in between: the rate the graph replays see
    hipEvent_t a, b2; CK(hipEventCreate(&a)); CK(hipEventCreate(&b2));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) launch(i % SLOTS);
    CK(hipEventRecord(b2, st)); CK(hipStreamSynchronize(st));
    float msbb; CK(hipEventElapsedTime(&msbb, a, b2));
    // two (three) streams, alternating launches with no dependencies between them: does the tail of one launch overlap
    // the ramp of the next when they sit on different hardware queues?  (outputs of their own; host clock)
    for (int ns : {2, 3}) {
        hipStream_t ss[3] = {st, nullptr, nullptr};
        float* outs[3] = {out, nullptr, nullptr};
        for (int q = 1; q < ns; ++q) { CK(hipStreamCreate(&ss[q])); CK(hipMalloc(&outs[q], B * N * 4)); }
        auto launch_on = [&](int slot, int q) {
            LoadCF32TimesWindow pro{in + (size_t)slot * B * N, win, 1};
#ifdef FB_TRIVIAL_EPI
            Epi epi{outs[q]};
#else
            Epi epi{outs[q], coeff, scale, offset, BinGuard{(FB_FAST && guard) ? 256.0f : 0.0f, 0.0f}};
#endif
#ifdef FB_SPLIT
            k<<<grid, N / 4, lds, ss[q]>>>(L, W, pro, epi);
#else
            k<<<grid, N / 8, lds, ss[q]>>>(L, W, pro, epi);
#endif
        };
        for (int i = 0; i < 30; ++i) launch_on(i % SLOTS, i % ns);
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps * 4; ++i) launch_on(i % SLOTS, i % ns);
        CK(hipDeviceSynchronize());
        const double us_total = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%-28s %d streams alternating, no dependencies: %.2f us per launch (host clock, %d launches)\n", name, ns,
               us_total / (reps * 4), reps * 4);
    }
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps * 4; ++i) launch(i % SLOTS);
        CK(hipDeviceSynchronize());
        const double us_total = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%-28s 1 stream: %.2f us per launch (host clock, %d launches)\n", name, us_total / (reps * 4), reps * 4);
    }
    // checksum of the output for slot 3 (fixed), order independent
    launch(3); CK(hipStreamSynchronize(st));
    checksum_kernel<<<1024, 256, 0, st>>>((const uint32_t*)out, B * N, acc);
    unsigned long long hacc[2];
    CK(hipMemcpy(hacc, acc, 16, hipMemcpyDeviceToHost));
    printf("%-28s mode %d grid %d  events: median %.2f us  min %.2f  mean %.2f | back-to-back %.2f us/launch | checksum %016llx %016llx\n",
           name, mode, grid, us[reps / 2], us[0], mean, msbb * 1e3 / reps, hacc[0], hacc[1]);
    return 0;
}
