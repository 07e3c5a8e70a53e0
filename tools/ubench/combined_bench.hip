// Feasibility experiment: ONE launch that carries the 4096-point fused spectrum kernel's 512 workgroups (cycle k) AND
// the spectrogram's 256 tiles of cycle k-1 as 512-thread workgroups behind them (blockIdx >= 512): the spectrogram
// workgroups are dispatched as the first spectrum workgroups of each CU retire, i.e. into the launch's tail.
// Compared with the two kernels launched one after the other.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I cyberether_amd/csrc/kernels -I cyberether_amd/csrc
//         tools/ubench/combined_bench.hip
#include "../../cyberether_amd/csrc/kernels/spectrogram.hip"
#include "fft_lds_r03_variants.hh"  // the round-3 header with its A/B switches (the product header dropped them)

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace jst::kernels {
hipError_t raise_dynamic_lds(const void* kernel, int bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace jst::kernels

using namespace jst::dev;
using namespace jst::kernels;

#ifndef CB_FAST
#define CB_FAST false
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int N = 4096;
using Epi = StoreAmplitudeRangeT<CB_FAST>;

template <int SPEC_THREADS, int SPEC_DEPTH>
__global__ __launch_bounds__(512, 4) void combined_kernel(const FftLayout L, const float2* __restrict__ W,
                                                          const LoadCF32TimesWindow pro, const Epi epi,
                                                          float* __restrict__ bins, const float* __restrict__ spec_in,
                                                          uint32_t batches, uint32_t width, uint32_t height, float decay) {
#ifdef CB_SPEC_FIRST  // spectrogram tiles take the first slot of every CU: they overlap the spectrum kernel's start-up
    if (blockIdx.x >= 256u) {
        fft_pipe_body<N, true, true, LoadCF32TimesWindow, Epi>(L, W, pro, epi, blockIdx.x - 256u, 512u);
    } else {
        spectrogram_body<16, 4, SPEC_THREADS, SPEC_DEPTH, true>(bins, spec_in, 0, batches, width, height, (int64_t)width, 1,
                                                              decay, blockIdx.x, 256u);
    }
#else
    if (blockIdx.x < 512u) {
        fft_pipe_body<N, true, true, LoadCF32TimesWindow, Epi>(L, W, pro, epi, blockIdx.x, 512u);
    } else {
        spectrogram_body<16, 4, SPEC_THREADS, SPEC_DEPTH, true>(bins, spec_in, 0, batches, width, height, (int64_t)width, 1,
                                                              decay, blockIdx.x - 512u, 256u);
    }
#endif
}

static float gauss() {
    static unsigned long long s = 88172645463325252ull;
    double u[2];
    for (int i = 0; i < 2; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; u[i] = ((s >> 11) + 0.5) / 9007199254740992.0; }
    return (float)(sqrt(-2.0 * log(u[0])) * cos(6.283185307179586 * u[1]));
}

int main(int argc, char** argv) {
    const uint64_t B = 1024;
    const int SLOTS = 8, H = 256;
    const int reps = argc > 1 ? atoi(argv[1]) : 400;
    float2 *in, *win, *W; float* out[2]; float* bins[2];
    CK(hipMalloc(&in, (size_t)SLOTS * B * N * 8)); CK(hipMalloc(&win, N * 8)); CK(hipMalloc(&W, N * 8));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&out[i], B * N * 4)); CK(hipMalloc(&bins[i], (size_t)N * H * 4)); CK(hipMemset(bins[i], 0, (size_t)N * H * 4)); }
    std::vector<float2> h(N);
    for (int k = 0; k < N; ++k) { const double a = 6.283185307179586476925286766559 * k / N; h[k] = make_float2((float)cos(a), (float)sin(a)); }
    CK(hipMemcpy(W, h.data(), N * 8, hipMemcpyHostToDevice));
    for (int i = 0; i < N; ++i) {
        const double w = 0.42 - 0.5 * cos(6.283185307179586 * i / (N - 1)) + 0.08 * cos(2 * 6.283185307179586 * i / (N - 1));
        h[i] = make_float2((float)((i & 1) ? -w : w), 0.0f);
    }
    CK(hipMemcpy(win, h.data(), N * 8, hipMemcpyHostToDevice));
    {
        std::vector<float2> hin(B * N);
        for (int s = 0; s < SLOTS; ++s) {
            for (uint64_t b = 0; b < B; ++b) {
                const double f = fmod(100.25 + (double)b + s, (double)N) / N;
                for (int n = 0; n < N; ++n) {
                    const double ph = 6.283185307179586 * f * n;
                    hin[b * N + n] = make_float2((float)cos(ph) + 1e-3f * gauss(), (float)sin(ph) + 1e-3f * gauss());
                }
            }
            CK(hipMemcpy(in + (size_t)s * B * N, hin.data(), B * N * 8, hipMemcpyHostToDevice));
        }
    }
    FftLayout L{};
    L.transforms = B; L.outer_rank = 1; L.outer_shape[0] = B; L.in_outer_stride[0] = N; L.out_outer_stride[0] = N;
    L.in_axis_stride = 1; L.out_axis_stride = 1;
    const float coeff = 20.0f * log10f(1.0f / (float)N), scale = 1.0f / 100.0f, offset = 100.0f * scale;
    const float decay = powf(0.999f, (float)B);
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t lds_f = fft_pipe_lds_bytes(N), lds_s = spectrogram_lds_bytes(H), lds_c = std::max(lds_f, lds_s);
    auto kf = fft_pipe_kernel<N, true, true, LoadCF32TimesWindow, Epi>;
    auto ks = spectrogram_kernel<16, 4, 1024, 16, true>;
    auto kc = combined_kernel<512, 32>;
    CK(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
    CK(hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
    auto epi_of = [&](int k) { return Epi{out[k & 1], coeff, scale, offset, BinGuard{CB_FAST ? 256.0f : 0.0f, 0.0f}}; };
    auto separate = [&](int k) {  // cycle k: spectrum, then its spectrogram
        LoadCF32TimesWindow pro{in + (size_t)(k % SLOTS) * B * N, win, 1};
        kf<<<512, 512, lds_f, st>>>(L, W, pro, epi_of(k));
        ks<<<256, 1024, lds_s, st>>>(bins[0], out[k & 1], 0, (uint32_t)B, (uint32_t)N, (uint32_t)H, (int64_t)N, 1, decay);
    };
    auto combined = [&](int k) {  // launch k: spectrum of cycle k + spectrogram of cycle k - 1
        LoadCF32TimesWindow pro{in + (size_t)(k % SLOTS) * B * N, win, 1};
        kc<<<768, 512, lds_c, st>>>(L, W, pro, epi_of(k), bins[1], out[(k + 1) & 1], (uint32_t)B, (uint32_t)N, (uint32_t)H, decay);
    };
    // same sequence of cycles through both forms: cycles 0 .. reps-1 (+ the trailing spectrogram of the last cycle)
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(bins[0], 0, (size_t)N * H * 4)); CK(hipMemset(bins[1], 0, (size_t)N * H * 4));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < reps; ++k) separate(k);
        CK(hipDeviceSynchronize());
        const double us_sep = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        // combined: launch 0 has no previous cycle: run the spectrum alone, then reps - 1 combined launches, then the tail
        t0 = std::chrono::steady_clock::now();
        {
            LoadCF32TimesWindow pro{in, win, 1};
            kf<<<512, 512, lds_f, st>>>(L, W, pro, epi_of(0));
        }
        for (int k = 1; k < reps; ++k) combined(k);
        ks<<<256, 1024, lds_s, st>>>(bins[1], out[(reps - 1) & 1], 0, (uint32_t)B, (uint32_t)N, (uint32_t)H, (int64_t)N, 1, decay);
        CK(hipDeviceSynchronize());
        const double us_comb = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        // the two spectrogram states must be identical
        std::vector<float> b0((size_t)N * H), b1((size_t)N * H);
        CK(hipMemcpy(b0.data(), bins[0], b0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b1.data(), bins[1], b1.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0; double sum = 0;
        for (size_t i = 0; i < b0.size(); ++i) { diff += (b0[i] != b1[i]); sum += b0[i]; }
        printf("%s: separate %.2f us per cycle, combined %.2f us per cycle (%d cycles); spectrogram states differ in %zu cells (sum %.1f)\n",
               CB_FAST ? "fast" : "exact", us_sep, us_comb, reps, diff, sum);
    }
    // the same comparison under hipGraph replay (16 cycles per graph, like the runtime's period graphs)
    {
        hipGraph_t g[2]; hipGraphExec_t ge[2];
        for (int v = 0; v < 2; ++v) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < 16; ++k) { if (v == 0) separate(k); else combined(k); }
            CK(hipStreamEndCapture(st, &g[v]));
            CK(hipGraphInstantiate(&ge[v], g[v], nullptr, nullptr, 0));
        }
        for (int rep = 0; rep < 3; ++rep) {
            double us[2];
            for (int v = 0; v < 2; ++v) {
                for (int i = 0; i < 4; ++i) CK(hipGraphLaunch(ge[v], st));
                CK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 40; ++i) CK(hipGraphLaunch(ge[v], st));
                CK(hipDeviceSynchronize());
                us[v] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (40 * 16);
            }
            printf("%s, graph replay: separate %.2f us per cycle, combined %.2f us per cycle\n", CB_FAST ? "fast" : "exact", us[0], us[1]);
        }
    }
    return 0;
}
