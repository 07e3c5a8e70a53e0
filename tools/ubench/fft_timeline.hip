// Per-workgroup phase timeline of the pipelined fused spectrum kernel (s_memtime stamps by
// thread 0).  Diagnostic only.  hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off
//   -DJST_FFT_TIMELINE -I cyberether_amd/csrc/kernels tools/ubench/fft_timeline.hip
#include "fft_lds_r03_variants.hh"  // the round-3 header with its A/B switches (the product header dropped them)
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace jst::dev;

template <bool FAST>
void run(const char* name) {
    constexpr int N = 4096; const uint64_t B = 1024;
    float2 *in, *win, *W; float* out; unsigned long long* tl;
    hipMalloc(&in, B * N * 8); hipMalloc(&win, N * 8); hipMalloc(&W, N * 8); hipMalloc(&out, B * N * 4);
    hipMalloc(&tl, 512 * 64 * 8); hipMemset(tl, 0, 512 * 64 * 8);
    std::vector<float2> h(N);
    for (int i = 0; i < N; ++i) h[i] = make_float2(cosf(6.283185307f * i / N), sinf(6.283185307f * i / N));
    hipMemcpy(W, h.data(), N * 8, hipMemcpyHostToDevice);
    for (int i = 0; i < N; ++i) h[i] = make_float2((i & 1) ? -0.5f : 0.5f, (i & 1) ? -0.0f : 0.0f);
    hipMemcpy(win, h.data(), N * 8, hipMemcpyHostToDevice);
    std::vector<float2> hin(B * N);
    for (size_t i = 0; i < hin.size(); ++i) hin[i] = make_float2((float)((i * 7919) % 1000) / 1000.f - 0.5f, (float)((i * 104729) % 1000) / 1000.f - 0.5f);
    hipMemcpy(in, hin.data(), B * N * 8, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(jst_tl_base), &tl, sizeof(tl));
    FftLayout L{}; L.transforms = B; L.outer_rank = 1; L.outer_shape[0] = B; L.in_outer_stride[0] = N; L.out_outer_stride[0] = N; L.in_axis_stride = 1; L.out_axis_stride = 1;
    LoadCF32TimesWindow pro{in, win, 1};
    StoreAmplitudeRangeT<FAST> epi{out, -72.0f, 0.01f, 1.0f};
    auto k = fft_pipe_kernel<N, true, true, LoadCF32TimesWindow, StoreAmplitudeRangeT<FAST>>;
    const size_t lds = fft_pipe_lds_bytes(N);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<<<512, 512, lds>>>(L, W, pro, epi);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> t(512 * 64);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < 512; ++b) { t0 = std::min(t0, t[b * 64]); for (int s = 0; s < 32; ++s) t1 = std::max(t1, t[b * 64 + s]); }
    printf("== %s: kernel %.2f us by events\n", name, ms * 1e3);
    {   // calibrate clock64 against the constant 100 MHz wall clock over workgroup 0's lifetime
        const double ticks = (double)(t[0 * 64 + 16 + 8] - t[0]);
        const double wall = (double)(t[0 * 64 + 63] - t[0 * 64 + 62]) * 10.0;  // ns
        printf("   wg0 lifetime: %.0f clock64 ticks = %.0f ns by wall_clock64 -> %.3f ns/tick\n", ticks, wall, wall / ticks);
    }
    {
        unsigned long long w0 = ~0ull, w1 = 0, lastStart = 0, firstEnd = ~0ull;
        for (int b = 0; b < 512; ++b) {
            w0 = std::min(w0, t[b * 64 + 62]); lastStart = std::max(lastStart, t[b * 64 + 62]);
            w1 = std::max(w1, t[b * 64 + 63]); firstEnd = std::min(firstEnd, t[b * 64 + 63]);
        }
        printf("   device span first-start..last-end = %.2f us; starts spread over %.2f us; ends spread over %.2f us\n",
               (w1 - w0) * 0.01, (lastStart - w0) * 0.01, (w1 - firstEnd) * 0.01);
    }
    const char* names[] = {"iter start", "input ready+window", "pass0 done", "bar0", "pass1 done", "bar1", "pass2 done", "bar2", "pass3+epilogue done"};
    for (int b : {0, 1, 255, 256, 511}) {
        for (int it = 0; it < 2; ++it) {
            printf("  wg %3d transform %d:", b, it);
            for (int s = 0; s < 9; ++s) printf(" %6llu", t[b * 64 + it * 16 + s] - t0);
            printf("\n");
        }
    }
    // averages over all workgroups: duration of each phase
    for (int it = 0; it < 2; ++it) {
        printf("  mean phase ticks, transform %d:", it);
        for (int s = 1; s < 9; ++s) { double a = 0; for (int b = 0; b < 512; ++b) a += (double)(t[b * 64 + it * 16 + s] - t[b * 64 + it * 16 + s - 1]); printf(" %s=%.0f", names[s], a / 512); }
        printf("\n");
    }
}
int main() { run<true>("fast"); run<false>("exact"); return 0; }
