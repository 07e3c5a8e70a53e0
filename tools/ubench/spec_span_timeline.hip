// spec_span_timeline.hip -- where the cycle-batched Spectrogram kernel's time goes: shader-clock cycles per phase, summed
// over the 16 cycles of a span per wavefront (spectrogram.hip: spectrogram_index_span_kernel, JST_SPAN_TIMELINE).
//   1 rows counted (wait for the rows + rotate + 16 LDS atomics) | 2 barrier | 3 counts read + zeroed | 4 barrier |
//   5 decay + hit update | 6 state stores
#ifndef JST_SPAN_NO_TL  // -DJST_SPAN_NO_TL: the product kernel as it is (timing only, no stamps)
#define JST_SPAN_TIMELINE
#endif
#include "../../cyberether_amd/csrc/kernels/spectrogram.hip"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

namespace jst::kernels {
hipError_t raise_dynamic_lds(const void* kernel, int bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace jst::kernels

int main(int argc, char** argv) {
    const uint32_t B = 1024, N = 4096, H = 256, C = argc > 1 ? atoi(argv[1]) : 16;
    std::vector<uint8_t> idx((size_t)B * N * C);
    std::mt19937 rng(1);
    std::normal_distribution<float> noise(60.0f, 14.0f);  // a noise floor ~14 rows wide, like the bench's
    for (auto& v : idx) {
        const float f = noise(rng);
        v = (uint8_t)(f < 1 ? 0 : (f > 255 ? 255 : f));
    }
    if (argc > 2) {  // real indices (tools/dump_bench_indices.py): as many cycles as the file holds, repeated
        FILE* f = fopen(argv[2], "rb");
        if (f) {
            const size_t got = fread(idx.data(), 1, idx.size(), f);
            fclose(f);
            for (size_t i = got; i < idx.size() && got; ++i) idx[i] = idx[i % got];
            printf("indices from %s (%zu bytes)\n", argv[2], got);
        }
    }
    uint8_t* d_idx;
    float* d_bins;
    unsigned long long* d_tl;
    hipMalloc(&d_idx, idx.size());
    hipMalloc(&d_bins, (size_t)H * N * 4);
    hipMalloc(&d_tl, 256 * 16 * 8 * 8);
    hipMemcpy(d_idx, idx.data(), idx.size(), hipMemcpyHostToDevice);
    hipMemset(d_bins, 0, (size_t)H * N * 4);
#ifdef JST_SPAN_TIMELINE
    jst::kernels::jst_span_tl_host = d_tl;
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) jst::kernels::launch_spectrogram_index_span(d_bins, d_idx, B, B, N, H, 0.36f, C, 0, 0, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) jst::kernels::launch_spectrogram_index_span(d_bins, d_idx, B, B, N, H, 0.36f, C, 0, 0, nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("span of %u cycles, back-to-back launches: %.2f us each = %.2f us per cycle\n", C, ms * 1000 / 50, ms * 1000 / 50 / C);
    if (argc > 3) {  // every launch behind a 768 MiB fill: the index tensors come from HBM, not from the Infinity Cache
        void* big;
        hipMalloc(&big, 768u << 20);
        float sum = 0;
        for (int i = 0; i < 20; ++i) {
            hipMemsetAsync(big, i, 768u << 20, nullptr);
            hipEventRecord(e0);
            jst::kernels::launch_spectrogram_index_span(d_bins, d_idx, B, B, N, H, 0.36f, C, 0, 0, nullptr);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
            sum += ms;
        }
        printf("behind a 768 MiB fill (cold Infinity Cache), event pair per launch: %.2f us = %.2f us per cycle\n", sum * 1000 / 20, sum * 1000 / 20 / C);
    }
#ifndef JST_SPAN_TIMELINE
    return 0;
#endif
    std::vector<unsigned long long> tl(256 * 16 * 8);
    hipMemcpy(tl.data(), d_tl, tl.size() * 8, hipMemcpyDeviceToHost);
    const char* names[8] = {"-", "rows counted (wait+atomics)", "barrier 1", "counts read + zeroed", "barrier 2", "decay + hits", "stores", "-"};
    for (int p = 1; p <= 6; ++p) {
        std::vector<double> v;
        for (int w = 0; w < 256 * 16; ++w) v.push_back((double)tl[w * 8 + p] / C);
        std::sort(v.begin(), v.end());
        printf("  %-30s per cycle (shader clocks): min %8.0f  median %8.0f  p90 %8.0f  max %8.0f\n", names[p], v[0], v[v.size() / 2],
               v[v.size() * 9 / 10], v.back());
    }
    std::vector<double> tot;
    for (int w = 0; w < 256 * 16; ++w) {
        double t = 0;
        for (int p = 1; p <= 6; ++p) t += (double)tl[w * 8 + p];
        tot.push_back(t / C);
    }
    std::sort(tot.begin(), tot.end());
    printf("  wavefront total per cycle: min %.0f median %.0f max %.0f clocks\n", tot[0], tot[tot.size() / 2], tot.back());
    return 0;
}
