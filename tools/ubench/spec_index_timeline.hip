// spec_index_timeline.hip -- where the index-fed Spectrogram kernel's 5.2 us go: wall-clock stamps (s_memrealtime,
// 100 MHz) of every workgroup's phases on a 1024 x 4096 cycle, after warm-up launches.
//   0 entry | 1 state + row requests issued | 2 histogram cleared, barrier | 3 this wavefront's atomics issued |
//   4 barrier | 5 counts read | 6 state stores issued
#define JST_SPEC_TIMELINE
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

int main() {
    const uint32_t B = 1024, N = 4096, H = 256;
    std::vector<uint8_t> idx((size_t)B * N);
    std::mt19937 rng(1);
    std::normal_distribution<float> noise(60.0f, 14.0f);  // a noise floor ~14 rows wide, like the bench's
    for (auto& v : idx) {
        const float f = noise(rng);
        v = (uint8_t)(f < 1 ? 0 : (f > 255 ? 255 : f));
    }
    uint8_t* d_idx;
    float* d_bins;
    unsigned long long* d_tl;
    hipMalloc(&d_idx, idx.size());
    hipMalloc(&d_bins, (size_t)H * N * 4);
    hipMalloc(&d_tl, 256 * 8 * 8);
    hipMemcpy(d_idx, idx.data(), idx.size(), hipMemcpyHostToDevice);
    hipMemset(d_bins, 0, (size_t)H * N * 4);
    jst::kernels::jst_spec_tl_host = d_tl;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) jst::kernels::launch_spectrogram_index(d_bins, d_idx, B, B, N, H, 0.36f, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) jst::kernels::launch_spectrogram_index(d_bins, d_idx, B, B, N, H, 0.36f, nullptr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("back-to-back launches: %.2f us each\n", ms * 1000 / 200);
    std::vector<unsigned long long> tl(256 * 8);
    hipMemcpy(tl.data(), d_tl, tl.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t_end = 0;
    for (int w = 0; w < 256; ++w) {
        t0 = std::min(t0, tl[w * 8]);
        t_end = std::max(t_end, tl[w * 8 + 6]);
    }
    printf("first entry -> last stores issued: %.2f us\n", (t_end - t0) * 0.01);
    const char* names[7] = {"entry (after first entry)", "requests issued", "cleared + barrier", "atomics issued (wave 0)",
                            "barrier", "counts read", "stores issued"};
    for (int s = 0; s < 7; ++s) {
        std::vector<double> v;
        for (int w = 0; w < 256; ++w) v.push_back(s == 0 ? (tl[w * 8] - t0) * 0.01 : (tl[w * 8 + s] - tl[w * 8 + s - 1]) * 0.01);
        std::sort(v.begin(), v.end());
        printf("  %-28s min %.2f  median %.2f  p90 %.2f  max %.2f us\n", names[s], v[0], v[128], v[230], v[255]);
    }
    std::vector<double> life;
    for (int w = 0; w < 256; ++w) life.push_back((tl[w * 8 + 6] - tl[w * 8]) * 0.01);
    std::sort(life.begin(), life.end());
    printf("  workgroup lifetime           min %.2f  median %.2f  max %.2f us\n", life[0], life[128], life[255]);
    return 0;
}
