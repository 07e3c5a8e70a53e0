// Per-workgroup phase timeline of the LDS-tiled FFT kernels at BASELINE config 5 (B x 65536 points, window fused in, provider fast
// epilogue): wall-clock stamps by thread 0 of every workgroup, for several static plans (lane counts).  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -DJST_TILED_TIMELINE -I cyberether_amd/csrc/kernels
//         -I cyberether_amd/csrc tools/ubench/tiled_timeline_c5.hip -o tools/ubench/bin/tiled_timeline_c5
#include "../../cyberether_amd/csrc/kernels/fft_tiled.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

namespace jst::kernels {  // the two externals fft_tiled.hip links against
int fft_plan_factors(uint64_t n, uint32_t* fact) { return plan_factors_ce(n, fact); }
hipError_t raise_dynamic_lds(const void* kernel, int bytes) {
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace jst::kernels

using namespace jst::kernels;
using namespace jst::dev;

static void report(const char* name, const std::vector<unsigned long long>& t, unsigned grid, int passes) {
    unsigned long long w0 = ~0ull, w1 = 0;
    for (unsigned b = 0; b < grid; ++b) { w0 = std::min(w0, t[b * 16]); w1 = std::max(w1, t[b * 16 + 15]); }
    std::vector<double> dur(2 + passes + 1, 0.0);
    double life = 0;
    for (unsigned b = 0; b < grid; ++b) {
        const unsigned long long* s = &t[b * 16];
        dur[0] += (double)(s[1] - s[0]);
        for (int p = 0; p < passes; ++p) dur[1 + p] += (double)(s[2 + p] - s[1 + p]);
        dur[1 + passes] += (double)(s[15] - s[1 + passes]);
        life += (double)(s[15] - s[0]);
    }
    printf("   %s: %u workgroups, device span %.2f us, mean workgroup lifetime %.2f us | load %.2f |", name, grid, (w1 - w0) * 0.01,
           life / grid * 0.01, dur[0] / grid * 0.01);
    for (int p = 0; p < passes; ++p) printf(" pass%d %.2f", p, dur[1 + p] / grid * 0.01);
    printf(" | store/epilogue %.2f", dur[1 + passes] / grid * 0.01);
    const unsigned long long mid = (w0 + w1) / 2;
    unsigned alive = 0;
    for (unsigned b = 0; b < grid; ++b) alive += (t[b * 16] <= mid && t[b * 16 + 15] >= mid);
    printf(" | alive at mid-span %.2f per CU\n", alive / 256.0);
}

template <int SP>
static void run(uint64_t B, float2* in, float2* win, float* out, float2* scratch, float2* Wp, unsigned long long* tl) {
    const uint64_t n = 65536;
    constexpr TiledPlan P = static_plan(SP);
    printf("== SP %d, %llu transforms: R1 %u S %u CA %u CB %u g %u nf %u\n", SP, (unsigned long long)B, P.R1, P.S, P.CA, P.CB, P.g, P.nf);
    const unsigned gridA = (unsigned)(B * ((P.S + P.CA - 1) / P.CA)), gridB = (unsigned)(B * ((P.R1 + P.CB - 1) / P.CB));
    const unsigned gmax = std::max(gridA, gridB);
    FftLayout L{}; L.transforms = B; L.outer_rank = 1; L.outer_shape[0] = B; L.in_outer_stride[0] = (int64_t)n; L.out_outer_stride[0] = (int64_t)n;
    L.in_axis_stride = 1; L.out_axis_stride = 1;
    const LoadCF32TimesWindow pro{in, win, 1};
    const StoreAmplitudeRangeT<true> epi{out, -96.3f, 0.01f, 1.0f, BinGuard{}};
    auto ka = fft_tile_columns_kernel<true, LoadCF32TimesWindow, SP>;
    constexpr bool kPersist = persist_eligible(P);
    auto kb = fft_tile_blocks_kernel<true, LoadCF32TimesWindow, StoreAmplitudeRangeT<true>, SP, false, kPersist>;
    hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileElems * 8));
    hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileElems * 8));
    const size_t lds_a = (size_t)P.R1 * P.CA * 8, lds_b = (size_t)P.S * (P.CB | 1u) * 8 + (size_t)block_twiddle_entries(P) * 8;
    const unsigned ta = threads_for((uint64_t)P.R1 * P.CA, min_threads_for_passes(P, 0, P.g, (uint64_t)P.R1 * P.CA));
    const unsigned tb = threads_for((uint64_t)P.S * P.CB, min_threads_for_passes(P, P.g, P.nf, (uint64_t)P.S * P.CB));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> t((size_t)gmax * 16);
    float ms;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(tl, 0, (size_t)gmax * 16 * 8);
        hipEventRecord(e0);
        ka<<<gridA, ta, lds_a>>>(L, P, Wp, pro, scratch);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    printf("   columns kernel: %.2f us by events, %u threads, %zu B LDS\n", ms * 1e3, ta, lds_a);
    report("columns", t, gridA, (int)P.g);
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(tl, 0, (size_t)gmax * 16 * 8);
        hipEventRecord(e0);
        kb<<<kPersist ? persistent_grid((const void*)kb, tb, lds_b, gridB) : gridB, tb, lds_b>>>(L, P, Wp, pro, epi, scratch, gridB);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    const unsigned gb = kPersist ? persistent_grid((const void*)kb, tb, lds_b, gridB) : gridB;
    printf("   blocks kernel: %.2f us by events, %u threads, %zu B LDS, %u workgroups for %u tiles (timeline: the LAST tile of each)\n", ms * 1e3, tb, lds_b, gb, gridB);
    report("blocks", t, gb, (int)(P.nf - P.g));
    if (kPersist && gb < gridB) {
        double tops[3] = {0, 0, 0}; unsigned n3 = 0;
        for (unsigned b = 0; b < gb; ++b) { const unsigned long long* s = &t[(size_t)b * 16]; if (s[8] > s[7] && s[7] > s[6]) { ++n3; tops[0] += (double)(s[6] - s[0]); tops[1] += (double)(s[7] - s[6]); tops[2] += (double)(s[8] - s[7]); } }
        if (n3) printf("   workgroups with three tiles (%u): start -> top of tile 0 %.2f us | tile 0 %.2f | tile 1 %.2f\n", n3, tops[0] / n3 * 0.01, tops[1] / n3 * 0.01, tops[2] / n3 * 0.01);
        double a[5] = {0, 0, 0, 0, 0};
        for (unsigned b = 0; b < gb; ++b) {
            const unsigned long long* s = &t[(size_t)b * 16];
            a[0] += (double)(s[12] - s[11]); a[1] += (double)(s[14] - s[12]); a[2] += (double)(s[1] - s[14]);
            a[3] += (double)(s[5] - s[1]); a[4] += (double)(s[11] - s[0]);
        }
        printf("   last tile, us: bases %.2f | wait + commit %.2f | barrier %.2f | passes + epilogue %.2f || kernel start -> top of the last tile %.2f\n",
               a[0] / gb * 0.01, a[1] / gb * 0.01, a[2] / gb * 0.01, a[3] / gb * 0.01, a[4] / gb * 0.01);
    }
}

int main() {
    const uint64_t n = 65536, Bmax = 128;
    float2 *in, *win, *scratch, *Wp;
    float* out;
    hipMalloc(&in, Bmax * n * 8); hipMalloc(&win, n * 8); hipMalloc(&out, Bmax * n * 4); hipMalloc(&scratch, Bmax * n * 8);
    std::vector<float2> h(Bmax * n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_float2((float)((i * 7919) % 1000) / 1000.f - 0.5f, (float)((i * 104729) % 1000) / 1000.f - 0.5f);
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (size_t i = 0; i < n; ++i) h[i] = make_float2(0.5f + 0.5f * (float)cos(6.283185307179586 * i / n), 0.0f);
    hipMemcpy(win, h.data(), n * 8, hipMemcpyHostToDevice);
    std::vector<float> w(2 * n);
    for (uint64_t k = 0; k < n; ++k) { const double a = 6.283185307179586 * k / n; w[2 * k] = (float)cos(a); w[2 * k + 1] = (float)sin(a); }
    const uint64_t cnt = fft_pass_twiddle_count(n);
    std::vector<float> pt(2 * cnt);
    fft_pass_twiddle_fill(n, w.data(), pt.data());
    hipMalloc(&Wp, cnt * 8); hipMemcpy(Wp, pt.data(), cnt * 8, hipMemcpyHostToDevice);
    unsigned long long* tl; hipMalloc(&tl, (size_t)Bmax * 1024 * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(jst_tiled_tl), &tl, sizeof(tl));
    for (uint64_t B : {128ull, 16ull}) {
        run<4>(B, in, win, out, scratch, Wp, tl);   // CA 16, CB 8: the product's plan
        run<5>(B, in, win, out, scratch, Wp, tl);   // CA 32
        run<6>(B, in, win, out, scratch, Wp, tl);   // CA 8, CB 16
    }
    return 0;
}
