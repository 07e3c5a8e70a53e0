// Phase timeline of ONE workgroup of the single-kernel tiled FFT at multi-fm.yml's Filter transform: 8 x 8050 points
// (2*5*5*7*23, 8000 valid samples, pad fused in), fold epilogue with 2 heads and fold 805 -- 8 workgroups in all, so the
// kernel's duration IS a workgroup's lifetime.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -DJST_TILED_TIMELINE -I cyberether_amd/csrc/kernels
//         -I cyberether_amd/csrc -I include tools/ubench/tiled_timeline_8050.hip
#include "../../cyberether_amd/csrc/kernels/fft_tiled.hip"

#include <algorithm>
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

int main() {
    const uint64_t n = 8050, valid = 8000, B = 8, F = 805, heads = 2;
    TiledPlan P;
    if (!make_tiled_plan(n, B, P) || !plan_fold_groups(P, F)) { printf("no plan\n"); return 1; }
    printf("plan: R1 %u S %u CB %u g %u nf %u factors", P.R1, P.S, P.CB, P.g, P.nf);
    for (uint32_t q = 0; q < P.nf; ++q) printf(" %u", P.fact[q]);
    printf("\n");
    float2 *in, *out, *Wp, *hf, *folded;
    hipMalloc(&in, B * valid * 8); hipMalloc(&out, B * n * 8); hipMalloc(&hf, heads * n * 8); hipMalloc(&folded, B * heads * F * 8);
    std::vector<float2> h(B * valid);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_float2((float)((i * 7919) % 1000) / 1000.f - 0.5f, (float)((i * 104729) % 1000) / 1000.f - 0.5f);
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(hf, h.data(), heads * n * 8, hipMemcpyHostToDevice);
    std::vector<float> w(2 * n);
    for (uint64_t k = 0; k < n; ++k) { const double a = 6.283185307179586 * k / n; w[2 * k] = (float)cos(a); w[2 * k + 1] = (float)sin(a); }
    const uint64_t cnt = fft_pass_twiddle_count(n);
    std::vector<float> pt(2 * cnt);
    fft_pass_twiddle_fill(n, w.data(), pt.data());
    hipMalloc(&Wp, cnt * 8); hipMemcpy(Wp, pt.data(), cnt * 8, hipMemcpyHostToDevice);
    const unsigned grid = (unsigned)((B + P.CB - 1) / P.CB);
    unsigned long long* tl; hipMalloc(&tl, (size_t)grid * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(jst_tiled_tl), &tl, sizeof(tl));
    FftLayout L{}; L.transforms = B; L.outer_rank = 1; L.outer_shape[0] = B; L.in_outer_stride[0] = (int64_t)valid; L.out_outer_stride[0] = (int64_t)n;
    L.in_axis_stride = 1; L.out_axis_stride = 1;
    const LoadCF32Padded pro{in, (uint32_t)valid};
    const size_t lds = (size_t)P.S * (P.CB | 1u) * 8;
    const unsigned threads = threads_for((uint64_t)P.S * P.CB, nullptr, min_threads_for_passes(P, P.g, P.nf, (uint64_t)P.S * P.CB));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> t((size_t)grid * 16);
    auto show = [&](const char* name, float ms, bool fold) {
        hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
        printf("== %s: %.2f us by events, %u workgroups x %u threads, %zu B LDS\n", name, ms * 1e3, grid, threads, lds);
        const int np = (int)(P.nf - P.g);
        std::vector<double> acc(16, 0.0);
        for (unsigned b = 0; b < grid; ++b) for (int q = 0; q < 16; ++q) acc[q] += (double)(t[b * 16 + q] - t[b * 16]) * 0.01 / grid;
        printf("   mean us since workgroup start: loaded %.2f |", acc[1]);
        for (int p = 0; p < np; ++p) printf(" pass%d(%u) %.2f", p, P.fact[P.g + p], acc[2 + p]);
        printf(" | generic: H ready %.2f, sums done %.2f", acc[12], acc[13]);
        if (fold) printf(" | fold: head0 walk %.2f, products %.2f, head1 walk %.2f", acc[9], acc[8], acc[10]);
        printf(" | end %.2f\n", acc[15]);
    };
    float ms;
    {
        auto kb = fft_tile_blocks_kernel<true, LoadCF32Padded, StoreCF32, 0, true>;
        hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileElems * 8));
        const StoreCF32 epi{out};
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(tl, 0, (size_t)grid * 16 * 8);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            kb<<<grid, threads, lds>>>(L, P, Wp, pro, epi, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        show("plain store", ms, false);
    }
    {
        auto kb = fft_tile_blocks_kernel<true, LoadCF32Padded, FoldProductEpi, 0, true>;
        hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileElems * 8));
        FoldProductEpi epi{folded, hf, 1, (uint32_t)F, (uint32_t)(n / F), 0u, nullptr, 1u, 1u, true, 0, 0, 0, 0, (uint32_t)heads, (int64_t)n};
        epi.dq = (uint32_t)F;
        epi.nq = P.n;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(tl, 0, (size_t)grid * 16 * 8);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            kb<<<grid, threads, lds>>>(L, P, Wp, pro, epi, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        show("fold epilogue, 2 heads", ms, true);
    }
    return 0;
}
