// Per-workgroup phase timeline of the LDS-tiled FFT kernels (columns / blocks) at SURVEY config 3's transform
// (100 x 160000 points, pad fused in): wall-clock stamps by thread 0 of every workgroup.  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -DJST_TILED_TIMELINE -I cyberether_amd/csrc/kernels
//         -I cyberether_amd/csrc tools/ubench/tiled_timeline.hip
#include "../../cyberether_amd/csrc/kernels/fft_tiled.hip"

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

namespace jst::kernels {  // the two externals fft_tiled.hip links against
int fft_plan_factors(uint64_t n, uint32_t* fact) {
    int nf = 0;
    uint64_t len = n;
    while ((len & 7) == 0) { fact[nf++] = 8; len >>= 3; }
    while ((len & 3) == 0) { fact[nf++] = 4; len >>= 2; }
    if ((len & 1) == 0) { len >>= 1; fact[nf++] = 2; std::swap(fact[0], fact[nf - 1]); }
    for (uint64_t d = 3; d * d <= len; d += 2)
        while (len % d == 0) { fact[nf++] = (uint32_t)d; len /= d; }
    if (len > 1) fact[nf++] = (uint32_t)len;
    return nf;
}
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
    printf("== %s: %u workgroups, device span %.2f us, mean workgroup lifetime %.2f us\n", name, grid, (w1 - w0) * 0.01,
           life / grid * 0.01);
    printf("   mean phase us: load %.2f |", dur[0] / grid * 0.01);
    for (int p = 0; p < passes; ++p) printf(" pass%d %.2f", p, dur[1 + p] / grid * 0.01);
    printf(" | store/epilogue %.2f\n", dur[1 + passes] / grid * 0.01);
    // concurrency: how many workgroups are alive at the middle of the span
    const unsigned long long mid = (w0 + w1) / 2;
    unsigned alive = 0;
    for (unsigned b = 0; b < grid; ++b) alive += (t[b * 16] <= mid && t[b * 16 + 15] >= mid);
    printf("   workgroups alive at mid-span: %u (%.2f per CU)\n", alive, alive / 256.0);
}

int main() {
    const uint64_t n = 160000, valid = 159750, B = 100;
    TiledPlan P;
    if (!make_tiled_plan(n, B, P)) { printf("no plan\n"); return 1; }
    printf("plan: R1 %u S %u CA %u CB %u g %u nf %u\n", P.R1, P.S, P.CA, P.CB, P.g, P.nf);
    float2 *in, *out, *scratch, *Wp;
    hipMalloc(&in, B * valid * 8); hipMalloc(&out, B * n * 8); hipMalloc(&scratch, B * n * 8);
    std::vector<float2> h(B * valid);
    for (size_t i = 0; i < h.size(); ++i) h[i] = make_float2((float)((i * 7919) % 1000) / 1000.f - 0.5f, (float)((i * 104729) % 1000) / 1000.f - 0.5f);
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<float> w(2 * n);
    for (uint64_t k = 0; k < n; ++k) { const double a = 6.283185307179586 * k / n; w[2 * k] = (float)cos(a); w[2 * k + 1] = (float)sin(a); }
    const uint64_t cnt = fft_pass_twiddle_count(n);
    std::vector<float> pt(2 * cnt);
    fft_pass_twiddle_fill(n, w.data(), pt.data());
    hipMalloc(&Wp, cnt * 8); hipMemcpy(Wp, pt.data(), cnt * 8, hipMemcpyHostToDevice);
    const unsigned gridA = (unsigned)(B * ((P.S + P.CA - 1) / P.CA)), gridB = (unsigned)(B * ((P.R1 + P.CB - 1) / P.CB));
    const unsigned gmax = std::max(gridA, gridB);
    unsigned long long* tl; hipMalloc(&tl, (size_t)gmax * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(jst_tiled_tl), &tl, sizeof(tl));
    FftLayout L{}; L.transforms = B; L.outer_rank = 1; L.outer_shape[0] = B; L.in_outer_stride[0] = (int64_t)valid; L.out_outer_stride[0] = (int64_t)n;
    L.in_axis_stride = 1; L.out_axis_stride = 1;
    const LoadCF32Padded pro{in, (uint32_t)valid};
    const StoreCF32 epi{out};
    auto ka = fft_tile_columns_kernel<true, LoadCF32Padded>;
    auto kb = fft_tile_blocks_kernel<true, LoadCF32Padded, StoreCF32>;
    hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileElems * 8));
    hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileElems * 8));
    const size_t lds_a = (size_t)P.R1 * P.CA * 8, lds_b = (size_t)P.S * (P.CB | 1u) * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<unsigned long long> t((size_t)gmax * 16);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(tl, 0, (size_t)gmax * 16 * 8);
        hipEventRecord(e0);
        ka<<<gridA, threads_for((uint64_t)P.R1 * P.CA), lds_a>>>(L, P, Wp, pro, scratch);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    printf("columns kernel: %.2f us by events, %u threads, %zu B LDS\n", ms * 1e3, threads_for((uint64_t)P.R1 * P.CA), lds_a);
    report("columns", t, gridA, (int)P.g);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(tl, 0, (size_t)gmax * 16 * 8);
        hipEventRecord(e0);
        kb<<<gridB, threads_for((uint64_t)P.S * P.CB), lds_b>>>(L, P, Wp, pro, epi, scratch, gridB);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    printf("blocks kernel: %.2f us by events, %u threads, %zu B LDS\n", ms * 1e3, threads_for((uint64_t)P.S * P.CB), lds_b);
    report("blocks", t, gridB, (int)(P.nf - P.g));
    // the same blocks kernel with the Filter's Multiply -> Fold epilogue (fold 16000, one head), constant plan 8
    {
        const uint64_t F = 16000;
        TiledPlan PF = P;
        if (!plan_fold_groups(PF, F)) { printf("no fold groups\n"); return 1; }
        float2 *hf, *folded;
        hipMalloc(&hf, n * 8); hipMalloc(&folded, B * F * 8);
        hipMemcpy(hf, h.data(), n * 8, hipMemcpyHostToDevice);
        FoldProductEpi fe{folded, hf, 1, (uint32_t)F, (uint32_t)(n / F), 0u, nullptr, 1u, 1u, true, 0, 0, 0, 0, 1u, 0};
        fe.dq = (uint32_t)(F / PF.R1);
        fe.dk = (uint32_t)(F % PF.R1);
        fe.nq = PF.n / PF.R1;
        fe.grp_step = PF.grp_stride ? fe.dk / PF.grp_stride : 0;
        printf("fold plan: grp_w %u grp_stride %u CB %u dq %u dk %u grp_step %u\n", PF.grp_w, PF.grp_stride, PF.CB, fe.dq, fe.dk, fe.grp_step);
        constexpr bool kPersist = persist_eligible(static_plan(8));
        auto kf = fft_tile_blocks_kernel<true, LoadCF32Padded, FoldProductEpi, 8, false, kPersist>;
        hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTileElems * 8));
        const bool is_static = same_plan(PF, static_plan(8));
        printf("constant plan 8 %s the run-time plan\n", is_static ? "equals" : "DIFFERS FROM");
        const unsigned gridF = (unsigned)(B * ((PF.R1 + PF.CB - 1) / PF.CB));
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(tl, 0, (size_t)gmax * 16 * 8);
            hipEventRecord(e0);
            kf<<<kPersist ? persistent_grid((const void*)kf, threads_for((uint64_t)PF.S * PF.CB), (size_t)PF.S * (PF.CB | 1u) * 8 + (size_t)block_twiddle_entries(static_plan(8)) * 8, gridF) : gridF, threads_for((uint64_t)PF.S * PF.CB), (size_t)PF.S * (PF.CB | 1u) * 8 + (size_t)block_twiddle_entries(static_plan(8)) * 8>>>(L, PF, Wp, pro, fe, scratch, gridF);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
        printf("blocks kernel + fold epilogue: %.2f us by events\n", ms * 1e3);
        const int np = (int)(PF.nf - PF.g);
        double acc[16] = {0};
        for (unsigned b = 0; b < gridF; ++b) for (int q = 0; q < 16; ++q) acc[q] += (double)(t[(size_t)b * 16 + q] - t[(size_t)b * 16]) * 0.01 / gridF;
        printf("   mean us since workgroup start: loaded %.2f | last pass %.2f | products in place %.2f | walk %.2f | end %.2f\n",
               acc[1], acc[1 + np], acc[8], acc[9], acc[15]);
        unsigned long long w0 = ~0ull, w1 = 0;
        for (unsigned b = 0; b < gridF; ++b) { w0 = std::min(w0, t[(size_t)b * 16]); w1 = std::max(w1, t[(size_t)b * 16 + 15]); }
        printf("   %u workgroups, device span %.2f us\n", gridF, (w1 - w0) * 0.01);
    }
    return 0;
}
