// A/B of the two fused 4096-point spectrum kernels at the bench line's launch size: fft_pipe_kernel (fft_lds.hh, 512 threads,
// 2 workgroups per CU) against fft_quad_kernel (fft_quad.hh, 256 threads, in-place exchange, rows by LDS-DMA, 4 workgroups per CU), both with
// the product's functors (real window operand resident, provider fast, row-index side output) and cache policies.
// Bit-compares every F32 value and every index byte of the two, then times them alternately over a 16-slot ring.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off [-DJST_QUAD_...] -I cyberether_amd/csrc/kernels \
//         -I cyberether_amd/csrc -I include tools/ubench/quad_bench.hip -o tools/ubench/bin/quad_bench_<variant>
// Diagnostic only; the product path is cyberether_amd/lib/libjetstream_hip.so.
#define JST_LOAD_AUX 2
#define JST_STORE_AUX 18
#define JST_SIDE_STORE_AUX 16
#define JST_OPND_RESIDENT 1
#include "fft_quad.hh"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jst::dev;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef QB_FAST
#define QB_FAST true
#endif


__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, uint64_t n, unsigned long long* acc, unsigned long long* first) {
    unsigned long long d = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (a[i] != b[i]) { ++d; atomicMin(first, (unsigned long long)i); }
    if (d) atomicAdd(acc, d);
}

static uint32_t lcg_state = 12345u;
static inline float urand() { lcg_state = lcg_state * 1664525u + 1013904223u; return ((lcg_state >> 8) + 0.5f) * (1.0f / 16777216.0f); }
static inline float gauss() { const float u1 = urand(), u2 = urand(); return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2); }

int main(int argc, char** argv) {
    constexpr int N = 4096;
    const uint64_t batches = 1024, H = 256;
    const int cycles = argc > 1 ? atoi(argv[1]) : 16;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const char* name = argc > 3 ? argv[3] : "quad";
    const uint64_t B = batches * cycles;
    const uint32_t pitch = (uint32_t)batches + 2;

    float2 *in, *win, *W; float *out_a, *out_b; uint8_t *side_a, *side_b; unsigned long long* acc;
    CK(hipMalloc(&in, B * N * 8)); CK(hipMalloc(&win, N * 8)); CK(hipMalloc(&W, N * 8));
    CK(hipMalloc(&out_a, B * N * 4)); CK(hipMalloc(&out_b, B * N * 4));
    const size_t side_bytes = (size_t)cycles * pitch * N;
    CK(hipMalloc(&side_a, side_bytes)); CK(hipMalloc(&side_b, side_bytes));
    CK(hipMemset(side_a, 0, side_bytes)); CK(hipMemset(side_b, 0, side_bytes));
    CK(hipMemset(out_a, 0xff, B * N * 4)); CK(hipMemset(out_b, 0xee, B * N * 4));
    CK(hipMalloc(&acc, 32)); CK(hipMemset(acc, 0, 32));
    std::vector<float2> h(N);
    for (int k = 0; k < N; ++k) {
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
        std::vector<float2> hin(batches * N);
        for (uint64_t b = 0; b < batches; ++b) {
            const double f = fmod(100.25 + (double)b, (double)N) / N;
            for (int n = 0; n < N; ++n) {
                const double ph = 6.283185307179586 * f * n;
                hin[b * N + n] = make_float2((float)cos(ph) + 1e-3f * gauss(), (float)sin(ph) + 1e-3f * gauss());
            }
        }
        // a few special rows: zeros, a huge value, a NaN (the guard's cold path and non-finite handling)
        // (QB_SPECIAL=1 only: such rows run the guard's exact ladder on every element, and all sixteen copies of a row fall to ONE
        // workgroup -- a tail that is the harness's, not the kernels')
        if (getenv("QB_SPECIAL")) {
            for (int n = 0; n < N; ++n) hin[5 * N + n] = make_float2(0.0f, 0.0f);
            hin[6 * N + 17] = make_float2(3.0e38f, -3.0e38f);
            hin[7 * N + 99] = make_float2(NAN, 1.0f);
        }
        for (int c = 0; c < cycles; ++c) CK(hipMemcpy(in + (size_t)c * batches * N, hin.data(), batches * N * 8, hipMemcpyHostToDevice));
    }
    FftLayout L{};
    L.transforms = B; L.outer_rank = 1; L.outer_shape[0] = B; L.in_outer_stride[0] = N; L.out_outer_stride[0] = N;
    L.in_axis_stride = 1; L.out_axis_stride = 1;
    const float coeff = 20.0f * log10f(1.0f / (float)N);
    const float scale = 1.0f / 100.0f, offset = 100.0f * scale;
    using Pro = RealOperand<LoadCF32TimesWindow>;
    using Epi = StoreAmplitudeRangeSideT<QB_FAST>;
    Pro pro{{in, win, 1}};
    const Epi epi_a{{out_a, coeff, scale, offset, BinGuard{QB_FAST ? (float)H : 0.0f, 0.0f}}, side_a, (float)H, (uint32_t)batches, pitch};
    const Epi epi_b{{out_b, coeff, scale, offset, BinGuard{QB_FAST ? (float)H : 0.0f, 0.0f}}, side_b, (float)H, (uint32_t)batches, pitch};

    auto ka = fft_pipe_kernel<N, true, true, Pro, Epi>;
    auto kb = fft_quad_kernel<true, Pro, Epi>;
    const size_t lds_a = fft_pipe_lds_bytes(N), lds_b = fft_quad_lds_bytes();
    CK(hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
    CK(hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    hipFuncAttributes fa{}, fb{};
    CK(hipFuncGetAttributes(&fa, (const void*)ka)); CK(hipFuncGetAttributes(&fb, (const void*)kb));
    int occ_a = 0, occ_b = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_a, (const void*)ka, 512, lds_a));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, (const void*)kb, 256, lds_b));
    int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("[%s] pipe: regs %d scratch %zu lds %zu occupancy %d/CU | quad: regs %d scratch %zu lds %zu occupancy %d/CU | CUs %d\n", name,
           fa.numRegs, (size_t)fa.localSizeBytes, lds_a, occ_a, fb.numRegs, (size_t)fb.localSizeBytes, lds_b, occ_b, cus);
    const int grid_a = 2 * cus, grid_b = (argc > 4 ? atoi(argv[4]) : 4) * cus;

    hipStream_t st; CK(hipStreamCreate(&st));
    uint32_t* sched = nullptr;  // QB_STATIC=1: the static round robin instead of the device counter
    if (!getenv("QB_STATIC")) { CK(hipMalloc(&sched, kQuadSchedWords * 4)); CK(hipMemset(sched, 0, kQuadSchedWords * 4)); }
#ifdef JST_QUAD_TIMELINE
    unsigned long long* tl; CK(hipMalloc(&tl, (size_t)grid_b * 16 * 8)); CK(hipMemset(tl, 0, (size_t)grid_b * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(jst_qtl_base), &tl, sizeof(tl)));
#endif
    const char* only = getenv("QB_ONLY");  // "a": the pipelined kernel alone, "b": the quad kernel alone (fault isolation)
    const bool run_a = !only || only[0] == 'a', run_b = !only || only[0] == 'b';
    if (run_a) ka<<<grid_a, 512, lds_a, st>>>(L, W, pro, epi_a);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    printf("[%s] pipe ran\n", name); fflush(stdout);
    if (run_b) kb<<<grid_b, 256, lds_b, st>>>(L, W, pro, epi_b, sched);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    printf("[%s] quad ran\n", name); fflush(stdout);
    // clocks: the first milliseconds of a fresh process run far below the sustained clock
    for (int r = 0; r < (getenv("QB_WARM") ? atoi(getenv("QB_WARM")) : 300); ++r) {
        if (run_a) ka<<<grid_a, 512, lds_a, st>>>(L, W, pro, epi_a);
        if (run_b) kb<<<grid_b, 256, lds_b, st>>>(L, W, pro, epi_b, sched);
    }
    CK(hipStreamSynchronize(st));
    {
        unsigned long long init[4] = {0, ~0ull, 0, ~0ull};
        CK(hipMemcpy(acc, init, 32, hipMemcpyHostToDevice));
        diff_kernel<<<1024, 256, 0, st>>>((const uint32_t*)out_a, (const uint32_t*)out_b, B * N, acc, acc + 1);
        diff_kernel<<<1024, 256, 0, st>>>((const uint32_t*)side_a, (const uint32_t*)side_b, side_bytes / 4, acc + 2, acc + 3);
        CK(hipStreamSynchronize(st));
        unsigned long long r[4];
        CK(hipMemcpy(r, acc, 32, hipMemcpyDeviceToHost));
        printf("[%s] values differing: %llu of %llu (first at %lld = transform %lld pos %lld) | index words differing: %llu (first word %lld)\n", name, r[0],
               (unsigned long long)(B * N), r[0] ? (long long)r[1] : -1ll, r[0] ? (long long)(r[1] / N) : -1ll, r[0] ? (long long)(r[1] % N) : -1ll,
               r[2], r[2] ? (long long)r[3] : -1ll);
        if (r[0]) {  // show the neighbourhood of the first mismatch
            const uint64_t at = r[1] / N * N;
            std::vector<float> va(N), vb(N);
            CK(hipMemcpy(va.data(), out_a + at, N * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(vb.data(), out_b + at, N * 4, hipMemcpyDeviceToHost));
            int shown = 0, bad = 0;
            for (int i = 0; i < N; ++i) if (memcmp(&va[i], &vb[i], 4)) { ++bad; if (shown < 12) { printf("   pos %d: pipe %.9g quad %.9g\n", i, va[i], vb[i]); ++shown; } }
            printf("   %d of %d positions differ in that transform\n", bad, N);
        }
    }
#ifdef JST_QUAD_TIMELINE
    {
        CK(hipMemset(tl, 0, (size_t)grid_b * 16 * 8));
        kb<<<grid_b, 256, lds_b, st>>>(L, W, pro, epi_b, sched);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(grid_b * 16);
        CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
        const char* names[] = {"top barrier", "pass 0", "barrier 0", "pass 1", "barrier 1", "pass 2", "barrier 2", "p3 reads + barrier", "pieces issued", "pass 3 + epilogue", "own pieces landed"};
        double tot = 0, per[11];
        for (int k = 0; k < 11; ++k) {
            double a = 0;
            for (int b = 0; b < grid_b; ++b) a += (double)h[b * 16 + k] / (double)h[b * 16 + 12];
            per[k] = a / grid_b; tot += per[k];
        }
        printf("[%s] timeline, wavefront 0, mean ticks per transform (%.0f total):", name, tot);
        for (int k = 0; k < 11; ++k) printf(" %s=%.0f (%.1f%%)", names[k], per[k], 100.0 * per[k] / tot);
        printf("\n");
        // workgroup lifetimes on the 100 MHz wall clock: when do they start and end relative to the first start?
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < grid_b; ++b) { t0 = std::min(t0, h[b * 16 + 13]); t1 = std::max(t1, h[b * 16 + 14]); }
        std::vector<double> st(grid_b), en(grid_b);
        for (int b = 0; b < grid_b; ++b) { st[b] = (double)(h[b * 16 + 13] - t0) * 0.01; en[b] = (double)(h[b * 16 + 14] - t0) * 0.01; }
        std::vector<double> ss = st, es = en;
        std::sort(ss.begin(), ss.end()); std::sort(es.begin(), es.end());
        printf("[%s] workgroup starts (us after the first): median %.2f p90 %.2f max %.2f | ends: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f | span %.2f us\n", name,
               ss[grid_b / 2], ss[grid_b * 9 / 10], ss[grid_b - 1], es[0], es[grid_b / 10], es[grid_b / 2], es[grid_b * 9 / 10], es[grid_b - 1], (double)(t1 - t0) * 0.01);
        double xe[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < grid_b; ++b) xe[b % 8] += en[b] / (grid_b / 8);
        printf("[%s] mean end by blockIdx %% 8:", name);
        for (int x = 0; x < 8; ++x) printf(" %.2f", xe[x]);
        double idle = 0;
        for (int b = 0; b < grid_b; ++b) idle += (es[grid_b - 1] - en[b]) + st[b];
        printf(" | workgroup-time not running (before start + after end): %.1f%% of grid x span\n", 100.0 * idle / (grid_b * es[grid_b - 1]));
    }
#endif
    std::vector<hipEvent_t> ev(4 * reps);
    for (auto& e : ev) CK(hipEventCreate(&e));
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(ev[4 * r + 0], st));
        if (run_a) ka<<<grid_a, 512, lds_a, st>>>(L, W, pro, epi_a);
        CK(hipEventRecord(ev[4 * r + 1], st));
        CK(hipEventRecord(ev[4 * r + 2], st));
        if (run_b) kb<<<grid_b, 256, lds_b, st>>>(L, W, pro, epi_b, sched);
        CK(hipEventRecord(ev[4 * r + 3], st));
    }
    CK(hipStreamSynchronize(st));
    std::vector<float> ta, tb;
    for (int r = 0; r < reps; ++r) {
        float ms;
        CK(hipEventElapsedTime(&ms, ev[4 * r], ev[4 * r + 1])); ta.push_back(ms * 1e3f);
        CK(hipEventElapsedTime(&ms, ev[4 * r + 2], ev[4 * r + 3])); tb.push_back(ms * 1e3f);
    }
    std::sort(ta.begin(), ta.end()); std::sort(tb.begin(), tb.end());
    const double bytes = 12.0 * N * (double)B;
    printf("[%s] %d cycles per launch: pipe median %.2f us (min %.2f) = %.3f of 8 TB/s | quad median %.2f us (min %.2f) = %.3f | quad/pipe %.4f\n", name, cycles,
           ta[reps / 2], ta[0], bytes / (ta[reps / 2] * 1e-6) / 8e12, tb[reps / 2], tb[0], bytes / (tb[reps / 2] * 1e-6) / 8e12, tb[reps / 2] / ta[reps / 2]);
    return 0;
}
