// Steady-state phase timeline of the PRODUCT's pipelined fused spectrum kernel at the bench line's launch size:
// 16384 transforms of 4096 points (16 cycles of CF32[1024, 4096]) on 512 persistent workgroups, provider fast with the
// real window operand resident and the row-index side output.  Thread 0 of every workgroup stamps clock64 at the
// phase boundaries of every transform into a ring of three iterations: what is read back are the LAST three
// transforms of each workgroup (30, 31, 32 of 32: the last one has no successor to prefetch).  Diagnostic only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -DJST_FFT_TIMELINE -DJST_OPND_RESIDENT=1
//         -I cyberether_amd/csrc/kernels -I cyberether_amd/csrc -I include tools/ubench/fft_timeline_batched.hip
#include "fft_lds.hh"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace jst::dev;

int main() {
    constexpr int N = 4096;
    const uint64_t B = 16384, batches = 1024, H = 256;
    float2 *in, *win, *W; float* out; uint8_t* side; unsigned long long* tl;
    hipMalloc(&in, B * N * 8); hipMalloc(&win, N * 8); hipMalloc(&W, N * 8); hipMalloc(&out, B * N * 4); hipMalloc(&side, B * N);
    hipMalloc(&tl, 512 * (64 + 128) * 8); hipMemset(tl, 0, 512 * (64 + 128) * 8);
    std::vector<float2> h(N);
    for (int i = 0; i < N; ++i) h[i] = make_float2(cosf(6.283185307f * i / N), sinf(6.283185307f * i / N));
    hipMemcpy(W, h.data(), N * 8, hipMemcpyHostToDevice);
    for (int i = 0; i < N; ++i) h[i] = make_float2((i & 1) ? -0.5f : 0.5f, 0.0f);
    hipMemcpy(win, h.data(), N * 8, hipMemcpyHostToDevice);
    {
        std::vector<float2> hin(batches * N);
        for (size_t i = 0; i < hin.size(); ++i) hin[i] = make_float2((float)((i * 7919) % 1000) / 1000.f - 0.5f, (float)((i * 104729) % 1000) / 1000.f - 0.5f);
        for (uint64_t c = 0; c < B / batches; ++c) hipMemcpy(in + c * batches * N, hin.data(), batches * N * 8, hipMemcpyHostToDevice);
    }
    hipMemcpyToSymbol(HIP_SYMBOL(jst_tl_base), &tl, sizeof(tl));
    FftLayout L{}; L.transforms = B; L.outer_rank = 1; L.outer_shape[0] = B; L.in_outer_stride[0] = N; L.out_outer_stride[0] = N; L.in_axis_stride = 1; L.out_axis_stride = 1;
    using Pro = RealOperand<LoadCF32TimesWindow>;
    using Epi = StoreAmplitudeRangeSideT<true>;
    Pro pro{{in, win, 1}};
    const Epi epi{{out, -72.2472f, 0.01f, 1.0f, BinGuard{(float)H, 0.0f}}, side, (float)H, (uint32_t)batches, (uint32_t)batches};
    auto k = fft_pipe_kernel<N, true, true, Pro, Epi>;
    const size_t lds = fft_pipe_lds_bytes(N);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        k<<<512, 512, lds>>>(L, W, pro, epi);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.2f us by events (%s)\n", rep, ms * 1e3, hipGetErrorString(hipGetLastError()));
    }
    std::vector<unsigned long long> t(512 * 64);
    hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost);
    // ticks -> ns: workgroup lifetime by the 100 MHz wall clock (slots 62 / 63) against clock64 over the ring
    const char* names[] = {"iter start", "input ready+window", "pass0 done", "bar0", "pass1 done", "bar1", "pass2 done", "bar2", "pass3+epilogue done"};
    // 32 transforms per workgroup: tl_it 0..31, ring slot tl_it % 3 -> slots hold 30 (slot 0), 31 (slot 1), 29 (slot 2)
    const int order[3] = {2, 0, 1};
    const char* which[3] = {"transform 30 of 32", "transform 31 of 32", "transform 32 of 32 (last: nothing to prefetch)"};
    double per_tick = 0;
    {
        double a = 0; int cnt = 0;
        for (int b = 0; b < 512; ++b) {
            const double wall = (double)(t[b * 64 + 63] - t[b * 64 + 62]) * 10.0;  // ns, whole lifetime
            (void)wall;
            // one full iteration in ticks: start of transform 31 - start of transform 30
            const double it_ticks = (double)(t[b * 64 + 0 * 16 + 0] - t[b * 64 + 2 * 16 + 0]);
            a += wall / 32.0 / it_ticks; ++cnt;
        }
        per_tick = a / cnt;
        printf("mean workgroup lifetime / 32 per iteration-ticks -> about %.3f ns per clock64 tick (rough)\n", per_tick);
    }
    double life = 0;
    for (int b = 0; b < 512; ++b) life += (double)(t[b * 64 + 63] - t[b * 64 + 62]) * 0.01;
    printf("mean workgroup lifetime %.2f us = %.3f us per transform\n", life / 512, life / 512 / 32);
    for (int q = 0; q < 3; ++q) {
        const int sl = order[q];
        printf("%s, mean phase ticks:", which[q]);
        double tot = 0;
        for (int s = 1; s < 9; ++s) {
            double a = 0;
            for (int b = 0; b < 512; ++b) a += (double)(t[b * 64 + sl * 16 + s] - t[b * 64 + sl * 16 + s - 1]);
            printf(" %s=%.0f", names[s], a / 512);
            tot += a / 512;
        }
        printf(" | sum %.0f ticks\n", tot);
    }
    {   // gap between the end of one transform (slot 8) and the start of the next (slot 0)
        double a = 0;
        for (int b = 0; b < 512; ++b) a += (double)(t[b * 64 + 0 * 16 + 0] - t[b * 64 + 2 * 16 + 8]);
        printf("gap end of transform 30 -> start of 31: %.0f ticks\n", a / 512);
    }
    {   // per wavefront, transform 31 of 32 (tl_it == 30): arrival at each stamp relative to the workgroup's EARLIEST wavefront at
        // that stamp, mean over the workgroups; and which wavefront is last at barrier 0 (slot 2 = pass 0 done)
        std::vector<unsigned long long> tw(512 * 128);
        hipMemcpy(tw.data(), tl + 512 * 64, tw.size() * 8, hipMemcpyDeviceToHost);
        const int slots[] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
        printf("per wavefront (transform 31 of 32): mean lag in ticks behind the workgroup's first wavefront at each stamp\n");
        printf("  stamp:            ");
        for (int s : slots) printf(" %9s", (const char*[]){"it start", "in ready", "p0 done", "bar0", "p1 done", "bar1", "p2 done", "bar2", "epi done"}[s]);
        printf("\n");
        int last_hist[8] = {0};
        for (int w = 0; w < 8; ++w) {
            printf("  wavefront %d:      ", w);
            for (int s : slots) {
                double a = 0;
                for (int b = 0; b < 512; ++b) {
                    unsigned long long mn = ~0ull;
                    for (int v = 0; v < 8; ++v) mn = std::min(mn, tw[(size_t)b * 128 + v * 16 + s]);
                    a += (double)(tw[(size_t)b * 128 + w * 16 + s] - mn);
                }
                printf(" %9.0f", a / 512);
            }
            printf("\n");
        }
        for (int b = 0; b < 512; ++b) {
            int lw = 0;
            for (int v = 1; v < 8; ++v) if (tw[(size_t)b * 128 + v * 16 + 2] > tw[(size_t)b * 128 + lw * 16 + 2]) lw = v;
            ++last_hist[lw];
        }
        printf("  last wavefront at barrier 0, count over 512 workgroups:");
        for (int w = 0; w < 8; ++w) printf(" w%d=%d", w, last_hist[w]);
        printf("\n");
        // spread of arrivals at barrier 0 and at the end of the epilogue
        for (int s : {8, 0, 1, 2}) {
            double a = 0;
            for (int b = 0; b < 512; ++b) {
                unsigned long long mn = ~0ull, mx = 0;
                for (int v = 0; v < 8; ++v) { mn = std::min(mn, tw[(size_t)b * 128 + v * 16 + s]); mx = std::max(mx, tw[(size_t)b * 128 + v * 16 + s]); }
                a += (double)(mx - mn);
            }
            printf("  spread (last - first wavefront) at stamp %d: %.0f ticks\n", s, a / 512);
        }
    }
    return 0;
}
