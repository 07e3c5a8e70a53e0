// Issue-rate microbenchmark of the exact Amplitude -> Range epilogue and of a radix-8 butterfly + twiddle pass,
// in registers, no memory traffic: ns per element per wavefront and per SIMD at 1, 2, 4, 8 wavefronts per SIMD.
// Answers: what does the VALU really sustain on this instruction mix (literals, SGPR operands, transcendentals,
// half-rate conversions) compared with the 1.1 ns per plain v_add/v_mul of valu_ops.hip?
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -I cyberether_amd/csrc/kernels tools/ubench/epi_rate.hip
#include "device_math.hh"

#include <cstdio>
#include <cstdlib>
using namespace jst::dev;

#ifndef ER_FAST
#define ER_FAST 0
#endif

template <int ILP>
__global__ __launch_bounds__(1024) void epi_kernel(float* out, const float2* in, float coeff, float scale, float offset, int iters) {
    float2 v[ILP];
    float acc[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) { v[j] = in[threadIdx.x + j * 1024]; acc[j] = 0.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
#if ER_FAST
            const float r = amplitude_range_fast_guarded(v[j], coeff, scale, offset, BinGuard{256.0f, 0.0f});
#else
            const float r = amplitude_range_exact(v[j], coeff, scale, offset);
#endif
            acc[j] += r;
            v[j].x = v[j].x * 1.0000001f;  // keeps the chain data-dependent, stays in the same tanh class
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < ILP; ++j) s += acc[j];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

__global__ __launch_bounds__(1024) void fft_kernel(float2* out, const float2* in, int iters) {
    float2 x[8], w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = in[threadIdx.x + j * 1024]; w[j] = in[threadIdx.x + (8 + j) * 1024]; }
    for (int it = 0; it < iters; ++it) {
        butterfly8<true>(x);
#pragma unroll
        for (int c = 1; c < 8; ++c) x[c] = special_mul<true>(x[c], w[c]);
#pragma unroll
        for (int c = 0; c < 8; ++c) { x[c].x *= 0.3535f; x[c].y *= 0.3535f; }  // keep magnitudes bounded (16 extra mul)
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) out[blockIdx.x * 8192 + threadIdx.x + j * 1024] = x[j];
}

int main() {
    float* out; float2* in; float2* out2;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 16 * 1024 * 8); hipMalloc(&out2, 256 * 8192 * 8);
    float2 h[16 * 1024];
    for (int i = 0; i < 16 * 1024; ++i) h[i] = make_float2(0.01f + 1e-4f * (i % 977), 0.02f - 1e-4f * (i % 331));
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const float coeff = 20.0f * log10f(1.0f / 4096.0f), scale = 0.01f, offset = 1.0f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int w : {1, 2, 4, 8}) {
        const int threads = 256 * w;  // w wavefronts per SIMD
        for (int ilp : {1, 2}) {
            auto k = ilp == 1 ? epi_kernel<1> : epi_kernel<2>;
            k<<<256, threads>>>(out, in, coeff, scale, offset, 10);
            hipEventRecord(e0);
            k<<<256, threads>>>(out, in, coeff, scale, offset, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double elems = (double)iters * ilp;
            printf("epilogue(%s) ilp %d  waves/SIMD %d : %.1f ns per element per wave, %.1f ns per element per SIMD\n",
                   ER_FAST ? "fast+guard" : "exact", ilp, w, ms * 1e6 / elems, ms * 1e6 / (elems * w));
        }
        fft_kernel<<<256, threads>>>(out2, in, 10);
        hipEventRecord(e0);
        fft_kernel<<<256, threads>>>(out2, in, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("butterfly8+7 twiddles+16 mul  waves/SIMD %d : %.1f ns per butterfly per wave, %.1f ns per SIMD\n", w,
               ms * 1e6 / iters, ms * 1e6 / (iters * (double)w));
    }
    return 0;
}
