// LDS instruction throughput per CU at 16 wavefronts per CU (the fused kernel's occupancy): ds_read_b64 vs ds_read2_b64,
// ds_write_b64 vs ds_write2st64_b64, conflict-free addresses (lane-contiguous 8-byte elements).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ubench/lds_rate.hip -o tools/ubench/bin/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned addr = (unsigned)(size_t)smem + threadIdx.x * 8u;  // LDS byte address (low 32 bits of the flat ptr)
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = make_float2(threadIdx.x + j, 1.0f);
    const unsigned a = threadIdx.x * 8u;
    (void)addr;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {  // 8 x ds_read_b64, rows 8 KiB apart
            asm volatile(
                "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8192\n ds_read_b64 %2, %8 offset:16384\n ds_read_b64 %3, %8 offset:24576\n"
                "ds_read_b64 %4, %8 offset:32768\n ds_read_b64 %5, %8 offset:40960\n ds_read_b64 %6, %8 offset:49152\n ds_read_b64 %7, %8 offset:57344\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(a) : "memory");
        } else if constexpr (MODE == 1) {  // 4 x ds_read2st64_b64 (offsets in units of 64*8 = 512 B)
            asm volatile(
                "ds_read2st64_b64 %0, %4 offset0:0 offset1:16\n ds_read2st64_b64 %1, %4 offset0:32 offset1:48\n"
                "ds_read2st64_b64 %2, %4 offset0:64 offset1:80\n ds_read2st64_b64 %3, %4 offset0:96 offset1:112\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(*(float4*)&v[0]), "=&v"(*(float4*)&v[2]), "=&v"(*(float4*)&v[4]), "=&v"(*(float4*)&v[6]) : "v"(a) : "memory");
        } else if constexpr (MODE == 2) {  // 8 x ds_write_b64
            asm volatile(
                "ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:8192\n ds_write_b64 %8, %2 offset:16384\n ds_write_b64 %8, %3 offset:24576\n"
                "ds_write_b64 %8, %4 offset:32768\n ds_write_b64 %8, %5 offset:40960\n ds_write_b64 %8, %6 offset:49152\n ds_write_b64 %8, %7 offset:57344\n"
                "s_waitcnt lgkmcnt(0)\n"
                :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(a) : "memory");
        } else if constexpr (MODE == 3) {  // 4 x ds_write2st64_b64
            asm volatile(
                "ds_write2st64_b64 %4, %0, %1 offset0:0 offset1:16\n ds_write2st64_b64 %4, %2, %3 offset0:32 offset1:48\n"
                "ds_write2st64_b64 %4, %5, %6 offset0:64 offset1:80\n ds_write2st64_b64 %4, %7, %8 offset0:96 offset1:112\n"
                "s_waitcnt lgkmcnt(0)\n"
                :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(a), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]) : "memory");
        } else if constexpr (MODE == 4) {  // 4 x ds_read_b128 (two adjacent elements per lane)
            const unsigned a2 = threadIdx.x * 16u;
            asm volatile(
                "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16384\n ds_read_b128 %2, %4 offset:32768\n ds_read_b128 %3, %4 offset:49152\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=&v"(*(float4*)&v[0]), "=&v"(*(float4*)&v[2]), "=&v"(*(float4*)&v[4]), "=&v"(*(float4*)&v[6]) : "v"(a2) : "memory");
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j].x + v[j].y;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {512, 1024}) {  // 1024 threads = 16 wavefronts per CU = 4 per SIMD
        k<MODE><<<256, threads, 72 * 1024>>>(out, 10);
        hipEventRecord(e0);
        k<MODE><<<256, threads, 72 * 1024>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ns_iter = ms * 1e6 / iters;  // all waves of a CU do one 8-element group each
        const int waves = threads / 64;
        printf("%-22s %2d waves/CU: %.1f ns per round of %d wave-groups = %.1f ns (%.1f clk @1.92GHz) per wave per 8 x 512 B, %.0f B/clk/CU\n",
               name, waves, ns_iter, waves, ns_iter / waves, ns_iter / waves * 1.92, waves * 4096.0 / (ns_iter * 1.92));
    }
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    run<0>("8 x ds_read_b64", out);
    run<1>("4 x ds_read2st64_b64", out);
    run<4>("4 x ds_read_b128", out);
    run<2>("8 x ds_write_b64", out);
    run<3>("4 x ds_write2st64_b64", out);
    return 0;
}
