// lds_atomic_rate.hip -- what an LDS atomic (ds_add_u32, no return) costs per wavefront instruction on gfx950 with all 16
// wavefronts of a 1024-thread workgroup issuing them, as a function of the bank pattern:
//   0  two lanes per bank, different addresses (the minimum for 64 lanes on 32 banks)
//   1  the spectrogram's layout: bank = 16 * (bin & 1) + column, 4 rows x 16 columns per instruction, random bins
//   2  random dword addresses in a 16 KiB histogram
//   3  plain ds_write_b32 with pattern 0 (for scale)
//   4  bank = 16 * (row & 1) + column (bin * 32 words pitch, two interleaved copies): deterministic two lanes per bank
// hipcc --offload-arch=gfx950 -O3 -o bin/lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t* out, const uint32_t* rnd, int iters, unsigned long long* clk) {
    extern __shared__ uint32_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t e = tid; e < 16384u; e += 1024u) lds[e] = 0;
    __syncthreads();
    uint32_t r = rnd[tid];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            r = r * 1664525u + 1013904223u;
            const uint32_t bin = (r >> 16) & 0xffu;
            uint32_t w;
            if (MODE == 0 || MODE == 3) w = (lane & 31u) + 32u * ((lane >> 5) + 2u * (bin & 63u));
            else if (MODE == 1) w = bin * 16u + ((lane + j) & 15u);
            else if (MODE == 2) w = (r >> 8) & 4095u;
            else w = bin * 32u + 16u * ((lane >> 4) & 1u) + ((lane + j) & 15u);
            if (MODE == 3) lds[w] = r;
            else __hip_atomic_fetch_add(&lds[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
    uint32_t s = 0;
    for (uint32_t e = tid; e < 16384u; e += 1024u) s += lds[e];
    out[blockIdx.x * 1024 + tid] = s;
}

int main() {
    uint32_t *out, *rnd;
    unsigned long long* clk;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&rnd, 1024 * 4);
    hipMalloc(&clk, 256 * 8);
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 12345u * (i + 1) ^ (i << 13);
    hipMemcpy(rnd, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 64;
    auto run = [&](auto kern, const char* name) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 65536, 0, out, rnd, iters, clk);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 65536, 0, out, rnd, iters, clk);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(256);
        hipMemcpy(c.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
        const double instr = 16.0 * iters * 16;  // wave instructions per workgroup
        printf("%-58s %.1f us, %.2f ns per wave instruction (CU), %.1f counter ticks\n", name, ms * 1e3, ms * 1e6 / instr,
               (double)c[7] / instr);
    };
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run(k<0>, "0 ds_add, two lanes per bank (minimum)");
    run(k<1>, "1 ds_add, spectrogram layout (bank = 16*(bin&1)+col)");
    run(k<2>, "2 ds_add, random addresses");
    run(k<3>, "3 ds_write_b32, two lanes per bank");
    run(k<4>, "4 ds_add, bin*32 pitch, two interleaved copies");
    return 0;
}
