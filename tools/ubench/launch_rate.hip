// How fast does the chip START wavefronts?  Empty workgroups (one store each) of W wavefronts with L bytes of LDS and a register
// footprint of 64 / 128 VGPRs: grid / (event time - empty-launch time) = workgroups per microsecond.  Diagnostic (round 6: the
// LDS-tiled FFT kernels' one-workgroup-per-tile launches run at this rate, not at their tiles' cost).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_rate.hip -o tools/ubench/bin/launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int VG>
__global__ __launch_bounds__(1024, VG == 128 ? 4 : 8) void empty_kernel(unsigned* out, int spin) {
    extern __shared__ unsigned lds[];
    if (spin < 0) lds[threadIdx.x] = blockIdx.x;  // never: keeps the allocation
    if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x;
    if constexpr (VG == 128) {  // force a 128-VGPR allocation
        float v[100];
#pragma unroll
        for (int i = 0; i < 100; ++i) v[i] = (float)(threadIdx.x + i);
        if (spin < 0) {
#pragma unroll
            for (int i = 0; i < 100; ++i) __asm__ volatile("" : "+v"(v[i]));
            float s = 0;
#pragma unroll
            for (int i = 0; i < 100; ++i) s += v[i];
            out[threadIdx.x] = (unsigned)s;
        }
    }
}

template <int VG>
static void run(unsigned* out, int grid, int threads, int lds) {
    hipFuncSetAttribute((const void*)empty_kernel<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        empty_kernel<VG><<<grid, threads, lds>>>(out, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 1e3, waves = (double)grid * (threads / 64);
    printf("VGPR %3d | grid %6d x %4d threads (%2d waves), LDS %6d B: %8.2f us | %7.1f workgroups/us | %7.1f waves/us\n", VG, grid, threads,
           threads / 64, lds, us, grid / us, waves / us);
}

int main() {
    unsigned* out; hipMalloc(&out, 1 << 22);
    for (int threads : {64, 256, 512, 1024})
        for (int lds : {0, 16384, 65536})
            for (int grid : {2048, 16384}) run<64>(out, grid, threads, lds);
    for (int threads : {256, 1024}) for (int grid : {1024, 16384}) run<128>(out, grid, threads, 36 * 1024);
    return 0;
}
