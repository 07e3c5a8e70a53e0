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

// Workgroups that LIVE for a while: every thread waits `ticks` of the 100 MHz wall clock, thread 0 then stores.  grid x lifetime / span
// = workgroups alive on average; against the slots the chip has (occupancy x CUs) it says how fast a freed slot is refilled.
__global__ __launch_bounds__(1024) void living_kernel(unsigned* out, unsigned ticks, unsigned long long* stamps) {
    extern __shared__ unsigned lds[];
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    if (ticks == 0xffffffffu) lds[threadIdx.x] = blockIdx.x;  // never: keeps the allocation
    while ((unsigned long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[blockIdx.x] = blockIdx.x;
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
    }
}
static void run_living(unsigned* out, unsigned long long* stamps, int grid, int threads, int lds, unsigned ticks) {
    hipFuncSetAttribute((const void*)living_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)living_kernel, threads, lds);
    for (int rep = 0; rep < 3; ++rep) { living_kernel<<<grid, threads, lds>>>(out, ticks, stamps); hipDeviceSynchronize(); }
    static unsigned long long h[2 * 65536];
    hipMemcpy(h, stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    unsigned long long w0 = ~0ull, w1 = 0; double life = 0;
    for (int b = 0; b < grid; ++b) { if (h[2 * b] < w0) w0 = h[2 * b]; if (h[2 * b + 1] > w1) w1 = h[2 * b + 1]; life += (double)(h[2 * b + 1] - h[2 * b]); }
    const double span = (double)(w1 - w0) * 0.01, mean_life = life / grid * 0.01, alive = life / (double)(w1 - w0);
    const double slots = (double)per_cu * 256.0, rounds = grid / slots;
    printf("living | grid %6d x %4d threads, LDS %6d B, lifetime %5.2f us: span %7.2f us | alive %7.1f of %5.0f slots (%4.1f %%) | ideal span %6.2f us "
           "-> refill gap per slot and round %5.2f us\n", grid, threads, lds, mean_life, span, alive, slots, 100.0 * alive / slots, rounds * mean_life,
           rounds > 1.0 ? (span - rounds * mean_life) / (rounds - 1.0 > 0.5 ? rounds : 1.0) : 0.0);
}

int main() {
    unsigned* out; hipMalloc(&out, 1 << 22);
    unsigned long long* stamps; hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 65536);
    // the tiled kernels' shapes: config 5 blocks (512 threads, 40 KB), columns (256 threads, 16 KB), config 3 (768 / 1024 threads)
    for (unsigned ticks : {200u, 500u, 1000u}) {
        run_living(out, stamps, 2048, 512, 40 * 1024, ticks);
        run_living(out, stamps, 4096, 256, 16 * 1024, ticks);
        run_living(out, stamps, 3200, 768, 50 * 1024, ticks);
        run_living(out, stamps, 8192, 256, 16 * 1024, ticks);
    }
    for (int threads : {64, 256, 512, 1024})
        for (int lds : {0, 16384, 65536})
            for (int grid : {2048, 16384}) run<64>(out, grid, threads, lds);
    for (int threads : {256, 1024}) for (int grid : {1024, 16384}) run<128>(out, grid, threads, 36 * 1024);
    return 0;
}
