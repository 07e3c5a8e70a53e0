// Dependent-issue latency of VALU ops on gfx950: NCH independent chains per wave, W waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_dep.hip -o tools/ubench/valu_dep
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NCH>
__global__ __launch_bounds__(1024) void k(float* out, float seed, int iters) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = seed * 1.0000001f;
    for (int it = 0; it < iters; ++it) {
        if (NCH == 1)
            asm volatile("v_mul_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n"
                         "v_mul_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        if (NCH == 2)
            asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n"
                         "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        if (NCH == 4)
            asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
                         "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        if (NCH == 8)  // eight independent instructions, then eight that depend on them pairwise
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int NCH>
void run(int waves_per_simd) {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, threads = 256 * waves_per_simd;
    k<NCH><<<256, threads>>>(out, 1.0f, 10);
    hipEventRecord(e0);
    k<NCH><<<256, threads>>>(out, 1.0f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = (double)iters * 8;
    printf("chains/wave %d  waves/SIMD %d : %.2f ns per instruction per wave, %.2f ns per instr per SIMD\n", NCH,
           waves_per_simd, ms * 1e6 / per_wave, ms * 1e6 / (per_wave * waves_per_simd));
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
