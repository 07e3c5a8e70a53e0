// VALU issue-rate microbenchmark for gfx950: scalar f32 add/mul vs packed (v_pk_*) vs fma,
// 4 waves/SIMD, register-resident.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float* out, float seed, int iters) {
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 0.5f}; }
    const float m = seed * 1.0000001f, c = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { a[i] = a[i] * m; a[i] = a[i] + c; }           // 2 scalar VALU
                if (MODE == 1) { p[i] = p[i] * v2f{m, m}; p[i] = p[i] + v2f{c, c}; }  // 2 packed
                if (MODE == 2) { a[i] = __builtin_fmaf(a[i], m, c); }           // 1 fma
                if (MODE == 3) { a[i] = (a[i] > c) ? a[i] * m : a[i] + c; }     // cmp+mul+add+cndmask
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double ops_per_iter_lane) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<MODE><<<1024, 256>>>(out, 1.0f, 10);
    hipEventRecord(e0);
    k<MODE><<<1024, 256>>>(out, 1.0f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = 4096.0 * iters * 32 * (MODE == 2 ? 1 : (MODE == 3 ? 4 : 2));
    const double per_simd = wave_instr / 1024.0;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz), %.1f Tlane-op/s\n",
           name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4,
           4096.0 * 64 * iters * 32 * ops_per_iter_lane / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    run<0>("scalar mul+add", 2);
    run<1>("packed mul+add (v_pk)", 4);
    run<2>("fma", 2);
    run<3>("cmp+mul+add+cndmask", 2);
    return 0;
}
