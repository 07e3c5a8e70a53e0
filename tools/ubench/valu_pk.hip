// Packed-f32 VALU throughput on gfx950 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) next to the scalar forms:
// does one packed instruction cost one issue slot (2x the flops) or two?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_pk.hip -o tools/ubench/valu_pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256, 4) void k(float* out, float seed, int iters) {
    v2f a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f,
        a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    v2f m = {seed * 1.0000001f, seed * 0.9999999f}, c = {seed * 0.5f, seed * 0.25f};
    for (int it = 0; it < iters; ++it) {
#define E(op, n) op " %" #n ", %" #n ", %8\n"
#define E3(op, n) op " %" #n ", %" #n ", %8, %9\n"
#define ES(op, n) op " %" #n ", %" #n ", %8\n"
#define ASM(TXT) asm volatile(TXT TXT TXT TXT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c))
#define ALL(M, op) M(op, 0) M(op, 1) M(op, 2) M(op, 3) M(op, 4) M(op, 5) M(op, 6) M(op, 7)
        if (OP == 0) ASM(ALL(E, "v_pk_add_f32"));
        if (OP == 1) ASM(ALL(E, "v_pk_mul_f32"));
        if (OP == 2) ASM(ALL(E3, "v_pk_fma_f32"));
#define EPS(op, n) op " %" #n ", s[20:21], %8, %" #n " op_sel_hi:[0,1,1]\n"
#define EFS(op, n) op " %" #n ", s20, %8, %" #n "\n"
        if (OP == 3) { asm volatile("s_mov_b32 s20, 0x3f800001\n s_mov_b32 s21, 0x3f800002" ::: "s20", "s21"); ASM(ALL(EPS, "v_pk_fma_f32")); }
    }
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
template <int OP>
void run(const char* name) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    k<OP><<<1024, 256>>>(out, 1.0f, 10);
    hipEventRecord(e0);
    k<OP><<<1024, 256>>>(out, 1.0f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s %7.3f ms  %.2f ns per packed wave-instr per SIMD\n", name, ms, ms * 1e6 / (4.0 * iters * 32));
    hipFree(out);
}
int main() { run<0>("v_pk_add_f32"); run<1>("v_pk_mul_f32"); run<2>("v_pk_fma_f32"); run<3>("v_pk_fma_f32 sgpr splat"); return 0; }
