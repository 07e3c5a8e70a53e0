// Per-opcode VALU throughput on gfx950: 32 independent instructions of ONE opcode per loop
// iteration (inline asm, 8 independent register chains), 4 waves per SIMD on every SIMD.
// Reports cycles per wave-instruction per SIMD at the measured shader clock (s_memtime vs wall).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_ops.hip -o tools/ubench/valu_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY4(OPS) OPS OPS OPS OPS

template <int OP>
__global__ __launch_bounds__(256, 4) void k(float* out, float seed, int iters, unsigned long long* clk) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = seed * 1.0000001f, c = seed * 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define V2(n) "%" #n
#define E(op, n) op " %" #n ", %" #n ", %8\n"
#define E3(op, n) op " %" #n ", %" #n ", %8, %9\n"
#define E1(op, n) op " %" #n ", %" #n "\n"
#define ASM(TXT) asm volatile(TXT TXT TXT TXT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc")
#define ALL(M, op) M(op, 0) M(op, 1) M(op, 2) M(op, 3) M(op, 4) M(op, 5) M(op, 6) M(op, 7)
        if (OP == 0) ASM(ALL(E, "v_add_f32"));
        if (OP == 1) ASM(ALL(E, "v_mul_f32"));
        if (OP == 2) ASM(ALL(E3, "v_fma_f32"));
        if (OP == 3) ASM(ALL(E, "v_cndmask_b32"));      // vcc select
        if (OP == 4) ASM(ALL(E, "v_and_b32"));
        if (OP == 5) ASM(ALL(E, "v_add_u32"));
        if (OP == 6) ASM(ALL(E, "v_lshlrev_b32"));
        if (OP == 7) ASM(ALL(E1, "v_cvt_i32_f32"));
        if (OP == 8) ASM(ALL(E1, "v_rcp_f32"));
        if (OP == 9) ASM(ALL(E1, "v_sqrt_f32"));
        if (OP == 10) ASM(ALL(E1, "v_exp_f32"));
        if (OP == 11) ASM(ALL(E1, "v_frexp_mant_f32"));
        if (OP == 12) ASM(ALL(E, "v_max_f32"));
        if (OP == 13) ASM(ALL(E1, "v_mov_b32"));
        if (OP == 14) ASM(ALL(E3, "v_div_fixup_f32"));
        if (OP == 15) ASM(ALL(E3, "v_div_fmas_f32"));
        if (OP == 16) ASM(ALL(E, "v_sub_f32"));
        if (OP == 17) ASM(ALL(E3, "v_bfe_u32"));
        if (OP == 18) ASM(ALL(E, "v_ldexp_f32"));
        if (OP == 19) ASM(ALL(E3, "v_med3_f32"));
        if (OP == 20) ASM(ALL(E3, "v_mad_u32_u24"));
        if (OP == 21) ASM(ALL(E1, "v_cvt_f32_i32"));
        if (OP == 22) ASM(ALL(E1, "v_rndne_f32"));
        if (OP == 23) ASM(ALL(E1, "v_log_f32"));
        if (OP == 24) ASM(ALL(E1, "v_frexp_exp_i32_f32"));
        if (OP == 25) ASM(ALL(E3, "v_add3_u32"));
        if (OP == 26) ASM(ALL(E3, "v_lshl_add_u32"));
        if (OP == 27) ASM(ALL(E, "v_mul_lo_u32"));
        if (OP == 28) ASM(ALL(E, "v_xor_b32"));
        if (OP == 29) ASM(ALL(E, "v_min_u32"));
        if (OP == 30) ASM(ALL(E3, "v_bfi_b32"));
        if (OP == 31) { asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555" ::: "s20", "s21");
#define E64(op, n) op " %" #n ", %" #n ", %8, s[20:21]\n"
            ASM(ALL(E64, "v_cndmask_b32_e64")); }
        if (OP == 32) {  // compare + select pairs, vcc rewritten every time (the compiler's usual shape)
#define EC(op, n) "v_cmp_lt_f32 vcc, %" #n ", %9\n v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
            ASM(ALL(EC, "")); }
        if (OP == 33) ASM(ALL(E3, "v_perm_b32"));
        if (OP == 34) ASM(ALL(E, "v_or_b32"));
        if (OP == 35) ASM(ALL(E, "v_ashrrev_i32"));
        if (OP == 36) ASM(ALL(E, "v_min_f32"));
        if (OP == 37) ASM(ALL(E, "v_sub_u32"));
        if (OP == 38) ASM(ALL(E3, "v_and_or_b32"));
        if (OP == 39) ASM(ALL(E, "v_max_i32"));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

// compare ops write vcc/sgpr pairs: separate kernel
__global__ __launch_bounds__(256, 4) void kcmp(float* out, float seed, int iters) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, m = seed * 1.01f;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned long long s0, s1, s2, s3;
        asm volatile(
            "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %4, %6\n v_cmp_gt_f32 %3, %5, %6\n"
            "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %4, %6\n v_cmp_gt_f32 %3, %5, %6\n"
            "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %4, %6\n v_cmp_gt_f32 %3, %5, %6\n"
            "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %4, %6\n v_cmp_gt_f32 %3, %5, %6\n"
            "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %4, %6\n v_cmp_gt_f32 %3, %5, %6\n"
            "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %4, %6\n v_cmp_gt_f32 %3, %5, %6\n"
            "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %4, %6\n v_cmp_gt_f32 %3, %5, %6\n"
            "v_cmp_lt_f32 %0, %4, %6\n v_cmp_lt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %4, %6\n v_cmp_gt_f32 %3, %5, %6\n"
            : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(m));
        acc += s0 ^ s1 ^ s2 ^ s3;
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(acc & 0xff);
}

template <int OP>
double run(const char* name) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    k<OP><<<1024, 256>>>(out, 1.0f, 10, clk);
    hipEventRecord(e0);
    k<OP><<<1024, 256>>>(out, 1.0f, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double per_simd = 4.0 * iters * 32;  // wave-instructions per SIMD
    printf("%-22s %7.3f ms  %.2f ns/instr/SIMD   %.2f shader-cycles/instr (wave0 counted %llu cycles)\n", name, ms,
           ms * 1e6 / per_simd, (double)c / per_simd, c);
    hipFree(out); hipFree(clk);
    return ms;
}
int main() {
    run<0>("v_add_f32"); run<1>("v_mul_f32"); run<2>("v_fma_f32"); run<16>("v_sub_f32"); run<12>("v_max_f32");
    run<19>("v_med3_f32"); run<3>("v_cndmask_b32"); run<13>("v_mov_b32"); run<4>("v_and_b32"); run<28>("v_xor_b32");
    run<5>("v_add_u32"); run<25>("v_add3_u32"); run<26>("v_lshl_add_u32"); run<6>("v_lshlrev_b32"); run<17>("v_bfe_u32");
    run<29>("v_min_u32"); run<20>("v_mad_u32_u24"); run<27>("v_mul_lo_u32");
    run<7>("v_cvt_i32_f32"); run<21>("v_cvt_f32_i32"); run<22>("v_rndne_f32"); run<18>("v_ldexp_f32");
    run<11>("v_frexp_mant_f32"); run<24>("v_frexp_exp_i32_f32");
    run<8>("v_rcp_f32"); run<9>("v_sqrt_f32"); run<10>("v_exp_f32"); run<23>("v_log_f32");
    run<14>("v_div_fixup_f32"); run<15>("v_div_fmas_f32");
    run<30>("v_bfi_b32"); run<31>("v_cndmask_e64 sgpr"); run<32>("v_cmp+v_cndmask (x2)"); run<33>("v_perm_b32");
    run<34>("v_or_b32"); run<35>("v_ashrrev_i32"); run<36>("v_min_f32"); run<37>("v_sub_u32"); run<38>("v_and_or_b32"); run<39>("v_max_i32");
    {
        float* out; hipMalloc(&out, 1024 * 256 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kcmp<<<1024, 256>>>(out, 1.0f, 10);
        hipEventRecord(e0); kcmp<<<1024, 256>>>(out, 1.0f, 4000); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-22s %7.3f ms  %.2f ns/instr/SIMD (incl. 3 salu per 32 cmp)\n", "v_cmp_*_f32 -> sgpr", ms, ms * 1e6 / (4.0 * 4000 * 32));
    }
    return 0;
}
