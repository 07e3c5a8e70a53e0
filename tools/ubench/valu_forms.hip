// VALU throughput per SIMD on gfx950 by instruction FORM (operand kinds), W wavefronts per SIMD, 16 independent
// destinations per wavefront (no dependency stalls): what does a wave64 instruction cost the SIMD when both sources
// are VGPRs, when one is an SGPR, when it is packed, when it is an FMA ...
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_forms.hip -o tools/ubench/bin/valu_forms
#include <hip/hip_runtime.h>
#include <cstdio>

#define R16(OP)                                                                                            \
    OP("%0", "%16") OP("%1", "%17") OP("%2", "%18") OP("%3", "%19") OP("%4", "%20") OP("%5", "%21") OP("%6", "%22") OP("%7", "%23") \
    OP("%8", "%24") OP("%9", "%25") OP("%10", "%26") OP("%11", "%27") OP("%12", "%28") OP("%13", "%29") OP("%14", "%30") OP("%15", "%31")

#define F_ADD_VV(d, s) "v_add_f32 " d ", " d ", " s "\n"
#define F_MUL_VV(d, s) "v_mul_f32 " d ", " d ", " s "\n"
#define F_ADD_SV(d, s) "v_add_f32 " d ", %32, " d "\n"
#define F_ADD_VV3(d, s) "v_add_f32 " d ", " s ", " s "\n"  /* dst not a source */
#define F_FMA_VVV(d, s) "v_fma_f32 " d ", " d ", " s ", " s "\n"
#define F_FMA_SVV(d, s) "v_fma_f32 " d ", %32, " d ", " s "\n"
#define F_MOV(d, s) "v_mov_b32 " d ", " s "\n"
#define F_CNDMASK(d, s) "v_cndmask_b32 " d ", " d ", " s ", vcc\n"
#define F_EXP(d, s) "v_exp_f32 " d ", " s "\n"
#define F_RCP(d, s) "v_rcp_f32 " d ", " s "\n"
#define F_SQRT(d, s) "v_sqrt_f32 " d ", " s "\n"
#define F_CVT(d, s) "v_cvt_u32_f32 " d ", " s "\n"
#define F_SUB_VV(d, s) "v_sub_f32 " d ", " d ", " s "\n"
#define F_DPP(d, s) "v_mov_b32_dpp " d ", " s " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_CMP_VCC(d, s) "v_cmp_lt_f32 vcc, " d ", " s "\n"
#define F_CMP_SGPR(d, s) "v_cmp_lt_f32_e64 s[40:41], " d ", " s "\n"
#define F_CMP_CND(d, s) "v_cmp_lt_f32 vcc, " d ", " s "\n v_cndmask_b32 " d ", " d ", " s ", vcc\n"
#define F_CND_SGPR(d, s) "v_cndmask_b32_e64 " d ", " d ", " s ", s[42:43]\n"
#define F_AND(d, s) "v_and_b32 " d ", " d ", " s "\n"
#define F_ASHR(d, s) "v_ashrrev_i32 " d ", 31, " s "\n"
#define F_FREXP_M(d, s) "v_frexp_mant_f32 " d ", " s "\n"
#define F_FREXP_E(d, s) "v_frexp_exp_i32_f32 " d ", " s "\n"
#define F_CVT_FI(d, s) "v_cvt_f32_i32 " d ", " s "\n"
#define F_FRACT(d, s) "v_fract_f32 " d ", " s "\n"
#define F_MIN(d, s) "v_min_f32 " d ", " d ", " s "\n"
#define F_MED3(d, s) "v_med3_f32 " d ", " d ", " s ", " s "\n"
#define F_MUL_LIT(d, s) "v_mul_f32 " d ", 0x3f7ec46d, " d "\n"
#define F_ADD_ABS(d, s) "v_add_f32_e64 " d ", |" d "|, " s "\n"
#define F_LOG(d, s) "v_log_f32 " d ", " s "\n"
#define F_ADD_U32(d, s) "v_add_u32 " d ", " d ", " s "\n"
#define F_MUL_LO(d, s) "v_mul_lo_u32 " d ", " d ", " s "\n"

template <int FORM>
__global__ __launch_bounds__(1024) void k(float* out, float seed, int iters) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x + i; b[i] = seed * 0.5f + i; }
    const int sc = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed * 1.0000001f));
    for (int it = 0; it < iters; ++it) {
#define BODY(OPS)                                                                                                      \
    asm volatile(R16(OPS) R16(OPS)                                                                                     \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),     \
                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) \
                 : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]),  \
                   "v"(b[9]), "v"(b[10]), "v"(b[11]), "v"(b[12]), "v"(b[13]), "v"(b[14]), "v"(b[15]), "s"(sc)          \
                 : "vcc", "s40", "s41", "s42", "s43")
        if (FORM == 0) BODY(F_ADD_VV);
        if (FORM == 1) BODY(F_MUL_VV);
        if (FORM == 2) BODY(F_ADD_SV);
        if (FORM == 3) BODY(F_ADD_VV3);
        if (FORM == 4) BODY(F_FMA_VVV);
        if (FORM == 5) BODY(F_FMA_SVV);
        if (FORM == 6) BODY(F_MOV);
        if (FORM == 7) BODY(F_CNDMASK);
        if (FORM == 8) BODY(F_EXP);
        if (FORM == 9) BODY(F_RCP);
        if (FORM == 10) BODY(F_SQRT);
        if (FORM == 11) BODY(F_CVT);
        if (FORM == 12) BODY(F_SUB_VV);
        if (FORM == 13) BODY(F_DPP);
        if (FORM == 14) BODY(F_CMP_VCC);
        if (FORM == 15) BODY(F_CMP_SGPR);
        if (FORM == 16) BODY(F_CMP_CND);
        if (FORM == 17) BODY(F_CND_SGPR);
        if (FORM == 18) BODY(F_AND);
        if (FORM == 19) BODY(F_ASHR);
        if (FORM == 20) BODY(F_FREXP_M);
        if (FORM == 21) BODY(F_FREXP_E);
        if (FORM == 22) BODY(F_CVT_FI);
        if (FORM == 23) BODY(F_FRACT);
        if (FORM == 24) BODY(F_MIN);
        if (FORM == 25) BODY(F_MED3);
        if (FORM == 26) BODY(F_MUL_LIT);
        if (FORM == 27) BODY(F_ADD_ABS);
        if (FORM == 28) BODY(F_LOG);
        if (FORM == 29) BODY(F_ADD_U32);
        if (FORM == 30) BODY(F_MUL_LO);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

// packed forms: 8 register pairs
typedef float v2f __attribute__((ext_vector_type(2)));
#define R8(OP) OP("%0", "%8") OP("%1", "%9") OP("%2", "%10") OP("%3", "%11") OP("%4", "%12") OP("%5", "%13") OP("%6", "%14") OP("%7", "%15")
#define P_ADD_VV(d, s) "v_pk_add_f32 " d ", " d ", " s "\n"
#define P_MUL_VV(d, s) "v_pk_mul_f32 " d ", " d ", " s "\n"
#define P_ADD_SV(d, s) "v_pk_add_f32 " d ", %16, " d "\n"
#define P_FMA_VVV(d, s) "v_pk_fma_f32 " d ", " d ", " s ", " s "\n"
#define P_ADD_NEG(d, s) "v_pk_add_f32 " d ", " d ", " s " neg_lo:[0,1] neg_hi:[0,1]\n"
#define P_ADD_SWZ(d, s) "v_pk_add_f32 " d ", " d ", " s " op_sel:[0,1] op_sel_hi:[1,0]\n"
template <int FORM>
__global__ __launch_bounds__(1024) void kp(float* out, float seed, int iters) {
    v2f a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = v2f{seed + threadIdx.x + i, seed + i}; b[i] = v2f{seed * 0.5f + i, seed}; }
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i sc = v2i{__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed * 1.0000001f)),
                       __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed))};
    for (int it = 0; it < iters; ++it) {
#define PBODY(OPS)                                                                                                   \
    asm volatile(R8(OPS) R8(OPS) R8(OPS) R8(OPS)                                                                      \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])     \
                 : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "s"(sc))
        if (FORM == 0) PBODY(P_ADD_VV);
        if (FORM == 1) PBODY(P_MUL_VV);
        if (FORM == 2) PBODY(P_ADD_SV);
        if (FORM == 3) PBODY(P_FMA_VVV);
        if (FORM == 4) PBODY(P_ADD_NEG);
        if (FORM == 5) PBODY(P_ADD_SWZ);
    }
    v2f s = v2f{0, 0};
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s.x + s.y;
}

template <class K>
void run(const char* name, K kern, int waves_per_simd) {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, threads = 256 * waves_per_simd;
    kern<<<256, threads>>>(out, 1.0f, 10);
    hipEventRecord(e0);
    kern<<<256, threads>>>(out, 1.0f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = (double)iters * 32;
    printf("%-28s waves/SIMD %d : %.3f ns per instruction per SIMD\n", name, waves_per_simd, ms * 1e6 / (per_wave * waves_per_simd));
    hipFree(out);
}
int main() {
    for (int w : {2, 4}) {
        run("v_cmp_lt_f32 vcc", k<14>, w);
        run("v_cmp_lt_f32_e64 sgpr", k<15>, w);
        run("v_cmp + v_cndmask (per pair/2)", k<16>, w);
        run("v_cndmask_e64 sgpr pair", k<17>, w);
        run("v_and_b32", k<18>, w);
        run("v_ashrrev_i32", k<19>, w);
        run("v_frexp_mant_f32", k<20>, w);
        run("v_frexp_exp_i32_f32", k<21>, w);
        run("v_cvt_f32_i32", k<22>, w);
        run("v_fract_f32", k<23>, w);
        run("v_min_f32", k<24>, w);
        run("v_med3_f32", k<25>, w);
        run("v_mul_f32 literal", k<26>, w);
        run("v_add_f32_e64 abs", k<27>, w);
        run("v_log_f32", k<28>, w);
        run("v_add_u32", k<29>, w);
        run("v_mul_lo_u32", k<30>, w);
    }
    for (int w : {1, 2, 4}) {
        run("v_add_f32 v,v,v (d=s0)", k<0>, w);
        run("v_mul_f32 v,v,v (d=s0)", k<1>, w);
        run("v_sub_f32 v,v,v (d=s0)", k<12>, w);
        run("v_add_f32 v,s,v", k<2>, w);
        run("v_add_f32 d,v,v (same src)", k<3>, w);
        run("v_fma_f32 v,v,v,v", k<4>, w);
        run("v_fma_f32 v,s,v,v", k<5>, w);
        run("v_mov_b32", k<6>, w);
        run("v_cndmask_b32 vcc", k<7>, w);
        run("v_exp_f32", k<8>, w);
        run("v_rcp_f32", k<9>, w);
        run("v_sqrt_f32", k<10>, w);
        run("v_cvt_u32_f32", k<11>, w);
        run("v_mov_b32_dpp quad_perm", k<13>, w);
        run("v_pk_add_f32 v,v,v", kp<0>, w);
        run("v_pk_mul_f32 v,v,v", kp<1>, w);
        run("v_pk_add_f32 v,s,v", kp<2>, w);
        run("v_pk_fma_f32 v,v,v,v", kp<3>, w);
        run("v_pk_add_f32 neg", kp<4>, w);
        run("v_pk_add_f32 op_sel swz", kp<5>, w);
    }
    return 0;
}
