// fft_split_experiment.hh -- a REJECTED experiment of round 3, kept outside the product (tools/ubench/fused_bench.hip
// -DFB_SPLIT builds it): the role-split form of the fused 4096-point kernel.  Bit-identical to the pipelined kernel
// (same checksums) but slower on MI355X: exact 21.9 vs 19.7 us, lean fast 22.5 vs 17.7 us, trivial epilogue 16.0 vs
// 15.6 us per 1024 x 4096 launch, same box (profiles/r03_experiments/e_role_split_kernel.log).  Why: a radix-8 pass on
// eight wavefronts is LATENCY bound (LDS round trip -> ~110 dependent VALU instructions -> LDS write -> barrier:
// ~1200 cycles even with an idle VALU), so a CU needs at least two transforms in their passes at once to hide it; the
// pipelined kernel has them (two workgroups), this form has one FFT role per CU (109 KiB of LDS per workgroup) and its
// stage time -- ~3.9 us per transform against ~2.75 us per transform and CU for the pipelined kernel -- is set by that
// chain, whatever the epilogue weighs.  Two FFT roles plus an epilogue role per CU would need 217 KiB of LDS.
#pragma once
#include "fft_lds_r03_variants.hh"  // the round-3 header with its A/B switches (the product header dropped them)

namespace jst::dev {

// =============================================================================================
// Role-split variant of the pipelined kernel for N = 4096 on dense rows (round 3).
//
// What the timelines of the pipelined kernel show (profiles/r03_experiments/b_timeline.log): its two co-resident
// workgroups run the same phases at the same time -- four passes whose exchanges (write drain, barrier skew, LDS read
// latency: ~750 cycles each) leave the VALU idle in BOTH, then two epilogues that saturate it together -- and the
// launch lasts as long as one workgroup's serial chain load -> 4 passes -> epilogue -> 4 passes -> epilogue.  Here
// ONE workgroup of 16 wavefronts per CU splits into two roles that run DIFFERENT phases at the same time:
//   * wavefronts 0..7, the FFT role: the pipelined kernel's transform (prefetch registers, register / LDS twiddles,
//     ping-pong exchange buffers) up to the last butterfly, whose outputs go to a hand-off buffer H in LDS
//     (H[c][u]: lane-contiguous, conflict-free) instead of through the epilogue;
//   * wavefronts 8..15, the EPILOGUE role: one transform behind, thread u takes the eight outputs u + 512 c out of H
//     into registers and runs Amplitude / Range / store on them, two per interval.
// A stage (one transform) is four intervals closed by four workgroup barriers: the three exchange barriers the FFT
// role needs anyway plus the hand-off barrier; the epilogue role passes the same four, so the hardware barrier keeps
// the two roles in step as a two-deep software pipeline across wavefronts.  In every interval a SIMD holds two FFT
// wavefronts (short VALU bursts around LDS round trips, priority 3) and two epilogue wavefronts (long dependent
// VALU streams, priority 0): the epilogue fills the exchange gaps of the passes, and the chain of a CU's four
// transforms becomes  load -> passes -> 3 x (passes || epilogue) -> epilogue.  Same arithmetic in the same order as
// every other path: bit-identical results.  LDS: 2 x 36 KiB exchange + 4.5 KiB twiddles + 32 KiB hand-off = 109 KiB,
// one workgroup per CU; <= 128 VGPRs (16 wavefronts per CU).
template <int N, bool FWD, class Pro, class Epi>
__global__ __launch_bounds__(N / 4, 4) void fft_split_kernel(const FftLayout L, const float2* __restrict__ W,
                                                            const Pro pro, const Epi epi) {
    constexpr int T = N / 8;
    constexpr Plan plan = make_plan(N);
    constexpr TwPlan tp = make_twplan(N);
    static_assert(plan.nf == 4 && plan.ip[0] == 8 && plan.ip[1] == 8 && plan.ip[2] == 8 && plan.ip[3] == 8,
                  "the role-split kernel is written for four radix-8 passes (N = 4096)");
    static_assert(tp.reg_off[0] >= 0 && tp.lds_off[1] >= 0 && tp.lds_off[2] >= 0, "twiddle placement of N = 4096");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* bufA = reinterpret_cast<float2*>(smem_raw);
    float2* bufB = bufA + lds_elems(N);
    float2* twl = bufB + lds_elems(N);
    float2* H = twl + tp.lds_entries;
    const int role = (int)threadIdx.x / T;  // wave-uniform: 0 = FFT, 1 = epilogue
    const int tid = (int)threadIdx.x - role * T;
    const uint32_t bid = blockIdx.x, grid = gridDim.x;
    const uint32_t count = bid < L.transforms ? (uint32_t)((L.transforms - bid + grid - 1) / grid) : 0u;
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_ptr;
    constexpr uint32_t RB = Pro::kRawBytes;

    if (role == 0) {
        // ================================= FFT role ===============================================
        __builtin_amdgcn_s_setprio(3);
        float2 tw0[7];  // pass-0 twiddles of butterfly i = tid (IDO0 = T)
#pragma unroll
        for (int c = 1; c < 8; ++c) tw0[c - 1] = W[(unsigned)(c * plan.l1[0]) * (unsigned)tid];
        constexpr int TWL_PER_THREAD = (tp.lds_entries + T - 1) / T;
        float2 twv[TWL_PER_THREAD + 1];
#pragma unroll
        for (int q = 0; q < TWL_PER_THREAD; ++q) {
            const int g = tid + q * T;
            unsigned widx = 0;
#pragma unroll
            for (int p = 1; p < 3; ++p) {
                const int entries = plan.ido[p] * 7;
                const int e = g - tp.lds_off[p];
                if (e >= 0 && e < entries) widx = ((unsigned)(e % 7) + 1u) * (unsigned)plan.l1[p] * (unsigned)(e / 7);
            }
            twv[q] = W[widx];
        }
        typename Pro::raw_t raw[8];
        float2 opnd[8];
        const rsrc_t r_opnd0 = make_rsrc(pro.operand_row(), count ? (uint32_t)N * 8u : 0u);
        {
            int64_t in_base, out_base;
            fft_bases(L, count ? bid : 0, in_base, out_base);
            const rsrc_t r_in = make_rsrc(pro.row(in_base), count ? (uint32_t)N * RB : 0u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (Pro::kHasOperand) opnd[e] = buf_load_f2(r_opnd0, (uint32_t)tid * 8u, (uint32_t)(T * e) * 8u);
                raw[e] = Pro::load_raw_buf(r_in, (uint32_t)tid * RB, (uint32_t)(T * e) * RB);
            }
        }
#pragma unroll
        for (int q = 0; q < TWL_PER_THREAD; ++q)
            if (tid + q * T < tp.lds_entries) twl[tid + q * T] = twv[q];
        for (uint32_t s = 0; s <= count; ++s) {
            if (s < count) {
                float2 x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = pro.apply(raw[e], opnd[e]);
                {   // prefetch the next transform of this workgroup (zero-record descriptor past the last one)
                    const bool more = s + 1 < count;
                    int64_t nin, nout;
                    fft_bases(L, (uint64_t)bid + (uint64_t)(more ? s + 1 : s) * grid, nin, nout);
                    const rsrc_t r_in = make_rsrc(pro.row(nin), more ? (uint32_t)N * RB : 0u);
#pragma unroll
                    for (int e = 0; e < 8; ++e) raw[e] = Pro::load_raw_buf(r_in, (uint32_t)tid * RB, (uint32_t)(T * e) * RB);
                }
                // ---- pass 0: IDO = T, butterfly u = tid, twiddles in registers ----------------------------
                butterfly<8, FWD>(x);
                twiddle_inplace3<FWD>((unsigned)tid, x[1], x[2], x[3], tw0[0], tw0[1], tw0[2]);
                twiddle_inplace4<FWD>((unsigned)tid, x[4], x[5], x[6], x[7], tw0[3], tw0[4], tw0[5], tw0[6]);
                {
                    float2* wr = bufA + pphys(tid);
#pragma unroll
                    for (int c = 0; c < 8; ++c) wr[pcphys(c * T)] = x[c];
                }
                lds_barrier();
                // ---- passes 1 and 2: exchange read, butterfly, LDS twiddles, exchange write -----------------
#pragma unroll
                for (int P = 1; P <= 2; ++P) {
                    const int IDO = P == 1 ? plan.ido[1] : plan.ido[2];
                    float2* src = P == 1 ? bufA : bufB;
                    float2* dst = P == 1 ? bufB : bufA;
                    const unsigned i = (unsigned)(tid & (IDO - 1));
                    const int k = tid / IDO;
                    const float2* rd = src + pphys((int)i + IDO * 8 * k);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const unsigned long long bits = *(lds_u64_ptr)(rd + pcphys(IDO * b));
                        x[b] = __builtin_bit_cast(float2, bits);
                    }
                    float2 w[8];
                    const int off = P == 1 ? tp.lds_off[1] : tp.lds_off[2];
#pragma unroll
                    for (int c = 1; c < 8; ++c) w[c] = twl[off + (int)i * 7 + (c - 1)];
                    butterfly<8, FWD>(x);
                    twiddle_inplace3<FWD>(i, x[1], x[2], x[3], w[1], w[2], w[3]);
                    twiddle_inplace4<FWD>(i, x[4], x[5], x[6], x[7], w[4], w[5], w[6], w[7]);
                    float2* wr = dst + pphys(tid);
#pragma unroll
                    for (int c = 0; c < 8; ++c) wr[pcphys(c * T)] = x[c];
                    lds_barrier();
                }
                // ---- pass 3: IDO = 1 (no twiddles); outputs to the hand-off buffer -------------------------
                {
                    const float2* rd = bufA + pphys(8 * tid);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const unsigned long long bits = *(lds_u64_ptr)(rd + pcphys(b));
                        x[b] = __builtin_bit_cast(float2, bits);
                    }
                    butterfly<8, FWD>(x);
#pragma unroll
                    for (int c = 0; c < 8; ++c) H[c * T + tid] = x[c];
                    if constexpr (Pro::kHasOperand) {  // the operand of the next transform (L2), into the registers just freed
                        const rsrc_t r_opnd = make_rsrc(pro.operand_row(), s + 1 < count ? (uint32_t)N * 8u : 0u);
#pragma unroll
                        for (int e = 0; e < 8; ++e) opnd[e] = buf_load_f2(r_opnd, (uint32_t)tid * 8u, (uint32_t)(T * e) * 8u);
                    }
                }
                lds_barrier();
            } else {  // drain stage: the epilogue role finishes the last transform
                lds_barrier();
                lds_barrier();
                lds_barrier();
                lds_barrier();
            }
        }
    } else {
        // ================================= epilogue role ==========================================
        __builtin_amdgcn_s_setprio(0);
        for (uint32_t s = 0; s <= count; ++s) {
            if (s >= 1) {
                int64_t in_base, out_base;
                fft_bases(L, (uint64_t)bid + (uint64_t)(s - 1) * grid, in_base, out_base);
                const rsrc_t r_out = make_rsrc(epi.row(out_base), (uint32_t)N * Epi::kElemBytes);
                float2 y[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const unsigned long long bits = *(lds_u64_ptr)(H + c * T + tid);
                    y[c] = __builtin_bit_cast(float2, bits);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int c = 2 * q; c < 2 * q + 2; ++c) {
                        epi.store_buf(r_out, (uint32_t)tid * Epi::kElemBytes, (uint32_t)(c * T) * Epi::kElemBytes, y[c]);
#ifndef JST_NO_EPI_SCHED_BARRIER
                        __builtin_amdgcn_sched_barrier(0);
#endif
                    }
                    lds_barrier();
                }
            } else {
                lds_barrier();
                lds_barrier();
                lds_barrier();
                lds_barrier();
            }
        }
    }
}
constexpr size_t fft_split_lds_bytes(int n) {
    return (2 * (size_t)lds_elems(n) + (size_t)make_twplan(n).lds_entries + (size_t)n) * sizeof(float2);
}

}  // namespace jst::dev
