// fft_wave.hh -- the 4096-point transform with ONE WAVEFRONT PER TRANSFORM (gfx950, wave64).
//
// What it replaces: the same reference functions as fft_lds.hh (pocketfft::c2c behind FftImplNativeCpu::kernelC2C,
// src/domains/dsp/fft/module_impl_native_cpu.cc:125-140, with Multiply / Amplitude / Range fused through the same
// prologue / epilogue functors).  The ARITHMETIC is unchanged -- four radix-8 passes in pocketfft's order
// (pocketfft.hh:1141-1225), the same twiddle table, the same butterflies (device_math.hh), no FMA -- so the output bits
// are those of fft_pipe_kernel.  The MAPPING is new, and it is the one 4096 = 64 x 64 and a 64-lane wavefront ask for:
//
//   * lane i holds column i of the transform seen as a 64 x 64 matrix: the 64 elements i + 64 m (one 512-byte run per
//     wave-wide load instruction).  Pass 0 (ido = 512) and pass 1 (ido = 64) only ever combine elements of ONE column
//     (butterfly (i + 64 b) of pass 0 reads m = b + 8 b', butterfly (i, k) of pass 1 reads m = b' + 8 k): both run in
//     registers, eight independent radix-8 butterflies per pass and lane, no exchange in between;
//   * ONE transposition through LDS (column i -> row k: lane k then holds the elements 64 k + j), as two F32 planes
//     through the same 64 x 65-word buffer (pitch 65: the lane-contiguous writes and the stride-65 reads are both
//     conflict-free); a wavefront's LDS operations execute in order, so the exchange needs NO barrier -- there is
//     no s_barrier anywhere in the transform loop;
//   * pass 2 (ido = 8) and pass 3 (ido = 1) only combine elements of one ROW: in registers again; the twiddles of
//     pass 2 do not depend on the lane (W[64 c i''], i'' = the butterfly's position in the lane's row): 64 table
//     entries in LDS, read as broadcasts (as literal operands they were hoisted into ~100 SGPRs and spilled);
//   * lane k ends up with the outputs k + 64 r: a wave-wide store is one 256-byte run of the F32 row (and one 64-byte
//     run of the one-byte side output), as in fft_pipe_kernel;
//   * as the outputs of the last pass retire, their registers take the NEXT transform's input loads (all 64 in flight
//     by the end of the epilogue): the HBM round trip hides behind the epilogue.
//
// Against fft_pipe_kernel (8 wavefronts per transform, 3 LDS exchanges with a workgroup barrier each): a third of the
// LDS traffic, no barriers, eight independent butterflies of instruction-level parallelism per wavefront instead of one,
// and every address a base register plus an immediate.  Cost: 128 data VGPRs per lane -> 2 wavefronts per SIMD
// (launch bounds 512 x 2), the pass-0 twiddles W[c (i + 64 b)] (lane dependent, 56 per lane) from a 28 KiB LDS table,
// the window operand from L1/L2 once per transform.
// LDS: 28 KiB twiddle table + 8 wavefronts x 16.25 KiB transposition planes = 158 KiB of the CU's 160: one workgroup
// of 8 wavefronts per CU.
#pragma once

#include "fft_lds.hh"

#ifndef JST_WAVE_PREFETCH  // A/B switch: the next transform's loads behind the retiring outputs (1) or after the epilogue (0)
#define JST_WAVE_PREFETCH 1
#endif

#ifndef JST_WAVE_PASS_SB_OFF  // A/B switch: no scheduling barrier between the butterflies of passes 0-2
#define JST_WAVE_PASS_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define JST_WAVE_PASS_SB() do {} while (0)
#endif

namespace jst::dev {

constexpr int kWaveN = 4096;
constexpr int kWaveWaves = 8;                   // wavefronts per workgroup (one workgroup per CU)
constexpr int kWaveTw0Entries = 7 * 512;        // W[c i]: c = 1..7 (major), i = 0..511
constexpr int kWaveTw2Entries = 64;             // W[64 k], k = 0..63: pass 2's twiddles (lane independent)
constexpr int kWavePitch = 65;                  // words per row of a transposition plane
constexpr int kWavePlaneWords = 64 * kWavePitch;
constexpr size_t fft_wave_lds_bytes() {
    return (size_t)(kWaveTw0Entries + kWaveTw2Entries) * sizeof(float2) + (size_t)kWaveWaves * kWavePlaneWords * sizeof(float);
}

template <class Pro>
constexpr bool pro_real_operand() {
    if constexpr (requires { Pro::kRealOperand; }) return Pro::kRealOperand;
    else return false;
}

// Byte offset of element lane + 64 m of a row of BYTES-byte elements, split the way a buffer instruction takes it: the
// per-lane part with the low twelve bits of the wave-uniform part folded into the instruction's immediate field
// (lane_off) and a multiple of 4096 as the scalar offset (wave_soff) -- 64 distinct scalar offsets per access stream
// were 64 SGPRs each, hoisted out of the transform loop and spilled.
template <uint32_t BYTES>
__device__ __forceinline__ uint32_t lane_off(int lane, int m) {
    return (uint32_t)lane * BYTES + (((uint32_t)m * 64u * BYTES) & 4095u);
}
template <uint32_t BYTES>
constexpr uint32_t wave_soff(int m) {
    return ((uint32_t)m * 64u * BYTES) & ~4095u;
}

// Pins a value to the place it is computed: the epilogue's branches (cold paths, wave-uniform tests) split the transform
// loop into hundreds of basic blocks, and LLVM's code sinking moves every computation whose users all sit in later
// blocks down to them -- e.g. seven eighths of pass 2 below the first epilogue branch, with all of its inputs (128
// registers) and twiddles kept live on the way.  An empty volatile asm that "rewrites" the value is a use in this block.
__device__ __forceinline__ void pin(float2& v) { __asm__ volatile("" : "+v"(v.x), "+v"(v.y)); }

// orders the compiler's LDS accesses of ONE wavefront (the hardware executes them in order: nothing is emitted)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

template <bool FWD, class Pro, class Epi>
__global__ __launch_bounds__(kWaveWaves * 64, 2) void fft_wave4096_kernel(const FftLayout L, const float2* __restrict__ W,
                                                                          const Pro pro, const Epi epi_arg) {
    constexpr int N = kWaveN;
    Epi epi = epi_arg;  // this thread's copy: an epilogue may pin constants in VGPRs for the whole transform loop
    if constexpr (requires { epi.pin_constants(); }) epi.pin_constants();
    constexpr uint32_t RB = Pro::kRawBytes;
    constexpr uint32_t EB = Epi::kElemBytes;
    using raw_t = typename Pro::raw_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* tw0 = reinterpret_cast<float2*>(smem_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    float2* tw2 = tw0 + kWaveTw0Entries;
    float* plane = reinterpret_cast<float*>(tw2 + kWaveTw2Entries) + wave * kWavePlaneWords;

    // pass-0 twiddle table (shared by the workgroup) and this lane's pass-1 twiddles W[8 c lane]
#pragma unroll
    for (int q = 0; q < kWaveTw0Entries / (kWaveWaves * 64); ++q) {
        const int e = tid + q * kWaveWaves * 64;   // q = c - 1 (512 entries per c, 512 threads)
        tw0[e] = W[(unsigned)((q + 1) * (e & 511))];
    }
    if (tid < kWaveTw2Entries) tw2[tid] = W[(unsigned)(64 * tid)];
    float2 tw1[8];
#pragma unroll
    for (int c = 1; c < 8; ++c) tw1[c] = W[(unsigned)(c * 8 * lane)];
    __syncthreads();

    // wavefront w of workgroup g takes the transforms g + grid * (w + 8 j): a launch with fewer transforms than wavefront
    // slots spreads over all CUs first
    const uint32_t slots = gridDim.x * kWaveWaves;
    uint64_t t = (uint64_t)wave * gridDim.x + blockIdx.x;
    if (t >= L.transforms) return;

    const float2* tw0l = tw0 + lane;
    float* wr_plane = plane + lane;                  // element (q, lane) at + 65 q
    const float* rd_plane = plane + lane * kWavePitch;  // element (lane, j) at + j
    const rsrc_t r_opnd = make_rsrc(pro.operand_row(), (uint32_t)N * 8u);

    // dense [B, N] tensors only (one batch axis; the launcher checks): row of transform t, possibly on a ring.  BRANCH-FREE
    // (no ring = a ring of 2^32 - 1 rows starting at 0: rows are below 2^31): a branch here splits the transform loop's
    // body into basic blocks, and the compiler then sinks a whole pass of arithmetic below the split while the loads
    // that feed it stay above -- every loaded value live at once (172 spilled VGPRs in the first build).
    const uint32_t ring_rows = L.ring_transforms ? (uint32_t)L.ring_transforms : 0xffffffffu;
    const uint32_t ring_first = L.ring_transforms ? (uint32_t)L.ring_first : 0u;
    const auto ring_row = [&](uint64_t tt) { return (ring_first + (uint32_t)tt) % ring_rows; };
    const auto bases = [&](uint64_t tt, int64_t& ib, int64_t& ob) {
        const uint64_t row = ring_row(tt);
        ib = (int64_t)L.in_offset + (int64_t)row * L.in_outer_stride[0];
        ob = (int64_t)L.out_offset + (int64_t)row * L.out_outer_stride[0];
    };
    int64_t in_base, out_base;
    bases(t, in_base, out_base);
    raw_t raw[64];
    {
        const rsrc_t r_in = make_rsrc(pro.row(in_base), (uint32_t)N * RB);
#pragma unroll
        for (int m = 0; m < 64; ++m) raw[m] = Pro::load_raw_buf(r_in, lane_off<RB>(lane, m), wave_soff<RB>(m));
    }

    while (true) {
        // ---- prologue: x[m] = input[lane + 64 m] (x) operand[lane + 64 m], the operand in two halves from L1 / L2 ----
        float2 x[64];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float2 w[32];
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                if constexpr (!Pro::kHasOperand) w[m] = mk(0.0f, 0.0f);
                else if constexpr (pro_real_operand<Pro>())
                    w[m] = mk(u2f(__builtin_amdgcn_raw_buffer_load_b32(r_opnd, lane_off<8>(lane, h * 32 + m), wave_soff<8>(h * 32 + m), 0)), 0.0f);
                else
                    w[m] = buf_load_f2(r_opnd, lane_off<8>(lane, h * 32 + m), wave_soff<8>(h * 32 + m));
            }
#pragma unroll
            for (int m = 0; m < 32; ++m) x[h * 32 + m] = pro.apply(raw[h * 32 + m], w[m]);
        }

        // ---- pass 0: butterfly i = lane + 64 b reads m = b + 8 b', writes m = b + 8 c, twiddle W[c i] ----------------
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            float2 y[8], w[8];
#pragma unroll
            for (int c = 1; c < 8; ++c) w[c] = tw0l[(c - 1) * 512 + 64 * b];
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) y[bb] = x[b + 8 * bb];
            butterfly<8, FWD>(y);
            if (b == 0) {  // i == 0 only here, in lane 0: pocketfft leaves those outputs untouched (pocketfft.hh:1141-1225)
                twiddle_inplace3<FWD>((unsigned)lane, y[1], y[2], y[3], w[1], w[2], w[3]);
                twiddle_inplace4<FWD>((unsigned)lane, y[4], y[5], y[6], y[7], w[4], w[5], w[6], w[7]);
            } else {
#pragma unroll
                for (int c = 1; c < 8; ++c) y[c] = special_mul<FWD>(y[c], w[c]);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) x[b + 8 * c] = y[c];
            JST_WAVE_PASS_SB();
        }

        // ---- pass 1: butterfly (lane, k) reads m = b' + 8 k, writes q = k + 8 c, twiddle W[8 c lane] -----------------
        float2 z[64];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float2 y[8];
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) y[bb] = x[bb + 8 * k];
            butterfly<8, FWD>(y);
            twiddle_inplace3<FWD>((unsigned)lane, y[1], y[2], y[3], tw1[1], tw1[2], tw1[3]);
            twiddle_inplace4<FWD>((unsigned)lane, y[4], y[5], y[6], y[7], tw1[4], tw1[5], tw1[6], tw1[7]);
#pragma unroll
            for (int c = 0; c < 8; ++c) z[k + 8 * c] = y[c];
            JST_WAVE_PASS_SB();
        }

        // ---- transposition: lane i holds (q, i) for all q; lane k receives (k, j) for all j -- real parts, then imaginary
        // volatile: ONE ds_write_b32 / ds_read_b32 per element from one base register each (16-bit immediate offsets).
        // Left to pair them into ds_write2 / ds_read2 (8-bit offsets in dwords) the compiler materialised 48 base
        // addresses in VGPRs -- at the point of the kernel where all 128 data registers are live.
        typedef volatile __attribute__((address_space(3))) float* lds_f32_ptr;
        typedef const volatile __attribute__((address_space(3))) float* lds_cf32_ptr;
        float2 v[64];
#pragma unroll
        for (int q = 0; q < 64; ++q) *(lds_f32_ptr)(wr_plane + q * kWavePitch) = z[q].x;
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 64; ++j) v[j].x = *(lds_cf32_ptr)(rd_plane + j);
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 64; ++q) *(lds_f32_ptr)(wr_plane + q * kWavePitch) = z[q].y;
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 64; ++j) v[j].y = *(lds_cf32_ptr)(rd_plane + j);
        wave_lds_fence();

        // ---- pass 2: butterfly (i, lane) reads j = i + 8 b', twiddle W[64 c i] (lane independent); its output c is
        //      input i of pass-3 butterfly lane + 64 c
        float2 u[64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float2 y[8];
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) y[bb] = v[i + 8 * bb];
            butterfly<8, FWD>(y);
            if (i > 0) {  // every lane reads the same entry: one broadcast LDS access each
                float2 w[8];
#pragma unroll
                for (int c = 1; c < 8; ++c) w[c] = tw2[c * i];
#pragma unroll
                for (int c = 1; c < 8; ++c) y[c] = special_mul<FWD>(y[c], w[c]);
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pin(y[c]);
                u[c * 8 + i] = y[c];
            }
            JST_WAVE_PASS_SB();
        }

        // ---- next transform's loads ride behind the retiring outputs ------------------------------------------------
        const uint64_t tn = t + slots;
        const bool more = tn < L.transforms;
        int64_t nin = 0, nout = 0;
        bases(more ? tn : t, nin, nout);
        // past the last transform the descriptor has zero records: the loads return 0 without touching memory (no
        // conditional load: see fft_lds.hh on what a phi behind a load costs)
        const rsrc_t r_next = make_rsrc(pro.row(nin), more ? (uint32_t)N * RB : 0u);
        const rsrc_t r_out = make_rsrc(epi.row(out_base), (uint32_t)N * EB);
        rsrc_t r_side = r_out;
        uint32_t side_group_stride = 0;
        if constexpr (epi_has_side<Epi>()) {
            r_side = epi.template side_rsrc<false>(ring_row(t), (uint32_t)N, 0);
            side_group_stride = epi.side_pitch * 128u;
            // not a loop invariant for the compiler: hoisted, the 32 products (r >> 1) * stride took 32 SGPRs (spilled)
            __asm__ volatile("" : "+s"(side_group_stride));
        }

        // ---- pass 3 (no twiddles) + epilogue: butterfly lane + 64 c, output c3 is element lane + 64 (c + 8 c3) ---------
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float2 y[8];
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) y[bb] = u[c * 8 + bb];
            butterfly<8, FWD>(y);
#pragma unroll
            for (int c3 = 0; c3 < 8; ++c3) pin(y[c3]);
#pragma unroll
            for (int c3 = 0; c3 < 8; ++c3) {
                const int r = c + 8 * c3;
                if constexpr (epi_has_side<Epi>()) {
                    float val;
                    uint32_t index;
                    epi.side_compute(y[c3], val, index);
                    buf_store_f1(r_out, lane_off<4>(lane, r), wave_soff<4>(r), val);
                    buf_store_u8(r_side, (uint32_t)lane + (uint32_t)(r & 1) * 64u, (uint32_t)(r >> 1) * side_group_stride, index);
                } else {
                    epi.store_buf(r_out, lane_off<EB>(lane, r), wave_soff<EB>(r), y[c3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#if JST_WAVE_PREFETCH
#pragma unroll
            for (int q = 0; q < 8; ++q)
                raw[c * 8 + q] = Pro::load_raw_buf(r_next, lane_off<RB>(lane, c * 8 + q), wave_soff<RB>(c * 8 + q));
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if (!more) break;
        t = tn;
        out_base = nout;
#if !JST_WAVE_PREFETCH
#pragma unroll
        for (int m = 0; m < 64; ++m) raw[m] = Pro::load_raw_buf(r_next, lane_off<RB>(lane, m), wave_soff<RB>(m));
#endif
    }
}

}  // namespace jst::dev
