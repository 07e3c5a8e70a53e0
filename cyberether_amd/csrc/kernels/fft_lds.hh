// fft_lds.hh -- LDS-resident Stockham FFT for power-of-two lengths on gfx950 (CDNA4).
//
// What it replaces: FftImplNativeCpu::kernelC2C -> pocketfft::c2c
// (src/domains/dsp/fft/module_impl_native_cpu.cc:125-140, pocketfft.hh:3387) and the cuFFT call
// sites (module_impl_native_cuda.cc:321,433,439,463).  NOT a translation of either: the pass
// *arithmetic* is pocketfft's (same factor order 8..8,4,[2 first], same butterflies, same
// twiddle table, no FMA) so every output bit equals the reference CPU path, but the execution
// is designed for CDNA4:
//
//   * one transform is owned by T = N/16 threads (4096-pt: 256 threads = 4 wavefronts), every
//     thread holding 16 complex points in VGPRs (two radix-8 butterflies per pass);
//   * passes exchange data through ONE LDS buffer per transform (N*9/8 cf32, 36 KiB at 4096 ->
//     4 transforms resident per CU, 16 waves/CU).  Stockham writes are always lane-contiguous
//     (u + c*N/ip); the strided reads of the late passes (ido = 8, 4, 1) are made bank-conflict
//     free for ds_read_b64 by padding one element per 8 (phys = p + p/8);
//   * the first pass reads HBM directly (64 lanes x 8 B = 512 B contiguous per instruction) and
//     can apply the Multiply module's broadcast window on the fly; the last pass feeds its
//     outputs to an epilogue functor (plain store, Amplitude, Amplitude+Range) so that
//     Window(.)x -> FFT -> Amplitude -> Range moves 8 B in + 4 B out per sample and nothing else;
//   * twiddles come from a per-length table W[k] = exp(+j 2 pi k/N) built on the host with
//     pocketfft's own two-table double-precision scheme (fft_plan.cc) and stay L2/L1 resident.
#pragma once

#include "device_math.hh"
#include "kernels.hh"

namespace jst::dev {

// Row of the tensors that transform t of the launch works on: t itself, or -- a span of a ring (FftLayout::ring_*) --
// (ring_first + t) mod ring_transforms.  Both below 2^31 (the launchers check), so the remainder is a 32-bit one.
__device__ __forceinline__ uint64_t fft_ring_row(const FftLayout& L, uint64_t t) {
    if (L.ring_transforms == 0) return t;
    return (uint64_t)(((uint32_t)L.ring_first + (uint32_t)t) % (uint32_t)L.ring_transforms);
}
__device__ __forceinline__ void fft_bases(const FftLayout& L, uint64_t t, int64_t& in_base,
                                          int64_t& out_base) {
    in_base = (int64_t)L.in_offset;
    out_base = (int64_t)L.out_offset;
    if (L.outer_rank == 1) {  // one batch axis (every dense [B, N] tensor): no 64-bit divisions
        t = fft_ring_row(L, t);
        in_base += (int64_t)t * L.in_outer_stride[0];
        out_base += (int64_t)t * L.out_outer_stride[0];
        return;
    }
    for (int a = L.outer_rank - 1; a >= 0; --a) {
        const uint64_t c = t % L.outer_shape[a];
        t /= L.outer_shape[a];
        in_base += (int64_t)c * L.in_outer_stride[a];
        out_base += (int64_t)c * L.out_outer_stride[a];
    }
}

// ---- compile-time plan: pocketfft cfftp::factorize for n = 2^m (pocketfft.hh:1476-1497) -------
struct Plan {
    int nf;
    int ip[8];
    int l1[8];
    int ido[8];
};
constexpr Plan make_plan(int n) {
    Plan p{};
    int len = n, nf = 0;
    while ((len & 7) == 0) { p.ip[nf++] = 8; len >>= 3; }
    while ((len & 3) == 0) { p.ip[nf++] = 4; len >>= 2; }
    if ((len & 1) == 0) {
        len >>= 1;
        p.ip[nf++] = 2;
        const int t = p.ip[0];
        p.ip[0] = p.ip[nf - 1];
        p.ip[nf - 1] = t;
    }
    p.nf = nf;
    int l1 = 1;
    for (int k = 0; k < nf; ++k) {
        p.l1[k] = l1;
        p.ido[k] = n / (l1 * p.ip[k]);
        l1 *= p.ip[k];
    }
    return p;
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a full workgroup fence:
// hipcc emits s_waitcnt vmcnt(0) in front of s_barrier, which would drain the in-flight
// prefetch of the next transform (and the previous transform's output stores) at every
// exchange.  The passes only communicate through LDS, so release/acquire on the local address
// space is all that is needed: s_waitcnt lgkmcnt(0) + s_barrier.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS physical index: one pad element per 8 keeps the stride-8 / stride-64 read patterns of the
// ido = 1 and ido = 8 passes on distinct banks (see header comment).
__device__ __forceinline__ int phys(int p) { return p + (p >> 3); }
constexpr int lds_elems(int n) { return n + (n >> 3); }

// Buffer-descriptor addressing for the dense (CONTIG) case: one 128-bit descriptor in SGPRs per
// tensor row, a single 32-bit per-lane byte offset shared by all of a thread's accesses and a
// wave-uniform (SGPR/immediate) offset per access.  With flat addressing hipcc materialises a
// 64-bit VGPR address pair per access and carries them through the transform loop.
using rsrc_t = __amdgpu_buffer_rsrc_t;
// Cache policy of the kernels' output stores (gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1).  sc1 = AGENT scope: the
// store is written through to the device-coherent level as it is issued.  With the default (wave scope) the 16 MiB of
// a launch's output sit dirty in the eight per-XCD L2s until the end-of-kernel release writes them back, and that
// write-back is serial time at the end of every launch: 24.0 -> 22.3 us per 1024 x 4096 launch by events, step
// 29.3 -> 27.5 us, same box (profiles/r02_experiments/r_store_policy.log).  nt (streaming) shortens the kernel as
// much but evicts the lines, and the Spectrogram that reads them next pays it back (9.6 -> 11.0 us).
#ifndef JST_STORE_AUX
#define JST_STORE_AUX 16
#endif
// The same policy for plain global stores: a relaxed atomic store at agent scope is `global_store ... sc1`, nothing else.
__device__ __forceinline__ void store_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_agent(float2* p, float2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
#ifndef JST_LOAD_AUX  // A/B switch: cache policy of the 8-byte stream loads (2 = nt)
#define JST_LOAD_AUX 0
#endif
#ifndef JST_SIDE_STORE_AUX  // A/B switch: cache policy of the one-byte side-output stores
#define JST_SIDE_STORE_AUX JST_STORE_AUX
#endif
__device__ __forceinline__ float2 buf_load_f2(rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff_bytes, soff_bytes, 0));
    return mk(v.x, v.y);
}
// the input stream: every element is read once, by one CU (the window operand and tables keep the default policy)
__device__ __forceinline__ float2 buf_load_f2_stream(rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff_bytes, soff_bytes, JST_LOAD_AUX));
    return mk(v.x, v.y);
}
__device__ __forceinline__ void buf_store_f2(rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes, float2 v) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(v2u{f2u(v.x), f2u(v.y)}, r, voff_bytes, soff_bytes, JST_STORE_AUX);
}
__device__ __forceinline__ void buf_store_f1(rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(f2u(v), r, voff_bytes, soff_bytes, JST_STORE_AUX);
}
__device__ __forceinline__ void buf_store_u8(rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)v, r, voff_bytes, soff_bytes, JST_SIDE_STORE_AUX);
}
// Measured and rejected, kept out of this header (tools/ubench/fft_lds_r03_variants.hh has them as A/B switches, logs in
// profiles/r03_experiments/): 16-byte loads / stores with quad transposes in front (+0.5..1.0 us per 1024 x 4096 launch:
// the kernel is bound by its per-workgroup latency chain, not by request count), the device-decided real-operand form
// (123 VGPRs cost more than the re-requests), exec-branch / select forms of the twiddle multiply, paired LDS reads,
// the one-pad-per-8 exchange layout, plain (wave-scope) stores.
// The Multiply operand of the prologue (the window taps of this thread's eight pass-0 positions) stays in VGPRs across
// the transforms of a workgroup instead of being re-requested from L2 behind every retired output: a thread visits
// the same positions in every transform.  Round 2 could not afford the 16 registers (127 VGPRs); the in-place twiddles
// left room (104).  In the memory-only skeleton the re-requests cost 2.4 us per launch and their L2 round trip is
// exposed at the top of every transform (profiles/r03_experiments/a_floor_bisect.log, rows D2 -> E1).
#ifndef JST_OPND_RESIDENT
#define JST_OPND_RESIDENT 0
#endif

// ---- prologues (how pass 0 obtains CC(i,b,k)) -------------------------------------------------
// operator()(base, axis_stride, pos): element 'pos' along the transform axis of the transform
// whose first element sits at 'base'.  CONTIG instantiations (axis stride 1 on every operand)
// index with 32-bit offsets from a per-transform pointer so that addresses are base + constant.
struct LoadCF32 {
    const float2* in;
    using raw_t = float2;  // what the prefetch registers of the pipelined kernel hold
    static constexpr uint32_t kRawBytes = 8;
    static __device__ __forceinline__ raw_t load_raw_buf(rsrc_t r, uint32_t voff, uint32_t soff) { return buf_load_f2_stream(r, voff, soff); }
    template <bool CONTIG>
    __device__ __forceinline__ float2 load(int64_t base, int64_t axis_stride, int pos) const {
        if constexpr (CONTIG) return (in + base)[(unsigned)pos];
        else return in[base + (int64_t)pos * axis_stride];
    }
    // pipelined kernel: raw load (prefetchable), per-position operand (none), apply (identity)
    // position = upos (wave-uniform) + lpos (per lane)
    __device__ __forceinline__ const void* row(int64_t base) const { return in + base; }
    __device__ __forceinline__ const void* operand_row() const { return in; }
    template <bool CONTIG>
    __device__ __forceinline__ float2 load_raw(int64_t base, int64_t axis_stride, int upos,
                                               int lpos) const {
        return in[base + (int64_t)(upos + lpos) * axis_stride];
    }
    static constexpr bool kHasOperand = false;
    template <bool CONTIG>
    __device__ __forceinline__ float2 load_operand(int, int) const { return mk(0.0f, 0.0f); }
    __device__ __forceinline__ float2 apply(float2 v, float2) const { return v; }
};
// Multiply fused in: signal[...,n] * window[n] with the Multiply module's arithmetic
// (core/multiply/module_impl_native_cpu.cc:94-100); window broadcast over every outer axis
// (stride 0 there), element stride wstride along the transform axis.
struct LoadCF32TimesWindow {
    const float2* in;
    const float2* window;
    int64_t wstride;
    using raw_t = float2;
    static constexpr uint32_t kRawBytes = 8;
    static __device__ __forceinline__ raw_t load_raw_buf(rsrc_t r, uint32_t voff, uint32_t soff) { return buf_load_f2_stream(r, voff, soff); }
    template <bool CONTIG>
    __device__ __forceinline__ float2 load(int64_t base, int64_t axis_stride, int pos) const {
        if constexpr (CONTIG) {
            return cmul_full((in + base)[(unsigned)pos], window[(unsigned)pos]);
        } else {
            return cmul_full(in[base + (int64_t)pos * axis_stride],
                             window[(int64_t)pos * wstride]);
        }
    }
    // pipelined kernel: raw load (prefetchable), then the window multiply (taps from L1/L2)
    __device__ __forceinline__ const void* row(int64_t base) const { return in + base; }
    __device__ __forceinline__ const void* operand_row() const { return window; }
    template <bool CONTIG>
    __device__ __forceinline__ float2 load_raw(int64_t base, int64_t axis_stride, int upos,
                                               int lpos) const {
        return in[base + (int64_t)(upos + lpos) * axis_stride];
    }
    static constexpr bool kHasOperand = true;
    template <bool CONTIG>
    __device__ __forceinline__ float2 load_operand(int upos, int lpos) const {
        return window[(int64_t)(upos + lpos) * wstride];
    }
    __device__ __forceinline__ float2 apply(float2 v, float2 w) const { return cmul_full(v, w); }
    // the same product for a REAL operand (imaginary part +-0), without the terms that are +-0: see RealOperand below
    __device__ __forceinline__ float2 apply_real(float2 v, float wx) const { return mk(v.x * wx, v.y * wx); }
};

// Raw SDR sample formats fused in: Cast (CI16 / CI8 / CU8 -> CF32, core/cast/module_impl_native_cpu.cc:150-229:
// static_cast<F32>(x) / scaler with scaler = 32768 or 128 -- a power of two, so x * (1 / scaler) is the same float,
// exactly) followed by the Multiply with the broadcast window.  The pipelined kernel prefetches the RAW words (4 or 2
// bytes per complex sample instead of 8: half or a quarter of the input stream, and of the prefetch registers) and
// converts when the transform starts.  RAW = uint32_t: CI16 (re in the low half); RAW = uint16_t: CI8 / CU8.
template <class RAW, bool SIGNED>
struct LoadCITimesWindow {
    const RAW* in;
    const float2* window;
    int64_t wstride;
    float inv_scaler;  // 1 / 32768 or 1 / 128
    using raw_t = RAW;
    static constexpr uint32_t kRawBytes = sizeof(RAW);
    static __device__ __forceinline__ raw_t load_raw_buf(rsrc_t r, uint32_t voff, uint32_t soff) {
        if constexpr (sizeof(RAW) == 4) return (RAW)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
        else return (RAW)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
    }
    __device__ __forceinline__ float2 convert(RAW w) const {
        float re, im;
        if constexpr (sizeof(RAW) == 4) {  // CI16: always signed
            re = (float)(int16_t)(w & 0xffffu);
            im = (float)(int16_t)(w >> 16);
        } else if constexpr (SIGNED) {
            re = (float)(int8_t)(w & 0xffu);
            im = (float)(int8_t)(w >> 8);
        } else {
            re = (float)(uint8_t)(w & 0xffu);
            im = (float)(uint8_t)(w >> 8);
        }
        return mk(re * inv_scaler, im * inv_scaler);
    }
    template <bool CONTIG>
    __device__ __forceinline__ float2 load(int64_t base, int64_t axis_stride, int pos) const {
        if constexpr (CONTIG) return cmul_full(convert((in + base)[(unsigned)pos]), window[(unsigned)pos]);
        else return cmul_full(convert(in[base + (int64_t)pos * axis_stride]), window[(int64_t)pos * wstride]);
    }
    __device__ __forceinline__ const void* row(int64_t base) const { return in + base; }
    __device__ __forceinline__ const void* operand_row() const { return window; }
    template <bool CONTIG>
    __device__ __forceinline__ raw_t load_raw(int64_t base, int64_t axis_stride, int upos, int lpos) const {
        return in[base + (int64_t)(upos + lpos) * axis_stride];
    }
    static constexpr bool kHasOperand = true;
    template <bool CONTIG>
    __device__ __forceinline__ float2 load_operand(int upos, int lpos) const {
        return window[(int64_t)(upos + lpos) * wstride];
    }
    __device__ __forceinline__ float2 apply(raw_t v, float2 w) const { return cmul_full(convert(v), w); }
    __device__ __forceinline__ float2 apply_real(raw_t v, float wx) const {
        const float2 c = convert(v);
        return mk(c.x * wx, c.y * wx);
    }
};
using LoadCI16TimesWindow = LoadCITimesWindow<uint32_t, true>;
using LoadCI8TimesWindow = LoadCITimesWindow<uint16_t, true>;
using LoadCU8TimesWindow = LoadCITimesWindow<uint16_t, false>;

// A prologue whose Multiply operand is known (on the host) to be REAL -- a window: every imaginary part +-0.  Provider
// "fast" only (bins exact, floats within tolerance): (a + bi)(c +- 0i) = (ac - b(+-0)) + (a(+-0) + bc)i, and for finite a, b
// the terms b(+-0), a(+-0) are zeros that change nothing but the sign of a zero result -- ac and bc ARE the products, bit for
// bit, wherever they are not zero.  A zero's sign can only turn into another zero's sign in the butterflies (sums,
// differences and products of zeros), and the power re^2 + im^2 the epilogue starts from does not see it: for finite input
// the output is bit-identical, with two VALU instructions per sample instead of seven (four products, two sums, the
// unordered test of __mulsc3's recovery branch) -- 40 of ~610 per wavefront and transform in a kernel that is VALU-issue
// bound in its steady state -- and the imaginary halves of the resident operand are dead: 8 VGPRs.
// Non-finite input (where inf * 0 = NaN and the recovery branch differ from a real multiply) is outside what provider
// "fast" promises bit for bit; provider "generic" never uses this.  A SEPARATE instantiation, chosen on the host: with
// both products in one kernel (decided per wavefront on the device) both operand forms stayed live across the loop --
// 128 VGPRs + 96 B of scratch, 203 instead of 187 us per 16384-transform launch.
template <class Pro>
struct RealOperand : Pro {
    static constexpr bool kRealOperand = true;  // only the operand's real parts are read (fft_wave.hh loads just those)
    __device__ __forceinline__ float2 apply(typename Pro::raw_t v, float2 w) const { return Pro::apply_real(v, w.x); }
};

// ---- epilogues (what happens to CH of the last pass) ------------------------------------------
struct StoreCF32 {
    float2* out;
    static constexpr uint32_t kElemBytes = 8;
    __device__ __forceinline__ const void* row(int64_t base) const { return out + base; }
    __device__ __forceinline__ void store_buf(rsrc_t r, uint32_t voff, uint32_t soff, float2 v) const {
        buf_store_f2(r, voff, soff, v);
    }
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t axis_stride, int pos,
                                          float2 v) const {
        if constexpr (CONTIG) store_agent(out + base + (unsigned)pos, v);
        else store_agent(out + (base + (int64_t)pos * axis_stride), v);
    }
};
template <bool FAST>
struct StoreAmplitudeT {  // Amplitude module fused (amplitude/module_impl_native_cpu.cc:73-86)
    float* out;
    float coeff;
    static constexpr uint32_t kElemBytes = 4;
    __device__ __forceinline__ const void* row(int64_t base) const { return out + base; }
    __device__ __forceinline__ float value(float2 v) const {
        return FAST ? amplitude_cf32_fast(v, coeff) : amplitude_exact(v, coeff);
    }
    __device__ __forceinline__ void store_buf(rsrc_t r, uint32_t voff, uint32_t soff, float2 v) const {
        buf_store_f1(r, voff, soff, value(v));
    }
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t axis_stride, int pos,
                                          float2 v) const {
        const float r = FAST ? amplitude_cf32_fast(v, coeff) : amplitude_exact(v, coeff);
        if constexpr (CONTIG) store_agent(out + base + (unsigned)pos, r);
        else store_agent(out + (base + (int64_t)pos * axis_stride), r);
    }
};
template <bool FAST>
struct StoreAmplitudeRangeT {  // Amplitude -> Range fused (range/module_impl_native_cpu.cc:67-82)
    float* out;
    float coeff, scale, offset;
    BinGuard guard;  // FAST only: heights of the Spectrogram consumers (device_math.hh)
    FastRangePoly poly = make_fast_range_poly(coeff, scale, offset);  // FAST only: folded constants, computed on the host
    static constexpr uint32_t kElemBytes = 4;
    __device__ __forceinline__ const void* row(int64_t base) const { return out + base; }
    // FAST: called once per thread on the kernel's own copy of the functor, before the transform loop.  The second
    // scalar operand of the polynomial's fused multiply-adds and the guard widths become VGPR residents (an empty asm
    // "rewrites" them, so they are neither re-materialised from SGPRs per use nor recomputed per output).
    __device__ __forceinline__ void pin_constants() {
        if constexpr (FAST) {
            guard.t0 = guard.h0 * 7.5e-7f;
            guard.t1 = guard.h1 * 7.5e-7f;
            __asm__ volatile("" : "+v"(poly.k2), "+v"(poly.k0), "+v"(guard.t0), "+v"(guard.t1));
        }
    }
    // the same with the second guard width left where it is (fft_quad.hh, at its VGPR limit: a second consumer height is the
    // rare case, and without one the width is never read)
    __device__ __forceinline__ void pin_constants_lean() {
        if constexpr (FAST) {
            guard.t0 = guard.h0 * 7.5e-7f;
            guard.t1 = guard.h1 * 7.5e-7f;
            __asm__ volatile("" : "+v"(poly.k2), "+v"(poly.k0), "+v"(guard.t0));
        }
    }
    __device__ __forceinline__ float value(float2 v) const {
        if constexpr (FAST) return amplitude_range_fast_guarded(v, coeff, scale, offset, guard, poly);
        else return amplitude_range_exact(v, coeff, scale, offset);
    }
    __device__ __forceinline__ void store_buf(rsrc_t r, uint32_t voff, uint32_t soff, float2 v) const {
        buf_store_f1(r, voff, soff, value(v));
    }
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t axis_stride, int pos,
                                          float2 v) const {
        const float r = value(v);
        if constexpr (CONTIG) store_agent(out + base + (unsigned)pos, r);
        else store_agent(out + (base + (int64_t)pos * axis_stride), r);
    }
};

// Amplitude -> Range with a SIDE OUTPUT for one known Spectrogram consumer: beside every F32 value the row index the
// Spectrogram derives from it (spectrogram/module_impl_native_cpu.cc:70-77: index = value * height, a hit when
// 0 < index < height) goes out as one byte, U8[transforms][N] dense, 0 = no hit.  The index is formed from the very
// float that is stored, so it is the consumer's own arithmetic moved in front of the store; the Spectrogram then reads
// 1 byte per sample (one 16-byte request per row and tile) instead of re-reading the 4-byte values.  height <= 256.
// Layout of the side tensor: U8[cycle][n / 128][side_pitch][128] (rows [0, side_batches) used) -- TILE-MAJOR: the 128 columns [128 q, 128 q + 128)
// of all rows of one compute cycle lie one behind the other (side_batches = the rows of one cycle; a cycle-batched launch
// carries several cycles).  The consumer works on tiles of 16 columns x ALL rows: with row-major indices each of its
// 16-byte requests opened another 4096-byte row (64 MiB per 16-cycle span at ~1.3 TB/s when they come from HBM: 50 us
// where the Infinity Cache would have served them in 30); tile-major it walks 128 KiB contiguously.  On this side
// nothing changes: a wavefront's 64 one-byte stores are still one 64-byte run, and the two wavefronts of a 128-column
// group still complete a 128-byte line together; the group index (tid >> 7, wave-uniform) goes into the descriptor base.
template <bool FAST>
struct StoreAmplitudeRangeSideT : StoreAmplitudeRangeT<FAST> {
    uint8_t* side;
    float side_height;
    uint32_t side_batches;
    uint32_t side_pitch;  // rows a column group occupies in memory (>= side_batches: see kernels.hh, the pad skews the groups over the HBM channels)
    static constexpr bool kHasSide = true;
    // IN_BASE: every thread stores ONE butterfly of the last pass (u = tid), so the column group u >> 7 is wave-uniform and
    // rides in the descriptor base; otherwise (several butterflies per thread) it is formed per lane.
    template <bool IN_BASE>
    __device__ __forceinline__ rsrc_t side_rsrc(uint64_t transform, uint32_t n, int tid) const {
        const uint32_t t = (uint32_t)transform;
        const uint32_t cycle = t / side_batches, row = t - cycle * side_batches;
        const uint32_t group = IN_BASE ? (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 7) : 0u;
        const uint32_t in_cycle = (group * side_pitch + row) * 128u;
        return make_rsrc(side + (size_t)cycle * side_pitch * n + in_cycle, side_pitch * n - in_cycle);
    }
    // the same descriptor for an explicit (wave-uniform) column group: fft_quad.hh, two butterflies per thread
    __device__ __forceinline__ rsrc_t side_rsrc_group(uint64_t transform, uint32_t n, uint32_t group) const {
        const uint32_t t = (uint32_t)transform;
        const uint32_t cycle = t / side_batches, row = t - cycle * side_batches;
        const uint32_t in_cycle = (group * side_pitch + row) * 128u;
        return make_rsrc(side + (size_t)cycle * side_pitch * n + in_cycle, side_pitch * n - in_cycle);
    }
    // voff / soff: the F32 store's byte offsets (4 * u, 4 * c * BUT with BUT a multiple of 128).
    // FAST: guard.h0 == side_height (fft_side.hip puts the fed Spectrogram's height first), so the product value * height
    // the bin guard formed IS the quantiser's: the index comes from it, no second multiply.  The index rule
    // `1 <= f < height ? (u32)f : 0` is `f < height ? (u32)f : 0` for f >= 0: the conversion truncates everything below 1 to 0.
    // the value that is stored and the row index the Spectrogram derives from it (both kernels' epilogues)
    __device__ __forceinline__ void side_compute(float2 v, float& y, uint32_t& index) const {
        float f;
        if constexpr (FAST) {
            y = amplitude_range_fast_guarded_from_power((v.x * v.x) + (v.y * v.y), this->coeff, this->scale, this->offset,
                                                        this->guard, this->poly, f);
        } else {
            y = this->value(v);
            f = y * side_height;
        }
#if JST_EPI_V2
        // FAST: 0 <= f < height or f == 0 -- a value ON an integer (f == height included: the one non-hit the conversion
        // does not map to 0 by itself) only comes out of the guard's exact recomputation, which zeroes it there
        // (device_math.hh); NaN converts to 0.  No compare, no select on the hot path.
        if constexpr (FAST) index = (uint32_t)f;
        else index = (f < side_height) ? (uint32_t)f : 0u;
#else
        index = (f < side_height) ? (uint32_t)f : 0u;
#endif
    }
    template <bool IN_BASE>
    __device__ __forceinline__ void store_buf_side(rsrc_t r, rsrc_t rs, uint32_t voff, uint32_t soff, float2 v) const {
        float y;
        uint32_t index;
        side_compute(v, y, index);
        buf_store_f1(r, voff, soff, y);
        const uint32_t u = voff >> 2;
        const uint32_t lane_off = IN_BASE ? (u & 127u) : (u >> 7) * (side_pitch * 128u) + (u & 127u);
        buf_store_u8(rs, lane_off, (soff >> 2) * side_pitch, index);
    }
};
template <class Epi>
constexpr bool epi_has_side() {
    if constexpr (requires { Epi::kHasSide; }) return Epi::kHasSide;
    else return false;
}

using StoreAmplitude = StoreAmplitudeT<false>;
using StoreAmplitudeRange = StoreAmplitudeRangeT<false>;

constexpr int cphys(int q) { return q + (q >> 3); }

// The pipelined kernel's exchange layout: TWO pad elements per 16 (same footprint).  Its writes are 16-lane groups of
// consecutive butterflies (ds_write_b64: four groups of 16 lanes, banks = element mod 16): with one pad per 8 a group
// straddles a pad slot and its first and last lane meet on a bank -- every group of every write took two LDS cycles
// (96 of the 112 conflict cycles per wavefront and transform that SQ_LDS_BANK_CONFLICT counted for N = 4096).  With
// the pads at multiples of 16 an aligned group of 16 is contiguous: writes and the ido = 8 reads are conflict-free,
// the contiguous reads keep their one extra cycle per 32-lane group and the ido = 1 reads (stride 8) gain one:
// 32 conflict cycles instead of 112, no extra instruction (base + compile-time offsets as before: no carry into
// bit 4 between a base and its in-butterfly offsets for power-of-two plans).
__device__ __forceinline__ int pphys(int p) { return p + ((p >> 4) << 1); }
constexpr int pcphys(int q) { return q + ((q >> 4) << 1); }

// One butterfly per thread in the last pass (u = tid): the side output's column group is wave-uniform (see
// StoreAmplitudeRangeSideT); the last pass's butterfly count is a multiple of 128 for every n the side output supports.
template <int N, int T>
constexpr bool kSideGroupInBase = (N / make_plan(N).ip[make_plan(N).nf - 1]) == T;

// ---- one Stockham pass -----------------------------------------------------------------------
// Butterfly u in [0, N/IP): i = u % IDO, k = u / IDO.
//   reads  CC(i,b,k) = src[i + IDO*(b + IP*k)]         (pocketfft.hh CC macro)
//   writes CH(i,k,c) = dst[i + IDO*(k + L1*c)] = dst[u + c*N/IP]
//   twiddle on output c >= 1 when i > 0: W[c*L1*i]  (comp_twiddle, pocketfft.hh:1513-1535)
// LDS addresses are one per-butterfly base plus compile-time offsets: for power-of-two IDO,
// floor((i + IDO*b)/8) = floor(IDO*b/8) (i < IDO and IDO | 8 or 8 | IDO), so
// phys(src0 + IDO*b) = phys(src0) + cphys(IDO*b); likewise for the writes when 8 | N/IP.
template <int N, int T, bool FWD, bool CONTIG, int P, class Pro, class Epi>
__device__ __forceinline__ void run_passes(float2* lds, const float2* __restrict__ W, int tid,
                                           bool active, int64_t in_base, int64_t in_as,
                                           int64_t out_base, int64_t out_as, const Pro& pro,
                                           const Epi& epi) {
    constexpr Plan plan = make_plan(N);
    constexpr int IP = plan.ip[P], L1 = plan.l1[P], IDO = plan.ido[P];
    constexpr int BUT = N / IP;
    constexpr int NB = BUT / T;  // butterflies per thread
    constexpr bool FIRST = (P == 0), LAST = (P == plan.nf - 1);
    static_assert(BUT % T == 0 && NB >= 1, "thread count must divide the butterfly count");

    float2 x[NB][IP];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int u = tid + j * T;
        const int i = u & (IDO - 1), k = u / IDO;
        const int src0 = i + IDO * IP * k;
        if constexpr (FIRST) {
#pragma unroll
            for (int b = 0; b < IP; ++b)
                x[j][b] = active ? pro.template load<CONTIG>(in_base, in_as, src0 + IDO * b)
                                 : mk(0.0f, 0.0f);
        } else {
            const float2* rd = lds + phys(src0);
#pragma unroll
            for (int b = 0; b < IP; ++b) x[j][b] = rd[cphys(IDO * b)];
        }
    }
    if constexpr (!FIRST && !LAST) lds_barrier();  // all reads done before anyone overwrites
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int u = tid + j * T;
        const unsigned i = (unsigned)(u & (IDO - 1));
        butterfly<IP, FWD>(x[j]);
        if constexpr (IDO > 1) {
#pragma unroll
            for (int c = 1; c < IP; ++c) {
                const float2 w = W[(unsigned)(c * L1) * i];
                const float2 y = special_mul<FWD>(x[j][c], w);
                x[j][c] = (i != 0u) ? y : x[j][c];
            }
        }
        if constexpr (LAST) {
            if (active) {
#pragma unroll
                for (int c = 0; c < IP; ++c)
                    epi.template store<CONTIG>(out_base, out_as, u + c * BUT, x[j][c]);
            }
        } else if constexpr (BUT % 8 == 0) {
            float2* wr = lds + phys(u);
#pragma unroll
            for (int c = 0; c < IP; ++c) wr[cphys(c * BUT)] = x[j][c];
        } else {
#pragma unroll
            for (int c = 0; c < IP; ++c) lds[phys(u + c * BUT)] = x[j][c];
        }
    }
    if constexpr (!LAST) {
        lds_barrier();  // writes visible before the next pass reads
        run_passes<N, T, FWD, CONTIG, P + 1, Pro, Epi>(lds, W, tid, active, in_base, in_as,
                                                       out_base, out_as, pro, epi);
    }
}

// =============================================================================================
// Pipelined variant (512 <= N <= 8192): one transform per workgroup at a time, T = N/8 threads
// (4096-pt: 512 threads = 8 wavefronts, ONE radix-8 butterfly per thread per pass), persistent
// over the transforms blockIdx.x, blockIdx.x + gridDim.x, ...
//   * the raw input of the NEXT transform is prefetched into registers while the current one is
//     computed (HBM latency and the whole load phase hide behind VALU work from the second
//     transform on);
//   * twiddles of every pass and the window taps of this thread's pass-0 positions are loaded
//     once per workgroup and stay in VGPRs across transforms;
//   * passes exchange through TWO LDS buffers used alternately: one barrier per exchange (the
//     buffer being written is never the one other waves may still be reading), 3 barriers per
//     4096-pt transform instead of 6;
//   * 2 x 36 KiB LDS, <=128 VGPR -> 2 workgroups = 16 waves per CU.
// Same arithmetic, same order as the slot kernel above (bit-identical results).
// Timeline instrumentation for tools/ubench/fft_timeline_batched.hip only (never defined in the product).  The stamps
// of the last three transforms of a workgroup collect in LDS and reach memory in one piece when the workgroup ends: a
// global store per stamp sits in the one vmcnt in front of the kernel's own waits and stretched a transform from 5.4
// to 7.2 us.
#ifdef JST_FFT_TIMELINE
__device__ unsigned long long* jst_tl_base;
__shared__ unsigned long long jst_tl_lds[64];
__shared__ unsigned long long jst_tl_wave[8 * 16];  // per wavefront, transform JST_TL_WAVE_IT of the workgroup
#ifndef JST_TL_WAVE_IT
#define JST_TL_WAVE_IT 30
#endif
#define JST_STAMP(slot)                                                                   \
    do {                                                                                  \
        if (threadIdx.x == 0) jst_tl_lds[(tl_it % 3) * 16 + (slot)] = clock64();          \
        if ((threadIdx.x & 63) == 0 && tl_it == JST_TL_WAVE_IT) jst_tl_wave[(threadIdx.x >> 6) * 16 + (slot)] = clock64(); \
    } while (0)
#define JST_TL_ARG , int tl_it
#define JST_TL_PASS , tl_it
#else
#define JST_STAMP(slot) do {} while (0)
#define JST_TL_ARG
#define JST_TL_PASS
#endif

// Where pass P's twiddles live in the pipelined kernel: a pass whose table IDO*(IP-1) is small
// (<= 512 entries: every pass but the first one or two) reads it from an LDS copy shared by the
// workgroup; the large, lane-unique tables of the early passes are held in VGPRs.
struct TwPlan {
    int reg_off[8];   // first register slot of pass p (or -1)
    int lds_off[8];   // first LDS table entry of pass p (or -1)
    int regs;         // float2 registers per thread
    int lds_entries;  // float2 entries in the LDS table
};
constexpr TwPlan make_twplan(int n) {
    const Plan plan = make_plan(n);
    TwPlan t{};
    for (int p = 0; p < 8; ++p) t.reg_off[p] = t.lds_off[p] = -1;
    for (int p = 0; p < plan.nf; ++p) {
        if (plan.ido[p] <= 1) continue;
        const int entries = plan.ido[p] * (plan.ip[p] - 1);
        if (entries <= 512) {
            t.lds_off[p] = t.lds_entries;
            t.lds_entries += entries;
        } else {
            t.reg_off[p] = t.regs;
            t.regs += (8 / plan.ip[p]) * (plan.ip[p] - 1);
        }
    }
    return t;
}

// The output twiddles of one butterfly, IN PLACE under an exec mask that leaves the lanes with i == 0
// untouched: pocketfft multiplies by WA(c-1,i) only when i > 0 (pocketfft.hh:1141-1225; signed zeros and
// non-finite values of the i == 0 butterflies survive as they are).  Written as one asm statement per group
// of outputs because the compiler's own handling of a divergent `if` around the multiplies merges old and new
// values through a second register set (+16 VGPRs at the join, which spills at the 128 cap), and its
// if-converted form is fourteen v_cndmask (half rate, plus SGPR hazard nops) per butterfly.  Same operations
// and order as special_mul<FWD>: four rounded products, one sum, one difference.
#define JST_TW1(YX, YY, WX, WY)         \
    "v_mul_f32 %[t0], " YX ", " WX "\n" \
    "v_mul_f32 %[t1], " YY ", " WY "\n" \
    "v_mul_f32 %[t2], " YY ", " WX "\n" \
    "v_mul_f32 %[t3], " YX ", " WY "\n"
template <bool FWD>
__device__ __forceinline__ void twiddle_inplace3(unsigned i, float2& y0, float2& y1, float2& y2,
                                                 float2 w0, float2 w1, float2 w2) {
    float t0, t1, t2, t3;
    unsigned long long sv;
    if constexpr (FWD) {
        __asm__ volatile(
            "v_cmp_ne_u32_e32 vcc, 0, %[i]\n"
            "s_and_saveexec_b64 %[sv], vcc\n"
            JST_TW1("%[y0x]", "%[y0y]", "%[w0x]", "%[w0y]") "v_add_f32 %[y0x], %[t0], %[t1]\n v_sub_f32 %[y0y], %[t2], %[t3]\n"
            JST_TW1("%[y1x]", "%[y1y]", "%[w1x]", "%[w1y]") "v_add_f32 %[y1x], %[t0], %[t1]\n v_sub_f32 %[y1y], %[t2], %[t3]\n"
            JST_TW1("%[y2x]", "%[y2y]", "%[w2x]", "%[w2y]") "v_add_f32 %[y2x], %[t0], %[t1]\n v_sub_f32 %[y2y], %[t2], %[t3]\n"
            "s_or_b64 exec, exec, %[sv]\n"
            : [y0x] "+v"(y0.x), [y0y] "+v"(y0.y), [y1x] "+v"(y1.x), [y1y] "+v"(y1.y), [y2x] "+v"(y2.x), [y2y] "+v"(y2.y), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(sv)
            : [i] "v"(i), [w0x] "v"(w0.x), [w0y] "v"(w0.y), [w1x] "v"(w1.x), [w1y] "v"(w1.y), [w2x] "v"(w2.x), [w2y] "v"(w2.y)
            : "vcc");
    } else {
        __asm__ volatile(
            "v_cmp_ne_u32_e32 vcc, 0, %[i]\n"
            "s_and_saveexec_b64 %[sv], vcc\n"
            JST_TW1("%[y0x]", "%[y0y]", "%[w0x]", "%[w0y]") "v_sub_f32 %[y0x], %[t0], %[t1]\n v_add_f32 %[y0y], %[t3], %[t2]\n"
            JST_TW1("%[y1x]", "%[y1y]", "%[w1x]", "%[w1y]") "v_sub_f32 %[y1x], %[t0], %[t1]\n v_add_f32 %[y1y], %[t3], %[t2]\n"
            JST_TW1("%[y2x]", "%[y2y]", "%[w2x]", "%[w2y]") "v_sub_f32 %[y2x], %[t0], %[t1]\n v_add_f32 %[y2y], %[t3], %[t2]\n"
            "s_or_b64 exec, exec, %[sv]\n"
            : [y0x] "+v"(y0.x), [y0y] "+v"(y0.y), [y1x] "+v"(y1.x), [y1y] "+v"(y1.y), [y2x] "+v"(y2.x), [y2y] "+v"(y2.y), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(sv)
            : [i] "v"(i), [w0x] "v"(w0.x), [w0y] "v"(w0.y), [w1x] "v"(w1.x), [w1y] "v"(w1.y), [w2x] "v"(w2.x), [w2y] "v"(w2.y)
            : "vcc");
    }
}
template <bool FWD>
__device__ __forceinline__ void twiddle_inplace4(unsigned i, float2& y0, float2& y1, float2& y2, float2& y3,
                                                 float2 w0, float2 w1, float2 w2, float2 w3) {
    float t0, t1, t2, t3;
    unsigned long long sv;
    if constexpr (FWD) {
        __asm__ volatile(
            "v_cmp_ne_u32_e32 vcc, 0, %[i]\n"
            "s_and_saveexec_b64 %[sv], vcc\n"
            JST_TW1("%[y0x]", "%[y0y]", "%[w0x]", "%[w0y]") "v_add_f32 %[y0x], %[t0], %[t1]\n v_sub_f32 %[y0y], %[t2], %[t3]\n"
            JST_TW1("%[y1x]", "%[y1y]", "%[w1x]", "%[w1y]") "v_add_f32 %[y1x], %[t0], %[t1]\n v_sub_f32 %[y1y], %[t2], %[t3]\n"
            JST_TW1("%[y2x]", "%[y2y]", "%[w2x]", "%[w2y]") "v_add_f32 %[y2x], %[t0], %[t1]\n v_sub_f32 %[y2y], %[t2], %[t3]\n"
            JST_TW1("%[y3x]", "%[y3y]", "%[w3x]", "%[w3y]") "v_add_f32 %[y3x], %[t0], %[t1]\n v_sub_f32 %[y3y], %[t2], %[t3]\n"
            "s_or_b64 exec, exec, %[sv]\n"
            : [y0x] "+v"(y0.x), [y0y] "+v"(y0.y), [y1x] "+v"(y1.x), [y1y] "+v"(y1.y), [y2x] "+v"(y2.x), [y2y] "+v"(y2.y), [y3x] "+v"(y3.x), [y3y] "+v"(y3.y), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(sv)
            : [i] "v"(i), [w0x] "v"(w0.x), [w0y] "v"(w0.y), [w1x] "v"(w1.x), [w1y] "v"(w1.y), [w2x] "v"(w2.x), [w2y] "v"(w2.y), [w3x] "v"(w3.x), [w3y] "v"(w3.y)
            : "vcc");
    } else {
        __asm__ volatile(
            "v_cmp_ne_u32_e32 vcc, 0, %[i]\n"
            "s_and_saveexec_b64 %[sv], vcc\n"
            JST_TW1("%[y0x]", "%[y0y]", "%[w0x]", "%[w0y]") "v_sub_f32 %[y0x], %[t0], %[t1]\n v_add_f32 %[y0y], %[t3], %[t2]\n"
            JST_TW1("%[y1x]", "%[y1y]", "%[w1x]", "%[w1y]") "v_sub_f32 %[y1x], %[t0], %[t1]\n v_add_f32 %[y1y], %[t3], %[t2]\n"
            JST_TW1("%[y2x]", "%[y2y]", "%[w2x]", "%[w2y]") "v_sub_f32 %[y2x], %[t0], %[t1]\n v_add_f32 %[y2y], %[t3], %[t2]\n"
            JST_TW1("%[y3x]", "%[y3y]", "%[w3x]", "%[w3y]") "v_sub_f32 %[y3x], %[t0], %[t1]\n v_add_f32 %[y3y], %[t3], %[t2]\n"
            "s_or_b64 exec, exec, %[sv]\n"
            : [y0x] "+v"(y0.x), [y0y] "+v"(y0.y), [y1x] "+v"(y1.x), [y1y] "+v"(y1.y), [y2x] "+v"(y2.x), [y2y] "+v"(y2.y), [y3x] "+v"(y3.x), [y3y] "+v"(y3.y), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(sv)
            : [i] "v"(i), [w0x] "v"(w0.x), [w0y] "v"(w0.y), [w1x] "v"(w1.x), [w1y] "v"(w1.y), [w2x] "v"(w2.x), [w2y] "v"(w2.y), [w3x] "v"(w3.x), [w3y] "v"(w3.y)
            : "vcc");
    }
}
#undef JST_TW1

template <int N, int T, bool FWD, bool CONTIG, int P, class Pro, class Epi>
__device__ __forceinline__ void pipe_passes(float2 (&x)[8], float2* buf0, float2* buf1,
                                            const float2 (&twr)[make_twplan(N).regs + 1],
                                            const float2* twl, int tid, int64_t out_base,
                                            int64_t out_as, const Epi& epi, const Pro& pro,
                                            float2 (&opnd)[8], bool more, rsrc_t r_out,
                                            rsrc_t r_opnd, rsrc_t r_side, bool young JST_TL_ARG) {
    constexpr Plan plan = make_plan(N);
    constexpr TwPlan tp = make_twplan(N);
    constexpr int IP = plan.ip[P], IDO = plan.ido[P];
    constexpr int BUT = N / IP;
    constexpr int NB = 8 / IP;  // butterflies per thread (8 points per thread)
    constexpr bool LAST = (P == plan.nf - 1);
    static_assert(BUT == NB * T, "T must be N/8");
    // Wave priority: the passes are coupled by workgroup barriers (a late wave stalls seven others),
    // the epilogue is a long barrier-free VALU stream.  With both co-resident workgroups at equal
    // priority the epilogue of one delays the passes of the other; passes at priority 3 and the
    // epilogue at 0 gave 30.8 -> 28.9 us (exact) and 21.0 -> 20.1 us (fast) per 1024 x 4096 launch.
// Epilogue priority leapfrog (see the last pass below): switch point in outputs (0 = off) and the lift.  Same-box A/B per
// 16384-transform launch (profiles/r04_experiments/n_epilogue_leapfrog.log): off 171.9 / 169.8 us, 4 outputs + 1: 168.4 /
// 165.9, 4 outputs + 2: 166.7 / 165.7, every 2 outputs or every output: no gain.
#ifndef JST_EPI_LEAPFROG
#define JST_EPI_LEAPFROG 4
#endif
#ifndef JST_EPI_LIFT
#define JST_EPI_LIFT 2
#endif
#ifndef JST_PRIO_PA
#define JST_PRIO_PA 3
#define JST_PRIO_PB 3
#define JST_PRIO_EA 0
#define JST_PRIO_EB 1
#endif
    {
        // The second workgroup of a CU (dispatched ~1.5-2.5 us after the first: blockIdx >= grid/2 with two resident
        // workgroups per CU) holds the YOUNGER wavefronts, which lose every VALU arbitration tie against the older
        // workgroup's: left alone it finishes ~5 us after its neighbour and runs that stretch at half occupancy.
        if constexpr (P == 0) {
            if (young) __builtin_amdgcn_s_setprio(JST_PRIO_PB);
            else __builtin_amdgcn_s_setprio(JST_PRIO_PA);
        }
        if constexpr (LAST) {
            if (young) __builtin_amdgcn_s_setprio(JST_PRIO_EB);
            else __builtin_amdgcn_s_setprio(JST_PRIO_EA);
        }
    }
    // x[] holds CC(i,b,k) for butterfly j at x[j*IP + b]
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int u = tid + j * T;
        const unsigned i = (unsigned)(u & (IDO - 1));
        float2 y[IP];
#pragma unroll
        for (int b = 0; b < IP; ++b) y[b] = x[j * IP + b];
        butterfly<IP, FWD>(y);
        if constexpr (IDO > 1) {
            // pocketfft multiplies by WA(c-1,i) only when i > 0 (signed zeros and non-finite values survive
            // the i == 0 butterflies untouched).  One exec-masked region around the seven multiplies instead
            // of fourteen compare/select pairs: the lanes with i == 0 sit out, everyone else pays nothing.
            float2 w[IP];
#pragma unroll
            for (int c = 1; c < IP; ++c) {
                if constexpr (tp.reg_off[P] >= 0) w[c] = twr[tp.reg_off[P] + j * (IP - 1) + (c - 1)];
                else w[c] = twl[tp.lds_off[P] + i * (IP - 1) + (c - 1)];
            }
            if constexpr (IP == 8) {
                twiddle_inplace3<FWD>(i, y[1], y[2], y[3], w[1], w[2], w[3]);  // y[0] takes no twiddle
                twiddle_inplace4<FWD>(i, y[4], y[5], y[6], y[7], w[4], w[5], w[6], w[7]);
            } else {
#pragma unroll
                for (int c = 1; c < IP; ++c) {
                    const float2 z = special_mul<FWD>(y[c], w[c]);
                    y[c] = (i != 0u) ? z : y[c];
                }
            }
        }
        if constexpr (LAST) {
#pragma unroll
            for (int c = 0; c < IP; ++c) {
#if JST_EPI_LEAPFROG
                // The two wavefronts of a workgroup that share a SIMD (w and w + 4 of eight) take turns at the higher
                // priority through the barrier-free epilogue (every JST_EPI_LEAPFROG outputs): the arbiter's oldest-first
                // rule otherwise lets the older one run ahead for the whole stretch and the next barrier waits for the
                // younger.
                if (c % JST_EPI_LEAPFROG == 0) {
                    const bool upper = (tid >> 8) & 1;  // wavefronts 4..7 of the 512-thread workgroup
                    const bool lift = ((c / JST_EPI_LEAPFROG) & 1) ? !upper : upper;
                    if (young) {
                        if (lift) __builtin_amdgcn_s_setprio(JST_PRIO_EB + JST_EPI_LIFT);
                        else __builtin_amdgcn_s_setprio(JST_PRIO_EB);
                    } else {
                        if (lift) __builtin_amdgcn_s_setprio(JST_PRIO_EA + JST_EPI_LIFT);
                        else __builtin_amdgcn_s_setprio(JST_PRIO_EA);
                    }
                }
#endif
                if constexpr (CONTIG && epi_has_side<Epi>())
                    epi.template store_buf_side<kSideGroupInBase<N, T>>(r_out, r_side, (uint32_t)u * Epi::kElemBytes,
                                       (uint32_t)(c * BUT) * Epi::kElemBytes, y[c]);
                else if constexpr (CONTIG)
                    epi.store_buf(r_out, (uint32_t)u * Epi::kElemBytes,
                                  (uint32_t)(c * BUT) * Epi::kElemBytes, y[c]);
                else
                    epi.template store<CONTIG>(out_base, out_as, u + c * BUT, y[c]);
                // keep at most two epilogues in flight: interleaving all eight costs ~40 VGPRs
                // of temporaries and pushes the prefetch registers into scratch
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (Pro::kHasOperand && !(JST_OPND_RESIDENT && CONTIG)) {
                    // The per-position operand of the prologue (window taps) is not kept live
                    // across the passes: element e is re-requested from L2 as soon as output e
                    // has retired, into the registers that output just freed.
                    // CONTIG: unconditional (r_opnd has zero records past the last transform) -- a load
                    // under `if (more)` is copied out of a phi right behind the epilogue, and the wait
                    // for that copy also waits for every store issued before it.
                    if (CONTIG || more) {
                        constexpr int IP0 = plan.ip[0], IDO0 = plan.ido[0];
                        const int e = j * IP + c;  // constant after unrolling
                        const int u0 = tid + (e / IP0) * T;
                        const int l0 = (u0 & (IDO0 - 1)) + IDO0 * IP0 * (u0 / IDO0);
                        if constexpr (CONTIG)
                            opnd[e] = buf_load_f2(r_opnd, (uint32_t)l0 * 8u,
                                                  (uint32_t)(IDO0 * (e % IP0)) * 8u);
                        else
                            opnd[e] = pro.template load_operand<CONTIG>(IDO0 * (e % IP0), l0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
            float2* wr = buf0 + pphys(u);
#pragma unroll
            for (int c = 0; c < IP; ++c) wr[pcphys(c * BUT)] = y[c];
        }
    }
    JST_STAMP(2 + 2 * P);  // pass P computed, results issued to LDS / HBM
    if constexpr (!LAST) {
        lds_barrier();
        JST_STAMP(3 + 2 * P);  // barrier passed
        constexpr int IP2 = plan.ip[P + 1], IDO2 = plan.ido[P + 1];
        constexpr int NB2 = 8 / IP2;
#pragma unroll
        for (int j = 0; j < NB2; ++j) {
            const int u = tid + j * T;
            const int i = u & (IDO2 - 1), k = u / IDO2;
            const float2* rd = buf0 + pphys(i + IDO2 * IP2 * k);
#pragma unroll
            for (int b = 0; b < IP2; ++b) {
                // One ds_read_b64 per element: the load/store optimiser would pair these into ds_read2_b64, which the
                // LDS serves at 128 B/clk against 256 B/clk for the single form (tools/ubench/lds_rate.hip: 27 vs 16
                // clocks per wavefront for the eight elements of a butterfly).  A volatile access is never merged.
                typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_ptr;
                const unsigned long long bits = *(lds_u64_ptr)(rd + pcphys(IDO2 * b));
                x[j * IP2 + b] = __builtin_bit_cast(float2, bits);
            }
        }
        pipe_passes<N, T, FWD, CONTIG, P + 1, Pro, Epi>(x, buf1, buf0, twr, twl, tid, out_base,
                                                        out_as, epi, pro, opnd, more, r_out,
                                                        r_opnd, r_side, young JST_TL_PASS);
    }
}

constexpr bool fft_pipe_supported(int n) { return n >= 512 && n <= 8192; }
constexpr size_t fft_pipe_lds_bytes(int n) {
    return (2 * (size_t)lds_elems(n) + (size_t)make_twplan(n).lds_entries) * sizeof(float2);
}

// The kernel's body as a device function of (workgroup index, workgroup count): the plain kernel below passes
// blockIdx.x / gridDim.x, a launch that carries other work beside the transforms passes its own numbering.
template <int N, bool FWD, bool CONTIG, class Pro, class Epi>
__device__ __forceinline__ void fft_pipe_body(const FftLayout& L, const float2* __restrict__ W, const Pro& pro,
                                              const Epi& epi_arg, const uint32_t bid, const uint32_t grid) {
    constexpr int T = N / 8;
    constexpr Plan plan = make_plan(N);
    constexpr TwPlan tp = make_twplan(N);
    constexpr int NEX = plan.nf - 1;  // LDS exchanges per transform
    Epi epi = epi_arg;  // this thread's copy: an epilogue may pin constants in VGPRs for the whole transform loop
    if constexpr (requires { epi.pin_constants(); }) epi.pin_constants();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* bufA = reinterpret_cast<float2*>(smem_raw);
    float2* bufB = bufA + lds_elems(N);
    float2* twl = bufB + lds_elems(N);
    const int tid = threadIdx.x;

    // ---- per-workgroup constants and the first transform's loads ---------------------------------
    // Issue order: twiddle loads (L2), then the first transform's operand and input loads (HBM), then the LDS
    // copy of the small twiddle tables -- the s_waitcnt in front of those LDS writes covers only the (older)
    // twiddle loads, so the HBM round trip of the first transform overlaps the table set-up.  The table is first
    // read after the first exchange barrier (unless pass 0 itself reads it), so it needs no barrier of its own.
    float2 twr[tp.regs + 1];
    constexpr int TWL_PER_THREAD = (tp.lds_entries + T - 1) / T;
    float2 twv[TWL_PER_THREAD + 1];
#pragma unroll
    for (int p = 0; p < plan.nf; ++p) {
        if (plan.ido[p] <= 1) continue;
        if (tp.reg_off[p] >= 0) {
#pragma unroll
            for (int j = 0; j < 8 / plan.ip[p]; ++j) {
                const unsigned i = (unsigned)((tid + j * T) & (plan.ido[p] - 1));
#pragma unroll
                for (int c = 1; c < plan.ip[p]; ++c)
                    twr[tp.reg_off[p] + j * (plan.ip[p] - 1) + (c - 1)] =
                        W[(unsigned)(c * plan.l1[p]) * i];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < TWL_PER_THREAD; ++q) {
        const int g = tid + q * T;  // entry of the concatenated LDS table
        unsigned widx = 0;
#pragma unroll
        for (int p = 0; p < plan.nf; ++p) {
            if (tp.lds_off[p] < 0) continue;
            const int entries = plan.ido[p] * (plan.ip[p] - 1);
            const int e = g - tp.lds_off[p];
            if (e >= 0 && e < entries)
                widx = ((unsigned)(e % (plan.ip[p] - 1)) + 1u) * (unsigned)plan.l1[p] * (unsigned)(e / (plan.ip[p] - 1));
        }
        twv[q] = W[widx];
    }

    // pass-0 element positions of this thread: pos0[j] + IDO0*b
    constexpr int IP0 = plan.ip[0], IDO0 = plan.ido[0], NB0 = 8 / IP0;
    int pos0[NB0];
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
        const int u = tid + j * T;
        pos0[j] = (u & (IDO0 - 1)) + IDO0 * IP0 * (u / IDO0);
    }

    uint64_t t = bid;
    if (t >= L.transforms) return;
    const bool young = bid >= (grid >> 1);
    int64_t in_base, out_base;
    fft_bases(L, t, in_base, out_base);
    typename Pro::raw_t raw[8];
    float2 opnd[8];
    constexpr uint32_t RB = Pro::kRawBytes;  // bytes per complex sample of the input stream (8: cf32, 4: ci16, 2: ci8 / cu8)
    const rsrc_t r_opnd = make_rsrc(pro.operand_row(), (uint32_t)N * 8u);
    {
        const rsrc_t r_in = make_rsrc(pro.row(in_base), (uint32_t)N * RB);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (CONTIG) {
                if constexpr (Pro::kHasOperand)
                    opnd[e] = buf_load_f2(r_opnd, (uint32_t)pos0[e / IP0] * 8u,
                                          (uint32_t)(IDO0 * (e % IP0)) * 8u);
                raw[e] = Pro::load_raw_buf(r_in, (uint32_t)pos0[e / IP0] * RB, (uint32_t)(IDO0 * (e % IP0)) * RB);
            } else {
                opnd[e] = pro.template load_operand<CONTIG>(IDO0 * (e % IP0), pos0[e / IP0]);
                raw[e] = pro.template load_raw<CONTIG>(in_base, L.in_axis_stride,
                                                       IDO0 * (e % IP0), pos0[e / IP0]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < TWL_PER_THREAD; ++q)
        if (tid + q * T < tp.lds_entries) twl[tid + q * T] = twv[q];
    if constexpr (tp.lds_entries > 0 && tp.lds_off[0] >= 0) lds_barrier();  // pass 0 reads the table

    bool flip = false;
#ifdef JST_FFT_TIMELINE
    int tl_it = 0;
    if (threadIdx.x == 0) jst_tl_lds[62] = wall_clock64();
#endif
    while (true) {
        JST_STAMP(0);  // iteration start
        float2 x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = pro.apply(raw[e], opnd[e]);
        JST_STAMP(1);  // input arrived + prologue applied
        // prefetch the next transform of this workgroup while this one is computed
        const uint64_t tn = t + grid;
        const bool more = tn < L.transforms;
        int64_t nin = 0, nout = 0;
        if constexpr (CONTIG) {
            // UNCONDITIONAL loads: under `if (more)` the loaded registers meet the old ones in a phi, the
            // compiler copies them right behind the loads and the s_waitcnt for those copies exposes the
            // whole HBM round trip in front of the passes.  Past the last transform the descriptor has
            // zero records: the loads return 0 without touching memory.
            fft_bases(L, more ? tn : t, nin, nout);
            const rsrc_t r_in = make_rsrc(pro.row(nin), more ? (uint32_t)N * RB : 0u);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                raw[e] = Pro::load_raw_buf(r_in, (uint32_t)pos0[e / IP0] * RB, (uint32_t)(IDO0 * (e % IP0)) * RB);
        } else if (more) {
            fft_bases(L, tn, nin, nout);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                raw[e] = pro.template load_raw<CONTIG>(nin, L.in_axis_stride, IDO0 * (e % IP0), pos0[e / IP0]);
        }
        const rsrc_t r_out = make_rsrc(epi.row(out_base), (uint32_t)N * Epi::kElemBytes);
        const rsrc_t r_opnd_next = make_rsrc(pro.operand_row(), more ? (uint32_t)N * 8u : 0u);
        rsrc_t r_side = r_out;  // unused unless the epilogue has a side output
        if constexpr (CONTIG && epi_has_side<Epi>())
            r_side = epi.template side_rsrc<kSideGroupInBase<N, T>>(fft_ring_row(L, t), (uint32_t)N, tid);
        if (flip)
            pipe_passes<N, T, FWD, CONTIG, 0, Pro, Epi>(x, bufB, bufA, twr, twl, tid, out_base,
                                                        L.out_axis_stride, epi, pro, opnd,
                                                        more, r_out, r_opnd_next, r_side, young JST_TL_PASS);
        else
            pipe_passes<N, T, FWD, CONTIG, 0, Pro, Epi>(x, bufA, bufB, twr, twl, tid, out_base,
                                                        L.out_axis_stride, epi, pro, opnd,
                                                        more, r_out, r_opnd_next, r_side, young JST_TL_PASS);
#ifdef JST_FFT_TIMELINE
        ++tl_it;
        if (threadIdx.x == 0) jst_tl_lds[63] = wall_clock64();
#endif
        if (!more) break;
        if constexpr (NEX & 1) flip = !flip;  // next transform starts on the buffer written longest ago
        t = tn;
        out_base = nout;
    }
#ifdef JST_FFT_TIMELINE
    if (threadIdx.x == 0) {
        for (int q = 0; q < 64; ++q) jst_tl_base[blockIdx.x * 64 + q] = jst_tl_lds[q];
        for (int q = 0; q < 128; ++q) jst_tl_base[(size_t)gridDim.x * 64 + (size_t)blockIdx.x * 128 + q] = jst_tl_wave[q];
    }
#endif
}

template <int N, bool FWD, bool CONTIG, class Pro, class Epi>
#ifndef JST_PIPE_WAVES
#define JST_PIPE_WAVES ((N / 8) * 2 / 256 >= 4 ? 4 : (N / 8) * 2 / 256)
#endif
__global__ __launch_bounds__(N / 8, JST_PIPE_WAVES) void fft_pipe_kernel(
    const FftLayout L, const float2* __restrict__ W, const Pro pro, const Epi epi) {
    fft_pipe_body<N, FWD, CONTIG, Pro, Epi>(L, W, pro, epi, blockIdx.x, gridDim.x);
}

constexpr int fft_threads_per_transform(int n) { return n >= 16 ? n / 16 : 1; }
constexpr int fft_block_threads(int n) {
    return fft_threads_per_transform(n) >= 256 ? fft_threads_per_transform(n) : 256;
}
constexpr int fft_transforms_per_block(int n) {
    return fft_block_threads(n) / fft_threads_per_transform(n);
}
constexpr size_t fft_lds_bytes(int n) {
    return (size_t)fft_transforms_per_block(n) * lds_elems(n) * sizeof(float2);
}

template <int N, bool FWD, bool CONTIG, class Pro, class Epi>
__global__ __launch_bounds__(fft_block_threads(N), 4) void fft_lds_kernel(
    const FftLayout L, const float2* __restrict__ W, const Pro pro, const Epi epi) {
    constexpr int T = fft_threads_per_transform(N);
    constexpr int TPB = fft_transforms_per_block(N);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);

    const int slot = threadIdx.x / T;
    const int tid0 = threadIdx.x % T;
    float2* lds = smem + (size_t)slot * lds_elems(N);

    for (uint64_t t0 = (uint64_t)blockIdx.x * TPB; t0 < L.transforms;
         t0 += (uint64_t)gridDim.x * TPB) {
        const uint64_t t = t0 + slot;
        const bool active = t < L.transforms;
        int64_t in_base = 0, out_base = 0;
        if (active) fft_bases(L, t, in_base, out_base);
        if constexpr (T % 64 == 0) {
            // one transform per set of whole wavefronts: its base offsets are wave-uniform
            in_base = ((int64_t)__builtin_amdgcn_readfirstlane((int)(in_base >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)in_base);
            out_base = ((int64_t)__builtin_amdgcn_readfirstlane((int)(out_base >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)out_base);
        }
        // Keep the per-thread address arithmetic inside the transform loop: hoisted out (it is
        // loop invariant) it pins ~150 VGPRs and spills.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        if constexpr (N == 1) {
            if (active)
                epi.template store<CONTIG>(out_base, L.out_axis_stride, 0,
                                           pro.template load<CONTIG>(in_base, L.in_axis_stride, 0));
        } else {
            run_passes<N, T, FWD, CONTIG, 0, Pro, Epi>(lds, W, tid, active, in_base,
                                                       L.in_axis_stride, out_base,
                                                       L.out_axis_stride, pro, epi);
            lds_barrier();  // LDS reuse by the next transform of this slot
        }
    }
}

// =============================================================================================
// One transform per workgroup, EIGHT points per thread (T = N/8: 4096-pt = 512 threads), no
// persistence and no register prefetch: <= 64 VGPRs -> 8 wavefronts per SIMD, one LDS buffer ->
// four workgroups (transforms) resident per CU.  A single wavefront issues a VALU instruction
// only every ~3.4 ns, so the exact epilogue (a ~1000-instruction dependent stream per wavefront)
// is bound by per-wavefront issue latency unless >= 4 wavefronts per SIMD are in VALU phases at
// the same time; with four independent workgroups per CU in different phases (waiting for HBM,
// exchanging through LDS, in the epilogue) the SIMDs stay saturated and the hardware dispatcher
// does the load/compute/store pipelining that the persistent kernel does by hand.
template <int N, bool FWD, bool CONTIG, class Pro, class Epi>
__global__ __launch_bounds__(N / 8, (N / 8) / 64) void fft_wg_kernel(
    const FftLayout L, const float2* __restrict__ W, const Pro pro, const Epi epi) {
    constexpr int T = N / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    const uint64_t t = blockIdx.x;
    int64_t in_base = 0, out_base = 0;
    fft_bases(L, t, in_base, out_base);
    run_passes<N, T, FWD, CONTIG, 0, Pro, Epi>(lds, W, (int)threadIdx.x, true, in_base,
                                               L.in_axis_stride, out_base, L.out_axis_stride, pro, epi);
}

}  // namespace jst::dev
