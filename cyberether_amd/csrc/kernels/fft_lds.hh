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

__device__ __forceinline__ void fft_bases(const FftLayout& L, uint64_t t, int64_t& in_base,
                                          int64_t& out_base) {
    in_base = (int64_t)L.in_offset;
    out_base = (int64_t)L.out_offset;
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

// LDS physical index: one pad element per 8 keeps the stride-8 / stride-64 read patterns of the
// ido = 1 and ido = 8 passes on distinct banks (see header comment).
__device__ __forceinline__ int phys(int p) { return p + (p >> 3); }
constexpr int lds_elems(int n) { return n + (n >> 3); }

// ---- prologues (how pass 0 obtains CC(i,b,k)) -------------------------------------------------
// operator()(base, axis_stride, pos): element 'pos' along the transform axis of the transform
// whose first element sits at 'base'.  CONTIG instantiations (axis stride 1 on every operand)
// index with 32-bit offsets from a per-transform pointer so that addresses are base + constant.
struct LoadCF32 {
    const float2* in;
    template <bool CONTIG>
    __device__ __forceinline__ float2 load(int64_t base, int64_t axis_stride, int pos) const {
        if constexpr (CONTIG) return (in + base)[(unsigned)pos];
        else return in[base + (int64_t)pos * axis_stride];
    }
};
// Multiply fused in: signal[...,n] * window[n] with the Multiply module's arithmetic
// (core/multiply/module_impl_native_cpu.cc:94-100); window broadcast over every outer axis
// (stride 0 there), element stride wstride along the transform axis.
struct LoadCF32TimesWindow {
    const float2* in;
    const float2* window;
    int64_t wstride;
    template <bool CONTIG>
    __device__ __forceinline__ float2 load(int64_t base, int64_t axis_stride, int pos) const {
        if constexpr (CONTIG) {
            return cmul_full((in + base)[(unsigned)pos], window[(unsigned)pos]);
        } else {
            return cmul_full(in[base + (int64_t)pos * axis_stride],
                             window[(int64_t)pos * wstride]);
        }
    }
};

// ---- epilogues (what happens to CH of the last pass) ------------------------------------------
struct StoreCF32 {
    float2* out;
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t axis_stride, int pos,
                                          float2 v) const {
        if constexpr (CONTIG) (out + base)[(unsigned)pos] = v;
        else out[base + (int64_t)pos * axis_stride] = v;
    }
};
template <bool FAST>
struct StoreAmplitudeT {  // Amplitude module fused (amplitude/module_impl_native_cpu.cc:73-86)
    float* out;
    float coeff;
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t axis_stride, int pos,
                                          float2 v) const {
        const float r = FAST ? amplitude_cf32_fast(v, coeff) : amplitude_cf32(v, coeff);
        if constexpr (CONTIG) (out + base)[(unsigned)pos] = r;
        else out[base + (int64_t)pos * axis_stride] = r;
    }
};
template <bool FAST>
struct StoreAmplitudeRangeT {  // Amplitude -> Range fused (range/module_impl_native_cpu.cc:67-82)
    float* out;
    float coeff, scale, offset;
    template <bool CONTIG>
    __device__ __forceinline__ void store(int64_t base, int64_t axis_stride, int pos,
                                          float2 v) const {
        const float r = FAST ? range_f32_fast(amplitude_cf32_fast(v, coeff), scale, offset)
                             : range_f32(amplitude_cf32(v, coeff), scale, offset);
        if constexpr (CONTIG) (out + base)[(unsigned)pos] = r;
        else out[base + (int64_t)pos * axis_stride] = r;
    }
};

using StoreAmplitude = StoreAmplitudeT<false>;
using StoreAmplitudeRange = StoreAmplitudeRangeT<false>;

constexpr int cphys(int q) { return q + (q >> 3); }

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
    if constexpr (!FIRST && !LAST) __syncthreads();  // all reads done before anyone overwrites
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
        __syncthreads();  // writes visible before the next pass reads
        run_passes<N, T, FWD, CONTIG, P + 1, Pro, Epi>(lds, W, tid, active, in_base, in_as,
                                                       out_base, out_as, pro, epi);
    }
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
            __syncthreads();  // LDS reuse by the next transform of this slot
        }
    }
}

}  // namespace jst::dev
