// fir.hip -- direct-form polyphase FIR + decimation of a continuous CF32 stream: the "fast"
// provider of the Filter block (BASELINE config 3, "LDS tap stencil").
//
// The reference's Filter block (src/domains/dsp/filter/block_impl.cc:350-582) convolves through
// FFT overlap-add and decimates by folding the spectrum.  In the time domain that block computes,
// for real low-pass taps h (centre frequency 0) and decimation r, on the stream x obtained by
// walking the batch rows in order (overlap_add carries the tail of a row into the next one, and
// the tail of the last row into the next cycle):
//
//     y[g] = sum_{k < T} h[k] * x[g*r - k]            (x[n] = 0 before the stream starts)
//
// (tests/test_gpu_filter_fast.py checks this identity against the oracle's FFT chain to 1e-5 of
// peak, the reference's own tolerance in filter_engine/block_tests.cc:55-61).  With T/r taps per
// polyphase branch a direct evaluation needs r*8 bytes of input and 2*T fused multiply-adds per
// output: memory bound for the block's typical 10x decimation, and one launch instead of eleven.
//
// Mapping: a workgroup owns OW consecutive outputs of one row.  Their (OW + Q - 1) * r input samples
// (Q = ceil(T/r)) are staged with coalesced loads into LDS, de-interleaved by polyphase branch:
// lds[p][s] = x[u*r - p], s = u - (m0 - Q + 1), so a branch's samples are contiguous.  Rows before the
// first come from the previous row or the module's history tensor.  Thread t evaluates J consecutive
// outputs; per branch it walks samples s = J*t + i (i < J + Q - 1) in chunks of 32 -- sixteen 16-byte
// LDS reads at compile-time offsets, dense across the wavefront for J = 2 -- and feeds each sample into
// its J accumulators (even and odd samples separately: 2*J independent FMA chains), so one 16-byte LDS
// read serves 4*J FMAs.  Tap values are wave-uniform: the window of taps a chunk needs comes from a
// step-major table (built once by the statically settled fir_taps module) through scalar loads and
// enters the FMAs as SGPR operands.
#include <cstdio>
#include <cstdlib>

#include "kernels.hh"

namespace jst::kernels {

namespace {

constexpr int kFirThreads = 256;  // upper bound; the launch picks 64..256
constexpr int kFirChunk = 32;                       // samples per inner-loop chunk
constexpr int kFirMaxJ = 5;
constexpr int kFirStage = 12;                       // global loads in flight per thread while staging

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float2 buf_load_f2(rsrc_t r, uint32_t voff_bytes, uint32_t soff_bytes) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff_bytes, soff_bytes, 0));
    return make_float2(v.x, v.y);
}

struct FirDims {
    uint32_t S, M, T, r, Q, heads, tiles_per_row, OW, UL, chunks, magic_r, rows, update_history;
};

// Tap windows, step major: step = p * chunks + c (branch p, chunk c of kFirChunk samples) needs the
// taps q = (Q-1) - kFirChunk*c - (kFirChunk-1) + k, k = 0 .. kFirChunk+J-2, of branch p: entry
// [head][step][k], zero outside the filter, kFirWin floats per step (aligned scalar loads).
constexpr int kFirWin = 64;
__global__ void fir_taps_kernel(float* __restrict__ hp, const float2* __restrict__ taps, FirDims d) {
    const uint32_t steps = d.r * d.chunks, total = d.heads * steps * kFirWin;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t k = e % kFirWin, step = (e / kFirWin) % steps, head = e / (kFirWin * steps);
        const uint32_t p = step / d.chunks, c = step % d.chunks;
        const int32_t q = (int32_t)(d.Q - 1) - kFirChunk * (int32_t)c - (kFirChunk - 1) + (int32_t)k;
        float v = 0.0f;
        if (q >= 0 && k < kFirChunk + kFirMaxJ - 1) {
            const uint32_t tap = (uint32_t)q * d.r + p;
            if (tap < d.T) v = taps[(uint64_t)head * d.T + tap].x;
        }
        hp[e] = v;
    }
}

template <int J>
__global__ __launch_bounds__(kFirThreads) void fir_decimate_kernel(float2* __restrict__ out,
                                                                   const float2* __restrict__ in,
                                                                   const float2* hist,
                                                                   const float* __restrict__ hp, FirDims d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);  // [r][UL]
    const uint32_t tid = threadIdx.x;
    const uint32_t row = blockIdx.x / d.tiles_per_row, m0 = (blockIdx.x % d.tiles_per_row) * d.OW;

    // stage samples n = n_base + e = u*r - p into slot p*UL + s (s = e / r, p = r - 1 - e % r).
    // One buffer descriptor per row: [row start - back, row end) with back = T-1 samples of the
    // previous row; everything outside (before the stream, past the row) reads as zero by the
    // descriptor's range check.  kFirStage loads are in flight per thread.
    const uint32_t threads = blockDim.x;
    const int32_t n_base = ((int32_t)m0 - (int32_t)d.Q) * (int32_t)d.r + 1;
    const uint32_t real = (d.OW + d.Q - 1) * d.r;
    const uint32_t back = row > 0 ? d.T - 1 : 0;
    const float2* cur = in + (uint64_t)row * d.S;
    const rsrc_t rs = make_rsrc(cur - back, (d.S + back) * 8u);
    auto slot_of = [&](uint32_t e) {
        const uint32_t sidx = d.r == 1 ? e : __umulhi(e, d.magic_r);
        return (d.r - 1 - (e - sidx * d.r)) * d.UL + sidx;
    };
    {
        // A staging stride that is a multiple of r keeps every thread on ONE polyphase branch: element
        // e = tid + k*stride lands in slot (r-1-tid%r)*UL + tid/r + k*stride/r, so each step is two adds
        // (the few threads past the stride sit the staging out).
        const uint32_t q_t = d.r == 1 ? threads : __umulhi(threads, d.magic_r), stride = q_t * d.r;
        const uint32_t full = real / stride;  // iterations in which every participating thread has a sample
        const bool stager = tid < stride;
        uint32_t slot = slot_of(tid);
        uint32_t off = (uint32_t)(n_base + (int32_t)back + (int32_t)tid) * 8u;  // negative n: far out of range
        if (stager) {
            for (uint32_t k0 = 0; k0 <= full; k0 += kFirStage) {  // uniform trip count
                float2 v[kFirStage];
#pragma unroll
                for (int k = 0; k < kFirStage; ++k) v[k] = buf_load_f2(rs, off + k * stride * 8u, 0);
#pragma unroll
                for (int k = 0; k < kFirStage; ++k) {
                    const uint32_t it = k0 + k;  // uniform
                    if (it < full || (it == full && tid + it * stride < real)) lds[slot + k * q_t] = v[k];
                }
                off += kFirStage * stride * 8u;
                slot += kFirStage * q_t;
            }
        }
    }
    {   // slots past the tile's last sample (chunk round-up) only ever meet zero taps: keep them finite
        const uint32_t first = d.OW + d.Q - 1, ns = d.UL - first;
        for (uint32_t i = tid; i < ns * d.r; i += threads) lds[(i / ns) * d.UL + first + i % ns] = make_float2(0.0f, 0.0f);
    }
    const bool reads_history = row == 0 && n_base < 0;  // workgroup-uniform
    if (reads_history) {  // samples before the stream's first row: the previous cycle's tail
        __syncthreads();
        for (uint32_t e = tid; e < (uint32_t)(-n_base) && e < real; e += threads) {
            const int32_t n = n_base + (int32_t)e;
            if (n >= -(int32_t)(d.T - 1)) lds[slot_of(e)] = hist[(int32_t)(d.T - 1) + n];
        }
    }
    __syncthreads();
    if (reads_history && d.update_history) {  // the only reader of the history also renews it
        float2* hw = const_cast<float2*>(hist);
        const float2* tail = in + (uint64_t)d.rows * d.S - (d.T - 1);
        for (uint32_t i = tid; i < d.T - 1; i += threads) hw[i] = tail[i];
    }

    const bool active = m0 + tid * J < d.M;
    for (uint32_t head = 0; head < d.heads; ++head) {
        // iteration i of branch p feeds sample s = J*t + i into output j with tap q = j - i + Q - 1.
        // One step = one chunk of kFirChunk samples of one branch: its LDS reads and its scalar tap
        // window are requested together, then 2 * J * kFirChunk FMAs run; the other resident
        // wavefronts cover the wait.  Even and odd samples accumulate separately (2*J independent
        // FMA chains per wavefront).
        float2 acc0[J], acc1[J];
#pragma unroll
        for (int j = 0; j < J; ++j) acc0[j] = acc1[j] = make_float2(0.0f, 0.0f);
        const uint32_t branches = d.r;
        const float* tp = hp + (uint64_t)head * d.r * d.chunks * kFirWin;
        for (uint32_t p = 0; p < branches; ++p) {
            const float2* lp = lds + p * d.UL + tid * J;
            for (uint32_t c = 0; c < d.chunks; ++c, tp += kFirWin, lp += kFirChunk) {
                float2 x[kFirChunk];
                if constexpr (J % 2 == 0) {
                    const float4* lp4 = reinterpret_cast<const float4*>(lp);
#pragma unroll
                    for (int k = 0; k < kFirChunk / 2; ++k) {
                        const float4 v = lp4[k];
                        x[2 * k] = make_float2(v.x, v.y);
                        x[2 * k + 1] = make_float2(v.z, v.w);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < kFirChunk; ++k) x[k] = lp[k];
                }
                float tw[kFirChunk + J - 1];
#pragma unroll
                for (int k = 0; k < kFirChunk + J - 1; ++k) tw[k] = tp[k];
#pragma unroll
                for (int ii = 0; ii < kFirChunk; ++ii)
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const float h = tw[(kFirChunk - 1) + j - ii];
                        float2& a = (ii & 1) ? acc1[j] : acc0[j];
                        a.x = __builtin_fmaf(h, x[ii].x, a.x);
                        a.y = __builtin_fmaf(h, x[ii].y, a.y);
                    }
            }
        }
        float2 acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = make_float2(acc0[j].x + acc1[j].x, acc0[j].y + acc1[j].y);
        if (active) {
            float2* dst = out + ((uint64_t)row * d.heads + head) * d.M + m0 + tid * J;
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (m0 + tid * J + j < d.M) dst[j] = acc[j];
        }
    }
}

// history <- the last T-1 samples of the stream (the tail of the last row)
__global__ void fir_history_kernel(float2* __restrict__ hist, const float2* __restrict__ in, uint32_t S,
                                   uint32_t rows, uint32_t keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < keep) hist[i] = in[(uint64_t)rows * S - keep + i];
}

}  // namespace

bool fir_decimate_supported(uint64_t row_samples, uint64_t taps, uint64_t decimation) {
    if (decimation == 0 || taps == 0 || row_samples == 0) return false;
    if (row_samples % decimation != 0 || taps - 1 > row_samples || decimation > taps) return false;
    if (decimation > 32 || taps > 16384 || row_samples > (1ull << 28)) return false;
    return true;
}

namespace {

// J outputs per thread decide the FMAs per LDS byte (a CU reads 16 samples per clock from LDS and
// issues 128 FMA lanes; a sample feeds 2*J FMAs) against the LDS footprint of a tile of threads*J
// outputs (r * 8 bytes per output).  J = 2 is LDS-bandwidth bound at half the FMA peak (measured:
// VALU busy 48 %, LDS 50-80 %); even J > 2 reads 16 bytes at a 32-byte lane stride (2-way bank
// conflict: no gain), odd J is conflict-free with 8-byte reads but its tile no longer leaves room
// for three workgroups per CU.  tools/fir_sweep.py on the 251-tap /10 case: (2, 256) 48.7 us,
// (3, 256) 52.6, (5, 128) 69, (4, 128) 69 per cycle.
struct FirPlan {
    int J = 2, threads = 256;
    uint32_t UL = 0, chunks = 0;
    size_t lds = 0;
};
FirPlan fir_plan(uint32_t Q, uint32_t r) {
    auto slots = [&](int j, int threads) {  // per branch: J*(threads-1) + chunk-rounded (J + Q - 1)
        const uint32_t it = (uint32_t)((j + Q - 1 + kFirChunk - 1) / kFirChunk * kFirChunk);
        uint32_t ul = (uint32_t)(threads - 1) * j + it;
        if (j % 2 == 0) {  // 16-byte rows; 2 mod 16 spreads the staging writes of one sample time
            while (ul % 16 != 2) ++ul;
        } else {
            ul |= 1u;      // odd branch stride: conflict-free staging writes
        }
        return ul;
    };
    FirPlan p;
    p.J = (size_t)slots(4, 256) * r * sizeof(float2) <= 26 * 1024 ? 4 : 2;
    constexpr size_t kLdsBudget = 53 * 1024;  // three workgroups per CU
    while ((size_t)slots(p.J, p.threads) * r * sizeof(float2) > kLdsBudget && (p.J > 1 || p.threads > 64)) {
        if (p.threads > 64) p.threads -= 64;
        else --p.J;
    }
    p.UL = slots(p.J, p.threads);
    p.lds = (size_t)p.UL * r * sizeof(float2);
    p.chunks = (uint32_t)((p.J + Q - 1 + kFirChunk - 1) / kFirChunk);
    return p;
}
bool fir_dims(FirDims& d, FirPlan& plan, uint64_t rows, uint64_t row_samples, uint64_t ntaps, uint64_t decimation,
              uint64_t heads) {
    if (!fir_decimate_supported(row_samples, ntaps, decimation)) return false;
    d.S = (uint32_t)row_samples;
    d.r = (uint32_t)decimation;
    d.M = d.S / d.r;
    d.T = (uint32_t)ntaps;
    d.Q = (d.T + d.r - 1) / d.r;
    d.heads = (uint32_t)heads;
    d.rows = (uint32_t)rows;
    d.magic_r = d.r == 1 ? 0u : (uint32_t)(0x100000000ull / d.r) + 1;  // e / r for e * r < 2^32
    plan = fir_plan(d.Q, d.r);
    if (plan.lds > 160 * 1024) return false;
    d.OW = (uint32_t)(plan.threads * plan.J);
    d.UL = plan.UL;
    d.chunks = plan.chunks;
    d.tiles_per_row = (d.M + d.OW - 1) / d.OW;
    d.update_history = (d.Q <= d.OW && ntaps > 1) ? 1u : 0u;  // one workgroup reads the history: it renews it too
    return true;
}

}  // namespace

size_t fir_table_floats(uint64_t taps, uint64_t decimation, uint64_t heads) {
    const uint64_t Q = (taps + decimation - 1) / decimation;
    const uint64_t chunks = (kFirMaxJ + Q - 1 + kFirChunk - 1) / kFirChunk;  // the most any J needs
    return (size_t)(heads * decimation * chunks * kFirWin);
}

hipError_t launch_fir_table(float* table, const float2* taps, uint64_t ntaps, uint64_t decimation,
                            uint64_t heads, hipStream_t s) {
    FirDims d;
    FirPlan plan;
    if (!fir_dims(d, plan, 1, ntaps > 1 ? ntaps - 1 + decimation - (ntaps - 1) % decimation : decimation, ntaps,
                  decimation, heads))
        return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fir_taps_kernel, dim3(4), dim3(256), 0, s, table, taps, d);
    return hipGetLastError();
}

hipError_t launch_fir_decimate(float2* out, const float2* in, const float* table, float2* history,
                               uint64_t rows, uint64_t row_samples, uint64_t ntaps, uint64_t decimation,
                               uint64_t heads, hipStream_t s) {
    FirDims d;
    FirPlan plan;
    if (!fir_dims(d, plan, rows, row_samples, ntaps, decimation, heads)) return hipErrorInvalidValue;
    if (rows == 0 || heads == 0) return hipSuccess;
    const int J = plan.J, threads = plan.threads;
    const size_t lds = plan.lds;
    (void)hipGetLastError();
    const dim3 grid((unsigned)(rows * d.tiles_per_row));
#define JST_FIR(JJ)                                                                                   \
    do {                                                                                              \
        {                                                                                             \
            const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(fir_decimate_kernel<JJ>), 160 * 1024); \
            if (e != hipSuccess) return e;                                                            \
        }                                                                                             \
        hipLaunchKernelGGL((fir_decimate_kernel<JJ>), grid, dim3(threads), lds, s, out, in,           \
                           (const float2*)history, table, d);                           \
    } while (0)
    switch (J) {
        case 5: JST_FIR(5); break;
        case 4: JST_FIR(4); break;
        case 3: JST_FIR(3); break;
        case 2: JST_FIR(2); break;
        default: JST_FIR(1); break;
    }
#undef JST_FIR
    if (ntaps > 1 && !d.update_history) {
        const uint32_t keep = (uint32_t)ntaps - 1;
        hipLaunchKernelGGL(fir_history_kernel, dim3((keep + 255) / 256), dim3(256), 0, s, history, in, d.S,
                           d.rows, keep);
    }
    return hipGetLastError();
}

}  // namespace jst::kernels
