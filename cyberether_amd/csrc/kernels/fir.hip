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
#include <algorithm>
#include <type_traits>

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

// One tile (OW consecutive outputs of one row) of the direct form; `block` = row * tiles_per_row + tile.  The workgroup may
// have more threads than the tile has output slots (the MFMA kernel's fix-up workgroup, below): the extra ones only stage.
template <int J>
__device__ __forceinline__ void fir_tile(float2* __restrict__ out, const float2* __restrict__ in, const float2* hist,
                                         const float* __restrict__ hp, const FirDims& d, uint32_t block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);  // [r][UL]
    const uint32_t tid = threadIdx.x;
    const uint32_t row = block / d.tiles_per_row, m0 = (block % d.tiles_per_row) * d.OW;

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

    const bool active = tid * J < d.OW && m0 + tid * J < d.M;
    if (tid * J >= d.OW) return;  // (no barrier below this point)
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

template <int J>
__global__ __launch_bounds__(kFirThreads) void fir_decimate_kernel(float2* __restrict__ out,
                                                                   const float2* __restrict__ in,
                                                                   const float2* hist,
                                                                   const float* __restrict__ hp, FirDims d) {
    fir_tile<J>(out, in, hist, hp, d, blockIdx.x);
}

// =============================================================================================================================
// The same sums on the MATRIX cores (round 6).  The direct form above is LDS-bandwidth bound at half the vector FMA rate (a
// sample read from LDS feeds 2 * J FMAs); v_mfma_f32_16x16x4_f32 issues at the same 64 FLOP / clk / SIMD as the vector FMA
// but takes each operand ONCE from a register for 16 uses, and runs in a pipe of its own.  The FIR as a banded-Toeplitz
// product: the rows of one cycle are ONE continuous stream (row k + 1 continues row k).  A wavefront takes 128 consecutive
// outputs at a time -- 8 segments of 16 outputs: 16 real streams (re and im of each segment) -- and forms
//
//     D[stream m][output n] = sum_s A[m][s] * B[s][n],   A[m][s] = x_m[s0 + s],   B[s][n] = h[n*R + lead - s]   (0 outside the taps)
//
// over the 4 * NT samples s a 16-output tile can see (15 R + T of them carry a tap: 62 % of the products for 251 taps at R = 10).
// B is the same for every tile: NT registers per lane, loaded once per wavefront (fir_taps builds the table).  A: lane l holds
// stream l % 16, sample 4 j + l / 16 of step j -- one ds_read_b32 per MFMA from the wavefront's OWN LDS image of the 1280 + 4 NT - 160
// consecutive samples its 8 windows cover (12 KiB, loaded as twelve 1 KiB dwordx4 bursts per tile: the first attempt read the
// operand straight from global memory, 32 bytes per segment and step, and the 8192 interleaved sequential streams it made of
// the input ran the HBM at 1.5 TB/s).  The image carries 16 pad bytes per 160 samples, so that the 16 streams of a step --
// 8 addresses 1296 bytes apart, 2 dwords each, times four k -- fall on 64 different banks.  Two images per wavefront: the next
// tile's bursts are in flight while this tile's 101 steps run, and go to LDS behind them.  No barrier anywhere: a wavefront
// reads only what it wrote.  Even and odd steps accumulate separately (an MFMA on the accumulator of the previous one would wait
// 40 cycles for a 32-cycle issue slot).  Outputs whose taps reach back into the PREVIOUS cycle (the history tensor) are left to one
// extra workgroup that runs the direct form's first tile (it also renews the history, as before): every other sample before the
// stream reads as zero through the buffer descriptor's range check.  Same sums as the direct form up to the order of the
// additions; against the bit-exact FFT chain 1.2e-6 of the peak (tests/test_gpu_filter_fast.py).
struct FirMfmaDims {
    uint32_t tiles;       // wave tiles of 128 outputs
    uint32_t total;       // outputs of the whole stream (rows * M)
    uint32_t skip;        // outputs [0, skip) belong to the fix-up workgroup
    uint32_t in_bytes;    // the stream's size
    int32_t lead;         // samples a window starts before its first output's sample 0
    uint32_t lds_fixup;   // bytes the fix-up tile needs (the MFMA images start behind them)
};

typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

constexpr int kMfmaWaves = 4;       // wavefronts per workgroup, one per SIMD
constexpr int kMfmaPieces = 12;     // 16-byte pieces per lane and image: 64 * 12 * 2 = 1536 samples >= 1280 + 4 NT - 160

template <int R, int NT>
__global__ __launch_bounds__(kMfmaWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) void fir_mfma_kernel(float2* __restrict__ out, const float2* __restrict__ in,
                                                                      const float2* hist, const float* __restrict__ hp,
                                                                      const float* __restrict__ btab, FirDims d, FirMfmaDims md) {
    if (blockIdx.x == gridDim.x - 1) {  // the tile that reads (and renews) the history: the direct form, 64 outputs
        fir_tile<1>(out, in, hist, hp, d, 0);
        return;
    }
    static_assert(R == 10, "the image's pad rule is written for 160 samples per segment");
    static_assert(8 * 16 * R + 4 * NT - 16 * R <= 64 * kMfmaPieces * 2, "the image does not hold a tile's windows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr uint32_t kImageBytes = (64u * kMfmaPieces + (64u * kMfmaPieces) / 80u + 1u) * 16u;  // pieces + one pad piece per 80
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    unsigned char* image0 = smem_raw + (size_t)wv * 2u * kImageBytes;
    const uint32_t stream = lane & 15u, k = lane >> 4;
    float tb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) tb[j] = btab[j * 64 + lane];
    const rsrc_t rs = make_rsrc(in, md.in_bytes);
    // staging: piece p = lane + 64 i holds image samples 2p, 2p + 1; in LDS behind p / 80 pads
    uint32_t wr[kMfmaPieces];
#pragma unroll
    for (int i = 0; i < kMfmaPieces; ++i) {
        const uint32_t p = lane + 64u * (uint32_t)i;
        wr[i] = (p + p / 80u) * 16u;
    }
    // operand reads: segment sigma = stream / 2 starts at image sample 160 sigma -> byte 1296 sigma; sample k, component stream & 1
    const uint32_t rd = (stream >> 1) * 1296u + k * 8u + (stream & 1u) * 4u;
    const uint32_t q = lane >> 4, n = lane & 15u;
    const uint32_t waves_total = (gridDim.x - 1u) * (uint32_t)kMfmaWaves;
    uint32_t tile = blockIdx.x * (uint32_t)kMfmaWaves + wv;

    auto fetch = [&](uint32_t t, v4u (&v)[kMfmaPieces]) {
        // first image sample of tile t: t * 128 * R - lead (before the stream: the offset wraps far out of range -> 0)
        const uint32_t base = (uint32_t)(((int64_t)t * 128 * R - md.lead) * 8) + lane * 16u;
#pragma unroll
        for (int i = 0; i < kMfmaPieces; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (uint32_t)(i * 1024), 0u, 0);
    };
    auto stash = [&](unsigned char* image, const v4u (&v)[kMfmaPieces]) {
#pragma unroll
        for (int i = 0; i < kMfmaPieces; ++i) *reinterpret_cast<v4u*>(image + wr[i]) = v[i];
    };
    auto put = [&](uint64_t g, float re, float im) {
        if (g >= md.skip && g < md.total) out[g] = make_float2(re, im);
    };

    auto compute = [&](const unsigned char* image, uint32_t t) {
        v4f acc0 = v4f{0.0f, 0.0f, 0.0f, 0.0f}, acc1 = acc0;
        const unsigned char* a_ptr = image + rd;
        // step j reads image samples 160 sigma + 4 j + k: one more pad behind every 160 samples (40 steps).  The operands
        // of the NEXT chunk of steps are requested before this chunk's MFMAs issue (the scheduler is held to that order:
        // left alone it waited for every operand pair right in front of its two MFMAs -- an LDS round trip per 64 cycles of MFMA).
        constexpr int kChunk = 12;
        constexpr int kChunks = (NT + kChunk - 1) / kChunk;
        float a_op[2][kChunk];
        auto request = [&](int c, float (&dst)[kChunk]) {
#pragma unroll
            for (int i = 0; i < kChunk; ++i) {
                const int j = c * kChunk + i;
                if (j < NT) dst[i] = *reinterpret_cast<const float*>(a_ptr + j * 32 + (j / 40) * 16);
            }
        };
        request(0, a_op[0]);
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            if (c + 1 < kChunks) request(c + 1, a_op[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kChunk; ++i) {
                const int j = c * kChunk + i;
                if (j < NT) {
                    if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op[c & 1][i], tb[j], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op[c & 1][i], tb[j], acc0, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // registers 0,1 = (re, im) of segment 2 q, registers 2,3 = of segment 2 q + 1; the lane's output within the segment: n
        const uint64_t g0 = (uint64_t)t * 128u + 32u * q + n, g1 = g0 + 16u;
        put(g0, acc0[0] + acc1[0], acc0[1] + acc1[1]);
        put(g1, acc0[2] + acc1[2], acc0[3] + acc1[3]);
    };

    // Two tiles in the pipe per wavefront: tile i runs out of one image while the bursts of tile i + 1 are in flight; they go
    // to the other image behind tile i's steps.  (Measured, same box: a third tile in flight -- bursts requested two tiles
    // ahead, with hipcc's waits 32.5 us, with inline-asm bursts and our own vmcnt accounting 32.3 us -- against 29.5 us for
    // this form; two wavefronts per SIMD on one image each: 32.5 us.  The kernel moves 4.8 TB/s: it does not wait for latency.)
    v4u stage[kMfmaPieces];
    if (tile < md.tiles) {
        fetch(tile, stage);
        stash(image0, stage);
    }
    uint32_t flip = 0;
    for (; tile < md.tiles; tile += waves_total, flip ^= 1u) {
        const uint32_t next = tile + waves_total;
        if (next < md.tiles) fetch(next, stage);  // in flight behind this tile's steps
        compute(image0 + flip * kImageBytes, tile);
        if (next < md.tiles) stash(image0 + (flip ^ 1u) * kImageBytes, stage);
    }
}

// B operand of the MFMA form, [NT][64]: lane l of step j holds h[n*R + lead - 4j - k], n = l % 16, k = l / 16
__global__ void fir_mfma_taps_kernel(float* __restrict__ btab, const float2* __restrict__ taps, uint32_t T, uint32_t R, uint32_t NT,
                                     int32_t lead) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NT * 64u) return;
    const uint32_t lane = e & 63u, j = e >> 6;
    const int32_t idx = (int32_t)((lane & 15u) * R) + lead - 4 * (int32_t)j - (int32_t)(lane >> 4);
    btab[e] = (idx >= 0 && idx < (int32_t)T) ? taps[idx].x : 0.0f;
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

namespace {
// MFMA form: the (R, NT) pairs fir_mfma_kernel is instantiated for; NT steps of 4 samples must cover 15 R + T samples
constexpr int kFirMfmaMaxSteps = 101;
int fir_mfma_steps(uint64_t taps, uint64_t decimation, uint64_t heads) {
    if (heads != 1 || decimation != 10) return 0;
    if (taps + 150 <= 4 * 51) return 51;     // <= 54 taps (multi-fm.yml's 51)
    if (taps + 150 <= 4 * 101) return 101;   // <= 254 taps (BASELINE config 3's 251)
    return 0;
}
size_t fir_direct_table_floats(uint64_t taps, uint64_t decimation, uint64_t heads) {
    const uint64_t Q = (taps + decimation - 1) / decimation;
    const uint64_t chunks = (kFirMaxJ + Q - 1 + kFirChunk - 1) / kFirChunk;  // the most any J needs
    return (size_t)(heads * decimation * chunks * kFirWin);
}
}  // namespace

// the direct form's step-major tap windows, then (when the MFMA form takes the shape) its B operand [NT][64]
size_t fir_table_floats(uint64_t taps, uint64_t decimation, uint64_t heads) {
    return fir_direct_table_floats(taps, decimation, heads) + (size_t)kFirMfmaMaxSteps * 64;
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
    if (const int nt = fir_mfma_steps(ntaps, decimation, heads)) {
        const int32_t lead = 4 * nt - 1 - 15 * (int32_t)decimation;
        hipLaunchKernelGGL(fir_mfma_taps_kernel, dim3((unsigned)((nt * 64 + 255) / 256)), dim3(256), 0, s,
                           table + fir_direct_table_floats(ntaps, decimation, heads), taps, (uint32_t)ntaps, (uint32_t)decimation,
                           (uint32_t)nt, lead);
    }
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
    // The MFMA form: one head, the instantiated (R, NT) shapes, a stream the buffer descriptor can address, and a direct-form
    // first tile of (J = 2, 256 threads) for the outputs that reach into the history.  JST_FIR_DIRECT (switch) keeps the direct form.
    const int nt = switch_value(SW_FIR_DIRECT) ? 0 : fir_mfma_steps(ntaps, decimation, heads);
    const uint64_t total = rows * d.M, in_bytes = rows * row_samples * sizeof(float2);
    // the fix-up tile: the direct form at its smallest (J = 1, 64 output slots) -- it only has to cover the outputs whose taps
    // reach into the history (Q of them), and it is ONE workgroup beside 255 that each do 1/255 of the rest
    FirDims fix = d;
    FirPlan small;
    small.J = 1;
    small.threads = 64;
    small.chunks = (uint32_t)((1 + d.Q - 1 + kFirChunk - 1) / kFirChunk);
    small.UL = ((uint32_t)(small.threads - 1) + small.chunks * kFirChunk) | 1u;
    small.lds = (size_t)small.UL * d.r * sizeof(float2);
    fix.OW = 64;
    fix.UL = small.UL;
    fix.chunks = small.chunks;
    fix.tiles_per_row = (d.M + fix.OW - 1) / fix.OW;
    fix.update_history = (d.Q <= fix.OW && ntaps > 1) ? 1u : 0u;
    // (the tap windows fir_taps built are laid out for plan.chunks chunks per branch: the fix-up tile must walk the same layout)
    if (nt && fix.update_history && small.chunks == plan.chunks && in_bytes < (1ull << 31) && total >= 4096 && d.M >= 64) {
        int cus = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        FirMfmaDims md;
        md.tiles = (uint32_t)((total + 127) / 128);  // 8 segments x 16 outputs per wavefront and tile
        uint64_t groups = (uint64_t)(cus > 1 ? cus - 1 : 255);  // + the fix-up workgroup: every workgroup resident at once
        if (groups * kMfmaWaves > md.tiles) groups = (md.tiles + kMfmaWaves - 1) / kMfmaWaves;
        md.total = (uint32_t)total;
        md.skip = fix.OW;  // the fix-up workgroup's tile: outputs [0, 64) of row 0
        md.in_bytes = (uint32_t)in_bytes;
        md.lead = 4 * nt - 1 - 15 * (int32_t)decimation;
        md.lds_fixup = (uint32_t)small.lds;
        const size_t image = (size_t)(64 * kMfmaPieces + (64 * kMfmaPieces) / 80 + 1) * 16;
        const size_t lds_m = std::max(small.lds, (size_t)kMfmaWaves * 2 * image);
        const float* btab = table + fir_direct_table_floats(ntaps, decimation, heads);
        const dim3 grid_m((unsigned)(groups + 1));
        if (nt == 101) {
            const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(fir_mfma_kernel<10, 101>), 160 * 1024);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fir_mfma_kernel<10, 101>), grid_m, dim3(kMfmaWaves * 64), lds_m, s, out, in, (const float2*)history, table, btab, fix, md);
        } else {
            const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(fir_mfma_kernel<10, 51>), 160 * 1024);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((fir_mfma_kernel<10, 51>), grid_m, dim3(kMfmaWaves * 64), lds_m, s, out, in, (const float2*)history, table, btab, fix, md);
        }
        return hipGetLastError();
    }
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
