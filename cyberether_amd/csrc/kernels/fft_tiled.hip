// fft_tiled.hip -- LDS-tiled mixed-radix FFT for lengths the register kernels (fft_lds.hh) do not
// cover: any n whose cfftp plan (pocketfft.hh:1476-1497) uses radices <= 11, in ONE kernel when a
// transform fits an LDS tile, otherwise in TWO kernels (instead of one launch per pass through HBM,
// fft_global.hip):
//
//   plan factors f0..f(nf-1), split at g:  R1 = f0*..*f(g-1),  S = n / R1
//   kernel A ("columns"): passes 0..g-1.  For these passes the low part of the index,
//       i mod S, never changes (ido_p is a multiple of S), so the n-point array is S independent
//       columns of R1 elements at stride S; a workgroup keeps CA adjacent columns in LDS and runs
//       the g passes on them with local ido' = ido_p / S and twiddle index i = column + S*i'.
//   kernel B ("blocks"): passes g..nf-1.  After pass g-1 everything derived from the contiguous
//       block k = [k*S, (k+1)*S) stays together (Stockham: k'' = k + l1*c), so a workgroup keeps CB
//       adjacent blocks in LDS, runs the remaining passes with local l1' = l1 / R1, and writes
//       result q of block k to its autosorted position k + R1*q (CB adjacent k -> contiguous).
//   g = 0 (single kernel): R1 = 1, a "block" is a whole transform read from the strided input.
//
// Same butterflies, same twiddle table W[k] = exp(2 pi j k / n) and the same `i == 0 skips the
// twiddle` rule as every other FFT path here, so results stay bit-identical to pocketfft.  The
// window multiply (prologue) and amplitude / range (epilogue) functors of fft_lds.hh fuse in, which
// gives BASELINE config 5 (65536 points) the same one-pass-over-HBM shape as config 2:
// 8 B read + 8+8 B scratch + 4 B write per sample instead of 6 passes x 16 B + 3 elementwise passes.
#include "fft_lds.hh"
#include "fft_radix.hh"
#include "kernels.hh"

#include <cstring>
#include <type_traits>

namespace jst::kernels {

using namespace jst::dev;

namespace {

#ifndef JST_TILED_MIN_WAVES
#define JST_TILED_MIN_WAVES 6  // wavefronts per SIMD the register budget must allow (80 VGPRs)
#endif
constexpr int kMaxThreads = 1024;  // workgroup size follows the tile: about 4 elements per thread
constexpr uint32_t kTileElems = 8192;  // upper bound of a tile (64 KiB of LDS, one buffer: passes run in place)
constexpr uint64_t kWantGroups = 1024; // enough workgroups to cover 256 CUs several times

// Pad fused into the first load (core/pad/module_impl_native_cpu.cc:75-140): positions at or beyond
// `valid` along the transform axis read as zero, everything else comes from the unpadded tensor.
struct LoadCF32Padded {
    const float2* in;
    uint32_t valid;
    template <bool CONTIG>
    __device__ __forceinline__ float2 load(int64_t base, int64_t axis_stride, int pos) const {
        // branch-free: a conditional load is a branch, and hipcc drains vmcnt at every such branch when eight of them
        // are unrolled back to back (one HBM round trip per element); load a clamped position, select afterwards
        const uint32_t p = (uint32_t)pos < valid ? (uint32_t)pos : (valid ? valid - 1u : 0u);
        const float2 v = in[base + (int64_t)p * axis_stride];
        return (uint32_t)pos < valid ? v : mk(0.0f, 0.0f);
    }
};

struct TiledPlan {
    uint32_t n, nf, g, R1, S;
    uint32_t CA, CB;        // columns / blocks per workgroup: powers of two (ragged last tile)
    uint32_t ca_shift, cb_shift;
    uint32_t fact[20];
    uint32_t magic[20];     // ceil(2^32 / local ido of pass p): exact quotients for x < 2^16
    uint32_t tw_off[20];    // start of pass p in the per-pass twiddle table (fft_pass_twiddle_*)
};

// Lanes (columns / blocks) per workgroup.  Occupancy decides: a tile of <= 4096 elements means
// <= 512 threads and 32 KiB of LDS, i.e. three workgroups per CU whose load / pass / store phases
// overlap; so take the widest of {32,16} lanes that stays within 4096 elements and still yields
// kWantGroups workgroups, and fall back to 8 lanes (64-byte runs) up to the 8192-element cap.
uint32_t pick_lanes(uint32_t len, uint64_t lanes_total) {
    for (uint32_t c = 32; c >= 16; c /= 2)
        if ((uint64_t)len * c <= 4096 && (lanes_total + c - 1) / c >= kWantGroups) return c;
    return (uint64_t)len * 8 <= kTileElems ? 8u : 0u;
}
uint32_t ilog2(uint32_t v) {
    uint32_t s = 0;
    while ((1u << s) < v) ++s;
    return s;
}
uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d); }

bool make_tiled_plan(uint64_t n, uint64_t transforms, TiledPlan& p) {
    if (n < 2 || n > (1ull << 26)) return false;
    uint32_t fact[64];
    const int nf = fft_plan_factors(n, fact);
    if (nf <= 0 || nf > 20) return false;
    for (int i = 0; i < nf; ++i)
        if (fact[i] > 11) return false;
    std::memset(&p, 0, sizeof(p));
    p.n = (uint32_t)n;
    p.nf = (uint32_t)nf;
    for (int i = 0; i < nf; ++i) p.fact[i] = fact[i];
    if (transforms == 0) transforms = 1;
    if (n <= kTileElems) {  // one kernel, whole transforms per workgroup
        p.g = 0;
        p.R1 = 1;
        p.S = (uint32_t)n;
        uint32_t cb = 1;  // transforms per workgroup: keep kWantGroups workgroups when possible
        while (cb < 32 && (uint64_t)p.S * cb * 2 <= kTileElems && transforms / (cb * 2) >= kWantGroups)
            cb *= 2;
        p.CB = cb;
    } else {
        // two kernels: the split that balances R1 and S, both within a tile of >= 8 lanes
        uint32_t best = 0;
        double best_score = 1e300;
        uint64_t r1 = 1;
        for (int g = 1; g < nf; ++g) {
            r1 *= fact[g - 1];
            const uint64_t s = n / r1;
            if (r1 * 8 > kTileElems || s * 8 > kTileElems) continue;
            const double score = r1 > s ? (double)r1 / (double)s : (double)s / (double)r1;
            if (score < best_score) {
                best_score = score;
                best = (uint32_t)g;
            }
        }
        if (!best) return false;
        p.g = best;
        p.R1 = 1;
        for (uint32_t i = 0; i < best; ++i) p.R1 *= fact[i];
        p.S = p.n / p.R1;
        p.CA = pick_lanes(p.R1, transforms * p.S);
        p.CB = pick_lanes(p.S, transforms * p.R1);
        if (!p.CA || !p.CB) return false;
    }
    p.ca_shift = ilog2(p.CA ? p.CA : 1);
    p.cb_shift = ilog2(p.CB);
    // local ido of every pass: passes < g run on R1-point columns, the rest on S-point blocks
    uint32_t m = p.R1;
    for (uint32_t q = 0; q < p.g; ++q) {
        m /= p.fact[q];
        p.magic[q] = magic_of(m);
    }
    m = p.S;
    for (uint32_t q = p.g; q < p.nf; ++q) {
        m /= p.fact[q];
        p.magic[q] = magic_of(m);
    }
    uint64_t off = 0, l1 = 1;
    for (uint32_t q = 0; q < p.nf; ++q) {
        const uint64_t ido = n / (l1 * p.fact[q]);
        p.tw_off[q] = (uint32_t)off;
        off += (uint64_t)(p.fact[q] - 1) * ido;
        l1 *= p.fact[q];
    }
    return true;
}

// Workgroup b runs on XCD b % 8 (observed placement, used for speed only) and every XCD has an L2 of its own.
// Neighbouring tiles share cache lines (a column tile's 128-byte runs start 8 bytes off the line grid whenever the
// row length is odd in elements; a block tile writes 64-byte halves of 128-byte lines), so XCD k takes a CONTIGUOUS
// run of tiles: the second touch of a line is an L2 hit / an L2 merge instead of a second HBM transaction.
__device__ __forceinline__ uint32_t xcd_contiguous_tile(uint32_t b, uint32_t grid) {
    const uint32_t q = grid >> 3, r = grid & 7u, k = b & 7u;
    return k * q + (k < r ? k : r) + (b >> 3);
}

__device__ __forceinline__ void outer_bases(const FftLayout& L, uint64_t t, int64_t& in_base,
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

// One radix-IP pass over an LDS tile, IN PLACE: every thread first reads the inputs of all of its
// butterflies into registers, the workgroup synchronises, then the outputs go back to the same
// buffer (Stockham positions) -- one buffer instead of a ping-pong pair, i.e. half the LDS and
// twice the resident workgroups per CU.  The workgroup has at least tile/8 threads, so a thread
// owns at most ceil(8/IP) butterflies: NB below is a compile-time function of the radix.
// Element (x, lane) lives at x * pitch + lane; `lanes` (1 << lane_shift) independent
// sub-transforms of `len` points run side by side.
//   butterfly (i, k):  reads x = i + ido*(j + IP*k), writes x = i + ido*(k + l1loc*c)
//   twiddle:           PT[(c-1)*ido_glob + ig], ig = tw_i0 + lane*tw_lane + tw_is*i the GLOBAL i of
//                      the butterfly; PT is this pass's slice of the per-pass table (pocketfft's
//                      own layout, comp_twiddle :1513-1535: tw[(j-1)*(ido-1)+i-1] = W[j*l1*i]), so
//                      adjacent lanes read adjacent entries instead of gathering W at stride c*l1.
template <int IP>
constexpr int butterflies_per_thread() { return IP <= 3 ? 4 : (IP <= 7 ? 2 : 1); }

template <int IP, bool FWD>
__device__ __forceinline__ void tile_pass(float2* __restrict__ buf, const float2* __restrict__ PT,
                                          uint32_t len, uint32_t lane_shift, uint32_t live_lanes,
                                          uint32_t pitch, uint32_t ido, uint32_t ido_magic,
                                          uint32_t l1loc, uint32_t ido_glob, uint32_t tw_is,
                                          uint32_t tw_i0, uint32_t tw_lane) {
    constexpr int NB = butterflies_per_thread<IP>();
    const uint32_t nb = (len / IP) << lane_shift;
    float2 x[NB][IP];
    uint32_t wr_base[NB], step[NB];  // step = global i of the butterfly; 0xffffffff marks an idle slot
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const uint32_t b = threadIdx.x + (uint32_t)r * blockDim.x;
        const uint32_t lane = b & ((1u << lane_shift) - 1u), rest = b >> lane_shift;
        step[r] = 0xffffffffu;
        wr_base[r] = 0;
        if (b < nb && lane < live_lanes) {
            const uint32_t k = ido > 1 ? __umulhi(rest, ido_magic) : rest;
            const uint32_t i = rest - k * ido;
            const float2* rd = buf + (i + ido * IP * k) * pitch + lane;
#pragma unroll
            for (int j = 0; j < IP; ++j) x[r][j] = rd[(uint32_t)j * ido * pitch];
            wr_base[r] = (i + ido * k) * pitch + lane;
            step[r] = tw_i0 + lane * tw_lane + tw_is * i;
        }
    }
    __syncthreads();  // every input of this pass is in registers
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        if (step[r] == 0xffffffffu) continue;
        butterfly_any<IP, FWD>(x[r]);
        if (step[r] != 0) {
#pragma unroll
            for (int c = 1; c < IP; ++c)
                x[r][c] = special_mul<FWD>(x[r][c], PT[(uint32_t)(c - 1) * ido_glob + step[r]]);
        }
        float2* wr = buf + wr_base[r];
#pragma unroll
        for (int c = 0; c < IP; ++c) wr[(uint32_t)c * l1loc * ido * pitch] = x[r][c];
    }
    __syncthreads();  // outputs visible before the next pass (or the store) reads them
}

template <bool FWD>
__device__ __forceinline__ void tile_pass_any(uint32_t ip, float2* buf, const float2* PT, uint32_t len,
                                              uint32_t lane_shift, uint32_t live_lanes, uint32_t pitch,
                                              uint32_t ido, uint32_t ido_magic, uint32_t l1loc,
                                              uint32_t ido_glob, uint32_t tw_is, uint32_t tw_i0,
                                              uint32_t tw_lane) {
#define JST_TP(IP)                                                                                \
    tile_pass<IP, FWD>(buf, PT, len, lane_shift, live_lanes, pitch, ido, ido_magic, l1loc,        \
                       ido_glob, tw_is, tw_i0, tw_lane)
    switch (ip) {
        case 2: JST_TP(2); break;
        case 3: JST_TP(3); break;
        case 4: JST_TP(4); break;
        case 5: JST_TP(5); break;
        case 7: JST_TP(7); break;
        case 8: JST_TP(8); break;
        default: JST_TP(11); break;
    }
#undef JST_TP
}

// ---- kernel A: passes 0..g-1 on CA adjacent columns ---------------------------------------------
template <bool FWD, class Pro>
__global__ __launch_bounds__(kMaxThreads, JST_TILED_MIN_WAVES) void fft_tile_columns_kernel(const FftLayout L,
                                                                    const TiledPlan P,
                                                                    const float2* __restrict__ W,
                                                                    const Pro pro,
                                                                    float2* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* buf0 = reinterpret_cast<float2*>(smem_raw);
    const uint32_t tiles_per_t = (P.S + P.CA - 1) >> P.ca_shift;
    const uint32_t bid = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const uint64_t t = bid / tiles_per_t;
    const uint32_t c0 = (bid % tiles_per_t) << P.ca_shift;
    const uint32_t live = (P.S - c0 < P.CA) ? (P.S - c0) : P.CA;
    const uint32_t tile = P.R1 << P.ca_shift;
    int64_t in_base, out_base;
    outer_bases(L, t, in_base, out_base);
    // Eight loads in flight per thread (a tile is at most 8 elements per thread): written as a plain loop, hipcc
    // waits for every load before the LDS write that consumes it -- eight serial HBM round trips per workgroup
    // (round 1: ~10 of the ~20 us a column workgroup lived).
    for (uint32_t i0 = threadIdx.x; i0 < tile; i0 += 8 * blockDim.x) {
        float2 v[8];
        bool ok[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // unconditional loads from clamped positions: no branch between them
            const uint32_t idx = i0 + (uint32_t)k * blockDim.x;
            const uint32_t r = idx >> P.ca_shift, col = idx & (P.CA - 1u);
            ok[k] = idx < tile && col < live;
            const uint32_t rc = r < P.R1 ? r : P.R1 - 1u, cc = col < live ? col : live - 1u;
            v[k] = pro.template load<false>(in_base, L.in_axis_stride, (int)(c0 + cc + P.S * rc));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (ok[k]) buf0[i0 + (uint32_t)k * blockDim.x] = v[k];
    }
    __syncthreads();
    uint32_t l1 = 1, m = P.R1;
    for (uint32_t p = 0; p < P.g; ++p) {
        const uint32_t ip = P.fact[p];
        m /= ip;  // local ido'
        tile_pass_any<FWD>(ip, buf0, W + P.tw_off[p], P.R1, P.ca_shift, live, P.CA, m, P.magic[p], l1,
                           m * P.S, P.S, c0, 1u);
        l1 *= ip;
    }
    float2* o = scratch + t * P.n;
    for (uint32_t idx = threadIdx.x; idx < tile; idx += blockDim.x) {
        const uint32_t r = idx >> P.ca_shift, col = idx & (P.CA - 1u);
        if (col < live) o[c0 + col + P.S * r] = buf0[idx];
    }
}

// ---- kernel B: passes g..nf-1 on CB adjacent blocks (or whole transforms when g == 0) -----------
template <bool FWD, class Pro, class Epi>
__global__ __launch_bounds__(kMaxThreads, JST_TILED_MIN_WAVES) void fft_tile_blocks_kernel(const FftLayout L,
                                                                   const TiledPlan P,
                                                                   const float2* __restrict__ W,
                                                                   const Pro pro, const Epi epi,
                                                                   const float2* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int64_t lane_in[32], lane_out[32];  // per lane of the tile: tensor row bases
    const uint32_t pitch = P.CB | 1u;  // odd pitch: the x-major global loops stay conflict-free
    float2* buf0 = reinterpret_cast<float2*>(smem_raw);
    // R1 > 1: a tile is CB adjacent blocks of ONE transform; R1 == 1: CB adjacent transforms
    uint64_t t0;
    uint32_t k0, live;
    const uint32_t bid = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    if (P.R1 > 1) {
        const uint32_t tiles_per_t = (P.R1 + P.CB - 1) >> P.cb_shift;
        t0 = bid / tiles_per_t;
        k0 = (bid % tiles_per_t) << P.cb_shift;
        live = (P.R1 - k0 < P.CB) ? (P.R1 - k0) : P.CB;
    } else {
        t0 = (uint64_t)bid << P.cb_shift;
        k0 = 0;
        live = (uint32_t)((L.transforms - t0 < P.CB) ? (L.transforms - t0) : P.CB);
    }
    if (threadIdx.x < live) {
        int64_t ib, ob;
        outer_bases(L, P.R1 > 1 ? t0 : t0 + threadIdx.x, ib, ob);
        lane_in[threadIdx.x] = ib;
        lane_out[threadIdx.x] = ob;
    }
    __syncthreads();
    const uint32_t tile = P.S * live;
    // load: x fastest (contiguous in memory for both the dense scratch and a dense input row)
    const float2* blk = scratch + (t0 * P.R1 + k0) * P.S;  // only dereferenced when g > 0
    // eight loads in flight per thread: unconditional loads from clamped indices (a conditional load is a branch and
    // hipcc drains vmcnt at each of them), the g == 0 / g > 0 choice hoisted out of the unrolled body
    auto load_tile = [&](auto from_scratch) {
        for (uint32_t i0 = threadIdx.x; i0 < tile; i0 += 8 * blockDim.x) {
            float2 v[8];
            uint32_t slot[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t idx = i0 + (uint32_t)k * blockDim.x;
                const uint32_t cidx = idx < tile ? idx : tile - 1u;
                const uint32_t kb = cidx / P.S, x = cidx - kb * P.S;
                if constexpr (decltype(from_scratch)::value) v[k] = blk[cidx];
                else v[k] = pro.template load<false>(lane_in[kb], L.in_axis_stride, (int)x);
                slot[k] = idx < tile ? x * pitch + kb : 0xffffffffu;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (slot[k] != 0xffffffffu) buf0[slot[k]] = v[k];
        }
    };
    if (P.g == 0) load_tile(std::false_type{});
    else load_tile(std::true_type{});
    __syncthreads();
    const float2* src = buf0;
    uint32_t l1 = P.R1, ido = P.S;
    for (uint32_t p = P.g; p < P.nf; ++p) {
        const uint32_t ip = P.fact[p];
        ido /= ip;
        tile_pass_any<FWD>(ip, buf0, W + P.tw_off[p], P.S, P.cb_shift, live, pitch, ido, P.magic[p],
                           l1 / P.R1, ido, 1u, 0u, 0u);
        l1 *= ip;
    }
    // store result q of block (t, k) at k + R1*q: block index fastest when R1 > 1 (adjacent k are
    // adjacent in memory), q fastest for whole transforms
    if (P.R1 > 1) {
        const uint32_t total = P.S << P.cb_shift;
        for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const uint32_t kb = idx & (P.CB - 1u), q = idx >> P.cb_shift;
            if (kb < live)
                epi.template store<false>(lane_out[kb], L.out_axis_stride,
                                          (int)(k0 + kb + P.R1 * q), src[q * pitch + kb]);
        }
    } else {
        for (uint32_t idx = threadIdx.x; idx < tile; idx += blockDim.x) {
            const uint32_t kb = idx / P.S, q = idx - kb * P.S;
            epi.template store<false>(lane_out[kb], L.out_axis_stride, (int)q, src[q * pitch + kb]);
        }
    }
}

// at least tile/8 threads (the in-place passes hold <= 8 points per thread); a multiple of 256 so
// that every SIMD of the CU gets the same number of wavefronts and a second / third workgroup fits
inline unsigned threads_for(uint64_t tile_elems) {
    uint64_t t = ((tile_elems / 8 + 255) / 256) * 256;
    if (t < 256) t = 256;
    if (t > (uint64_t)kMaxThreads) t = kMaxThreads;
    return (unsigned)t;
}

template <bool FWD, class Pro, class Epi>
hipError_t launch_tiled(const TiledPlan& P, const FftLayout& L, const float2* W, const Pro& pro,
                        const Epi& epi, float2* scratch, hipStream_t s) {
    if (L.transforms == 0) return hipSuccess;
    (void)hipGetLastError();
    if (P.g > 0) {
        if (!scratch) return hipErrorInvalidValue;
        const size_t lds_a = (size_t)P.R1 * P.CA * sizeof(float2);
        auto ka = fft_tile_columns_kernel<FWD, Pro>;
        {  // tiles never exceed kTileElems
            const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(ka), (int)(kTileElems * sizeof(float2)));
            if (e != hipSuccess) return e;
        }
        const uint64_t blocks = L.transforms * ((P.S + P.CA - 1) / P.CA);
        if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ka, dim3((unsigned)blocks), dim3(threads_for((uint64_t)P.R1 * P.CA)), lds_a, s, L, P, W,
                           pro, scratch);
    }
    const size_t lds_b = (size_t)P.S * (P.CB | 1u) * sizeof(float2);
    auto kb = fft_tile_blocks_kernel<FWD, Pro, Epi>;
    {  // pitch CB|1 adds at most one lane of padding per row
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(kb), (int)(2 * kTileElems * sizeof(float2)));
        if (e != hipSuccess) return e;
    }
    if (lds_b > 2 * kTileElems * sizeof(float2)) return hipErrorInvalidValue;
    const uint64_t blocks = P.R1 > 1 ? L.transforms * ((P.R1 + P.CB - 1) / P.CB)
                                     : (L.transforms + P.CB - 1) / P.CB;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kb, dim3((unsigned)blocks), dim3(threads_for((uint64_t)P.S * P.CB)), lds_b, s, L, P, W, pro, epi,
                       (const float2*)scratch);
    return hipGetLastError();
}

template <class Pro, class Epi>
hipError_t dispatch_dir(bool forward, const TiledPlan& P, const FftLayout& L, const float2* W,
                        const Pro& pro, const Epi& epi, float2* scratch, hipStream_t s) {
    return forward ? launch_tiled<true>(P, L, W, pro, epi, scratch, s)
                   : launch_tiled<false>(P, L, W, pro, epi, scratch, s);
}

}  // namespace

// Per-pass twiddle table for the tiled kernels: pass p occupies (ip-1)*ido entries laid out
// [c-1][i] with value W[c * l1 * i] (i = 0 is present but never read).  Entry count / filler.
uint64_t fft_pass_twiddle_count(uint64_t n) {
    uint32_t fact[64];
    const int nf = fft_plan_factors(n, fact);
    uint64_t total = 0, l1 = 1;
    for (int q = 0; q < nf; ++q) {
        total += (uint64_t)(fact[q] - 1) * (n / (l1 * fact[q]));
        l1 *= fact[q];
    }
    return total;
}
void fft_pass_twiddle_fill(uint64_t n, const float* w_interleaved, float* out_interleaved) {
    uint32_t fact[64];
    const int nf = fft_plan_factors(n, fact);
    uint64_t off = 0, l1 = 1;
    for (int q = 0; q < nf; ++q) {
        const uint64_t ip = fact[q], ido = n / (l1 * ip);
        for (uint64_t c = 1; c < ip; ++c)
            for (uint64_t i = 0; i < ido; ++i) {
                const uint64_t src = c * l1 * i, dst = off + (c - 1) * ido + i;
                out_interleaved[2 * dst] = w_interleaved[2 * src];
                out_interleaved[2 * dst + 1] = w_interleaved[2 * src + 1];
            }
        off += (ip - 1) * ido;
        l1 *= ip;
    }
}

bool fft_tiled_supported(uint64_t n) {
    TiledPlan p;
    return make_tiled_plan(n, 1, p);
}
bool fft_tiled_needs_scratch(uint64_t n) {
    TiledPlan p;
    return make_tiled_plan(n, 1, p) && p.g > 0;
}

hipError_t launch_fft_c2c_tiled(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                                const float2* in, float2* out, float2* scratch, hipStream_t s) {
    TiledPlan p;
    if (!make_tiled_plan(n, L.transforms, p)) return hipErrorInvalidValue;
    return dispatch_dir(forward, p, L, W, LoadCF32{in}, StoreCF32{out}, scratch, s);
}

hipError_t launch_fft_c2c_tiled_padded(uint64_t n, uint64_t valid, bool forward, const FftLayout& L,
                                       const float2* W, const float2* in, float2* out,
                                       float2* scratch, hipStream_t s) {
    TiledPlan p;
    if (valid > n || !make_tiled_plan(n, L.transforms, p)) return hipErrorInvalidValue;
    return dispatch_dir(forward, p, L, W, LoadCF32Padded{in, (uint32_t)valid}, StoreCF32{out}, scratch, s);
}

hipError_t launch_spectrum_fused_tiled(uint64_t n, const FftLayout& L, const float2* W,
                                       const float2* in, const float2* window,
                                       int64_t window_stride, float* out, float amp_coeff,
                                       bool with_range, float range_scale, float range_offset,
                                       bool fast, float guard_h0, float guard_h1, float2* scratch,
                                       hipStream_t s) {
    TiledPlan p;
    if (!make_tiled_plan(n, L.transforms, p)) return hipErrorInvalidValue;
    const LoadCF32TimesWindow pro{in, window, window_stride};
    if (with_range) {
        if (fast)
            return launch_tiled<true>(p, L, W, pro,
                                      StoreAmplitudeRangeT<true>{out, amp_coeff, range_scale, range_offset, dev::BinGuard{guard_h0, guard_h1}},
                                      scratch, s);
        return launch_tiled<true>(p, L, W, pro,
                                  StoreAmplitudeRangeT<false>{out, amp_coeff, range_scale, range_offset, dev::BinGuard{}},
                                  scratch, s);
    }
    if (fast) return launch_tiled<true>(p, L, W, pro, StoreAmplitudeT<true>{out, amp_coeff}, scratch, s);
    return launch_tiled<true>(p, L, W, pro, StoreAmplitudeT<false>{out, amp_coeff}, scratch, s);
}

}  // namespace jst::kernels
