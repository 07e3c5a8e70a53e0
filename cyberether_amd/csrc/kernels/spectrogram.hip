// spectrogram.hip -- the Spectrogram module's compute (decaying 2-D persistence histogram) as a
// column-tiled LDS histogram on gfx950.  The reference has a CPU implementation only
// (src/domains/visualization/spectrogram/module_impl_native_cpu.cc:61-87):
//
//     for all bins: bins *= decay
//     for b, x:  index = (U64)(in[b,x] * height);  if 0 < index < height:
//                bins[x + index*width] = min(bins[...] + 0.02f, 1.0f)
//
// The float update is the SAME operation repeated once per hit, so the result depends only on
// the hit COUNT per bin, not on the order: we count hits with integer LDS atomics (exact, order
// free) and then apply min(v + 0.02f, 1.0f) count-times to the decayed value -- bit-identical
// to the sequential CPU loop.  The bin index rule is stated without the reference's undefined
// float->U64 cast: hit <=> 1.0f <= f < (float)height, index = (u32)f ("hard
// parts"; on x86-64 every other input fails 0 < index < height).
//
// Mapping: one workgroup owns a tile of TW adjacent columns for ALL batches (the only place a
// bin's hits can come from), so no inter-workgroup communication exists.  Rows of the F32[B,N]
// input are read TW*4 bytes at a time (64 B at TW = 16) -- they were just written by the
// spectrum kernel and sit in L2 / Infinity Cache.  The state tile is read-modified-written once.
#include <cstdlib>

#include "device_math.hh"
#include "kernels.hh"
#include "spectrogram_body.hh"

namespace jst::kernels {

namespace {
using namespace specdev;

template <int TW, int COPIES, int kThreads = kThreadsDefault, int kDepthT = 16, bool BUF = false>
__global__ __launch_bounds__(kThreads) void spectrogram_kernel(
    float* __restrict__ bins, const float* __restrict__ in, uint64_t in_offset, uint32_t batches,
    uint32_t width, uint32_t height, int64_t batch_stride, int64_t elem_stride, float decay) {
    spectrogram_body<TW, COPIES, kThreads, kDepthT, BUF>(bins, in, in_offset, batches, width, height, batch_stride,
                                                        elem_stride, decay, blockIdx.x, gridDim.x);
}

// The same histogram with the COUNTS as the result (U32[height][width], written over `counts`): the device half of
// the exact multi-GPU spectrogram merge -- spectrogram/module_impl_native_cpu.cc:61-87 applies min(v + 0.02f, 1.0f)
// once per hit, so the update depends on the hit count per bin only, and integer counts add exactly across ranks.
template <int TW, int COPIES>
__global__ __launch_bounds__(kThreadsDefault) void spectrogram_counts_kernel(
    uint32_t* __restrict__ counts, const float* __restrict__ in, uint64_t in_offset, uint32_t batches,
    uint32_t width, uint32_t height, int64_t batch_stride, int64_t elem_stride) {
    spectrogram_body<TW, COPIES, kThreadsDefault, 16, false, true>(reinterpret_cast<float*>(counts), in, in_offset, batches,
                                                                   width, height, batch_stride, elem_stride, 1.0f,
                                                                   blockIdx.x, gridDim.x);
}

// bins = decay * bins, then min(v + 0.02f, 1.0f) applied counts[i] times: what spectrogram_body does per tile, on
// counts that were summed over the ranks first.  decay = 0.999^(total batches of all ranks) (module_impl.cc:104).
__global__ __launch_bounds__(256) void spectrogram_apply_counts_kernel(float* __restrict__ bins,
                                                                       const uint32_t* __restrict__ counts,
                                                                       uint64_t cells, float decay) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (uint64_t)gridDim.x * 256) {
        const uint32_t k = counts[i];
        store_state(bins + i, apply_hits(bins[i] * decay, k < 64u ? k : 64u));  // 0.02 * 51 > 1: pinned at 1.0f long before 64 hits
    }
}


// The Spectrogram fed with ROW INDICES instead of values: `idx` is the tile-major U8[width / 128][batches][128] side output of the fused
// spectrum kernel (fft_lds.hh: StoreAmplitudeRangeSideT) -- per sample the index `(u32)(value * height)` the loop of
// spectrogram/module_impl_native_cpu.cc:70-77 would form, 0 where it does not hit.  One workgroup per tile of 16
// columns as above, but a thread takes whole ROWS of the tile: one 16-byte request per row instead of sixteen 4-byte
// ones (4 MiB instead of 16 MiB per 1024 x 4096 cycle), so a workgroup needs far fewer wavefronts to keep its reads in
// flight -- and the dispatch of the value kernel's 4096 wavefronts was ~40 % of its 7 us.  Lane l starts at column
// l % 16 and walks the tile cyclically: the 16 lanes of one LDS-atomic group always hold 16 different columns (no bank
// shared inside a group whatever the rows), and each of the four groups of a wavefront has a histogram copy of its own.
// width % 128 == 0 (the indices are tile-major in groups of 128 columns), height <= 256, batches * width < 2^31.
#ifdef JST_SPEC_TIMELINE  // tools/ubench/spec_index_timeline.hip: wall-clock stamps of workgroup phases (100 MHz)
unsigned long long* jst_spec_tl_host = nullptr;  // device buffer [workgroups][8], passed as a kernel argument
#define JST_SPEC_TL_PARAM , unsigned long long* __restrict__ jst_spec_tl
#define JST_SPEC_TL_ARG , jst_spec_tl_host
#define JST_SPEC_STAMP(slot) do { if (threadIdx.x == 0) jst_spec_tl[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define JST_SPEC_TL_PARAM
#define JST_SPEC_TL_ARG
#define JST_SPEC_STAMP(slot) do {} while (0)
#endif
template <int COPIES, int kThreads>
__global__ __launch_bounds__(kThreads) void spectrogram_index_kernel(float* __restrict__ bins, const uint8_t* __restrict__ idx,
                                                                     uint32_t batches, uint32_t pitch, uint32_t width,
                                                                     uint32_t height, float decay JST_SPEC_TL_PARAM) {
    constexpr uint32_t TW = 16;
    extern __shared__ __attribute__((aligned(64))) unsigned char smem_raw[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem_raw);  // [COPIES][height][TW] (+8 words between copies)
    const uint32_t tid = threadIdx.x;
    const uint32_t cells = height * TW;
    const uint32_t copy_stride = cells + 16u;  // 64-byte aligned copies (the column offset is OR-ed into the base), banks staggered
    JST_SPEC_STAMP(0);
    uint32_t tile = blockIdx.x;  // tiles sharing a 128-byte line of a row on one XCD (see spectrogram_body)
    if ((gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    constexpr uint32_t kCells = 4096 / kThreads;  // height <= 256: the whole state tile in registers
    float state[kCells];
#pragma unroll
    for (uint32_t j = 0; j < kCells; ++j) {
        const uint32_t e = tid + j * kThreads;
        state[j] = e < cells ? bins[(uint64_t)(e / TW) * width + tile * TW + (e % TW)] : 0.0f;
    }

    // rows tid, tid + kThreads, ...: kRows requests in flight per thread and round (a round covers 1024 rows)
    constexpr uint32_t kRows = 1024 / kThreads;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t r_idx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(idx), 0, pitch * width, 0x00020000);
    // tile-major indices (fft_lds.hh: StoreAmplitudeRangeSideT): U8[width / 128][batches][128], this tile's 16 bytes of row r at
#ifdef JST_SIDE_ROW_MAJOR  // A/B switch (fft_lds.hh)
    const uint32_t tile_base = tile * TW, row_bytes = width;
#else
    const uint32_t tile_base = (tile >> 3) * pitch * 128u + (tile & 7u) * TW, row_bytes = 128u;
#endif
    v4u q[kRows];
    auto request = [&](uint32_t first_row) {
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            const uint32_t row = first_row + tid + r * kThreads;  // a row that does not exist reads as 0: no hits
            q[r] = __builtin_amdgcn_raw_buffer_load_b128(r_idx, row < batches ? tile_base + row * row_bytes : 0xfffffff0u, 0, 0);
        }
    };
    request(0);
    JST_SPEC_STAMP(1);
    for (uint32_t e = tid * 4u; e < copy_stride * COPIES; e += kThreads * 4u)  // copy_stride % 4 == 0
        *reinterpret_cast<uint4*>(hist + e) = make_uint4(0u, 0u, 0u, 0u);
    lds_only_barrier();
    JST_SPEC_STAMP(2);

    const uint32_t rot = tid & 15u;  // this lane's first column
    // LDS byte address of (row i, column c) in this lane group's copy: base + 64 i + 4 c.  Index 0 (no hit) is NOT
    // predicated away: it counts into row 0, which no sample can hit (0 < index) and which the update below skips.
    // JST_SPAN_INTERLEAVED (two copies): U32[index][copy][16 columns] instead of two histograms one behind the other.  The 32
    // lanes of a half wavefront then hit 32 DIFFERENT banks whatever the indices are (lanes 0-15: copy 0, sixteen rotated
    // columns; lanes 16-31: copy 1): with separate histograms the bank is 16 x (index parity ^ copy) + column, and two lanes
    // of a half collide whenever their index parities differ the wrong way (every second pair).  Measured, same box, three
    // alternating bench.py runs each (tools/ubench/run_r05m.sh, profiles/r05_experiments/m_span_interleaved_copies.log): 31.5-31.8
    // us per 16-cycle span against 31.0-31.3 for the separate histograms -- the atomics are bound by the LDS atomic unit's
    // own rate (~12 clocks per wavefront instruction), not by bank conflicts.  Off.
#ifndef JST_SPAN_INTERLEAVED
#define JST_SPAN_INTERLEAVED 0
#endif
    constexpr bool kInterleaved = JST_SPAN_INTERLEAVED && COPIES == 2;
    const uint32_t my_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)hist +
                             (kInterleaved ? ((tid >> 4) & 1u) * 64u : ((tid >> 4) % COPIES) * copy_stride * 4u);
    typedef __attribute__((address_space(3))) uint32_t* lds_u32;
    for (uint32_t first = 0; first < batches; first += 1024u) {
        if (first != 0u) request(first);
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            // rotate the row's 16 bytes left by `rot`: whole dwords (two select stages), then bytes (v_alignbyte)
            const bool r1 = (rot & 4u) != 0u, r2 = (rot & 8u) != 0u;
            const uint32_t a0 = r1 ? q[r].y : q[r].x, a1 = r1 ? q[r].z : q[r].y, a2 = r1 ? q[r].w : q[r].z, a3 = r1 ? q[r].x : q[r].w;
            const uint32_t b0 = r2 ? a2 : a0, b1 = r2 ? a3 : a1, b2 = r2 ? a0 : a2, b3 = r2 ? a1 : a3;
            const uint32_t sh = rot & 3u;
            const uint32_t g[4] = {__builtin_amdgcn_alignbyte(b1, b0, sh), __builtin_amdgcn_alignbyte(b2, b1, sh),
                                   __builtin_amdgcn_alignbyte(b3, b2, sh), __builtin_amdgcn_alignbyte(b0, b3, sh)};
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) {
                const uint32_t i = (g[j >> 2] >> (8u * (j & 3u))) & 0xffu;  // the index at column (rot + j) % 16
                const uint32_t addr = (kInterleaved ? (i << 7) : (i << 6)) + ((((rot + j) & 15u) << 2) + my_base);
                __hip_atomic_fetch_add((lds_u32)(uintptr_t)addr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    JST_SPEC_STAMP(3);
    lds_only_barrier();
    JST_SPEC_STAMP(4);

    auto hits = [&](uint32_t e) {
        uint32_t k = 0;
#pragma unroll
        for (int cp = 0; cp < COPIES; ++cp) k += hist[cp * copy_stride + e];
        if (e < TW) k = 0u;        // row 0 collected the samples that do not hit
        return k < 64u ? k : 64u;  // 0.02 * 51 > 1: the value is pinned at 1.0f long before 64 hits
    };
    uint32_t k[kCells];
#pragma unroll
    for (uint32_t j = 0; j < kCells; ++j) k[j] = hits(tid + j * kThreads < cells ? tid + j * kThreads : 0u);
    JST_SPEC_STAMP(5);
#pragma unroll
    for (uint32_t j = 0; j < kCells; ++j) {
        const uint32_t e = tid + j * kThreads;
        if (e >= cells) continue;
        const float w = apply_hits(state[j] * decay, k[j]);
        store_state(bins + (uint64_t)(e / TW) * width + tile * TW + (e % TW), w);
    }
    JST_SPEC_STAMP(6);
}


// The index-fed Spectrogram over SEVERAL consecutive compute cycles in one launch (a cycle-batched runtime: the fused
// spectrum kernel of `cycles` ring slots ran as one launch and left `cycles` index tensors (tile-major, as above) one behind
// the other).  What spectrogram/module_impl_native_cpu.cc:61-87 does per cycle -- decay every bin by 0.999^batches, then
// min(v + 0.02f, 1.0f) once per hit -- happens here cycle after cycle on a state tile that STAYS IN REGISTERS: the state
// is read once and written once per launch instead of once per cycle, a launch (4096 wavefronts to dispatch, ~3 us) is
// paid once per span, and the next cycle's rows are in flight while this cycle's are counted.  Per cycle: the atomics,
// a barrier, every thread reads AND zeroes the counts of its own cells (no clear pass, no barrier between the two), a
// barrier.  Bit-identical to `cycles` launches of spectrogram_index_kernel by construction (same counts, same update).
#ifdef JST_SPAN_TIMELINE  // tools/ubench/spec_span_timeline.hip: shader-clock cycles per phase, summed per wavefront
unsigned long long* jst_span_tl_host = nullptr;  // device buffer [workgroups][16 waves][8 phases]
#define JST_SPAN_TL_PARAM , unsigned long long* __restrict__ jst_span_tl
#define JST_SPAN_TL_ARG , jst_span_tl_host
#define JST_SPAN_T0() unsigned long long jst_tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long jst_tl_t = __builtin_readcyclecounter()
#define JST_SPAN_MARK(ph) do { const unsigned long long n_ = __builtin_readcyclecounter(); jst_tl_acc[ph] += n_ - jst_tl_t; jst_tl_t = n_; } while (0)
#define JST_SPAN_DUMP() do { if ((threadIdx.x & 63u) == 0u) for (int p_ = 0; p_ < 8; ++p_) jst_span_tl[(blockIdx.x * 16u + (threadIdx.x >> 6)) * 8u + p_] = jst_tl_acc[p_]; } while (0)
#else
#define JST_SPAN_TL_PARAM
#define JST_SPAN_TL_ARG
#define JST_SPAN_T0() do {} while (0)
#define JST_SPAN_MARK(ph) do {} while (0)
#define JST_SPAN_DUMP() do {} while (0)
#endif
// PAIRED (round 5): two consecutive cycles count into ONE histogram pass -- cycle 2m adds 1, cycle 2m + 1 adds 65536 to the same
// word (a cell takes at most `batches` hits per cycle; the launcher pairs only below 65536 batches) -- then one barrier,
// one read-and-zero of the cells, one barrier, and the two decay + hit updates in order on the registers: the same counts
// per cycle and the same update order, with half the barriers and half the count reads per cycle.
template <int COPIES, int kThreads, bool PAIRED = false>
__global__ __launch_bounds__(kThreads) void spectrogram_index_span_kernel(float* __restrict__ bins, const uint8_t* __restrict__ idx,
                                                                          uint32_t batches, uint32_t pitch, uint32_t width,
                                                                          uint32_t height, float decay, uint32_t cycles,
                                                                          uint32_t first_slot, uint32_t ring_slots JST_SPAN_TL_PARAM) {
    constexpr uint32_t TW = 16;
    extern __shared__ __attribute__((aligned(64))) unsigned char smem_raw[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t cells = height * TW;
    const uint32_t copy_stride = cells + 16u;
    uint32_t tile = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    constexpr uint32_t kCells = 4096 / kThreads;
    float state[kCells];
#pragma unroll
    for (uint32_t j = 0; j < kCells; ++j) {
        const uint32_t e = tid + j * kThreads;
        state[j] = e < cells ? bins[(uint64_t)(e / TW) * width + tile * TW + (e % TW)] : 0.0f;
    }

    constexpr uint32_t kRows = 1024 / kThreads;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const uint32_t cycle_bytes = pitch * width;
    const __amdgpu_buffer_rsrc_t r_idx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(idx), 0, (ring_slots ? ring_slots : cycles) * cycle_bytes, 0x00020000);
    // cycle c of the span reads ring slot (first_slot + c) mod ring_slots (ring_slots = 0: tensor c behind idx)
    uint32_t req_slot = ring_slots ? first_slot : 0u;
    // A ROUND is 1024 rows of one cycle; rounds are numbered through the whole span.  Several register sets take the rounds
    // in turn (the loop below is unrolled over them): with ONE set refilled inside the loop hipcc keeps the rows being
    // counted in a copy made at the END of the previous iteration, i.e. behind `s_waitcnt vmcnt(0)` on the request
    // just issued -- the round trip to the index tensor was exposed in every cycle (4.2 us per cycle and workgroup).
#ifdef JST_SIDE_ROW_MAJOR
    const uint32_t tile_base = tile * TW, row_bytes = width;
#else
    const uint32_t tile_base = (tile >> 3) * pitch * 128u + (tile & 7u) * TW, row_bytes = 128u;  // tile-major indices, see spectrogram_index_kernel
#endif
    const uint32_t rounds_per_cycle = (batches + 1023u) >> 10;
    const uint32_t total_rounds = cycles * rounds_per_cycle;
    uint32_t req_cycle = 0, req_first = 0;  // the next round to request
    auto request = [&](v4u (&dst)[kRows]) {
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            const uint32_t row = req_first + tid + r * kThreads;  // a row (or a round) that does not exist reads as 0: no hits
            dst[r] = __builtin_amdgcn_raw_buffer_load_b128(
                r_idx, (row < batches && req_cycle < cycles) ? req_slot * cycle_bytes + tile_base + row * row_bytes : 0xfffffff0u, 0, 0);
        }
        req_first += 1024u;
        if (req_first >= batches) {
            req_first = 0u;
            ++req_cycle;
            ++req_slot;
            if (ring_slots && req_slot == ring_slots) req_slot = 0u;
        }
    };
    // Four register sets, requests three rounds ahead: behind the fused kernel of a batched span the index tensors come
    // from HBM (64 MiB written beside 256 MiB of values: little of it is left in the Infinity Cache), and one round of
    // counting (< 2 us) does not cover that round trip.
    v4u q0[kRows], q1[kRows], q2[kRows], q3[kRows];
    request(q0);
    request(q1);
    request(q2);
    for (uint32_t e = tid * 4u; e < copy_stride * COPIES; e += kThreads * 4u)
        *reinterpret_cast<uint4*>(hist + e) = make_uint4(0u, 0u, 0u, 0u);
    lds_only_barrier();

    const uint32_t rot = tid & 15u;
    // JST_SPAN_INTERLEAVED (two copies): U32[index][copy][16 columns] instead of two histograms one behind the other.  The 32
    // lanes of a half wavefront then hit 32 DIFFERENT banks whatever the indices are (lanes 0-15: copy 0, sixteen rotated
    // columns; lanes 16-31: copy 1): with separate histograms the bank is 16 x (index parity ^ copy) + column, and two lanes
    // of a half collide whenever their index parities differ the wrong way (every second pair).  Measured, same box, three
    // alternating bench.py runs each (tools/ubench/run_r05m.sh, profiles/r05_experiments/m_span_interleaved_copies.log): 31.5-31.8
    // us per 16-cycle span against 31.0-31.3 for the separate histograms -- the atomics are bound by the LDS atomic unit's
    // own rate (~12 clocks per wavefront instruction), not by bank conflicts.  Off.
#ifndef JST_SPAN_INTERLEAVED
#define JST_SPAN_INTERLEAVED 0
#endif
    constexpr bool kInterleaved = JST_SPAN_INTERLEAVED && COPIES == 2;
    const uint32_t my_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)hist +
                             (kInterleaved ? ((tid >> 4) & 1u) * 64u : ((tid >> 4) % COPIES) * copy_stride * 4u);
    typedef __attribute__((address_space(3))) uint32_t* lds_u32;
    uint32_t one = 1u;  // PAIRED: 1 for the first cycle of a pair, 65536 for the second
    bool second = false;
    auto count_rows = [&](const v4u (&cur)[kRows]) {
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            const bool r1 = (rot & 4u) != 0u, r2 = (rot & 8u) != 0u;
            const uint32_t a0 = r1 ? cur[r].y : cur[r].x, a1 = r1 ? cur[r].z : cur[r].y, a2 = r1 ? cur[r].w : cur[r].z,
                           a3 = r1 ? cur[r].x : cur[r].w;
            const uint32_t b0 = r2 ? a2 : a0, b1 = r2 ? a3 : a1, b2 = r2 ? a0 : a2, b3 = r2 ? a1 : a3;
            const uint32_t sh = rot & 3u;
            const uint32_t g[4] = {__builtin_amdgcn_alignbyte(b1, b0, sh), __builtin_amdgcn_alignbyte(b2, b1, sh),
                                   __builtin_amdgcn_alignbyte(b3, b2, sh), __builtin_amdgcn_alignbyte(b0, b3, sh)};
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) {
                const uint32_t i = (g[j >> 2] >> (8u * (j & 3u))) & 0xffu;
                const uint32_t addr = (kInterleaved ? (i << 7) : (i << 6)) + ((((rot + j) & 15u) << 2) + my_base);
#if defined(JST_SPAN_DIAG_NOATOMICS)  // timing diagnostics only
                if (addr == 0xffffffffu) hist[j] = i;
#else
                __hip_atomic_fetch_add((lds_u32)(uintptr_t)addr, PAIRED ? one : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
            }
        }
    };
    // End of a cycle: every thread reads AND zeroes the counts of its own cells (cells past the tile alias cell 0, whose
    // count -- row 0, the samples that do not hit -- nobody uses), then the cycle's decay and hit update on the registers.
    JST_SPAN_T0();
    auto end_cycle = [&]() {
        JST_SPAN_MARK(1);  // rows counted (incl. the wait for them)
        lds_only_barrier();
        JST_SPAN_MARK(2);  // barrier: every wavefront's atomics are in
        uint32_t k[kCells];
#pragma unroll
        for (uint32_t j = 0; j < kCells; ++j) {
            const uint32_t e = tid + j * kThreads < cells ? tid + j * kThreads : 0u;
            uint32_t n = 0;
#pragma unroll
            for (int cp = 0; cp < COPIES; ++cp) {
                const uint32_t at = kInterleaved ? (e >> 4) * 32u + (uint32_t)cp * 16u + (e & 15u) : cp * copy_stride + e;
                n += hist[at];
                hist[at] = 0u;
            }
            if (e < TW) n = 0u;
            k[j] = n;
        }
        JST_SPAN_MARK(3);  // counts read and zeroed
        lds_only_barrier();
        JST_SPAN_MARK(4);  // barrier
#if defined(JST_SPAN_DIAG_NOHITS)  // timing diagnostics only (tools/ubench/spec_span_timeline.hip)
#pragma unroll
        for (uint32_t j = 0; j < kCells; ++j) state[j] = state[j] * decay + (float)k[j];
#else
#pragma unroll
        for (uint32_t j = 0; j < kCells; ++j) {
            const uint32_t lo = PAIRED ? (k[j] & 0xffffu) : k[j];
            state[j] = apply_hits(state[j] * decay, lo < 64u ? lo : 64u);
        }
        if (PAIRED && second) {  // the pair's second cycle (the last cycle of an odd span stands alone)
#pragma unroll
            for (uint32_t j = 0; j < kCells; ++j) {
                const uint32_t hi = k[j] >> 16;
                state[j] = apply_hits(state[j] * decay, hi < 64u ? hi : 64u);
            }
        }
#endif
        JST_SPAN_MARK(5);  // decay + hits
    };
    uint32_t in_cycle = 0, cycles_done = 0;
    auto round_done = [&]() {
        if (++in_cycle == rounds_per_cycle) {
            in_cycle = 0;
            ++cycles_done;
            if (PAIRED && !second && cycles_done < cycles) {  // the next cycle counts into the high halves of the same words
                second = true;
                one = 65536u;
                return;
            }
            end_cycle();
            second = false;
            one = 1u;
        }
    };
    for (uint32_t round = 0; round < total_rounds; round += 4u) {
        // every request is issued whether or not its round exists (beyond the span it reads as zeros and is never
        // counted): conditional requests make hipcc's vmcnt bookkeeping conservative, i.e. the prefetch shallower
        request(q3);
        count_rows(q0);
        round_done();
        request(q0);
        if (round + 1u < total_rounds) {
            count_rows(q1);
            round_done();
        }
        request(q1);
        if (round + 2u < total_rounds) {
            count_rows(q2);
            round_done();
        }
        request(q2);
        if (round + 3u < total_rounds) {
            count_rows(q3);
            round_done();
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < kCells; ++j) {
        const uint32_t e = tid + j * kThreads;
        if (e >= cells) continue;
        store_state(bins + (uint64_t)(e / TW) * width + tile * TW + (e % TW), state[j]);
    }
    JST_SPAN_MARK(6);
    JST_SPAN_DUMP();
}


}  // namespace

namespace {
int spectrogram_copies(uint64_t height) { return height <= 256 ? 4 : (height <= 512 ? 2 : 1); }
}  // namespace

size_t spectrogram_lds_bytes(uint64_t height) {
    const int tw = height <= 1024 ? 16 : 8;
    return ((size_t)height * tw + 8) * spectrogram_copies(height) * sizeof(uint32_t);
}

hipError_t launch_spectrogram(float* bins, const float* in, uint64_t in_offset, uint64_t batches,
                              uint64_t width, uint64_t height, int64_t batch_stride,
                              int64_t elem_stride, float decay, hipStream_t stream) {
    if (width == 0 || height == 0) return hipSuccess;
    if (height > 2048 || batches > 0xffffffffull || width > 0xffffffffull)
        return hipErrorInvalidValue;
    const size_t lds = spectrogram_lds_bytes(height);
    const unsigned tiles16 = (unsigned)((width + 15) / 16), tiles8 = (unsigned)((width + 7) / 8);
    (void)hipGetLastError();  // drop any stale error: only this launch is judged
#define JST_SPEC_LAUNCH_B(TW, COPIES, THREADS, DEPTH, TILES, BUFV)                                  \
    do {                                                                                          \
        { /* the padded copies can exceed the 64 KiB default by a few words */                    \
            const hipError_t e = raise_dynamic_lds(                                               \
                reinterpret_cast<const void*>(spectrogram_kernel<TW, COPIES, THREADS, DEPTH, BUFV>), 80 * 1024); \
            if (e != hipSuccess) return e;                                                        \
        }                                                                                         \
        hipLaunchKernelGGL((spectrogram_kernel<TW, COPIES, THREADS, DEPTH, BUFV>), dim3(TILES),   \
                           dim3(THREADS), lds, stream, bins, in, in_offset, (uint32_t)batches,    \
                           (uint32_t)width, (uint32_t)height, batch_stride, elem_stride, decay);  \
    } while (0)
    // dense-enough input for the one-descriptor form: non-negative strides, rows that do not interleave, < 2 GiB
    const int64_t span = (int64_t)(batches ? batches - 1 : 0) * batch_stride + (int64_t)(width - 1) * elem_stride + 1;
    const bool buf_ok = elem_stride >= 0 && batch_stride >= (int64_t)(width - 1) * elem_stride + 1 && batches > 0 &&
                        span * 4 < (int64_t)0x7fff0000;
#define JST_SPEC_LAUNCH(TW, COPIES, THREADS, DEPTH, TILES)                          \
    do {                                                                            \
        if (buf_ok) JST_SPEC_LAUNCH_B(TW, COPIES, THREADS, DEPTH, TILES, true);     \
        else JST_SPEC_LAUNCH_B(TW, COPIES, THREADS, DEPTH, TILES, false);           \
    } while (0)
    if (height <= 256) {
        // 1024 threads: 512- and 256-thread workgroups (fewer wavefronts to dispatch) were +1.2 / +2.9 us per step (round 2)
        JST_SPEC_LAUNCH(16, 4, 1024, 16, tiles16);
    }
    else if (height <= 512) JST_SPEC_LAUNCH(16, 2, 1024, 16, tiles16);
    else if (height <= 1024) JST_SPEC_LAUNCH(16, 1, 1024, 16, tiles16);
    else JST_SPEC_LAUNCH(8, 1, 1024, 16, tiles8);
#undef JST_SPEC_LAUNCH
#undef JST_SPEC_LAUNCH_B
    return hipGetLastError();
}

bool spectrogram_index_supported(uint64_t batches, uint64_t width, uint64_t height) {
    // < 2^31 bytes of indices: a row that does not exist is requested at byte offset 0xfffffff0, which must lie beyond the
    // descriptor's range for every accepted shape
    return width > 0 && width % 128 == 0 && height >= 2 && height <= 256 && batches > 0 && batches * width < (1ull << 31);
}

hipError_t launch_spectrogram_index(float* bins, const uint8_t* idx, uint64_t batches, uint64_t pitch, uint64_t width,
                                    uint64_t height, float decay, hipStream_t stream) {
    if (!spectrogram_index_supported(batches, width, height) || pitch < batches || pitch * width >= (1ull << 31))
        return hipErrorInvalidValue;
    const size_t lds = ((size_t)height * 16 + 16) * 4 * sizeof(uint32_t);  // four copies, 64-byte aligned
    (void)hipGetLastError();
#define JST_SPEC_INDEX(THREADS)                                                                                      \
    do {                                                                                                             \
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(spectrogram_index_kernel<4, THREADS>),  \
                                               80 * 1024);                                                           \
        if (e != hipSuccess) return e;                                                                               \
        hipLaunchKernelGGL((spectrogram_index_kernel<4, THREADS>), dim3((unsigned)(width / 16)), dim3(THREADS), lds, \
                           stream, bins, idx, (uint32_t)batches, (uint32_t)pitch, (uint32_t)width, (uint32_t)height, \
                           decay JST_SPEC_TL_ARG);                                                                   \
    } while (0)
    JST_SPEC_INDEX(1024);  // 512- / 256-thread workgroups: 6.3 / 8.0 us against 4.9 (profiles/r03_experiments/l_...)
#undef JST_SPEC_INDEX
    return hipGetLastError();
}

// `cycles` consecutive index tensors behind `idx` (a cycle-batched span), one launch.
bool spectrogram_index_span_supported(uint64_t batches, uint64_t width, uint64_t height, uint64_t cycles) {
    return cycles >= 1 && spectrogram_index_supported(batches, width, height) && cycles * batches * width < (1ull << 31);
}

hipError_t launch_spectrogram_index_span(float* bins, const uint8_t* idx, uint64_t batches, uint64_t pitch, uint64_t width,
                                         uint64_t height, float decay, uint64_t cycles, uint64_t first_slot,
                                         uint64_t ring_slots, hipStream_t stream) {
    if (cycles < 1 || cycles >= (1ull << 31) || !spectrogram_index_supported(batches, width, height) || pitch < batches ||
        (ring_slots ? ring_slots : cycles) * pitch * width >= (1ull << 31) || (ring_slots && first_slot >= ring_slots))
        return hipErrorInvalidValue;
    // TWO private histogram copies per workgroup.  The single-cycle kernel
    // is launch bound and does not care (5.08 / 5.10 / 5.18 us); here the counting is what is left, and every copy is
    // 16 KiB more to read back and zero per cycle: 36.4 us per 16-cycle span with four, 32.2 with two, 32.8 with one (more
    // same-address collisions among the four rows of an atomic instruction) -- profiles/r03_experiments/w_span_kernel_diagnosis.log.
    // Round 4 tried the collision-free form (U32[index][4 copies][16 columns], two histograms used alternately: one barrier
    // per cycle): 36.4 us against 31.6 us for this kernel, same box -- what a cycle costs is reading back and zeroing the
    // copies and the hit update, not the atomics' collisions (tools/ubench/spectrogram_span2_experiment.hh,
    // profiles/r04_experiments/d_span_kernel_v2.log).
    (void)hipGetLastError();
    // Round 5: cycles counted in PAIRS (below 65536 batches, spans of more than one cycle); the overlapped two-histogram form
    // and other workgroup sizes were measured and rejected (profiles/r05_experiments/r_..., w_...).  Round 6 tried the overlap
    // once more -- two histograms, one barrier per pair, the next pair's atomics leaving in eight groups of four between the
    // eight hit updates, the counts forced into registers first (lgkmcnt counts to 15 and retires in order) -- bit-exact and
    // 37.3 us per 20-cycle span against 33.7: the wavefronts without hot cells hand the LDS unit their 32 atomics at once,
    // and a hot wavefront's first group queues behind all of them (profiles/r06_experiments/b_span_overlap.log).
    constexpr int copies = 2;
    const size_t lds = ((size_t)height * 16 + 16) * (size_t)copies * sizeof(uint32_t);
#define JST_SPEC_SPAN(COPIES, PAIRED, THREADS)                                                                                       \
    do {                                                                                                                         \
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(spectrogram_index_span_kernel<COPIES, THREADS, PAIRED>), \
                                               80 * 1024);                                                                       \
        if (e != hipSuccess) return e;                                                                                           \
        hipLaunchKernelGGL((spectrogram_index_span_kernel<COPIES, THREADS, PAIRED>), dim3((unsigned)(width / 16)), dim3(THREADS), lds, \
                           stream, bins, idx, (uint32_t)batches, (uint32_t)pitch, (uint32_t)width, (uint32_t)height,             \
                           decay, (uint32_t)cycles, (uint32_t)first_slot, (uint32_t)ring_slots JST_SPAN_TL_ARG);                \
    } while (0)
    if (batches < 65536 && cycles > 1) JST_SPEC_SPAN(2, true, 1024);
    else JST_SPEC_SPAN(2, false, 1024);
#undef JST_SPEC_SPAN
    return hipGetLastError();
}

hipError_t launch_spectrogram_counts(uint32_t* counts, const float* in, uint64_t in_offset, uint64_t batches,
                                     uint64_t width, uint64_t height, int64_t batch_stride, int64_t elem_stride,
                                     hipStream_t stream) {
    if (width == 0 || height == 0) return hipSuccess;
    if (height > 2048 || batches > 0xffffffffull || width > 0xffffffffull) return hipErrorInvalidValue;
    const size_t lds = spectrogram_lds_bytes(height);
    const unsigned tiles16 = (unsigned)((width + 15) / 16), tiles8 = (unsigned)((width + 7) / 8);
    (void)hipGetLastError();
#define JST_SPEC_COUNTS(TW, COPIES, TILES)                                                                          \
    do {                                                                                                            \
        const hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(spectrogram_counts_kernel<TW, COPIES>), \
                                               80 * 1024);                                                          \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL((spectrogram_counts_kernel<TW, COPIES>), dim3(TILES), dim3(kThreadsDefault), lds, stream, \
                           counts, in, in_offset, (uint32_t)batches, (uint32_t)width, (uint32_t)height,             \
                           batch_stride, elem_stride);                                                              \
    } while (0)
    if (height <= 256) JST_SPEC_COUNTS(16, 4, tiles16);
    else if (height <= 512) JST_SPEC_COUNTS(16, 2, tiles16);
    else if (height <= 1024) JST_SPEC_COUNTS(16, 1, tiles16);
    else JST_SPEC_COUNTS(8, 1, tiles8);
#undef JST_SPEC_COUNTS
    return hipGetLastError();
}

hipError_t launch_spectrogram_apply_counts(float* bins, const uint32_t* counts, uint64_t cells, float decay,
                                           hipStream_t stream) {
    if (cells == 0) return hipSuccess;
    (void)hipGetLastError();
    const unsigned grid = (unsigned)((cells + 255) / 256 < 4096 ? (cells + 255) / 256 : 4096);
    hipLaunchKernelGGL(spectrogram_apply_counts_kernel, dim3(grid), dim3(256), 0, stream, bins, counts, cells, decay);
    return hipGetLastError();
}

}  // namespace jst::kernels
