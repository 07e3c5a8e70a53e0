// spectrogram_body.hh -- the Spectrogram kernel's body as a device function of (workgroup index, workgroup count), shared
// by the stand-alone kernel (spectrogram.hip) and by the launch that carries the previous cycle's spectrogram beside
// the spectrum transforms (fft_kernels.hip).  See spectrogram.hip for the algorithm and its reference citations.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "hit_update.hh"

namespace jst::kernels::specdev {

// State stores written through at agent scope (`global_store ... sc1`), like the spectrum kernel's output (fft_lds.hh,
// JST_STORE_AUX): nothing dirty is left for the end-of-kernel release.
#ifndef JST_PLAIN_STORES  // A/B switch
__device__ __forceinline__ void store_state(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
__device__ __forceinline__ void store_state(float* p, float v) { *p = v; }
#endif

constexpr int kThreadsDefault = 1024;

// The saturating hit update applied k times (kernels/hit_update.hh, host-testable): the k dependent additions and one
// clamp.  JST_HITS_BINADE selects the binade form (the same floats in at most a dozen integer steps on the bit pattern,
// proven equal on the host by tests/test_hit_update.py): measured 33.3-33.9 us against 31.8 us per 16-cycle span for the
// additions (profiles/r04_experiments/e_hit_update_binade.log) -- its per-step conversion, multiply and two corrections cost
// more than the ~30 dependent additions of a noise-floor cell they replace -- so it is not the default.
#ifdef JST_HITS_BINADE
__device__ __forceinline__ float apply_hits(float w, uint32_t k) { return jst::dev::apply_hits_binade(w, k); }
#else
__device__ __forceinline__ float apply_hits(float w, uint32_t k) { return jst::dev::apply_hits(w, k); }
#endif

// Workgroup barrier that orders LDS traffic only: __syncthreads() is a full fence and puts s_waitcnt vmcnt(0)
// in front of s_barrier, which would drain the input loads in flight across the histogram clear.
__device__ __forceinline__ void lds_only_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// COPIES: private copies of the tile histogram, one per group of 16 lanes of a wavefront.  The 64
// lanes of one LDS-atomic instruction cover 4 consecutive batches x 16 columns, and a spectrum's
// neighbouring batches tend to land in the SAME bin of a column (noise floor): with one copy
// that is a 4-way same-address collision on nearly every instruction.  Copies are offset by 8
// words so the four lanes of a column fall on different banks.
// BUF: the input rows are addressed through ONE buffer descriptor (SGPRs), one 32-bit per-lane offset shared by all
// of a thread's loads and a wave-uniform row offset per load.  With flat 64-bit addresses the 16 loads in flight need
// 32 address VGPRs on top of their 16 destinations -- more than a 1024-thread workgroup's 64-register budget leaves, so
// hipcc recycled destination registers as addresses and drained vmcnt in front of the last four loads.  Rows at or
// beyond `batches` and the columns of a ragged last tile fall outside the descriptor's range and read as 0 (which never
// hits).  Used when the tensor spans < 2 GiB and its rows do not interleave; the flat form stays for everything else.
// The body as a device function of (workgroup index, workgroup count), see fft_pipe_body.
// COUNTS: the hit counts of this cycle are the RESULT -- `bins` then points at a U32[height][width] tensor that receives
// them (unclamped) and no state is read or updated: the device half of the exact multi-GPU merge (all-reduce the
// integer counts over the ranks, then spectrogram_apply_counts_kernel: one shared decay and the count-times update).
template <int TW, int COPIES, int kThreads = kThreadsDefault, int kDepthT = 16, bool BUF = false, bool COUNTS = false>
__device__ __forceinline__ void spectrogram_body(
    float* __restrict__ bins, const float* __restrict__ in, uint64_t in_offset, uint32_t batches,
    uint32_t width, uint32_t height, int64_t batch_stride, int64_t elem_stride, float decay,
    const uint32_t bid, const uint32_t grid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem_raw);  // [height][TW]

    const uint32_t tid = threadIdx.x;
    const uint32_t cells = height * TW;
    const uint32_t copy_stride = cells + 8u;

    // Tile order: workgroup b runs on XCD b % 8 (observed placement, used for speed only) and every XCD has an L2
    // of its own.  Tiles t and t+1 share each 128-byte line of a row (TW * 4 = 64 bytes per tile), so they go to
    // the SAME XCD: XCD k takes the contiguous run of tiles [k * tiles/8, (k+1) * tiles/8) and pulls every line
    // of its column band from the Infinity Cache once instead of twice.
    uint32_t tile = bid;
    if ((grid & 7u) == 0u) tile = (bid & 7u) * (grid >> 3) + (bid >> 3);

    // Nothing below depends on the histogram until the LDS atomics: the state tile (not touched by this cycle's
    // hits) and the first kDepth input rows per thread are requested BEFORE the histogram is cleared, and the
    // barriers order LDS traffic only (no s_waitcnt vmcnt(0)), so the L2 / Infinity-Cache round trips overlap the
    // clear and each other.
    constexpr uint32_t kCells = 4 * (kThreadsDefault / kThreads);  // 4096 cells (height 256 x 16 columns) in registers
    float state[kCells];
    float* cell[kCells];
#pragma unroll
    for (uint32_t j = 0; j < kCells; ++j) {
        const uint32_t e = tid + j * kThreads;
        const uint32_t xx = tile * TW + (e % TW);
        cell[j] = (e < cells && xx < width) ? bins + (uint64_t)(e / TW) * width + xx : nullptr;
        if constexpr (!COUNTS) state[j] = cell[j] ? *cell[j] : 0.0f;
    }

    const uint32_t c = tid % TW;
    const uint32_t x = tile * TW + c;
    const float fh = (float)height;
    uint32_t* my_hist = hist + ((tid / TW) % COPIES) * copy_stride;
    constexpr uint32_t rows_per_iter = kThreads / TW;
    constexpr uint32_t kDepth = kDepthT;  // loads in flight per thread: the reads are latency bound
    const float* col = in + in_offset + (int64_t)(x < width ? x : 0) * elem_stride;
    float v[kDepth];
    const uint32_t extent = BUF ? (uint32_t)(((int64_t)(batches - 1) * batch_stride + (int64_t)(width - 1) * elem_stride + 1) * 4) : 0u;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t r_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + in_offset), 0, extent, 0x00020000);
    [[maybe_unused]] uint32_t voff = 0, row_step = 0;
    if constexpr (BUF) {
        voff = x < width ? (uint32_t)(((int64_t)x * elem_stride + (int64_t)(tid / TW) * batch_stride) * 4) : 0x7ffffff0u;
        row_step = (uint32_t)((int64_t)rows_per_iter * batch_stride * 4);
        // Rows past the last batch must read as 0 (0 never hits).  The descriptor's range check covers the per-lane
        // voffset only -- LLVM documents soffset as excluded from it -- so the tail is NOT left to the wave-uniform
        // row offset: a lane whose row does not exist gets an out-of-range voffset (one compare on a row index).
#pragma unroll
        for (uint32_t j = 0; j < kDepth; ++j) {
            const bool row_ok = tid / TW + j * rows_per_iter < batches;
            v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_in, row_ok ? voff : 0x7ffffff0u, row_ok ? j * row_step : 0u, 0));
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < kDepth; ++j) {
            const uint32_t b = tid / TW + j * rows_per_iter;
            v[j] = (x < width && b < batches) ? col[(int64_t)b * batch_stride] : 0.0f;  // 0 never hits
        }
    }
    for (uint32_t e = tid * 4u; e < copy_stride * COPIES; e += kThreads * 4u) {  // copy_stride % 4 == 0
        *reinterpret_cast<uint4*>(hist + e) = make_uint4(0u, 0u, 0u, 0u);
    }
    lds_only_barrier();

    if (x < width) {
        for (uint32_t b0 = tid / TW; b0 < batches; b0 += rows_per_iter * kDepth) {
            if (b0 != tid / TW) {  // batches > rows_per_iter * kDepth: later rounds load here
                if constexpr (BUF) {
                    voff += kDepth * row_step;
#pragma unroll
                    for (uint32_t j = 0; j < kDepth; ++j) {
                        const bool row_ok = b0 + j * rows_per_iter < batches;
                        v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_in, row_ok ? voff : 0x7ffffff0u, row_ok ? j * row_step : 0u, 0));
                    }
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < kDepth; ++j) {
                        const uint32_t b = b0 + j * rows_per_iter;
                        v[j] = (b < batches) ? col[(int64_t)b * batch_stride] : 0.0f;
                    }
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < kDepth; ++j) {
                const float f = v[j] * fh;
                if (f >= 1.0f && f < fh) atomicAdd(&my_hist[(uint32_t)f * TW + c], 1u);
            }
        }
    }
    lds_only_barrier();

    auto hits = [&](uint32_t e) {
        uint32_t k = 0;
#pragma unroll
        for (int cp = 0; cp < COPIES; ++cp) k += hist[cp * copy_stride + e];
        if constexpr (COUNTS) return k;
        return k < 64u ? k : 64u;  // 0.02 * 51 > 1: the value is pinned at 1.0f long before 64 hits
    };
    if constexpr (COUNTS) {
        for (uint32_t e = tid; e < cells; e += kThreads) {
            const uint32_t xx = tile * TW + (e % TW);
            if (xx < width) reinterpret_cast<uint32_t*>(bins)[(uint64_t)(e / TW) * width + xx] = hits(e);
        }
        return;
    }
    auto apply = [&](float w, uint32_t k) { return apply_hits(w * decay, k); };
    uint32_t k[kCells];  // every count is read before the first (divergent, serial) update loop starts
#pragma unroll
    for (uint32_t j = 0; j < kCells; ++j) k[j] = hits(tid + j * kThreads < cells ? tid + j * kThreads : 0u);
#pragma unroll
    for (uint32_t j = 0; j < kCells; ++j)
        if (cell[j]) store_state(cell[j], apply(state[j], k[j]));
    for (uint32_t e = tid + kCells * kThreads; e < cells; e += kThreads) {  // height > 256
        const uint32_t xx = tile * TW + (e % TW);
        if (xx >= width) continue;
        float* p = bins + (uint64_t)(e / TW) * width + xx;
        store_state(p, apply(*p, hits(e)));
    }
}

}  // namespace jst::kernels::specdev
