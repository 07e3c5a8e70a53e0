// spectrogram_span2_experiment.hh -- REJECTED round-4 variant of spectrogram_index_span_kernel (kernels/spectrogram.hip), kept
// out of the product: collision-free LDS atomics (histogram U32[index][4 copies][16 columns]) and two histograms used
// alternately (one workgroup barrier per cycle).  Bit-identical state; 36.4 us per 16-cycle span against 31.6 us for the
// shipped kernel on the bench's own indices (profiles/r04_experiments/d_span_kernel_v2.log).  To rebuild the experiment:
// paste into the anonymous namespace of spectrogram.hip and launch with 2 * height * 64 * 4 bytes of dynamic LDS.

// Round 4 form of the span kernel: COLLISION-FREE atomics and ONE barrier per cycle.
//   * Histogram layout U32[index][4 copies][16 columns] (64 words = 256 B per index row): the four 16-lane groups of a
//     wavefront -- four rows of the tile, whose lanes with equal column tend to carry the SAME index (neighbouring batches
//     of a stationary spectrum) -- each count into their own copy, so the 64 addresses of one ds_add_u32 are distinct
//     (round 3: two copies, ~2x the conflict-free time on real indices), and lane l's bank is 16 (copy & 1) + column:
//     both 32-lane halves conflict-free.  A cell's four partial counts sit 64 B apart: two ds_read2_b32 fetch them, two
//     ds_write2_b32 zero them -- the same number of LDS instructions per cell as the two-copy form needed.
//   * TWO histograms used alternately by consecutive cycles: cycle c + 1 counts into the other one while the owners of
//     cycle c's cells still read, zero and apply, so the barrier between "counts read" and "next cycle's atomics" is gone;
//     the zeroes of cycle c are ordered before cycle c + 2's atomics by cycle c + 1's barrier.
// 2 x 64 KiB of LDS at height 256 (one 1024-thread workgroup per CU either way).  Same counts, same update order:
// bit-identical state.
template <int kThreads>
__global__ __launch_bounds__(kThreads) void spectrogram_index_span2_kernel(float* __restrict__ bins, const uint8_t* __restrict__ idx,
                                                                           uint32_t batches, uint32_t pitch, uint32_t width,
                                                                           uint32_t height, float decay, uint32_t cycles,
                                                                           uint32_t first_slot, uint32_t ring_slots) {
    constexpr uint32_t TW = 16;
    extern __shared__ __attribute__((aligned(64))) unsigned char smem_raw[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t cells = height * TW;
    const uint32_t hist_words = height * 64u;  // one histogram
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
    uint32_t req_slot = ring_slots ? first_slot : 0u;
    const uint32_t tile_base = (tile >> 3) * pitch * 128u + (tile & 7u) * TW, row_bytes = 128u;  // tile-major indices
    const uint32_t rounds_per_cycle = (batches + 1023u) >> 10;
    const uint32_t total_rounds = cycles * rounds_per_cycle;
    uint32_t req_cycle = 0, req_first = 0;
    auto request = [&](v4u (&dst)[kRows]) {
#pragma unroll
        for (uint32_t r = 0; r < kRows; ++r) {
            const uint32_t row = req_first + tid + r * kThreads;
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
    v4u q0[kRows], q1[kRows], q2[kRows], q3[kRows];
    request(q0);
    request(q1);
    request(q2);
    for (uint32_t e = tid * 4u; e < 2u * hist_words; e += kThreads * 4u)
        *reinterpret_cast<uint4*>(hist + e) = make_uint4(0u, 0u, 0u, 0u);
    lds_only_barrier();

    const uint32_t rot = tid & 15u;
    typedef __attribute__((address_space(3))) uint32_t* lds_u32;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)hist;
    const uint32_t group_off = ((tid >> 4) & 3u) * 64u;  // bytes: this 16-lane group's copy inside an index row
    uint32_t cur_buf = 0;                                  // bytes offset of the histogram the current cycle counts into
    auto count_rows = [&](const v4u (&cur)[kRows]) {
        const uint32_t my_base = lds_base + cur_buf + group_off;
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
                const uint32_t addr = (i << 8) + ((((rot + j) & 15u) << 2) + my_base);
                __hip_atomic_fetch_add((lds_u32)(uintptr_t)addr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    auto end_cycle = [&]() {
        lds_only_barrier();  // every wavefront's atomics of this cycle are in
        uint32_t k[kCells];
        uint32_t* h = hist + (cur_buf >> 2);
#pragma unroll
        for (uint32_t j = 0; j < kCells; ++j) {
            const uint32_t e = tid + j * kThreads < cells ? tid + j * kThreads : 0u;
            uint32_t* p = h + ((e >> 4) << 6) + (e & 15u);   // copy 0 of cell e; copies 16 words apart
            uint32_t n = p[0] + p[16] + p[32] + p[48];
            p[0] = 0u;
            p[16] = 0u;
            p[32] = 0u;
            p[48] = 0u;
            if (e < TW) n = 0u;   // index 0 collected the samples that do not hit
            k[j] = n < 64u ? n : 64u;
        }
        cur_buf ^= hist_words * 4u;  // the next cycle counts into the other histogram: no barrier in front of it
#pragma unroll
        for (uint32_t j = 0; j < kCells; ++j) state[j] = apply_hits(state[j] * decay, k[j]);
    };
    uint32_t in_cycle = 0;
    auto round_done = [&]() {
        if (++in_cycle == rounds_per_cycle) {
            in_cycle = 0;
            end_cycle();
        }
    };
    for (uint32_t round = 0; round < total_rounds; round += 4u) {
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
}

