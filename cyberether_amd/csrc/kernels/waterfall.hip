// waterfall.hip -- the Waterfall module's compute: copy the newest min(B, H) rows of an F32[B,N]
// spectrum batch into a ring of H rows (src/domains/visualization/waterfall/ring_state.hh:16-56,
// module_impl_native_cpu.cc:53-78; CUDA analogue module_impl_native_cuda.cc:18-42,108-149).
//
// Unlike the reference's CUDA path (cursor on the host, passed as a kernel argument every cycle)
// the ring cursor lives in device memory, so one captured launch is replayable from a hipGraph:
// every workgroup reads the cursor at entry, copies, then takes a ticket; the LAST workgroup to
// arrive -- by which time every workgroup has read the cursor -- advances it.  All index work is
// 64-bit integer and identical to PlanWaterfallWrite / WaterfallRingState::advance.
#include "device_math.hh"
#include "kernels.hh"

namespace jst::kernels {

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void waterfall_kernel(
    float* __restrict__ ring, uint64_t* state, const float* __restrict__ in, uint64_t in_offset,
    uint64_t batches, uint64_t width, uint64_t height, int64_t batch_stride, int64_t elem_stride) {
    const uint64_t write_index =
        __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // PlanWaterfallWrite (ring_state.hh:16-28)
    const uint64_t retained = batches < height ? batches : height;
    const uint64_t source_row = batches - retained;
    const uint64_t destination_row = (write_index + (source_row % height)) % height;

    const uint64_t total = retained * width;
    const uint64_t step = (uint64_t)gridDim.x * kThreads;
    for (uint64_t e = (uint64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += step) {
        const uint64_t row = e / width, column = e - row * width;
        const uint64_t dst = (destination_row + row) % height;
        ring[dst * width + column] =
            in[in_offset + (int64_t)(source_row + row) * batch_stride + (int64_t)column * elem_stride];
    }

    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t ticket =
            __hip_atomic_fetch_add(&state[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == (uint64_t)gridDim.x - 1) {
            // WaterfallRingState::advance (ring_state.hh:40-43)
            const uint64_t dirty =
                __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t room = height - dirty;
            __hip_atomic_store(&state[0], (write_index + (batches % height)) % height,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[1], dirty + (batches < room ? batches : room),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[2], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

hipError_t launch_waterfall(float* ring, uint64_t* state, const float* in, uint64_t in_offset,
                            uint64_t batches, uint64_t width, uint64_t height,
                            int64_t batch_stride, int64_t elem_stride, hipStream_t stream) {
    if (height == 0) return hipErrorInvalidValue;
    const uint64_t retained = batches < height ? batches : height;
    const uint64_t total = retained * width;
    uint64_t blocks = (total + kThreads - 1) / kThreads;
    if (blocks == 0) blocks = 1;  // the cursor still advances on an empty copy
    if (blocks > 4096) blocks = 4096;
    (void)hipGetLastError();  // drop any stale error: only this launch is judged
    hipLaunchKernelGGL(waterfall_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, ring,
                       state, in, in_offset, batches, width, height, batch_stride, elem_stride);
    return hipGetLastError();
}

// ---- Lineplot compute (visualization/lineplot/module_impl_native_cpu.cc:80-118; CUDA analogue
// module_impl_native_cuda.cc:21-49): per-bin batch sum (left to right from +0), normalisation to
// [-1, 1], then the moving average  avg -= avg/averaging; avg += amplitude/averaging.
// The averaged trace is the reference's only "averaged spectrum": it is what the optional
// cross-GPU all-reduce of BASELINE config 5 averages (cyberether_amd/distributed.py).
namespace {
// DEPTH loads in flight per thread.  The sum is ordered (left to right over the batch axis, F32, from +0: the reference's loop),
// the loads are not: a thread requests DEPTH rows, then adds them in order.  16 covers config 5's one stream (16 batches: one
// round trip); with 128 rows -- the 8-stream form -- 16 at a time are eight serial round trips of 1024 wavefronts, 4 MB in
// flight on the whole chip: 8.9 us for 33.5 MB (3.7 TB/s); 64 at a time: round 6.
template <int DEPTH>
__global__ __launch_bounds__(256) void lineplot_kernel(float* __restrict__ points,
                                                       float* __restrict__ average,
                                                       const float* __restrict__ in,
                                                       uint64_t in_offset, uint64_t batches,
                                                       uint64_t elements, int64_t batch_stride,
                                                       int64_t elem_stride, uint64_t decimation,
                                                       float normalization, float averaging) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elements) return;
    // Rows past the last batch read a clamped address and add +0.0f, which leaves a sum that started at +0.0f unchanged (it
    // can never be -0.0f).
    float sum = 0.0f;
    const float* col = in + in_offset + (int64_t)(i * decimation) * elem_stride;
    for (uint64_t b0 = 0; b0 < batches; b0 += DEPTH) {
        float v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const uint64_t b = b0 + (uint64_t)k;
            const float x = col[(int64_t)(b < batches ? b : batches - 1) * batch_stride];
            v[k] = b < batches ? x : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) sum += v[k];
    }
    const float amplitude = fminf(fmaxf((sum * normalization) - 1.0f, -1.0f), 1.0f);
    float avg = average[i];
    avg -= avg / averaging;
    avg += amplitude / averaging;
    average[i] = avg;
    points[(i * 2) + 1] = avg;
}
// The same over `cycles` consecutive compute cycles of a cycle-batched span in ONE launch: cycle c reads slot
// (first_slot + c) mod ring_slots of the input ring (slots slot_stride elements apart), the moving average stays in a
// register from the first cycle to the last -- the recursion is per bin, so the cycles of a bin are one thread's loop.
// DEPTH rows of CYC consecutive cycles are requested together (round 6): the SUMS of different cycles do not depend on each
// other, only the average's recursion does, so a span of config 5's one stream (16 rows per cycle) need not be one round trip
// per cycle (17.8 us per 16-cycle span with <16, 1>), and 128 rows (8 streams) need not be eight per cycle.  Rows and cycles
// past the end read a clamped address and are not added.  (A single stream of requests across cycle boundaries with a
// running cursor was tried first -- per-thread 64-bit pointers, then a scalar cursor with uniform branches: 14-32 instructions
// per load at one wavefront per SIMD, 23-54 us against 17.8-31.6: profiles/r06_experiments/d_lineplot_depth.log.)
template <int DEPTH, int CYC>
__global__ __launch_bounds__(256) void lineplot_span_kernel(float* __restrict__ points, float* __restrict__ average,
                                                            const float* __restrict__ in, uint64_t in_offset,
                                                            uint64_t slot_stride, uint32_t first_slot, uint32_t ring_slots,
                                                            uint32_t cycles, uint64_t batches, uint64_t elements,
                                                            int64_t batch_stride, int64_t elem_stride, uint64_t decimation,
                                                            float normalization, float averaging) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elements) return;
    float avg = average[i];
    uint32_t slot = first_slot;
    const float* lane = in + in_offset + (int64_t)(i * decimation) * elem_stride;
    for (uint32_t c = 0; c < cycles; c += CYC) {
        const float* col[CYC];
        {
            uint32_t sl = slot;
#pragma unroll
            for (int cc = 0; cc < CYC; ++cc) {  // a cycle past the span: the last one's rows again, not added
                col[cc] = lane + (uint64_t)sl * slot_stride;
                if (c + (uint32_t)cc + 1u < cycles && ++sl == ring_slots) sl = 0;
            }
        }
        float sum[CYC];
#pragma unroll
        for (int cc = 0; cc < CYC; ++cc) sum[cc] = 0.0f;
        for (uint64_t b0 = 0; b0 < batches; b0 += DEPTH) {
            float v[CYC][DEPTH];
#pragma unroll
            for (int cc = 0; cc < CYC; ++cc)
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) {
                    const uint64_t b = b0 + (uint64_t)k;
                    const float x = col[cc][(int64_t)(b < batches ? b : batches - 1) * batch_stride];
                    v[cc][k] = b < batches ? x : 0.0f;
                }
#pragma unroll
            for (int cc = 0; cc < CYC; ++cc)
#pragma unroll
                for (int k = 0; k < DEPTH; ++k) sum[cc] += v[cc][k];
        }
#pragma unroll
        for (int cc = 0; cc < CYC; ++cc) {
            if (c + (uint32_t)cc < cycles) {
                const float amplitude = fminf(fmaxf((sum[cc] * normalization) - 1.0f, -1.0f), 1.0f);
                avg -= avg / averaging;
                avg += amplitude / averaging;
                if (++slot == ring_slots) slot = 0;
            }
        }
    }
    average[i] = avg;
    points[(i * 2) + 1] = avg;
}
}  // namespace

hipError_t launch_lineplot_span(float* points, float* average, const float* in_ring, uint64_t in_offset, uint64_t slot_stride,
                                uint64_t first_slot, uint64_t ring_slots, uint64_t cycles, uint64_t batches, uint64_t elements,
                                int64_t batch_stride, int64_t elem_stride, uint64_t decimation, float normalization,
                                float averaging, hipStream_t stream) {
    if (elements == 0 || cycles == 0) return hipSuccess;
    if (ring_slots == 0 || first_slot >= ring_slots || cycles > 0xffffffffull) return hipErrorInvalidValue;
    (void)hipGetLastError();
#define JST_LP_SPAN(D, C)                                                                                                          \
    hipLaunchKernelGGL((lineplot_span_kernel<D, C>), dim3((unsigned)((elements + 255) / 256)), dim3(256), 0, stream, points, average, \
                       in_ring, in_offset, slot_stride, (uint32_t)first_slot, (uint32_t)ring_slots, (uint32_t)cycles, batches, elements,  \
                       batch_stride, elem_stride, decimation, normalization, averaging)
    if (batches >= 64) JST_LP_SPAN(64, 1);
    else if (batches <= 16 && cycles >= 2) JST_LP_SPAN(16, 4);
    else JST_LP_SPAN(16, 1);
#undef JST_LP_SPAN
    return hipGetLastError();
}

hipError_t launch_lineplot(float* points, float* average, const float* in, uint64_t in_offset,
                           uint64_t batches, uint64_t elements, int64_t batch_stride,
                           int64_t elem_stride, uint64_t decimation, float normalization,
                           float averaging, hipStream_t stream) {
    if (elements == 0) return hipSuccess;
    (void)hipGetLastError();
    if (batches >= 64)
        hipLaunchKernelGGL(lineplot_kernel<64>, dim3((unsigned)((elements + 255) / 256)), dim3(256), 0,
                           stream, points, average, in, in_offset, batches, elements, batch_stride,
                           elem_stride, decimation, normalization, averaging);
    else
        hipLaunchKernelGGL(lineplot_kernel<16>, dim3((unsigned)((elements + 255) / 256)), dim3(256), 0,
                           stream, points, average, in, in_offset, batches, elements, batch_stride,
                           elem_stride, decimation, normalization, averaging);
    return hipGetLastError();
}

}  // namespace jst::kernels
