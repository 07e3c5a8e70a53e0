// agc.hip -- tiled RMS automatic gain control (dsp/agc/module_impl_native_cpu.cc:20-160).
//
// Reference arithmetic, kept operation for operation (all in F64):
//   power(tile)  = sum over the tile, in sample order, of re*re + im*im          (:24-33,116-121)
//   raw(tile)    = clamp(reference / sqrt(power / len + epsilon), minGain, maxGain)  (:123-126)
//   start(0)     = raw(0); end(t) = t+1 < tiles ? LimitGainChange(raw(t+1), start(t)) : start(t);
//   start(t+1)   = end(t)                                                            (:128-150)
//   out[s]       = ApplyGain(in[s], start + (end-start)/len * s_in_tile)             (:41-64,136-147)
//
// The F64 tile sum is order dependent, so it stays sequential: one workgroup per (lane, tile)
// stages the per-sample powers in LDS with coalesced loads and lane 0 adds them in order.  The
// tile chain (LimitGainChange) is one thread per lane; the gain application is one thread per
// sample.  Three launches; all state is recomputed every call (the module is STATELESS).
#include <cstdlib>

#include "device_math.hh"
#include "kernels.hh"

namespace jst::kernels {

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 2048;  // F64 powers staged per step: 16 KiB of LDS

__device__ __forceinline__ void lane_bases(const AgcParams& p, uint64_t lane, int64_t& in_base,
                                           int64_t& out_base) {
    in_base = (int64_t)p.in_offset;
    out_base = (int64_t)p.out_offset;
    for (int a = p.lane_rank - 1; a >= 0; --a) {
        const uint64_t c = lane % p.lane_shape[a];
        lane /= p.lane_shape[a];
        in_base += (int64_t)c * p.in_lane_stride[a];
        out_base += (int64_t)c * p.out_lane_stride[a];
    }
}

__device__ __forceinline__ double sample_power(float v) {
    const double x = v;
    return x * x;
}
__device__ __forceinline__ double sample_power(float2 v) {
    const double re = v.x, im = v.y;
    return re * re + im * im;
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) {
    return (v < lo) ? lo : ((hi < v) ? hi : v);  // std::clamp
}

// The tile's power, summed in SAMPLE ORDER by one thread (agc/module_impl_native_cpu.cc: a plain F64 accumulation; no other
// order gives the reference's bits).  What can be taken off the chain is everything but the additions: sixteen addends come
// out of LDS as eight 16-byte reads issued one block AHEAD of the additions that consume them, so the chain runs at the F64
// add latency instead of an LDS round trip per eight elements (a lane of 805 samples: ~6 us -> ~3 us of the kernel's 11).
__device__ __forceinline__ double ordered_sum(const double* powers, uint64_t n, double sum) {
    const double2* pw = reinterpret_cast<const double2*>(powers);
    const uint64_t nb = n / 16;  // whole blocks of sixteen addends
    double2 a[8], c[8];
    auto load = [&](double2 (&dst)[8], uint64_t block) {  // (a block index past the end re-reads the last block: never added)
        const double2* q = pw + (block < nb ? block : nb - 1) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = q[k];
    };
    auto add = [&](const double2 (&src)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sum += src[k].x;
            sum += src[k].y;
        }
    };
    uint64_t blk = 0;
    if (nb > 0) load(a, 0);
    for (; blk + 2 <= nb; blk += 2) {  // two register sets in turn: no copies, the loads of one set fly over the other's additions
        load(c, blk + 1);
        add(a);
        load(a, blk + 2);
        add(c);
    }
    if (blk < nb) add(a);  // an odd number of blocks: the last one is in `a`
    for (uint64_t i = nb * 16; i < n; ++i) sum += powers[i];
    return sum;
}

template <class T>
__global__ __launch_bounds__(kBlock) void agc_power_kernel(const AgcParams p,
                                                           const T* __restrict__ in,
                                                           double* __restrict__ gains) {
    __shared__ __attribute__((aligned(16))) double powers[kChunk];
    const uint64_t lane = blockIdx.x / p.tiles, tile = blockIdx.x % p.tiles;
    int64_t in_base, out_base;
    lane_bases(p, lane, in_base, out_base);
    const uint64_t start = tile * p.tile;
    const uint64_t len = (p.tile < p.samples - start) ? p.tile : (p.samples - start);
    double sum = 0.0;
    for (uint64_t c0 = 0; c0 < len; c0 += kChunk) {
        const uint64_t n = (len - c0 < (uint64_t)kChunk) ? (len - c0) : (uint64_t)kChunk;
        for (uint64_t i = threadIdx.x; i < n; i += kBlock)
            powers[i] = sample_power(in[in_base + (int64_t)(start + c0 + i) * p.in_sample_stride]);
        __syncthreads();
        if (threadIdx.x == 0) sum = ordered_sum(powers, n, sum);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = sum / (double)len;
        gains[(lane * p.tiles + tile) * 2] =
            clampd(p.reference / sqrt(mean + p.epsilon), p.min_gain, p.max_gain);
    }
}

// LimitGainChange (:66-77) walked along the tiles of one lane; rewrites gains[][0..1] in place
// as (start, end) of every tile.
__global__ __launch_bounds__(kBlock) void agc_chain_kernel(const AgcParams p,
                                                           double* __restrict__ gains) {
    const uint64_t lane = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (lane >= p.lanes) return;
    double* g = gains + lane * p.tiles * 2;
    double start = g[0];
    for (uint64_t t = 0; t < p.tiles; ++t) {
        double end = start;
        if (t + 1 < p.tiles) {
            const double raw = g[(t + 1) * 2];
            const double q = start / p.max_gain_change;
            const double lo = (p.min_gain < q) ? q : p.min_gain;  // std::max(minGain, q)
            const double hi =
                (start > p.max_gain / p.max_gain_change) ? p.max_gain : start * p.max_gain_change;
            end = clampd(raw, lo, hi);
        }
        g[t * 2] = start;
        g[t * 2 + 1] = end;
        start = end;
    }
}

__device__ __forceinline__ double limit_gain(double magnitude, double gain, double limit) {
    return (magnitude > limit / gain) ? nextafter(limit / magnitude, 0.0) : gain;  // :35-41
}
__device__ __forceinline__ float clamp_to_f32(double v) {
    const double m = 3.40282346638528859811704183484516925e+38;  // numeric_limits<F32>::max()
    return (float)clampd(v, -m, m);
}
__device__ __forceinline__ float apply_gain(float v, double gain) {
    const double x = v;
    const double g = limit_gain(fabs(x), gain, 3.40282346638528859811704183484516925e+38);
    return clamp_to_f32(x * g);
}
__device__ __forceinline__ float2 apply_gain(float2 v, double gain) {
    const double re = v.x, im = v.y;
    // kMaxSafeCF32Magnitude = (F64)nextafterf(FLT_MAX, 0) (:20-21)
    const double g = limit_gain(hypot(re, im), gain, 3.40282326356119256160033759537265639e+38);
    return jst::dev::mk(clamp_to_f32(re * g), clamp_to_f32(im * g));
}

template <class T>
__global__ __launch_bounds__(kBlock) void agc_apply_kernel(const AgcParams p, T* __restrict__ out,
                                                           const T* __restrict__ in,
                                                           const double* __restrict__ gains) {
    const uint64_t total = p.lanes * p.samples;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * kBlock) {
        const uint64_t lane = idx / p.samples, s = idx % p.samples;
        int64_t in_base, out_base;
        lane_bases(p, lane, in_base, out_base);
        const uint64_t tile = s / p.tile, k = s % p.tile;
        const uint64_t start = tile * p.tile;
        const uint64_t len = (p.tile < p.samples - start) ? p.tile : (p.samples - start);
        const double g0 = gains[(lane * p.tiles + tile) * 2], g1 = gains[(lane * p.tiles + tile) * 2 + 1];
        const double step = (g1 - g0) / (double)len;
        const double gain = g0 + step * (double)k;
        out[out_base + (int64_t)s * p.out_sample_stride] =
            apply_gain(in[in_base + (int64_t)s * p.in_sample_stride], gain);
    }
}

// ONE launch when every lane is a single tile (the spectrum_engine block's AGC: one RMS tile per spectrum,
// spectrum_engine/block_impl.cc:186-190).  start = end = raw(0): no gain ramp and no neighbour tile to look at, so the
// workgroup that summed the tile's power applies the gain to it right away -- same F64 operations in the same order as the
// three kernels below (power in sample order, then ApplyGain with step = (g - g) / len = +0 and gain = g + 0 * k = g).
constexpr int kSingleThreads = 1024;  // a lane of the spectrum_engine's AGC is one spectrum: one or two samples per thread
template <class T, bool TAIL = false>
__global__ __launch_bounds__(kSingleThreads) void agc_single_tile_kernel(const AgcParams p, T* __restrict__ out, const T* __restrict__ in,
                                                                         double* __restrict__ gains, const AgcTail tail = AgcTail{}) {
    constexpr int kPer = kChunk / kSingleThreads;  // samples per thread and chunk
    __shared__ __attribute__((aligned(16))) double powers[kChunk];
    __shared__ double tile_gain;
    const uint64_t lane = blockIdx.x;
    // TAIL with a Waterfall: the ring cursor as waterfall_kernel reads it (kernels/waterfall.hip; PlanWaterfallWrite,
    // waterfall/ring_state.hh:16-28) -- every workgroup at entry, the last one to finish advances it
    uint64_t write_index = 0;
    if constexpr (TAIL)
        if (tail.ring) write_index = __hip_atomic_load(&tail.ring_state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int64_t in_base, out_base;
    lane_bases(p, lane, in_base, out_base);
    const uint64_t len = p.samples;
    // a lane of at most one chunk (2048 samples) is read ONCE: the samples wait in registers for the gain
    const bool resident = len <= (uint64_t)kChunk;
    T held[kPer];
    double sum = 0.0;
    for (uint64_t c0 = 0; c0 < len; c0 += kChunk) {
        const uint64_t n = (len - c0 < (uint64_t)kChunk) ? (len - c0) : (uint64_t)kChunk;
        T v[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {  // all of a thread's loads first
            const uint64_t i = threadIdx.x + (uint64_t)j * kSingleThreads;
            v[j] = in[in_base + (int64_t)(c0 + (i < n ? i : 0)) * p.in_sample_stride];
        }
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const uint64_t i = threadIdx.x + (uint64_t)j * kSingleThreads;
            if (i < n) powers[i] = sample_power(v[j]);
            held[j] = v[j];
        }
        __syncthreads();
        if (threadIdx.x == 0) sum = ordered_sum(powers, n, sum);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = sum / (double)len;
        const double g = clampd(p.reference / sqrt(mean + p.epsilon), p.min_gain, p.max_gain);
        gains[lane * 2] = g;      // (start, end) of the lane's only tile, as agc_chain_kernel leaves them
        gains[lane * 2 + 1] = g;
        tile_gain = g;
    }
    __syncthreads();
    const double g0 = tile_gain;
    const double step = (g0 - g0) / (double)len;
    // TAIL
    const uint64_t batches = p.lanes, height = tail.height;
    const uint64_t retained = (TAIL && tail.ring) ? (batches < height ? batches : height) : 0;
    const uint64_t source_row = batches - retained;
    const bool kept = TAIL && tail.ring && lane >= source_row;
    float* ring_row = nullptr;
    if (kept) ring_row = tail.ring + ((write_index + (source_row % height)) % height + (lane - source_row)) % height * len;
    float* level_row = TAIL ? tail.level + lane * len : nullptr;
    auto finish = [&](uint64_t k, T x) {
        const double gain = g0 + step * (double)k;
        const T y = apply_gain(x, gain);
        if constexpr (!TAIL) {
            out[out_base + (int64_t)k * p.out_sample_stride] = y;
        } else {
            if (out) out[out_base + (int64_t)k * p.out_sample_stride] = y;
            const float level = tail.fast ? jst::dev::range_f32_fast(jst::dev::amplitude_cf32_fast(y, tail.coeff), tail.scale, tail.offset)
                                          : jst::dev::range_f32(jst::dev::amplitude_exact(y, tail.coeff), tail.scale, tail.offset);
            level_row[k] = level;
            if (kept) ring_row[k] = level;
        }
    };
    if (resident) {
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const uint64_t k = threadIdx.x + (uint64_t)j * kSingleThreads;
            if (k < len) finish(k, held[j]);
        }
    } else {
        for (uint64_t k = threadIdx.x; k < len; k += kSingleThreads) finish(k, in[in_base + (int64_t)k * p.in_sample_stride]);
    }
    if constexpr (TAIL) {
        if (tail.ring) {
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint64_t ticket = __hip_atomic_fetch_add(&tail.ring_state[2], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ticket == (uint64_t)gridDim.x - 1) {  // WaterfallRingState::advance (ring_state.hh:40-43)
                    const uint64_t dirty = __hip_atomic_load(&tail.ring_state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint64_t room = height - dirty;
                    __hip_atomic_store(&tail.ring_state[0], (write_index + (batches % height)) % height, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&tail.ring_state[1], dirty + (batches < room ? batches : room), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&tail.ring_state[2], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

template <class T>
hipError_t run(T* out, const T* in, double* gains, const AgcParams& p, hipStream_t s) {
    if (p.lanes == 0 || p.samples == 0) return hipSuccess;
    (void)hipGetLastError();
    static const bool three_kernels = getenv("JST_AGC_THREE_KERNELS") != nullptr;  // A/B and tests
    if (p.tiles == 1 && !three_kernels) {
        hipLaunchKernelGGL((agc_single_tile_kernel<T, false>), dim3((unsigned)p.lanes), dim3(kSingleThreads), 0, s, p, out, in, gains, AgcTail{});
        return hipGetLastError();
    }
    hipLaunchKernelGGL((agc_power_kernel<T>), dim3((unsigned)(p.lanes * p.tiles)), dim3(kBlock), 0,
                       s, p, in, gains);
    hipLaunchKernelGGL(agc_chain_kernel, dim3((unsigned)((p.lanes + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p, gains);
    const uint64_t total = p.lanes * p.samples;
    uint64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256ull * 16ull) blocks = 256ull * 16ull;
    hipLaunchKernelGGL((agc_apply_kernel<T>), dim3((unsigned)blocks), dim3(kBlock), 0, s, p, out,
                       in, (const double*)gains);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_agc_tail(void* out, const void* in, double* gains, const AgcParams& p, const AgcTail& tail, hipStream_t s) {
    if (p.tiles != 1 || p.lanes == 0 || p.samples == 0 || p.lanes > 0x7fffffffull || !tail.level || (tail.ring && (!tail.ring_state || tail.height == 0)))
        return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL((agc_single_tile_kernel<float2, true>), dim3((unsigned)p.lanes), dim3(kSingleThreads), 0, s, p, static_cast<float2*>(out),
                       static_cast<const float2*>(in), gains, tail);
    return hipGetLastError();
}

hipError_t launch_agc(void* out, const void* in, bool complex, double* gains, const AgcParams& p,
                      hipStream_t s) {
    if (p.lanes * p.tiles > 0x7fffffffull) return hipErrorInvalidValue;
    if (complex) return run(static_cast<float2*>(out), static_cast<const float2*>(in), gains, p, s);
    return run(static_cast<float*>(out), static_cast<const float*>(in), gains, p, s);
}

}  // namespace jst::kernels
