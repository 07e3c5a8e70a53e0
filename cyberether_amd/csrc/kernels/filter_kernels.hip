// filter_kernels.hip -- the modules of the Filter (FFT overlap-add) and FM side chains that are not
// plain elementwise maps: Pad, Unpad, Fold, OverlapAdd, PhaseCorrection, FilterTaps, Arithmetic
// (axis reduction) and FM.  The reference has CPU implementations only for all of them
// (SURVEY 2b); every kernel restates the CPU arithmetic in the CPU's order (F64 where the CPU
// uses F64) so results are bit-identical -- FM included: its atan2f / sinf / cosf are the restatements of
// the host libm's routines in libm_float.hh (glibc 2.35, swept against libm.so.6 on every float).
#include <cstdlib>

#include "device_math.hh"
#include "kernels.hh"

namespace jst::kernels {

using namespace jst::dev;

namespace {
constexpr int kBlock = 256;
inline unsigned grid_for(uint64_t n) {
    const uint64_t need = (n + kBlock - 1) / kBlock;
    return (unsigned)(need < 4096 ? (need ? need : 1) : 4096);
}
#define JST_GRID_STRIDE(i, n)                                               \
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < (n); \
         i += (uint64_t)gridDim.x * kBlock)

// ---- Pad / Unpad (core/pad/module_impl_native_cpu.cc:75-140, core/unpad/...:66-135) ----------
// dense [outer, axis, inner]; T is float (F32) or float2 (CF32)
template <class T>
__global__ __launch_bounds__(kBlock) void pad_kernel(T* __restrict__ out, const T* __restrict__ in,
                                                     uint64_t outer, uint64_t in_axis,
                                                     uint64_t out_axis, uint64_t inner) {
    const uint64_t total = outer * out_axis * inner;
    JST_GRID_STRIDE(e, total) {
        const uint64_t i = e % inner, a = (e / inner) % out_axis, o = e / (inner * out_axis);
        T v{};
        if (a < in_axis) v = in[(o * in_axis + a) * inner + i];
        out[e] = v;
    }
}
template <class T>
__global__ __launch_bounds__(kBlock) void unpad_kernel(T* __restrict__ body, T* __restrict__ tail,
                                                       const T* __restrict__ in, uint64_t outer,
                                                       uint64_t in_axis, uint64_t body_axis,
                                                       uint64_t inner) {
    const uint64_t total = outer * in_axis * inner, tail_axis = in_axis - body_axis;
    JST_GRID_STRIDE(e, total) {
        const uint64_t i = e % inner, a = (e / inner) % in_axis, o = e / (inner * in_axis);
        if (a < body_axis) body[(o * body_axis + a) * inner + i] = in[e];
        else tail[(o * tail_axis + (a - body_axis)) * inner + i] = in[e];
    }
}

// ---- Fold (dsp/fold/module_impl_native_cpu.cc:103-172) ---------------------------------------
// dense [outer, axis, inner]; channel coordinate = (flat / chan_inner) % chan_count of the OUTPUT.
template <bool COMPLEX>
__global__ __launch_bounds__(kBlock) void fold_kernel(float* __restrict__ out,
                                                      const float* __restrict__ in, uint64_t outer,
                                                      uint64_t axis_size, uint64_t fold_size,
                                                      uint64_t inner, uint64_t scalar_offset,
                                                      const uint64_t* __restrict__ chan_offsets,
                                                      uint64_t chan_count, uint64_t chan_inner) {
    const uint64_t decim = axis_size / fold_size, total = outer * fold_size * inner;
    const double divisor = (double)decim;
    JST_GRID_STRIDE(e, total) {
        const uint64_t i = e % inner, k = (e / inner) % fold_size, o = e / (inner * fold_size);
        const uint64_t off = chan_offsets ? chan_offsets[(e / chan_inner) % chan_count] % axis_size
                                          : scalar_offset;
        double sr = 0.0, si = 0.0;
        for (uint64_t g = 0; g < decim; ++g) {
            const uint64_t shifted = k + g * fold_size;
            const uint64_t ia = shifted >= off ? shifted - off : axis_size - (off - shifted);
            const uint64_t idx = (o * axis_size + ia) * inner + i;
            if constexpr (COMPLEX) {
                sr += (double)in[2 * idx];
                si += (double)in[2 * idx + 1];
            } else {
                sr += (double)in[idx];
            }
        }
        if constexpr (COMPLEX) {
            out[2 * e] = (float)(sr / divisor);
            out[2 * e + 1] = (float)(si / divisor);
        } else {
            out[e] = (float)(sr / divisor);
        }
    }
}

// Fold over the last axis of a product a*b that is never written: same index walk as fold_kernel,
// each addend formed with the Multiply module's arithmetic.
__global__ __launch_bounds__(kBlock) void fold_product_kernel(float2* __restrict__ out,
                                                              const float2* __restrict__ a,
                                                              const float2* __restrict__ b,
                                                              const EwLayout P, uint64_t axis_size,
                                                              uint64_t fold_size, uint64_t scalar_offset,
                                                              const uint64_t* __restrict__ chan_offsets,
                                                              uint64_t chan_count, uint64_t chan_inner) {
    const uint64_t decim = axis_size / fold_size, outer = P.size / axis_size;
    const uint64_t total = outer * fold_size;
    const double divisor = (double)decim;
    JST_GRID_STRIDE(e, total) {
        const uint64_t k = e % fold_size, o = e / fold_size;
        const uint64_t off = chan_offsets ? chan_offsets[(e / chan_inner) % chan_count] % axis_size
                                          : scalar_offset;
        // operand offsets of row o of the product tensor (all axes but the last)
        int64_t oa = (int64_t)P.offset[1], ob = (int64_t)P.offset[2];
        uint64_t rem = o;
        for (int ax = P.rank - 2; ax >= 0; --ax) {
            const uint64_t c = rem % P.shape[ax];
            rem /= P.shape[ax];
            oa += (int64_t)c * P.stride[1][ax];
            ob += (int64_t)c * P.stride[2][ax];
        }
        const int64_t sa = P.stride[1][P.rank - 1], sb = P.stride[2][P.rank - 1];
        double sr = 0.0, si = 0.0;
        for (uint64_t g = 0; g < decim; ++g) {
            const uint64_t shifted = k + g * fold_size;
            const uint64_t ia = shifted >= off ? shifted - off : axis_size - (off - shifted);
            const float2 p = jst::dev::cmul_full(a[oa + (int64_t)ia * sa], b[ob + (int64_t)ia * sb]);
            sr += (double)p.x;
            si += (double)p.y;
        }
        out[e] = jst::dev::mk((float)(sr / divisor), (float)(si / divisor));
    }
}

// ---- OverlapAdd (dsp/overlap_add/module_impl_native_cpu.cc:121-202) --------------------------
struct OlaLayout {
    uint32_t rank;
    int32_t batch_axis;  // -1: none
    uint64_t buf_shape[kMaxRank], ovl_shape[kMaxRank];
};
template <class T>
__device__ __forceinline__ T add_t(T a, T b) {
    if constexpr (sizeof(T) == 8) return mk(a.x + b.x, a.y + b.y);
    else return a + b;
}
template <class T>
__global__ __launch_bounds__(kBlock) void overlap_add_kernel(T* __restrict__ out,
                                                             const T* __restrict__ buf,
                                                             const T* __restrict__ ovl,
                                                             const T* __restrict__ prev,
                                                             const OlaLayout L, uint64_t total) {
    JST_GRID_STRIDE(e, total) {
        uint64_t c[kMaxRank], rem = e;  // coordinates of e in the buffer
        for (int d = (int)L.rank - 1; d >= 0; --d) {
            c[d] = rem % L.buf_shape[d];
            rem /= L.buf_shape[d];
        }
        T v = buf[e];
        bool inside = true;
        for (uint32_t d = 0; d < L.rank; ++d) inside = inside && (c[d] < L.ovl_shape[d]);
        if (inside) {
            const bool first = L.batch_axis < 0 || c[L.batch_axis] == 0;
            uint64_t idx = 0;
            if (first) {  // previous-overlap state: overlap shape with batch extent 1
                for (uint32_t d = 0; d < L.rank; ++d) {
                    const bool is_batch = (int)d == L.batch_axis;
                    idx = idx * (is_batch ? 1 : L.ovl_shape[d]) + (is_batch ? 0 : c[d]);
                }
                v = add_t(v, prev[idx]);
            } else {
                for (uint32_t d = 0; d < L.rank; ++d)
                    idx = idx * L.ovl_shape[d] + (((int)d == L.batch_axis) ? c[d] - 1 : c[d]);
                v = add_t(v, ovl[idx]);
            }
        }
        out[e] = v;
    }
}
template <class T>
__global__ __launch_bounds__(kBlock) void overlap_state_kernel(T* __restrict__ prev,
                                                               const T* __restrict__ ovl,
                                                               const OlaLayout L, uint64_t total) {
    JST_GRID_STRIDE(e, total) {  // e indexes prev (overlap shape with batch extent 1)
        uint64_t c[kMaxRank], rem = e;
        for (int d = (int)L.rank - 1; d >= 0; --d) {
            const uint64_t ext = (d == L.batch_axis) ? 1 : L.ovl_shape[d];
            c[d] = rem % ext;
            rem /= ext;
        }
        if (L.batch_axis >= 0) c[L.batch_axis] = L.ovl_shape[L.batch_axis] - 1;
        uint64_t idx = 0;
        for (uint32_t d = 0; d < L.rank; ++d) idx = idx * L.ovl_shape[d] + c[d];
        prev[e] = ovl[idx];
    }
}

// The overlap region only: `out` already holds the buffer values (the fused inverse transform wrote them there).
// One thread per overlap element; the thread of a first-batch element is the only one that touches its state entry,
// so it reads it for the sum and then replaces it with the last batch's overlap.
// PhaseNext (the fused Filter tail, filter_modules.cc TryFuseFilter): the transform's epilogue of THIS cycle has read the
// correction table; the first workgroup advances the phase state exactly as phase_table_kernel does at the end of a cycle and
// writes the table of the NEXT cycle from the advanced state -- the same F64 operations on the same stored values, one
// cycle earlier in wall time.
struct PhaseNext {
    float2* corr = nullptr;
    double* phases = nullptr;
    const double* increments = nullptr;
    uint64_t channels = 0, batches = 0;
};
__device__ __forceinline__ void phase_table_row(float2* corr, double ph0, double wrapped, uint64_t c, uint64_t batches) {
    for (uint64_t b = 0; b < batches; ++b) {
        const double ph = ph0 + wrapped * (double)b;
        corr[c * batches + b] = mk((float)cos(ph), (float)sin(ph));
    }
}
template <class T>
__global__ __launch_bounds__(kBlock) void overlap_heads_kernel(T* __restrict__ out, const T* __restrict__ ovl,
                                                               T* __restrict__ prev, const OlaLayout L,
                                                               uint64_t total, const PhaseNext pn) {
    if (pn.corr != nullptr && blockIdx.x == 0) {
        const double two_pi = 2.0 * 3.14159265358979323846;
        for (uint64_t c = threadIdx.x; c < pn.channels; c += blockDim.x) {
            const double wrapped = remainder(pn.increments[c], two_pi);
            const double next = remainder(pn.phases[c] + wrapped * (double)pn.batches, two_pi);
            pn.phases[c] = next;
            phase_table_row(pn.corr, next, wrapped, c, pn.batches);
        }
    }
    JST_GRID_STRIDE(e, total) {  // e indexes the overlap tensor
        uint64_t c[kMaxRank], rem = e;
        for (int d = (int)L.rank - 1; d >= 0; --d) {
            c[d] = rem % L.ovl_shape[d];
            rem /= L.ovl_shape[d];
        }
        uint64_t bi = 0;  // the same coordinates in the buffer
        for (uint32_t d = 0; d < L.rank; ++d) bi = bi * L.buf_shape[d] + c[d];
        const bool first = L.batch_axis < 0 || c[L.batch_axis] == 0;
        uint64_t idx = 0;
        if (first) {
            uint64_t last = 0;  // the last batch's overlap element with these coordinates
            for (uint32_t d = 0; d < L.rank; ++d) {
                const bool is_batch = (int)d == L.batch_axis;
                idx = idx * (is_batch ? 1 : L.ovl_shape[d]) + (is_batch ? 0 : c[d]);
                last = last * L.ovl_shape[d] + (is_batch ? L.ovl_shape[d] - 1 : c[d]);
            }
            out[bi] = add_t(out[bi], prev[idx]);
            prev[idx] = ovl[last];
        } else {
            for (uint32_t d = 0; d < L.rank; ++d)
                idx = idx * L.ovl_shape[d] + (((int)d == L.batch_axis) ? c[d] - 1 : c[d]);
            out[bi] = add_t(out[bi], ovl[idx]);
        }
    }
}

// ---- PhaseCorrection (dsp/phase_correction/module_impl_native_cpu.cc:60-115) -----------------
__global__ void phase_table_kernel(float2* __restrict__ corr, double* __restrict__ phases,
                                   const double* __restrict__ increments, uint64_t channels,
                                   uint64_t batches, bool advance) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= channels) return;
    const double two_pi = 2.0 * 3.14159265358979323846;
    const double wrapped = remainder(increments[c], two_pi);
    const double ph0 = phases[c];
    phase_table_row(corr, ph0, wrapped, c, batches);
    if (advance) phases[c] = remainder(ph0 + wrapped * (double)batches, two_pi);
}
__global__ __launch_bounds__(kBlock) void phase_mul_kernel(const EwLayout L, float2* __restrict__ out,
                                                           const float2* __restrict__ in,
                                                           const float2* __restrict__ corr,
                                                           uint64_t batches, uint64_t batch_inner,
                                                           uint64_t channels, uint64_t channel_inner) {
    JST_GRID_STRIDE(idx, L.size) {
        int64_t o0 = (int64_t)L.offset[0], o1 = (int64_t)L.offset[1];
        if (L.contiguous) {
            o0 += (int64_t)idx;
            o1 += (int64_t)idx;
        } else {
            uint64_t rem = idx;
            for (int a = L.rank - 1; a >= 0; --a) {
                const uint64_t c = rem % L.shape[a];
                rem /= L.shape[a];
                o0 += (int64_t)c * L.stride[0][a];
                o1 += (int64_t)c * L.stride[1][a];
            }
        }
        const uint64_t b = batches == 1 ? 0 : (idx / batch_inner) % batches;
        const uint64_t c = channels == 1 ? 0 : (idx / channel_inner) % channels;
        out[o0] = cmul_full(in[o1], corr[c * batches + b]);
    }
}

// ---- FilterTaps (dsp/filter_taps/module_impl_native_cpu.cc:46-80) ----------------------------
__global__ void filter_taps_kernel(float2* __restrict__ out, double sample_rate, double bandwidth,
                                   const double* __restrict__ center, uint64_t heads,
                                   uint64_t taps) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= heads * taps) return;
    const double pi = 3.14159265358979323846;
    const uint64_t c = e / taps, i = e % taps;
    const double filter_width = (bandwidth / sample_rate) / 2.0;
    const double filter_offset = center[c] / sample_rate;
    const double fi = (double)i, half = (double)(taps - 1) / 2.0, n = fi - half;
    const double sinc =
        (n == 0.0) ? (2.0 * filter_width) : sin(2.0 * pi * filter_width * n) / (pi * n);
    const double win = (taps == 1) ? 1.0
                                   : 0.42 - 0.50 * cos(2.0 * pi * fi / (double)(taps - 1)) +
                                         0.08 * cos(4.0 * pi * fi / (double)(taps - 1));
    const double theta = ((2.0 * pi) * n) * filter_offset;
    const double sw = sinc * win;
    out[e] = mk((float)(sw * cos(theta)), (float)(sw * sin(theta)));
}

// ---- Arithmetic (core/arithmetic/module_impl_native_cpu.cc:98-146) ---------------------------
// L: operand 0 = output, operand 1 = input, both indexed over the OUTPUT shape (reduced axis has
// extent 1); the reduced axis is walked with (r, r_stride) in increasing order from +0.
template <class T, int OP>
__global__ __launch_bounds__(kBlock) void arithmetic_kernel(const EwLayout L, T* __restrict__ out,
                                                            const T* __restrict__ in, uint64_t r,
                                                            int64_t r_stride) {
    JST_GRID_STRIDE(idx, L.size) {
        int64_t o0 = (int64_t)L.offset[0], o1 = (int64_t)L.offset[1];
        uint64_t rem = idx;
        for (int a = L.rank - 1; a >= 0; --a) {
            const uint64_t c = rem % L.shape[a];
            rem /= L.shape[a];
            o0 += (int64_t)c * L.stride[0][a];
            o1 += (int64_t)c * L.stride[1][a];
        }
        T acc{};  // zeroKernel(): the output starts at +0
        for (uint64_t k = 0; k < r; ++k) {
            const T v = in[o1 + (int64_t)k * r_stride];
            if constexpr (sizeof(T) == 8) {
                if (OP == 0) acc = mk(acc.x + v.x, acc.y + v.y);
                else if (OP == 1) acc = mk(acc.x - v.x, acc.y - v.y);
                else acc = cmul_full(acc, v);
            } else {
                if (OP == 0) acc = acc + v;
                else if (OP == 1) acc = acc - v;
                else if (OP == 2) acc = acc * v;
                else acc = acc / v;
            }
        }
        out[o0] = acc;
    }
}

// ---- FM (dsp/fm/module_impl_native_cpu.cc:43-174) ---------------------------------------------
// One thread per lane walks its batches x samples in order (the stereo decoder and the
// de-emphasis are recursive filters: the lane is the only parallel dimension the reference's
// algorithm offers).  Narrow FM without de-emphasis has no recursion: one thread per SAMPLE.
struct FmState {
    float prev_re, prev_im;
    int has_prev;
    float narrow_deemph;
    float pilot_phase, pilot_cos_stage, pilot_sin_stage, pilot_cos, pilot_sin, left_de, right_de;
    float sum_notch[2], diff_notch[2], sum_filter[3][2], diff_filter[3][2];
};
__device__ __forceinline__ float fm_biquad(float x, const float* c /*b0 b1 b2 a1 a2*/, float* s) {
    const float y = c[0] * x + s[0];
    s[0] = c[1] * x - c[3] * y + s[1];
    s[1] = c[2] * x - c[4] * y;
    return y;
}
__device__ __forceinline__ float fm_lowpass(float x, const FmCoeffs& k, float (*s)[2]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) x = fm_biquad(x, k.lp[i], s[i]);
    return x;
}
__device__ __forceinline__ void fm_lane_offsets(const FmLayout& L, uint64_t lane, int64_t& in_off,
                                                int64_t& out_off) {
    in_off = (int64_t)L.in_offset;
    out_off = (int64_t)L.out_offset;
    for (int a = L.lane_rank - 1; a >= 0; --a) {
        const uint64_t c = lane % L.lane_shape[a];
        lane /= L.lane_shape[a];
        in_off += (int64_t)c * L.in_lane_stride[a];
        out_off += (int64_t)c * L.out_lane_stride[a];
    }
}
__device__ __forceinline__ float fm_discriminate(float2 prev, float2 cur, bool has, float ref) {
    const bool fin = __builtin_isfinite(cur.x) && __builtin_isfinite(cur.y) &&
                     __builtin_isfinite(prev.x) && __builtin_isfinite(prev.y);
    if (!has) return 0.0f;
    if (!fin) return __builtin_nanf("");
    const float2 p = cmul_full(mk(prev.x, -prev.y), cur);  // conj(previous) * current
    return libm_atan2f(p.y, p.x) * ref;
}
__global__ __launch_bounds__(64) void fm_kernel(float* __restrict__ out, const float2* __restrict__ in,
                                                FmState* __restrict__ states, const FmCoeffs k,
                                                const FmLayout L) {
    const uint64_t lane = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (lane >= L.lanes) return;
    int64_t in_off, out_off;
    fm_lane_offsets(L, lane, in_off, out_off);
    FmState st = states[lane];
    const double two_pi = 2.0f * 3.14159265358979323846;  // F32 op F64 like the reference
    float2 prev = mk(st.prev_re, st.prev_im);
    bool has = st.has_prev != 0;
    for (uint64_t b = 0; b < L.batches; ++b) {
        for (uint64_t s = 0; s < L.samples; ++s) {
            const float2 cur =
                in[in_off + (int64_t)b * L.in_batch_stride + (int64_t)s * L.in_sample_stride];
            const int64_t oo =
                out_off + (int64_t)b * L.out_batch_stride + (int64_t)s * L.out_sample_stride;
            const float d = fm_discriminate(prev, cur, has, k.ref);
            if (!__builtin_isfinite(d)) {
                out[oo] = d;
                if (k.wide) {
                    out[oo + L.out_channel_stride] = d;
                    st.pilot_phase += k.pilot_inc;
                    if ((double)st.pilot_phase >= two_pi)
                        st.pilot_phase = (float)((double)st.pilot_phase - two_pi);
                }
            } else if (!k.wide) {
                if (!k.deemph_enabled) out[oo] = d;
                else {
                    st.narrow_deemph += k.deemph_alpha * (d - st.narrow_deemph);
                    out[oo] = st.narrow_deemph;
                }
            } else {
                const float pc = libm_cosf(st.pilot_phase), ps = libm_sinf(st.pilot_phase);
                st.pilot_cos_stage += k.pilot_alpha * (d * pc - st.pilot_cos_stage);
                st.pilot_sin_stage += k.pilot_alpha * (d * ps - st.pilot_sin_stage);
                st.pilot_cos += k.pilot_alpha * (st.pilot_cos_stage - st.pilot_cos);
                st.pilot_sin += k.pilot_alpha * (st.pilot_sin_stage - st.pilot_sin);
                const float sum = fm_lowpass(fm_biquad(d, k.notch, st.sum_notch), k, st.sum_filter);
                const float po = libm_atan2f(st.pilot_cos, st.pilot_sin);
                const float carrier = libm_sinf(2.0f * (st.pilot_phase + po));
                const float diff = fm_lowpass(
                    fm_biquad(2.0f * d * carrier, k.notch, st.diff_notch), k, st.diff_filter);
                float left = sum + diff, right = sum - diff;
                if (k.deemph_enabled) {
                    st.left_de += k.deemph_alpha * (left - st.left_de);
                    st.right_de += k.deemph_alpha * (right - st.right_de);
                    left = st.left_de;
                    right = st.right_de;
                }
                out[oo] = left;
                out[oo + L.out_channel_stride] = right;
                st.pilot_phase += k.pilot_inc;
                if ((double)st.pilot_phase >= two_pi)
                    st.pilot_phase = (float)((double)st.pilot_phase - two_pi);
            }
            prev = cur;
            has = true;
        }
    }
    st.prev_re = prev.x;
    st.prev_im = prev.y;
    st.has_prev = 1;
    states[lane] = st;
}
// ---- wide FM as a wavefront pipeline -------------------------------------------------------------
// The stereo decoder is a chain of serial recurrences (pilot phase, four one-poles, eight biquads, two
// de-emphasis one-poles) with element-wise transcendental stages in between.  fm_kernel above walks
// it with one thread per lane, every dependent operation paying full latency: ~0.9 us per sample.
// Here one workgroup of eight wavefronts per lane runs the stages as a SOFTWARE PIPELINE over chunks of
// 1024 samples: in step s (steps are separated by one workgroup barrier)
//   wave 0     B  pilot phase recurrence of chunk s          (one lane; data independent)
//   waves 4-7  A  d = discriminator of chunk s               (element-wise)
//              C  xc, xs = d * cosf / sinf(phase) of chunk s-1
//              E  pilot offset atan2f, carrier sinf, xd = 2 d carrier of chunk s-3
//   wave 1     D  the pilot one-poles of chunk s-2 as a 2-deep LANE pipeline: lane 0/2 the first pole
//                 of sample n, lane 1/3 the second pole of sample n-1, fed by a DPP row shift
//   wave 2     F  notch + 3 low-pass biquads of the sum path (lanes 0-3) and of the difference path
//                 (lanes 4-7) of chunk s-4 as a 4-deep lane pipeline: lane k works on sample n - k%4
//              L  left = sum + diff, right = sum - diff of chunk s-5;  O  the output of chunk s-7
//   wave 3     G  de-emphasis of left / right of chunk s-6 (two lanes)
// so the four serial stretches run side by side on four SIMDs and the decode costs the time of the
// longest of them (the biquad cascade, ~50 cycles per sample) instead of their sum.  Chunks travel
// between stages through LDS rings; every chain keeps its operations and their order (same bits).
// A non-finite discriminator sample leaves every filter state untouched (fm/module_impl_native_cpu.cc:
// 96-112): it travels down the lane pipelines as a bubble, and each serial stage flushes its lane
// pipeline with bubbles at the end of a chunk so that a chunk's results are complete within its step.
__device__ __forceinline__ float dpp_row_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
// take ? a : b on a lane mask: the serial stretches must stay straight-line code (the compiler turns
// a ?: around a recurrence step into an exec-mask branch per sample)
__device__ __forceinline__ float fm_pick(uint32_t take_mask, float a, float b) {
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, a) & take_mask) |
                                         (__builtin_bit_cast(uint32_t, b) & ~take_mask));
}
constexpr int kFmChunk = 2048;   // narrow + de-emphasis: samples staged in LDS per serial stretch
constexpr int kFmThreads = 256;
constexpr int kFmWideThreads = 512, kFmCh = 1024, kFmSlot = kFmCh + 16;  // slot: chunk + flush/prefetch slack

__global__ __launch_bounds__(kFmWideThreads) void fm_wide_kernel(float* __restrict__ out,
                                                                  const float2* __restrict__ in,
                                                                  FmState* __restrict__ states, const FmCoeffs k,
                                                                  const FmLayout L) {
    // LDS rings, indexed by chunk % depth; depth = steps between the producer and the last consumer
    __shared__ float r_phase[4][kFmSlot], r_d[8][kFmSlot], r_xc[2][kFmSlot], r_xs[2][kFmSlot],
        r_pcos[2][kFmSlot], r_psin[2][kFmSlot], r_xd[2][kFmSlot], r_sum[2][kFmSlot], r_diff[2][kFmSlot],
        r_left[3][kFmSlot], r_right[3][kFmSlot],
        spill[kFmSlot];  // where lanes without a result write (never read)
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lid = tid & 63u;
    const uint64_t lane = blockIdx.x, n_total = L.batches * L.samples;
    const uint32_t chunks = (uint32_t)((n_total + kFmCh - 1) / kFmCh);
    int64_t in_off, out_off;
    fm_lane_offsets(L, lane, in_off, out_off);
    const FmState st = states[lane];
    const float nan = __builtin_nanf("");
    auto in_at = [&](uint64_t n) {
        return in[in_off + (int64_t)(n / L.samples) * L.in_batch_stride + (int64_t)(n % L.samples) * L.in_sample_stride];
    };
    auto count_of = [&](uint32_t c) {
        const uint64_t left = n_total - (uint64_t)c * kFmCh;
        return (uint32_t)(left < (uint64_t)kFmCh ? left : (uint64_t)kFmCh);
    };
    // per-stage state, alive in the registers of the wave that owns the stage
    float ph = st.pilot_phase;                       // B (wave 0, lane 0)
    float pole = 0.0f, pole_handed = nan;            // D (wave 1, lanes 0-3)
    float c5[5] = {0, 0, 0, 0, 0}, s0 = 0.0f, s1 = 0.0f, bq_handed = nan;  // F (wave 2, lanes 0-7)
    float de = 0.0f;                                 // G (wave 3, lanes 0-1)
    if (wave == 1) {
        if (lid == 0) pole = st.pilot_cos_stage;
        if (lid == 1) pole = st.pilot_cos;
        if (lid == 2) pole = st.pilot_sin_stage;
        if (lid == 3) pole = st.pilot_sin;
    }
    const uint32_t stage = lid & 3u;
    const bool diff_path = (lid & 4u) != 0;
    if (wave == 2 && lid < 8) {
        const float* cc = stage == 0 ? k.notch : k.lp[stage - 1];
        const float* ss = stage == 0 ? (diff_path ? st.diff_notch : st.sum_notch)
                                     : (diff_path ? st.diff_filter[stage - 1] : st.sum_filter[stage - 1]);
#pragma unroll
        for (int j = 0; j < 5; ++j) c5[j] = cc[j];
        s0 = ss[0];
        s1 = ss[1];
    }
    if (wave == 3) de = lid == 0 ? st.left_de : st.right_de;
    const double two_pi = 2.0f * 3.14159265358979323846;  // F32 op F64 like the reference
    // (double)ph >= two_pi (6.283185307179586) <=> ph >= the float just above it: the common step is
    // one F32 add and one F32 compare, the F64 wrap runs every ~10th sample
    const float wrap_at = __builtin_bit_cast(float, 0x40C90FDBu);  // 6.2831854820251465

    for (uint32_t s = 0; s < chunks + 7; ++s) {
        if (wave == 0) {
            // ---- B: chunk s ----
            if (s < chunks && lid == 0) {
                const uint32_t cnt = count_of(s);
                float* dst = r_phase[s & 3];
                for (uint32_t i = 0; i < cnt; ++i) {
                    dst[i] = ph;
                    ph += k.pilot_inc;
                    if (ph >= wrap_at) ph = (float)((double)ph - two_pi);
                }
            }
        } else if (wave == 1) {
            // ---- D: chunk s-2; lanes 0,1 = cosine poles (first, second), 2,3 = sine poles ----
            if (s >= 2 && s - 2 < chunks) {
                const uint32_t c = s - 2, cnt = count_of(c);
                const bool head = (lid & 1u) == 0;  // first pole of a pair: reads the staged input
                const float* src = lid < 2 ? r_xc[c & 1] : r_xs[c & 1];
                float* res = lid == 1 ? r_pcos[c & 1] : (lid == 3 ? r_psin[c & 1] : spill);
                for (uint32_t i0 = 0; i0 < cnt + 1; i0 += 8) {  // +1: flush the lane pipeline with a bubble
                    float xin[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xin[j] = src[i0 + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float from_left = dpp_row_shr1(pole_handed);
                        const float x = head ? xin[j] : from_left;
                        const uint32_t fin = __builtin_isfinite(x) ? 0xffffffffu : 0u;
                        const float upd = pole + k.pilot_alpha * (x - pole);
                        pole = fm_pick(fin, upd, pole);
                        pole_handed = fm_pick(fin, upd, x);
                        res[i0 + j] = pole_handed;  // second poles: the value of sample i0 + j - 1
                    }
                }
            }
        } else if (wave == 2) {
            // ---- F: chunk s-4; lanes 0-3 sum path (notch, lp0, lp1, lp2), lanes 4-7 difference path ----
            if (s >= 4 && s - 4 < chunks) {
                const uint32_t c = s - 4, cnt = count_of(c);
                const float* src = diff_path ? r_xd[c & 1] : r_d[c & 7];
                float* res = lid == 3 ? r_sum[c & 1] : (lid == 7 ? r_diff[c & 1] : spill);
                for (uint32_t i0 = 0; i0 < cnt + 3; i0 += 8) {  // +3: flush
                    float xin[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xin[j] = src[i0 + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float from_left = dpp_row_shr1(bq_handed);
                        const float x = stage == 0 ? xin[j] : from_left;
                        const uint32_t fin = __builtin_isfinite(x) ? 0xffffffffu : 0u;
                        const float y = c5[0] * x + s0;
                        const float n0 = c5[1] * x - c5[3] * y + s1;
                        const float n1 = c5[2] * x - c5[4] * y;
                        s0 = fm_pick(fin, n0, s0);
                        s1 = fm_pick(fin, n1, s1);
                        bq_handed = fm_pick(fin, y, x);
                        res[i0 + j] = bq_handed;  // last stages: the value of sample i0 + j - 3
                    }
                }
            }
        } else if (wave == 3) {
            // ---- G: chunk s-6: de-emphasis of left (lane 0) and right (lane 1), in place ----
            if (k.deemph_enabled && s >= 6 && s - 6 < chunks && lid < 2) {
                const uint32_t c = s - 6, cnt = count_of(c);
                float* src = lid == 0 ? r_left[c % 3] : r_right[c % 3];
                for (uint32_t i0 = 0; i0 < cnt; i0 += 8) {  // past the chunk and for non-finite d: bubbles
                    float xin[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xin[j] = src[i0 + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float x = xin[j];
                        const uint32_t fin = __builtin_isfinite(x) ? 0xffffffffu : 0u;
                        const float upd = de + k.deemph_alpha * (x - de);
                        de = fm_pick(fin, upd, de);
                        src[i0 + j] = fm_pick(fin, upd, x);
                    }
                }
            }
        } else {
            const uint32_t et = tid - 256;  // 256 element-wise threads
            // ---- A: chunk s ----
            if (s < chunks) {
                const uint32_t cnt = count_of(s);
                float* dst = r_d[s & 7];
                for (uint32_t i = et; i < kFmSlot; i += 256) {
                    float d = nan;  // past the chunk: bubbles (flush + prefetch slack of the serial stages)
                    if (i < cnt) {
                        const uint64_t n = (uint64_t)s * kFmCh + i;
                        const float2 cur = in_at(n);
                        const float2 prev = n ? in_at(n - 1) : mk(st.prev_re, st.prev_im);
                        d = fm_discriminate(prev, cur, n ? true : st.has_prev != 0, k.ref);
                    }
                    dst[i] = d;
                }
            }
            // ---- C: chunk s-1 ----
            if (s >= 1 && s - 1 < chunks) {
                const uint32_t c = s - 1, cnt = count_of(c);
                const float* dch = r_d[c & 7];
                const float* pch = r_phase[c & 3];
                for (uint32_t i = et; i < kFmSlot; i += 256) {
                    const float d = i < cnt ? dch[i] : nan;
                    const bool fin = __builtin_isfinite(d);
                    const float p = fin ? pch[i] : 0.0f;
                    r_xc[c & 1][i] = fin ? d * libm_cosf(p) : nan;  // non-finite discriminator sample = bubble
                    r_xs[c & 1][i] = fin ? d * libm_sinf(p) : nan;
                }
            }
            // ---- E: chunk s-3 ----
            if (s >= 3 && s - 3 < chunks) {
                const uint32_t c = s - 3, cnt = count_of(c);
                const float* dch = r_d[c & 7];
                const float* pch = r_phase[c & 3];
                for (uint32_t i = et; i < kFmSlot; i += 256) {
                    const float d = i < cnt ? dch[i] : nan;
                    float v = nan;
                    if (__builtin_isfinite(d)) {  // D's results sit 1 slot late
                        const float po = libm_atan2f(r_pcos[c & 1][i + 1], r_psin[c & 1][i + 1]);
                        const float carrier = libm_sinf(2.0f * (pch[i] + po));
                        v = 2.0f * d * carrier;
                    }
                    r_xd[c & 1][i] = v;
                }
            }
            // ---- L: chunk s-5: left = sum + diff, right = sum - diff (F's results sit 3 slots late) ----
            if (s >= 5 && s - 5 < chunks) {
                const uint32_t c = s - 5, cnt = count_of(c);
                const float* dch = r_d[c & 7];
                for (uint32_t i = et; i < kFmSlot; i += 256) {
                    const bool fin = i < cnt && __builtin_isfinite(dch[i]);
                    const float a = fin ? r_sum[c & 1][i + 3] : nan, b = fin ? r_diff[c & 1][i + 3] : nan;
                    r_left[c % 3][i] = a + b;
                    r_right[c % 3][i] = a - b;
                }
            }
            // ---- O: chunk s-7: the output ----
            if (s >= 7 && s - 7 < chunks) {
                const uint32_t c = s - 7, cnt = count_of(c);
                const float* dch = r_d[c & 7];
                for (uint32_t i = et; i < cnt; i += 256) {
                    const uint64_t n = (uint64_t)c * kFmCh + i;
                    const float d = dch[i];
                    const bool fin = __builtin_isfinite(d);
                    const int64_t oo = out_off + (int64_t)(n / L.samples) * L.out_batch_stride +
                                       (int64_t)(n % L.samples) * L.out_sample_stride;
                    out[oo] = fin ? r_left[c % 3][i] : d;
                    out[oo + L.out_channel_stride] = fin ? r_right[c % 3][i] : d;
                }
            }
        }
        __syncthreads();
    }
    // ---- state ----
    if (wave == 0 && lid == 0) {
        const float2 last = in_at(n_total - 1);
        states[lane].prev_re = last.x;
        states[lane].prev_im = last.y;
        states[lane].has_prev = 1;
        states[lane].pilot_phase = ph;
    }
    if (wave == 1) {
        if (lid == 0) states[lane].pilot_cos_stage = pole;
        if (lid == 1) states[lane].pilot_cos = pole;
        if (lid == 2) states[lane].pilot_sin_stage = pole;
        if (lid == 3) states[lane].pilot_sin = pole;
    }
    if (wave == 2 && lid < 8) {
        float* ss = stage == 0 ? (diff_path ? states[lane].diff_notch : states[lane].sum_notch)
                               : (diff_path ? states[lane].diff_filter[stage - 1] : states[lane].sum_filter[stage - 1]);
        ss[0] = s0;
        ss[1] = s1;
    }
    if (wave == 3 && k.deemph_enabled) {
        if (lid == 0) states[lane].left_de = de;
        if (lid == 1) states[lane].right_de = de;
    }
}

// Narrow FM with de-emphasis: the discriminator is element-wise, the de-emphasis one recurrence
// (narrow_deemph += alpha * (d - narrow_deemph), fm/module_impl_native_cpu.cc:114-120).  Same split as
// the wide decoder: all threads fill an LDS chunk with d, one lane walks it, all threads store.
__global__ __launch_bounds__(kFmThreads) void fm_narrow_deemph_kernel(float* __restrict__ out,
                                                                       const float2* __restrict__ in,
                                                                       FmState* __restrict__ states,
                                                                       const FmCoeffs k, const FmLayout L) {
    __shared__ float in_a[kFmChunk + 8], out_a[kFmChunk + 8];
    const uint32_t tid = threadIdx.x;
    const uint64_t lane = blockIdx.x, n_total = L.batches * L.samples;
    int64_t in_off, out_off;
    fm_lane_offsets(L, lane, in_off, out_off);
    const FmState st = states[lane];
    auto in_at = [&](uint64_t n) {
        return in[in_off + (int64_t)(n / L.samples) * L.in_batch_stride + (int64_t)(n % L.samples) * L.in_sample_stride];
    };
    float de = st.narrow_deemph;
    for (uint64_t c0 = 0; c0 < n_total; c0 += kFmChunk) {
        const uint32_t cnt = (uint32_t)((n_total - c0) < (uint64_t)kFmChunk ? (n_total - c0) : (uint64_t)kFmChunk);
        __syncthreads();
        for (uint32_t i = tid; i < cnt + 8; i += kFmThreads) {
            const uint64_t n = c0 + i;
            float d = __builtin_nanf("");
            if (i < cnt) {
                const float2 cur = in_at(n);
                const float2 prev = n ? in_at(n - 1) : mk(st.prev_re, st.prev_im);
                d = fm_discriminate(prev, cur, n ? true : st.has_prev != 0, k.ref);
            }
            in_a[i] = d;
        }
        __syncthreads();
        if (tid == 0) {
            for (uint32_t i0 = 0; i0 < cnt; i0 += 8) {
                float xin[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xin[j] = in_a[i0 + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = xin[j];
                    const uint32_t fin = __builtin_isfinite(x) ? 0xffffffffu : 0u;
                    const float upd = de + k.deemph_alpha * (x - de);
                    de = fm_pick(fin, upd, de);
                    out_a[i0 + j] = fm_pick(fin, upd, x);  // a non-finite sample goes out as it is
                }
            }
        }
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += kFmThreads) {
            const uint64_t n = c0 + i;
            out[out_off + (int64_t)(n / L.samples) * L.out_batch_stride + (int64_t)(n % L.samples) * L.out_sample_stride] =
                out_a[i];
        }
    }
    if (tid == 0) {
        const float2 last = in_at(n_total - 1);
        states[lane].prev_re = last.x;
        states[lane].prev_im = last.y;
        states[lane].has_prev = 1;
        states[lane].narrow_deemph = de;
    }
}

// ---- AM (dsp/am/module_impl_native_cpu.cc:40-101): envelope + one-pole DC blocker --------------
//     e[n] = |x[n]|  (std::abs of a complex<float> = libm hypotf);  y[n] = e[n] - e[n-1] + alpha*y[n-1]
// per lane, batches walked as one sequence.  The reference evaluates (e[n] - e[n-1]) + (alpha*y[n-1])
// left to right in F32 with no contraction, so the differences are computed by all threads (each
// is ONE rounding of two exactly known envelopes) and only the two-operation recurrence is walked
// by one thread per lane, out of LDS -- the same split as the narrow-FM de-emphasis kernel above.
// A lone wavefront issues one VALU instruction every ~3.4 ns on this part whatever its dependencies, so the walk
// costs its instruction count: mul + add per sample plus one exposed LDS wait per 16 = 9.8 ns (102 MS/s per lane).
struct AmState {
    float prev_envelope, prev_output;
};
__device__ __forceinline__ float am_envelope(float2 v) {
    if (__builtin_isinf(v.x) || __builtin_isinf(v.y)) return __builtin_inff();
    return (float)__builtin_sqrt((double)v.x * (double)v.x + (double)v.y * (double)v.y);
}
// Pipeline over chunks of kFmChunk samples, one workgroup barrier per step: in step s wave 0 walks the
// recurrence of chunk s (LDS -> LDS) while waves 1..3 write chunk s-1 to HBM and stage the differences of
// chunk s+1 (each envelope is computed once; a lane gets its left neighbour's by DPP-free shuffle).
__global__ __launch_bounds__(kFmThreads) void am_kernel(float* __restrict__ out, const float2* __restrict__ in,
                                                        AmState* __restrict__ states, const float alpha,
                                                        const FmLayout L) {
    __shared__ __attribute__((aligned(16))) float diff_a[2][kFmChunk + 32];
    __shared__ __attribute__((aligned(16))) float out_a[2][kFmChunk + 32];
    constexpr uint32_t kStagers = kFmThreads - 64;
    const uint32_t tid = threadIdx.x;
    const uint64_t lane = blockIdx.x, n_total = L.batches * L.samples;
    if (n_total == 0) return;
    int64_t in_off, out_off;
    fm_lane_offsets(L, lane, in_off, out_off);
    const AmState st = states[lane];
    auto env_at = [&](uint64_t n) {
        return am_envelope(
            in[in_off + (int64_t)(n / L.samples) * L.in_batch_stride + (int64_t)(n % L.samples) * L.in_sample_stride]);
    };
    const uint64_t chunks = (n_total + kFmChunk - 1) / kFmChunk;
    auto count = [&](uint64_t c) {
        const uint64_t left = n_total - c * kFmChunk;
        return (uint32_t)(left < (uint64_t)kFmChunk ? left : (uint64_t)kFmChunk);
    };
    // differences of chunk c by `width` threads numbered `t` (whole wavefronts).  All of a thread's samples are
    // requested before the first is used: one HBM round trip per chunk instead of one per sample.
    auto stage = [&](uint64_t c, uint32_t t, auto width_c) {
        constexpr uint32_t width = decltype(width_c)::value;
        constexpr uint32_t kSlack = 24;  // the walk reads two groups ahead of the last sample
        constexpr uint32_t kIter = (kFmChunk + kSlack + 63 + width - 1) / width;
        const uint32_t cnt = count(c);
        const uint32_t limit = (cnt + kSlack + 63) & ~63u;  // whole waves take the shuffle
        float* dst = diff_a[c & 1];
        auto at = [&](uint64_t n) {
            return in[in_off + (int64_t)(n / L.samples) * L.in_batch_stride + (int64_t)(n % L.samples) * L.in_sample_stride];
        };
        float2 v[kIter], edge[kIter];
#pragma unroll
        for (uint32_t k = 0; k < kIter; ++k) {
            const uint32_t i = t + k * width;
            const uint64_t n = c * kFmChunk + i;
            v[k] = i < cnt ? at(n) : mk(0.0f, 0.0f);
            edge[k] = ((i & 63u) == 0u && i < cnt && n) ? at(n - 1) : mk(0.0f, 0.0f);
        }
#pragma unroll
        for (uint32_t k = 0; k < kIter; ++k) {
            const uint32_t i = t + k * width;
            if (i >= limit) break;  // wave-uniform
            const float e = i < cnt ? am_envelope(v[k]) : 0.0f;
            float left = __shfl_up(e, 1);
            if ((i & 63u) == 0u && i < cnt) left = (c * kFmChunk + i) ? am_envelope(edge[k]) : st.prev_envelope;
            if (i < cnt + kSlack) dst[i] = i < cnt ? e - left : 0.0f;
        }
    };
    auto flush = [&](uint64_t c, uint32_t t, uint32_t width) {
        const uint32_t cnt = count(c);
        const float* src = out_a[c & 1] + 8;
        for (uint32_t i = t; i < cnt; i += width) {
            const uint64_t n = c * kFmChunk + i;
            out[out_off + (int64_t)(n / L.samples) * L.out_batch_stride + (int64_t)(n % L.samples) * L.out_sample_stride] =
                src[i];
        }
    };
    stage(0, tid, std::integral_constant<uint32_t, (uint32_t)kFmThreads>{});
    __syncthreads();
    float y = st.prev_output;
    for (uint64_t c = 0; c < chunks; ++c) {
        if (tid >= 64) {
            if (c) flush(c - 1, tid - 64, kStagers);
            if (c + 1 < chunks) stage(c + 1, tid - 64, std::integral_constant<uint32_t, kStagers>{});
        } else if (tid == 0) {
            const uint32_t cnt = count(c);
            const float* src = diff_a[c & 1];
            float* dst = out_a[c & 1];  // results start at dst[8]: the first iteration stores a dummy group in front
            // Two groups of 8 per iteration in registers of their own (no rotation moves): LDS traffic first --
            // the next group's loads and the PREVIOUS group's stores -- so that both round trips run under
            // the 16 dependent operations of the group being walked instead of in front of them.
            auto ld = [&](uint32_t at, float (&d)[8]) {
                const float4 lo = *reinterpret_cast<const float4*>(&src[at]);
                const float4 hi = *reinterpret_cast<const float4*>(&src[at + 4]);
                d[0] = lo.x, d[1] = lo.y, d[2] = lo.z, d[3] = lo.w, d[4] = hi.x, d[5] = hi.y, d[6] = hi.z, d[7] = hi.w;
            };
            auto st8 = [&](uint32_t at, const float (&d)[8]) {
                *reinterpret_cast<float4*>(&dst[at]) = make_float4(d[0], d[1], d[2], d[3]);
                *reinterpret_cast<float4*>(&dst[at + 4]) = make_float4(d[4], d[5], d[6], d[7]);
            };
            auto walk = [&](const float (&d)[8], float (&r)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = y = d[j] + alpha * y;
            };
            float a[8], b[8], ra[8], rb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            ld(0, a);
            for (uint32_t i0 = 0; i0 < cnt; i0 += 16) {
                ld(i0 + 8, b);
                st8(i0, rb);  // samples i0-8 .. i0-1 (a dummy group the first time)
                __builtin_amdgcn_sched_barrier(0);
                walk(a, ra);
                __builtin_amdgcn_sched_barrier(0);
                ld(i0 + 16, a);
                st8(i0 + 8, ra);
                __builtin_amdgcn_sched_barrier(0);
                walk(b, rb);
                __builtin_amdgcn_sched_barrier(0);
            }
            st8((cnt + 15u) & ~15u, rb);
            y = dst[8 + cnt - 1];  // the last iteration ran past cnt on zeros
        }
        __syncthreads();
    }
    flush(chunks - 1, tid, kFmThreads);
    if (tid == 0) {
        states[lane].prev_envelope = env_at(n_total - 1);
        states[lane].prev_output = y;
    }
}

// One launch: the thread that takes a lane's FIRST sample is the only reader of that lane's state, so it is also the one
// that replaces it with the lane's LAST sample (read, then written, in one thread's program order; the input is not
// written by this kernel) -- no second kernel to order the state update behind the readers.
__global__ __launch_bounds__(kBlock) void fm_narrow_parallel_kernel(
    float* __restrict__ out, const float2* __restrict__ in, FmState* states, const FmCoeffs k, const FmLayout L) {
    const uint64_t per_lane = L.batches * L.samples, total = L.lanes * per_lane;
    JST_GRID_STRIDE(e, total) {
        const uint64_t lane = e / per_lane, n = e % per_lane, b = n / L.samples, s = n % L.samples;
        int64_t in_off, out_off;
        fm_lane_offsets(L, lane, in_off, out_off);
        const float2 cur =
            in[in_off + (int64_t)b * L.in_batch_stride + (int64_t)s * L.in_sample_stride];
        float2 prev;
        bool has = true;
        if (n == 0) {
            prev = mk(states[lane].prev_re, states[lane].prev_im);
            has = states[lane].has_prev != 0;
            const float2 last = in[in_off + (int64_t)(L.batches - 1) * L.in_batch_stride +
                                   (int64_t)(L.samples - 1) * L.in_sample_stride];
            states[lane].prev_re = last.x;
            states[lane].prev_im = last.y;
            states[lane].has_prev = 1;
        } else {
            const uint64_t pb = (n - 1) / L.samples, ps = (n - 1) % L.samples;
            prev = in[in_off + (int64_t)pb * L.in_batch_stride + (int64_t)ps * L.in_sample_stride];
        }
        out[out_off + (int64_t)b * L.out_batch_stride + (int64_t)s * L.out_sample_stride] =
            fm_discriminate(prev, cur, has, k.ref);
    }
}

// ---- SignalGenerator, cosine / sine (dsp/signal_generator/module_impl_native_cpu.cc:20-23,
// 159-163, 200-231).  The phase is a serial F64 recurrence with an fmod wrap per sample; to keep
// every bit, ONE thread walks it (it is the synthetic INPUT generator of the benchmark configs,
// not a hot kernel), then all threads evaluate cos/sin in parallel.
__global__ void siggen_phase_kernel(double* __restrict__ phases, double* __restrict__ state,
                                    uint64_t count, double frequency, double sample_rate) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double period = 2.0 * 3.14159265358979323846;
    const double step = 2.0 * 3.14159265358979323846 * frequency / sample_rate;
    double ph = state[0];
    uint64_t i = 0;
    // fmod is exact, so for 0 <= ph < period and |step| < period the wrapped sum is one of
    // x, x - period (Sterbenz) or x + period: a short select chain instead of a libm call.
    // Anything outside that window (first sample of a non-canonical phase, steps beyond one
    // turn, non-finite values) takes the general path below.
    // Chunks of 32 samples run branch-free (the loop-carried chain is add, compare, subtract,
    // select); the window condition is accumulated and checked once per chunk, and a chunk that
    // left the window is redone by the general loop from its first sample.
    const bool up = step >= 0.0 && step < period, down = step < 0.0 && step > -period;
    while (i < count) {
        if ((up || down) && i + 32 <= count && ph >= 0.0 && ph < period) {
            double q = ph;
            bool ok = true;
            double buf[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                buf[k] = q;
                const double x = q + step;
                if (up) {
                    ok = ok && (x < 2.0 * period);
                    q = x < period ? x : x - period;
                } else {
                    q = x < 0.0 ? x + period : x;
                    ok = ok && (q < period);
                }
            }
            if (ok) {
#pragma unroll
                for (int k = 0; k < 32; ++k) phases[i + k] = buf[k];
                ph = q;
                i += 32;
                continue;
            }
        }
        const uint64_t stop = (count - i < 32) ? count : i + 32;  // general path for this chunk
        for (; i < stop; ++i) {
            phases[i] = ph;
            const double w = fmod(ph + step, period);
            ph = w < 0.0 ? w + period : w;
        }
    }
    state[0] = ph;
}
// advanceChirpPhase (:165-190): per sample the phase grows by 2*pi*cycles, cycles = the integral of
// the linear frequency sweep over dt, restarting at chirpDuration.  state = {phase, chirpTime}.
__global__ void siggen_chirp_phase_kernel(double* __restrict__ phases, double* __restrict__ state,
                                          uint64_t count, double sample_rate, double f0, double f1,
                                          double duration) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const double period = 2.0 * 3.14159265358979323846;
    const double dt = 1.0 / sample_rate;
    const double rate = (f1 - f0) / duration;
    double ph = state[0], tm = state[1];
    for (uint64_t i = 0; i < count; ++i) {
        phases[i] = ph;
        double cycles = 0.0;
        const double until = duration - tm;
        if (dt < until) {
            cycles = (f0 + rate * tm) * dt + 0.5 * rate * dt * dt;
            tm += dt;
        } else {
            cycles = (f0 + rate * tm) * until + 0.5 * rate * until * until;
            const double after = dt - until;
            tm = after;
            if (after > 0.0) cycles += (f0 + rate * 0.0) * after + 0.5 * rate * after * after;
        }
        const double w = fmod(ph + 2.0 * 3.14159265358979323846 * cycles, period);
        ph = w < 0.0 ? w + period : w;
    }
    state[0] = ph;
    state[1] = tm;
}

// waveform shapes of module_impl_native_cpu.cc:192-375, evaluated in F64 from the stored phase
enum : int { kSine = 0, kCosine = 1, kSquare = 2, kTriangle = 3, kSawtooth = 4, kDc = 5 };
__global__ __launch_bounds__(kBlock) void siggen_eval_kernel(float* __restrict__ out,
                                                             const double* __restrict__ phases,
                                                             uint64_t count, int complex_out,
                                                             int shape, double amplitude,
                                                             double dc_offset) {
    const double pi = 3.14159265358979323846;
    JST_GRID_STRIDE(i, count) {
        const double ph = (shape == kDc) ? 0.0 : phases[i];
        double re, im = 0.0;
        if (shape == kCosine) {          // :213-231
            re = amplitude * cos(ph) + dc_offset;
            im = amplitude * sin(ph);
        } else if (shape == kSine) {     // :192-211
            re = amplitude * sin(ph) + dc_offset;
            im = -amplitude * cos(ph);
        } else if (shape == kSquare) {   // :233-253
            re = amplitude * (ph < pi ? 1.0 : -1.0) + dc_offset;
        } else if (shape == kSawtooth) { // :255-275
            const double pv = ph / (2.0 * pi);
            re = amplitude * (2.0 * pv - 1.0) + dc_offset;
        } else if (shape == kTriangle) { // :277-299
            const double pv = ph / (2.0 * pi);
            re = amplitude * (pv < 0.5 ? 4.0 * pv - 1.0 : 3.0 - 4.0 * pv) + dc_offset;
        } else {                         // dc :329-348
            re = amplitude + dc_offset;
        }
        if (complex_out) {
            out[2 * i] = (float)re;
            out[2 * i + 1] = (float)im;
        } else {
            out[i] = (float)re;
        }
    }
}

// Noise (:301-327): clamp(scale * N(0,1) + dc).  The reference seeds std::mt19937 from
// std::random_device, so only the distribution is specified; here a counter-based generator
// (SplitMix64 over sample index + call counter) feeds Box-Muller in F64, one pair per sample.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(kBlock) void siggen_noise_kernel(float* __restrict__ out,
                                                              const uint64_t* __restrict__ state,
                                                              uint64_t count, int complex_out,
                                                              double scale, double variance,
                                                              double dc_offset) {
    const double fmax = 3.40282346638528859811704183484516925e+38;
    const uint64_t base = state[0];
    JST_GRID_STRIDE(i, count) {
        double n0 = 0.0, n1 = 0.0;
        if (variance > 0.0) {
            const uint64_t a = splitmix64(base + 2 * i), b = splitmix64(base + 2 * i + 1);
            const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740992.0);  // (0, 1]
            const double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);         // [0, 1)
            const double r = sqrt(-2.0 * log(u1));
            n0 = r * cos(2.0 * 3.14159265358979323846 * u2);
            n1 = r * sin(2.0 * 3.14159265358979323846 * u2);
        }
        double iv = scale * n0 + dc_offset, qv = scale * n1;
        iv = iv < -fmax ? -fmax : (fmax < iv ? fmax : iv);
        qv = qv < -fmax ? -fmax : (fmax < qv ? fmax : qv);
        if (complex_out) {
            out[2 * i] = (float)iv;
            out[2 * i + 1] = (float)qv;
        } else {
            out[i] = (float)iv;
        }
    }
}
__global__ void siggen_noise_advance_kernel(uint64_t* state, uint64_t count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) state[0] += 2 * count;
}

}  // namespace

hipError_t launch_signal_generator(float* out, double* phases, double* state, uint64_t count,
                                   bool complex_out, const SignalParams& p, hipStream_t s) {
    (void)hipGetLastError();
    const int cx = complex_out ? 1 : 0;
    switch (p.shape) {
        case SignalShape::Noise:
            hipLaunchKernelGGL(siggen_noise_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, out,
                               reinterpret_cast<const uint64_t*>(state + 2), count, cx,
                               p.amplitude * sqrt(p.noise_variance), p.noise_variance, p.dc_offset);
            hipLaunchKernelGGL(siggen_noise_advance_kernel, dim3(1), dim3(64), 0, s,
                               reinterpret_cast<uint64_t*>(state + 2), count);
            return hipGetLastError();
        case SignalShape::Dc:
            hipLaunchKernelGGL(siggen_eval_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, out,
                               (const double*)phases, count, cx, (int)kDc, p.amplitude, p.dc_offset);
            return hipGetLastError();
        case SignalShape::Chirp:
            hipLaunchKernelGGL(siggen_chirp_phase_kernel, dim3(1), dim3(64), 0, s, phases, state, count,
                               p.sample_rate, p.chirp_start, p.chirp_end, p.chirp_duration);
            hipLaunchKernelGGL(siggen_eval_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, out,
                               (const double*)phases, count, cx, (int)kCosine, p.amplitude, p.dc_offset);
            return hipGetLastError();
        default: break;
    }
    int shape = kCosine;
    if (p.shape == SignalShape::Sine) shape = kSine;
    if (p.shape == SignalShape::Square) shape = kSquare;
    if (p.shape == SignalShape::Triangle) shape = kTriangle;
    if (p.shape == SignalShape::Sawtooth) shape = kSawtooth;
    hipLaunchKernelGGL(siggen_phase_kernel, dim3(1), dim3(64), 0, s, phases, state, count, p.frequency,
                       p.sample_rate);
    hipLaunchKernelGGL(siggen_eval_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, out,
                       (const double*)phases, count, cx, shape, p.amplitude, p.dc_offset);
    return hipGetLastError();
}

size_t fm_state_bytes() { return sizeof(FmState); }

hipError_t launch_pad(void* out, const void* in, bool complex, uint64_t outer, uint64_t in_axis,
                      uint64_t out_axis, uint64_t inner, hipStream_t s) {
    const uint64_t total = outer * out_axis * inner;
    (void)hipGetLastError();
    if (complex)
        hipLaunchKernelGGL(pad_kernel<float2>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                           (float2*)out, (const float2*)in, outer, in_axis, out_axis, inner);
    else
        hipLaunchKernelGGL(pad_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                           (float*)out, (const float*)in, outer, in_axis, out_axis, inner);
    return hipGetLastError();
}
hipError_t launch_unpad(void* body, void* tail, const void* in, bool complex, uint64_t outer,
                        uint64_t in_axis, uint64_t body_axis, uint64_t inner, hipStream_t s) {
    const uint64_t total = outer * in_axis * inner;
    (void)hipGetLastError();
    if (complex)
        hipLaunchKernelGGL(unpad_kernel<float2>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                           (float2*)body, (float2*)tail, (const float2*)in, outer, in_axis,
                           body_axis, inner);
    else
        hipLaunchKernelGGL(unpad_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                           (float*)body, (float*)tail, (const float*)in, outer, in_axis, body_axis,
                           inner);
    return hipGetLastError();
}
hipError_t launch_fold(float* out, const float* in, bool complex, uint64_t outer, uint64_t axis_size,
                       uint64_t fold_size, uint64_t inner, uint64_t scalar_offset,
                       const uint64_t* chan_offsets, uint64_t chan_count, uint64_t chan_inner,
                       hipStream_t s) {
    const uint64_t total = outer * fold_size * inner;
    (void)hipGetLastError();
    if (complex)
        hipLaunchKernelGGL(fold_kernel<true>, dim3(grid_for(total)), dim3(kBlock), 0, s, out, in,
                           outer, axis_size, fold_size, inner, scalar_offset, chan_offsets,
                           chan_count, chan_inner);
    else
        hipLaunchKernelGGL(fold_kernel<false>, dim3(grid_for(total)), dim3(kBlock), 0, s, out, in,
                           outer, axis_size, fold_size, inner, scalar_offset, chan_offsets,
                           chan_count, chan_inner);
    return hipGetLastError();
}
hipError_t launch_fold_product_cf32(float2* out, const float2* a, const float2* b, const EwLayout& P,
                                    uint64_t axis_size, uint64_t fold_size, uint64_t scalar_offset,
                                    const uint64_t* chan_offsets, uint64_t chan_count,
                                    uint64_t chan_inner, hipStream_t s) {
    if (P.rank < 1 || axis_size == 0 || fold_size == 0) return hipErrorInvalidValue;
    const uint64_t total = (P.size / axis_size) * fold_size;
    (void)hipGetLastError();
    hipLaunchKernelGGL(fold_product_kernel, dim3(grid_for(total)), dim3(kBlock), 0, s, out, a, b, P,
                       axis_size, fold_size, scalar_offset, chan_offsets, chan_count, chan_inner);
    return hipGetLastError();
}
hipError_t launch_overlap_add(void* out, const void* buf, const void* ovl, void* prev, bool complex,
                              uint32_t rank, int32_t batch_axis, const uint64_t* buf_shape,
                              const uint64_t* ovl_shape, hipStream_t s) {
    OlaLayout L{};
    L.rank = rank;
    L.batch_axis = batch_axis;
    uint64_t total = 1, prev_total = 1;
    for (uint32_t d = 0; d < rank; ++d) {
        L.buf_shape[d] = buf_shape[d];
        L.ovl_shape[d] = ovl_shape[d];
        total *= buf_shape[d];
        prev_total *= ((int32_t)d == batch_axis) ? 1 : ovl_shape[d];
    }
    (void)hipGetLastError();
    if (complex) {
        hipLaunchKernelGGL(overlap_add_kernel<float2>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                           (float2*)out, (const float2*)buf, (const float2*)ovl,
                           (const float2*)prev, L, total);
        hipLaunchKernelGGL(overlap_state_kernel<float2>, dim3(grid_for(prev_total)), dim3(kBlock),
                           0, s, (float2*)prev, (const float2*)ovl, L, prev_total);
    } else {
        hipLaunchKernelGGL(overlap_add_kernel<float>, dim3(grid_for(total)), dim3(kBlock), 0, s,
                           (float*)out, (const float*)buf, (const float*)ovl, (const float*)prev, L,
                           total);
        hipLaunchKernelGGL(overlap_state_kernel<float>, dim3(grid_for(prev_total)), dim3(kBlock), 0,
                           s, (float*)prev, (const float*)ovl, L, prev_total);
    }
    return hipGetLastError();
}
// the table of the coming cycle from the state as it stands (no advance): primes the fused Filter tail at plan time
hipError_t launch_phase_table_prime(float2* corr, double* phases, const double* increments, uint64_t channels,
                                    uint64_t batches, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(phase_table_kernel, dim3((unsigned)((channels + 63) / 64)), dim3(64), 0, s, corr, phases,
                       increments, channels, batches, false);
    return hipGetLastError();
}
hipError_t launch_overlap_heads(void* out, const void* ovl, void* prev, bool complex, uint32_t rank,
                                int32_t batch_axis, const uint64_t* buf_shape, const uint64_t* ovl_shape,
                                hipStream_t s) {
    return launch_overlap_heads_phase(out, ovl, prev, complex, rank, batch_axis, buf_shape, ovl_shape, nullptr, nullptr,
                                      nullptr, 0, 0, s);
}
hipError_t launch_overlap_heads_phase(void* out, const void* ovl, void* prev, bool complex, uint32_t rank,
                                      int32_t batch_axis, const uint64_t* buf_shape, const uint64_t* ovl_shape,
                                      float2* corr, double* phases, const double* increments, uint64_t channels,
                                      uint64_t batches, hipStream_t s) {
    const PhaseNext pn{corr, phases, increments, channels, batches};
    OlaLayout L{};
    L.rank = rank;
    L.batch_axis = batch_axis;
    uint64_t total = 1;
    for (uint32_t d = 0; d < rank; ++d) {
        L.buf_shape[d] = buf_shape[d];
        L.ovl_shape[d] = ovl_shape[d];
        total *= ovl_shape[d];
    }
    (void)hipGetLastError();
    if (total == 0 && corr == nullptr) return hipSuccess;
    const unsigned grid = total ? grid_for(total) : 1u;
    if (complex)
        hipLaunchKernelGGL(overlap_heads_kernel<float2>, dim3(grid), dim3(kBlock), 0, s, (float2*)out,
                           (const float2*)ovl, (float2*)prev, L, total, pn);
    else
        hipLaunchKernelGGL(overlap_heads_kernel<float>, dim3(grid), dim3(kBlock), 0, s, (float*)out,
                           (const float*)ovl, (float*)prev, L, total, pn);
    return hipGetLastError();
}
hipError_t launch_phase_correction(const EwLayout& L, float2* out, const float2* in, float2* corr,
                                   double* phases, const double* increments, uint64_t batches,
                                   uint64_t batch_inner, uint64_t channels, uint64_t channel_inner,
                                   hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(phase_table_kernel, dim3((unsigned)((channels + 63) / 64)), dim3(64), 0, s,
                       corr, phases, increments, channels, batches, true);
    hipLaunchKernelGGL(phase_mul_kernel, dim3(grid_for(L.size)), dim3(kBlock), 0, s, L, out, in,
                       (const float2*)corr, batches, batch_inner, channels, channel_inner);
    return hipGetLastError();
}
hipError_t launch_filter_taps(float2* out, double sample_rate, double bandwidth, const double* center,
                              uint64_t heads, uint64_t taps, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(filter_taps_kernel, dim3((unsigned)((heads * taps + 255) / 256)), dim3(256),
                       0, s, out, sample_rate, bandwidth, center, heads, taps);
    return hipGetLastError();
}
hipError_t launch_arithmetic(const EwLayout& L, void* out, const void* in, bool complex, int op,
                             uint64_t r, int64_t r_stride, hipStream_t s) {
    (void)hipGetLastError();
#define JST_ARITH(T, OP)                                                                       \
    hipLaunchKernelGGL((arithmetic_kernel<T, OP>), dim3(grid_for(L.size)), dim3(kBlock), 0, s, \
                       L, (T*)out, (const T*)in, r, r_stride)
    if (complex) {
        if (op == 0) JST_ARITH(float2, 0);
        else if (op == 1) JST_ARITH(float2, 1);
        else if (op == 2) JST_ARITH(float2, 2);
        else return hipErrorInvalidValue;
    } else {
        if (op == 0) JST_ARITH(float, 0);
        else if (op == 1) JST_ARITH(float, 1);
        else if (op == 2) JST_ARITH(float, 2);
        else JST_ARITH(float, 3);
    }
#undef JST_ARITH
    return hipGetLastError();
}
size_t am_state_bytes() { return sizeof(AmState); }
hipError_t launch_am(float* out, const float2* in, void* states, float alpha, const FmLayout& L, hipStream_t s) {
    (void)hipGetLastError();
    if (L.lanes == 0 || L.batches * L.samples == 0) return hipSuccess;
    hipLaunchKernelGGL(am_kernel, dim3((unsigned)L.lanes), dim3(kFmThreads), 0, s, out, in, (AmState*)states, alpha,
                       L);
    return hipGetLastError();
}
hipError_t launch_fm(float* out, const float2* in, void* states, const FmCoeffs& k, const FmLayout& L,
                     hipStream_t s) {
    (void)hipGetLastError();
    if (!k.wide && k.deemph_enabled && L.batches * L.samples > 0 && !jst::switch_value(jst::SW_FM_SERIAL)) {
        hipLaunchKernelGGL(fm_narrow_deemph_kernel, dim3((unsigned)L.lanes), dim3(kFmThreads), 0, s, out, in,
                           (FmState*)states, k, L);
        return hipGetLastError();
    }
    if (k.wide && L.batches * L.samples > 0 && !jst::switch_value(jst::SW_FM_SERIAL)) {
        hipLaunchKernelGGL(fm_wide_kernel, dim3((unsigned)L.lanes), dim3(kFmWideThreads), 0, s, out, in,
                           (FmState*)states, k, L);
        return hipGetLastError();
    }
    if (!k.wide && !k.deemph_enabled) {
        const uint64_t total = L.lanes * L.batches * L.samples;
        if (total == 0) return hipSuccess;
        hipLaunchKernelGGL(fm_narrow_parallel_kernel, dim3(grid_for(total)), dim3(kBlock), 0, s, out,
                           in, (FmState*)states, k, L);
    } else {
        hipLaunchKernelGGL(fm_kernel, dim3((unsigned)((L.lanes + 63) / 64)), dim3(64), 0, s, out, in,
                           (FmState*)states, k, L);
    }
    return hipGetLastError();
}

}  // namespace jst::kernels
