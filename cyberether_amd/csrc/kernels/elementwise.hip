// elementwise.hip -- strided N-ary elementwise kernels: Multiply, Amplitude, Range,
// MultiplyConstant, Invert, Window.  These are the module-by-module ("API-literal") forms of the
// ops that fft_lds.hh also offers fused into the FFT; same arithmetic (device_math.hh).
//
// Traversal: one thread per output element, grid-stride; flat index -> coordinates by a
// mixed-radix decode over the common shape (row-major, like AutomaticIterator's odometer,
// include/jetstream/tools/automatic_iterator.hh:207-231); dense operands skip the decode.
#include <map>
#include <mutex>
#include <utility>

#include "device_math.hh"
#include "kernels.hh"

namespace jst::kernels {

using namespace jst::dev;

namespace {

constexpr int kBlock = 256;

template <int NOPS>
__device__ __forceinline__ void decode(const EwLayout& L, uint64_t idx, int64_t (&off)[NOPS]) {
#pragma unroll
    for (int o = 0; o < NOPS; ++o) off[o] = (int64_t)L.offset[o];
    if (L.contiguous) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o) off[o] += (int64_t)idx;
        return;
    }
    for (int a = L.rank - 1; a >= 0; --a) {
        const uint64_t c = idx % L.shape[a];
        idx /= L.shape[a];
#pragma unroll
        for (int o = 0; o < NOPS; ++o) off[o] += (int64_t)c * L.stride[o][a];
    }
}

template <class TO, class TA, class F>
__global__ __launch_bounds__(kBlock) void ew_unary(const EwLayout L, TO* __restrict__ out,
                                                   const TA* __restrict__ a, const F f) {
    const uint64_t step = (uint64_t)gridDim.x * kBlock;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < L.size; idx += step) {
        int64_t off[2];
        decode<2>(L, idx, off);
        out[off[0]] = f(idx, a[off[1]]);
    }
}

template <class TO, class TA, class TB, class F>
__global__ __launch_bounds__(kBlock) void ew_binary(const EwLayout L, TO* __restrict__ out,
                                                    const TA* __restrict__ a,
                                                    const TB* __restrict__ b, const F f) {
    const uint64_t step = (uint64_t)gridDim.x * kBlock;
    for (uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x; idx < L.size; idx += step) {
        int64_t off[3];
        decode<3>(L, idx, off);
        out[off[0]] = f(idx, a[off[1]], b[off[2]]);
    }
}

inline unsigned grid_for(uint64_t size) {
    const uint64_t need = (size + kBlock - 1) / kBlock;
    const uint64_t cap = 256ull * 16ull;  // 16 workgroups of 4 waves per CU
    return (unsigned)(need < cap ? (need ? need : 1) : cap);
}

template <class TO, class TA, class F>
hipError_t run_unary(const EwLayout& L, TO* out, const TA* a, const F& f, hipStream_t s) {
    if (L.size == 0) return hipSuccess;
    (void)hipGetLastError();  // drop any stale error: only this launch is judged
    hipLaunchKernelGGL((ew_unary<TO, TA, F>), dim3(grid_for(L.size)), dim3(kBlock), 0, s, L, out, a,
                       f);
    return hipGetLastError();
}
template <class TO, class TA, class TB, class F>
hipError_t run_binary(const EwLayout& L, TO* out, const TA* a, const TB* b, const F& f,
                      hipStream_t s) {
    if (L.size == 0) return hipSuccess;
    (void)hipGetLastError();  // drop any stale error: only this launch is judged
    hipLaunchKernelGGL((ew_binary<TO, TA, TB, F>), dim3(grid_for(L.size)), dim3(kBlock), 0, s, L,
                       out, a, b, f);
    return hipGetLastError();
}

struct MulCF32 {
    __device__ float2 operator()(uint64_t, float2 a, float2 b) const { return cmul_full(a, b); }
};
struct MulF32 {
    __device__ float operator()(uint64_t, float a, float b) const { return a * b; }
};
struct AmpCF32 {
    float coeff;
    __device__ float operator()(uint64_t, float2 v) const { return amplitude_exact(v, coeff); }
};
struct AmpCF32Fast {
    float coeff;
    __device__ float operator()(uint64_t, float2 v) const { return amplitude_cf32_fast(v, coeff); }
};
struct RangeF32Fast {
    float scale, offset;
    __device__ float operator()(uint64_t, float v) const { return range_f32_fast(v, scale, offset); }
};
struct AmpF32 {
    float coeff;
    __device__ float operator()(uint64_t, float v) const { return amplitude_f32(v, coeff); }
};
struct RangeF32 {
    float scale, offset;
    __device__ float operator()(uint64_t, float v) const { return range_f32(v, scale, offset); }
};
// Amplitude -> Range in one pass (amplitude/module_impl_native_cpu.cc:73-86 then range/module_impl_native_cpu.cc:67-82): the
// F32 level is formed and consumed in a register instead of going through memory between two launches.
struct AmpRangeCF32 {
    float coeff, scale, offset;
    __device__ float operator()(uint64_t, float2 v) const { return range_f32(amplitude_exact(v, coeff), scale, offset); }
};
struct AmpRangeCF32Fast {
    float coeff, scale, offset;
    __device__ float operator()(uint64_t, float2 v) const { return range_f32_fast(amplitude_cf32_fast(v, coeff), scale, offset); }
};
struct AmpRangeF32 {
    float coeff, scale, offset;
    __device__ float operator()(uint64_t, float v) const { return range_f32(amplitude_f32(v, coeff), scale, offset); }
};
// multiply_constant/module_impl_native_cpu.cc:92-100: CF32 * F32 scalar = (re*c, im*c).
struct MulConstCF32 {
    float c;
    __device__ float2 operator()(uint64_t, float2 v) const { return mk(v.x * c, v.y * c); }
};
struct MulConstF32 {
    float c;
    __device__ float operator()(uint64_t, float v) const { return v * c; }
};
// add/module_impl_native_cpu.cc:83-98: c = a + b (complex: component-wise).
struct AddCF32 {
    __device__ float2 operator()(uint64_t, float2 a, float2 b) const { return cadd(a, b); }
};
struct AddF32 {
    __device__ float operator()(uint64_t, float a, float b) const { return a + b; }
};
// cast/module_impl_native_cpu.cc:137-283: out = static_cast<F32>(in) / scaler per component
// (the scaler is a power of two: the division is exact; the int -> float conversion rounds to
// nearest even like x86 cvtsi2ss).  No offset is removed from unsigned formats, as in the reference.
template <class TI>
struct CastRealOp {
    float s;
    __device__ float operator()(uint64_t, TI v) const { return static_cast<float>(v) / s; }
};
template <class TI>
struct IntPair {
    TI re, im;
};
template <class TI>
struct CastComplexOp {
    float s;
    __device__ float2 operator()(uint64_t, IntPair<TI> v) const {
        return mk(static_cast<float>(v.re) / s, static_cast<float>(v.im) / s);
    }
};
struct CastF32ToCF32 {
    __device__ float2 operator()(uint64_t, float v) const { return mk(v, 0.0f); }
};
struct TanhProbe {
    // the variant range_f32 uses (device_math.hh), so the sweep test covers the shipped code path
#ifdef JST_TANH_SELECT_FORM
    __device__ float operator()(uint64_t, float v) const { return libm_tanhf(v); }
#else
    __device__ float operator()(uint64_t, float v) const { return libm_tanhf_branchy(v); }
#endif
};

// invert/module_impl_native_cpu.cc:79-103.
template <class TIN>
struct InvertOp {
    uint64_t inner, length;
    __device__ float2 operator()(uint64_t index, TIN in) const {
        float2 value;
        if constexpr (sizeof(TIN) == sizeof(float2)) value = in;
        else value = mk(in, 0.0f);
        const uint64_t coord = (index / inner) % length;
        if ((length & 1ull) == 0) return (coord & 1ull) ? mk(-value.x, -value.y) : value;
        const double phase = 2.0 * 3.14159265358979323846 * (double)(length / 2) * (double)coord /
                             (double)length;
        return cmul_full(value, mk((float)cos(phase), (float)sin(phase)));
    }
};

// window/module_impl_native_cpu.cc:20-37 -- Blackman, F64 evaluation, symmetric denominator.
__global__ __launch_bounds__(kBlock) void window_kernel(float2* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (n == 1) {
        out[0] = mk(1.0f, 0.0f);
        return;
    }
    const double pi = 3.14159265358979323846;
    const double tap = 0.42 - 0.50 * cos(2.0 * pi * (double)i / (double)(n - 1)) +
                       0.08 * cos(4.0 * pi * (double)i / (double)(n - 1));
    out[i] = mk((float)tap, 0.0f);
}

}  // namespace

hipError_t launch_multiply_cf32(const EwLayout& L, float2* c, const float2* a, const float2* b,
                                hipStream_t s) {
    return run_binary(L, c, a, b, MulCF32{}, s);
}
hipError_t launch_multiply_f32(const EwLayout& L, float* c, const float* a, const float* b,
                               hipStream_t s) {
    return run_binary(L, c, a, b, MulF32{}, s);
}
hipError_t launch_amplitude_cf32(const EwLayout& L, float* out, const float2* in, float coeff,
                                 bool fast, hipStream_t s) {
    if (fast) return run_unary(L, out, in, AmpCF32Fast{coeff}, s);
    return run_unary(L, out, in, AmpCF32{coeff}, s);
}
hipError_t launch_amplitude_f32(const EwLayout& L, float* out, const float* in, float coeff,
                                hipStream_t s) {
    return run_unary(L, out, in, AmpF32{coeff}, s);
}
hipError_t launch_range_f32(const EwLayout& L, float* out, const float* in, float scale,
                            float offset, bool fast, hipStream_t s) {
    if (fast) return run_unary(L, out, in, RangeF32Fast{scale, offset}, s);
    return run_unary(L, out, in, RangeF32{scale, offset}, s);
}
hipError_t launch_amplitude_range(const EwLayout& L, float* out, const void* in, bool in_is_complex, float coeff, float scale,
                                  float offset, bool fast, hipStream_t s) {
    if (!in_is_complex) return run_unary(L, out, static_cast<const float*>(in), AmpRangeF32{coeff, scale, offset}, s);
    if (fast) return run_unary(L, out, static_cast<const float2*>(in), AmpRangeCF32Fast{coeff, scale, offset}, s);
    return run_unary(L, out, static_cast<const float2*>(in), AmpRangeCF32{coeff, scale, offset}, s);
}
hipError_t launch_multiply_constant_cf32(const EwLayout& L, float2* out, const float2* in,
                                         float constant, hipStream_t s) {
    return run_unary(L, out, in, MulConstCF32{constant}, s);
}
hipError_t launch_multiply_constant_f32(const EwLayout& L, float* out, const float* in,
                                        float constant, hipStream_t s) {
    return run_unary(L, out, in, MulConstF32{constant}, s);
}
hipError_t launch_invert(const EwLayout& L, float2* out, const void* in, bool in_is_complex,
                         uint64_t inner, uint64_t length, hipStream_t s) {
    if (in_is_complex)
        return run_unary(L, out, static_cast<const float2*>(in), InvertOp<float2>{inner, length}, s);
    return run_unary(L, out, static_cast<const float*>(in), InvertOp<float>{inner, length}, s);
}
hipError_t launch_window(float2* out, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();  // drop any stale error: only this launch is judged
    hipLaunchKernelGGL(window_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       s, out, n);
    return hipGetLastError();
}
hipError_t launch_add_cf32(const EwLayout& L, float2* c, const float2* a, const float2* b,
                           hipStream_t s) {
    return run_binary(L, c, a, b, AddCF32{}, s);
}
hipError_t launch_add_f32(const EwLayout& L, float* c, const float* a, const float* b,
                          hipStream_t s) {
    return run_binary(L, c, a, b, AddF32{}, s);
}
hipError_t launch_cast(const EwLayout& L, void* out, const void* in, CastKind kind, float scaler,
                       hipStream_t s) {
#define JST_CAST_REAL(T) \
    return run_unary(L, static_cast<float*>(out), static_cast<const T*>(in), CastRealOp<T>{scaler}, s)
#define JST_CAST_CPLX(T)                                                                       \
    return run_unary(L, static_cast<float2*>(out), static_cast<const IntPair<T>*>(in),         \
                     CastComplexOp<T>{scaler}, s)
    switch (kind) {
        case CastKind::I8: JST_CAST_REAL(int8_t);
        case CastKind::U8: JST_CAST_REAL(uint8_t);
        case CastKind::I16: JST_CAST_REAL(int16_t);
        case CastKind::U16: JST_CAST_REAL(uint16_t);
        case CastKind::I32: JST_CAST_REAL(int32_t);
        case CastKind::U32: JST_CAST_REAL(uint32_t);
        case CastKind::CI8: JST_CAST_CPLX(int8_t);
        case CastKind::CU8: JST_CAST_CPLX(uint8_t);
        case CastKind::CI16: JST_CAST_CPLX(int16_t);
        case CastKind::CU16: JST_CAST_CPLX(uint16_t);
        case CastKind::CI32: JST_CAST_CPLX(int32_t);
        case CastKind::CU32: JST_CAST_CPLX(uint32_t);
        case CastKind::F32_TO_CF32:
            return run_unary(L, static_cast<float2*>(out), static_cast<const float*>(in),
                             CastF32ToCF32{}, s);
    }
#undef JST_CAST_REAL
#undef JST_CAST_CPLX
    return hipErrorInvalidValue;
}
// ---- Squelch (dsp/squelch/module_impl_native_cpu.cc:66-98): peak = max |x| over the whole tensor ----
// std::max(peak, v) keeps `peak` when v is NaN, so NaNs never win; |z| of a complex sample is libm
// hypotf, i.e. (float)sqrt((double)re*re + (double)im*im) with the infinity rule in front.  All values
// are >= +0, so their bit patterns order like the floats: one atomicMax on the bits per wavefront.
namespace {
__global__ __launch_bounds__(256) void peak_abs_kernel(uint32_t* __restrict__ peak_bits, const float* __restrict__ in,
                                                       uint64_t count, int complex) {
    float best = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        float v;
        if (complex) {
            const float re = in[2 * i], im = in[2 * i + 1];
            if (__builtin_isinf(re) || __builtin_isinf(im)) v = __builtin_inff();
            else v = (float)__builtin_sqrt((double)re * (double)re + (double)im * (double)im);
        } else {
            v = __builtin_fabsf(in[i]);
        }
        best = (v > best) ? v : best;  // false for NaN
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float other = __shfl_xor(best, off);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(peak_bits, __builtin_bit_cast(uint32_t, best));
}
}  // namespace
hipError_t launch_peak_abs(float* peak, const void* in, uint64_t count, bool complex, hipStream_t s) {
    (void)hipGetLastError();
    hipError_t e = hipMemsetAsync(peak, 0, sizeof(float), s);
    if (e != hipSuccess || count == 0) return e;
    const uint64_t blocks = (count + 255) / 256;
    hipLaunchKernelGGL(peak_abs_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s,
                       reinterpret_cast<uint32_t*>(peak), static_cast<const float*>(in), count, complex ? 1 : 0);
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void amplitude_range_probe_kernel(float* __restrict__ out_exact,
                                                                    float* __restrict__ out_fast,
                                                                    const float2* __restrict__ in, uint64_t count,
                                                                    float coeff, float scale, float offset,
                                                                    dev::BinGuard guard) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const float2 v = in[i];
        out_exact[i] = amplitude_range_exact(v, coeff, scale, offset);  // the fused epilogue's arithmetic
        out_fast[i] = amplitude_range_fast_guarded(v, coeff, scale, offset, guard);
    }
}
}  // namespace
hipError_t launch_amplitude_range_probe(float* out_exact, float* out_fast, const float2* in, uint64_t count,
                                        float amp_coeff, float range_scale, float range_offset,
                                        float guard_h0, float guard_h1, hipStream_t s) {
    if (count == 0) return hipSuccess;
    (void)hipGetLastError();
    const uint64_t blocks = (count + 255) / 256;
    hipLaunchKernelGGL(amplitude_range_probe_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                       s, out_exact, out_fast, in, count, amp_coeff, range_scale, range_offset,
                       dev::BinGuard{guard_h0, guard_h1});
    return hipGetLastError();
}
namespace {
template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* __restrict__ out, uint64_t count, T even, T odd) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256)
        out[i] = (i & 1) ? odd : even;
}
}  // namespace
hipError_t launch_fill_ones(void* out, uint64_t count, int elem_bytes, bool pair, hipStream_t s) {
    if (count == 0) return hipSuccess;
    (void)hipGetLastError();
    const bool f64 = (elem_bytes == 8 && !pair) || elem_bytes == 16;
    const uint64_t words = pair ? 2 * count : count;
    const uint64_t blocks = (words + 255) / 256;
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192));
    if (f64) hipLaunchKernelGGL(fill_kernel<double>, grid, dim3(256), 0, s, (double*)out, words, 1.0, pair ? 0.0 : 1.0);
    else hipLaunchKernelGGL(fill_kernel<float>, grid, dim3(256), 0, s, (float*)out, words, 1.0f, pair ? 0.0f : 1.0f);
    return hipGetLastError();
}
namespace {
__global__ __launch_bounds__(256) void divide_kernel(float* x, uint64_t count, float divisor) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256ull) x[i] = x[i] / divisor;
}
}  // namespace
// x[i] /= divisor in place (the cross-rank average of a summed trace: jst/comm.cc)
hipError_t launch_divide_f32(float* x, uint64_t count, float divisor, hipStream_t s) {
    if (count == 0) return hipSuccess;
    (void)hipGetLastError();
    const uint64_t blocks = (count + 255) / 256;
    hipLaunchKernelGGL(divide_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, x, count, divisor);
    return hipGetLastError();
}
hipError_t launch_tanhf_probe(float* out, const float* in, uint64_t count, hipStream_t s) {
    EwLayout L{};
    L.size = count;
    L.rank = 1;
    L.contiguous = 1;
    L.shape[0] = count;
    return run_unary(L, out, in, TanhProbe{}, s);
}

hipError_t raise_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> raised;  // (kernel, device) -> bytes granted
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    auto it = raised.find({kernel, dev});
    if (it != raised.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) raised[{kernel, dev}] = bytes;
    return e;
}

}  // namespace jst::kernels
