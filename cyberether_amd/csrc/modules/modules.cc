// modules.cc -- host side of the hot-path modules (see modules.hh): validation, output
// allocation and kernel submission.  No arithmetic on tensor data happens on the host.
#include "modules.hh"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <cmath>
#include <chrono>
#include <cstring>
#include <mutex>

namespace jst::modules {

using dev::EwLayout;
using dev::FftLayout;

// ---- helpers -----------------------------------------------------------------------------------
bool MakeEwLayout(const Tensor& out, const Tensor* a, const Tensor* b, EwLayout& L) {
    std::memset(&L, 0, sizeof(L));
    if (out.rank() > (Index)dev::kMaxRank) return false;
    L.size = out.size();
    L.rank = (int32_t)out.rank();
    const Tensor* ops[3] = {&out, a, b};
    bool dense = true;
    for (int o = 0; o < 3; ++o) {
        if (!ops[o]) continue;
        if (ops[o]->rank() != out.rank()) return false;
        L.offset[o] = ops[o]->offset();
        for (Index ax = 0; ax < out.rank(); ++ax) {
            if (ops[o]->shape(ax) != out.shape(ax)) return false;
            L.stride[o][ax] = (int64_t)ops[o]->stride(ax);
        }
        dense = dense && ops[o]->contiguous();
    }
    for (Index ax = 0; ax < out.rank(); ++ax) L.shape[ax] = out.shape(ax);
    L.contiguous = dense ? 1 : 0;
    return true;
}

namespace {

template <class T>
T* ptr(const Tensor& t) {
    return static_cast<T*>(t.data());
}

Result parse_shape(const std::string& s, Shape& out, const char* tag) {
    out.clear();
    if (s.size() < 2 || s.front() != '[' || s.back() != ']') {
        JST_ERROR("[%s] Shape must use bracket notation.", tag);
        return Result::ERROR;
    }
    size_t pos = 1;
    const size_t close = s.size() - 1;
    auto skip = [&] { while (pos < close && std::isspace((unsigned char)s[pos])) ++pos; };
    skip();
    if (pos == close) {
        JST_ERROR("[%s] Shape must have at least one dimension.", tag);
        return Result::ERROR;
    }
    while (pos < close) {
        const size_t begin = pos;
        U64 dim = 0;
        while (pos < close && s[pos] >= '0' && s[pos] <= '9') {
            if (dim > (~0ull - 9) / 10) {
                JST_ERROR("[%s] Shape dimension exceeds the supported numeric range.", tag);
                return Result::ERROR;
            }
            dim = dim * 10 + (U64)(s[pos] - '0');
            ++pos;
        }
        if (begin == pos) {
            JST_ERROR("[%s] Invalid shape syntax '%s'.", tag, s.c_str());
            return Result::ERROR;
        }
        if (dim == 0) {
            JST_ERROR("[%s] Shape dimensions cannot be zero.", tag);
            return Result::ERROR;
        }
        out.push_back(dim);
        skip();
        if (pos == close) break;
        if (s[pos] != ',') {
            JST_ERROR("[%s] Invalid shape syntax '%s'.", tag, s.c_str());
            return Result::ERROR;
        }
        ++pos;
        skip();
        if (pos == close) {
            JST_ERROR("[%s] Invalid shape syntax '%s'.", tag, s.c_str());
            return Result::ERROR;
        }
    }
    return Result::SUCCESS;
}

}  // namespace

// ---- twiddles ----------------------------------------------------------------------------------
// pocketfft's sincos_2pibyn<float> (fft/pocketfft.hh:296-372): exp(2 pi j k/n) as the double
// product of two short tables (fine x coarse), each entry evaluated with an octant reduction so
// the cos/sin arguments stay in [0, pi/4], then rounded to float.  Restated here so the device
// table is bit-identical to the one the reference CPU path multiplies by.
namespace {
struct cd {
    double r, i;
};
cd octant_sincos(U64 x, U64 n, double ang) {
    U64 y = x << 3;
    bool neg_im = false, rot = false;
    if (y >= 4 * n) {
        y = 8 * n - y;
        neg_im = true;
    }
    if (y >= 2 * n) {
        y -= 2 * n;
        rot = true;
    }
    double c, s;
    if (y < n) {
        c = std::cos((double)y * ang);
        s = std::sin((double)y * ang);
    } else {
        c = std::sin((double)(2 * n - y) * ang);
        s = std::cos((double)(2 * n - y) * ang);
    }
    if (rot) {
        const double t = c;
        c = -s;
        s = t;
    }
    if (neg_im) s = -s;
    return {c, s};
}
}  // namespace

void ComputeTwiddles(U64 n, float* out) {
    const long double pi = 3.141592653589793238462643383279502884197L;
    const double ang = (double)(0.25L * pi / (long double)n);
    const U64 nval = (n + 2) / 2;
    U64 shift = 1;
    while ((U64(1) << shift) * (U64(1) << shift) < nval) ++shift;
    const U64 mask = (U64(1) << shift) - 1;
    std::vector<cd> fine(mask + 1), coarse((nval + mask) / (mask + 1));
    fine[0] = {1.0, 0.0};
    for (U64 i = 1; i < fine.size(); ++i) fine[i] = octant_sincos(i, n, ang);
    coarse[0] = {1.0, 0.0};
    for (U64 i = 1; i < coarse.size(); ++i) coarse[i] = octant_sincos(i * (mask + 1), n, ang);
    for (U64 k = 0; k < n; ++k) {
        const bool mirror = 2 * k > n;
        const U64 idx = mirror ? n - k : k;
        const cd a = fine[idx & mask], b = coarse[idx >> shift];
        const float re = (float)(a.r * b.r - a.i * b.i);
        const float im = (float)(a.r * b.i + a.i * b.r);
        out[2 * k] = re;
        out[2 * k + 1] = mirror ? -im : im;
    }
}

Result GetTwiddles(U64 n, const float2** table) {
    static std::mutex mu;
    static std::map<U64, float2*> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(n);
    if (it == cache.end()) {
        std::vector<float> host(2 * n);
        ComputeTwiddles(n, host.data());
        float2* d = nullptr;
        JST_HIP_CHECK(hipMalloc(&d, n * sizeof(float2)), "hipMalloc(twiddles)");
        JST_HIP_CHECK(hipMemcpy(d, host.data(), n * sizeof(float2), hipMemcpyHostToDevice),
                      "hipMemcpy(twiddles)");
        it = cache.emplace(n, d).first;
    }
    *table = it->second;
    return Result::SUCCESS;
}

// Per-pass layout of the same values (tw[(j-1)*ido + i] = W[j*l1*i] for every pass of the plan),
// cached per n: the LDS-tiled kernels read it with unit stride across adjacent butterflies.
Result GetPassTwiddles(U64 n, const float2** table) {
    static std::mutex mu;
    static std::map<U64, float2*> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(n);
    if (it == cache.end()) {
        std::vector<float> w(2 * n);
        ComputeTwiddles(n, w.data());
        const U64 count = kernels::fft_pass_twiddle_count(n);
        std::vector<float> host(2 * (count ? count : 1));
        kernels::fft_pass_twiddle_fill(n, w.data(), host.data());
        float2* d = nullptr;
        JST_HIP_CHECK(hipMalloc(&d, (count ? count : 1) * sizeof(float2)), "hipMalloc(pass twiddles)");
        JST_HIP_CHECK(hipMemcpy(d, host.data(), count * sizeof(float2), hipMemcpyHostToDevice),
                      "hipMemcpy(pass twiddles)");
        it = cache.emplace(n, d).first;
    }
    *table = it->second;
    return Result::SUCCESS;
}

// ---- Window ------------------------------------------------------------------------------------
Result Window::validate() {
    bool ok = true;
    size = ConfigU64(config_, "size", 1024, &ok);
    if (!ok || size == 0) {
        JST_ERROR("[MODULE_WINDOW] Window size cannot be zero.");
        return Result::ERROR;
    }
    return Result::SUCCESS;
}
Result Window::define() {
    JST_CHECK(defineTaint(STATIC_OUTPUT));
    return defineInterfaceOutput("window");
}
Result Window::create() {
    JST_CHECK(output.create(device(), DataType::CF32, {size}));
    JST_CHECK(SetSignalAxes(output, {.sample = Index{0}}));
    produced("window", output);
    return Result::SUCCESS;
}
// The taps are a STATIC table (computed in the first cycle, never again): they are evaluated on the HOST with the
// host libm -- the very cos() the reference's CPU module calls (window/module_impl_native_cpu.cc:20-37) -- and
// uploaded, like the FFT twiddles.  A device evaluation would lean on the device math library's double cos being
// bit-equal to glibc's; one differing double that straddles a float rounding boundary would cost the whole chain
// its bit-exactness for that size.
Result Window::computeSubmit(hipStream_t stream) {
    hostTaps.assign(2 * size, 0.0f);
    if (size == 1) {
        hostTaps[0] = 1.0f;
    } else {
        const double pi = 3.14159265358979323846;
        for (U64 i = 0; i < size; ++i) {
            const double tap = 0.42 - 0.50 * std::cos(2.0 * pi * (double)i / (double)(size - 1)) +
                               0.08 * std::cos(4.0 * pi * (double)i / (double)(size - 1));
            hostTaps[2 * i] = (float)tap;
        }
    }
    JST_HIP_CHECK(hipMemcpyAsync(ptr<float2>(output) + output.offset(), hostTaps.data(),
                                 hostTaps.size() * sizeof(float), hipMemcpyHostToDevice, stream),
                  "window upload");
    JST_HIP_CHECK(hipStreamSynchronize(stream), "hipStreamSynchronize");  // pageable source: done before we return
    return Result::SUCCESS;
}

// ---- Invert ------------------------------------------------------------------------------------
Result Invert::validate() {
    if (!inputs_.count("signal")) return Result::SUCCESS;
    const Tensor& in = inputs_.at("signal");
    if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
    SignalAxes axes;
    if (ResolveSignalAxes(in, axes) != Result::SUCCESS) {
        JST_ERROR("[MODULE_INVERT] Input must contain valid signal axis metadata.");
        return Result::ERROR;
    }
    if (in.dtype() != DataType::F32 && in.dtype() != DataType::CF32) {
        JST_ERROR("[MODULE_INVERT_NATIVE_HIP] Unsupported data type '%s'.", DataTypeName(in.dtype()));
        return Result::ERROR;
    }
    resolvedAxis = *axes.sample;
    return Result::SUCCESS;
}
Result Invert::define() {
    JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
    JST_CHECK(defineInterfaceInput("signal"));
    return defineInterfaceOutput("signal");
}
Result Invert::create() {
    input = inputs_.at("signal");
    JST_CHECK(output.create(device(), DataType::CF32, input.shape()));
    JST_CHECK(output.propagateAttributes(input));
    axisInnerSize = 1;
    for (Index ax = resolvedAxis + 1; ax < input.rank(); ++ax) axisInnerSize *= input.shape(ax);
    axisLength = input.shape(resolvedAxis);
    produced("signal", output);
    return Result::SUCCESS;
}
Result Invert::computeSubmit(hipStream_t stream) {
    EwLayout L;
    if (!MakeEwLayout(output, &input, nullptr, L)) {
        JST_ERROR("[MODULE_INVERT] Unsupported tensor rank.");
        return Result::ERROR;
    }
    return hip_result(kernels::launch_invert(L, ptr<float2>(output), input.data(),
                                             input.dtype() == DataType::CF32, axisInnerSize,
                                             axisLength, stream),
                      "invert kernel");
}

// ---- Reshape / Cast (views) --------------------------------------------------------------------
Result Reshape::validate() {
    JST_CHECK(parse_shape(ConfigStr(config_, "shape", "[]"), target, "MODULE_RESHAPE"));
    if (!inputs_.count("buffer")) return Result::SUCCESS;
    const Tensor& in = inputs_.at("buffer");
    U64 n = 1;
    for (U64 d : target) n *= d;
    if (in.validShape() && n != in.size()) {
        JST_ERROR("[MODULE_RESHAPE] Cannot reshape %s into %s.", ShapeToString(in.shape()).c_str(),
                  ShapeToString(target).c_str());
        return Result::ERROR;
    }
    return Result::SUCCESS;
}
Result Reshape::define() {
    JST_CHECK(defineTaint(STATELESS));
    JST_CHECK(defineInterfaceInput("buffer"));
    return defineInterfaceOutput("buffer");
}
Result Reshape::create() {
    Tensor view = inputs_.at("buffer").clone();
    JST_CHECK(view.reshape(target));
    // Axis roles are positional: a reshape invalidates them (the block re-tags the view,
    // spectrum_engine/block_impl.cc:167-170).
    view.removeAttribute(SampleAxisAttribute);
    view.removeAttribute(BatchAxisAttribute);
    view.removeAttribute(ChannelAxisAttribute);
    produced("buffer", view);
    return Result::SUCCESS;
}

Result Cast::validate() {
    const std::string want = ConfigStr(config_, "outputType", "CF32");
    outputDtype = NameToDataType(want);
    if (outputDtype == DataType::None) {
        JST_ERROR("[MODULE_CAST] Invalid output type '%s'.", want.c_str());
        return Result::ERROR;
    }
    bypass = false;
    scaler = 1.0f;
    if (!inputs_.count("buffer")) return Result::SUCCESS;
    const Tensor& in = inputs_.at("buffer");
    bypass = in.dtype() == outputDtype;
    if (!in.validShape() || in.size() == 0 || bypass) return Result::SUCCESS;
    switch (in.dtype()) {  // cast/module_impl.cc:47-70
        case DataType::I8: case DataType::CI8: case DataType::U8: case DataType::CU8:
            scaler = 128.0f;
            break;
        case DataType::I16: case DataType::CI16: case DataType::U16: case DataType::CU16:
            scaler = 32768.0f;
            break;
        case DataType::I32: case DataType::CI32: case DataType::U32: case DataType::CU32:
            scaler = 2147483648.0f;
            break;
        default: break;
    }
    const DataType d = in.dtype();
    const bool real_int = d == DataType::I8 || d == DataType::U8 || d == DataType::I16 ||
                          d == DataType::U16 || d == DataType::I32 || d == DataType::U32;
    const bool cplx_int = d == DataType::CI8 || d == DataType::CU8 || d == DataType::CI16 ||
                          d == DataType::CU16 || d == DataType::CI32 || d == DataType::CU32;
    const bool ok = (outputDtype == DataType::F32 && real_int) ||
                    (outputDtype == DataType::CF32 && (d == DataType::F32 || cplx_int));
    if (!ok) {
        JST_ERROR("[MODULE_CAST_NATIVE_HIP] Unsupported conversion '%s' -> '%s'.", DataTypeName(d),
                  DataTypeName(outputDtype));
        return Result::ERROR;
    }
    return Result::SUCCESS;
}
Result Cast::define() {
    JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
    JST_CHECK(defineInterfaceInput("buffer"));
    return defineInterfaceOutput("buffer");
}
Result Cast::create() {
    input = inputs_.at("buffer");
    if (bypass) {
        produced("buffer", input);  // output aliases input (cast/module_impl.cc:96-100)
        return Result::SUCCESS;
    }
    JST_CHECK(output.create(device(), outputDtype, input.shape()));
    JST_CHECK(output.propagateAttributes(input));
    produced("buffer", output);
    return Result::SUCCESS;
}
Result Cast::computeSubmit(hipStream_t stream) {
    if (bypass || fusedIntoSpectrum) return Result::SUCCESS;
    EwLayout L;
    if (!MakeEwLayout(output, &input, nullptr, L)) {
        JST_ERROR("[MODULE_CAST] Unsupported tensor rank.");
        return Result::ERROR;
    }
    using kernels::CastKind;
    CastKind kind = CastKind::F32_TO_CF32;
    switch (input.dtype()) {
        case DataType::I8: kind = CastKind::I8; break;
        case DataType::U8: kind = CastKind::U8; break;
        case DataType::I16: kind = CastKind::I16; break;
        case DataType::U16: kind = CastKind::U16; break;
        case DataType::I32: kind = CastKind::I32; break;
        case DataType::U32: kind = CastKind::U32; break;
        case DataType::CI8: kind = CastKind::CI8; break;
        case DataType::CU8: kind = CastKind::CU8; break;
        case DataType::CI16: kind = CastKind::CI16; break;
        case DataType::CU16: kind = CastKind::CU16; break;
        case DataType::CI32: kind = CastKind::CI32; break;
        case DataType::CU32: kind = CastKind::CU32; break;
        default: break;
    }
    return hip_result(kernels::launch_cast(L, output.data(), input.data(), kind, scaler, stream),
                      "cast kernel");
}

// ---- Multiply ----------------------------------------------------------------------------------
namespace {
// MapSignalAxes with a right-aligned axis map (axis.cc:315-359).
Result map_axes_right_aligned(const Tensor& t, Index out_rank, SignalAxes& axes) {
    SignalAxes in;
    JST_CHECK(MapSignalAxes(t, in));
    const Index shift = out_rank - t.rank();
    axes = {};
    if (in.sample) axes.sample = *in.sample + shift;
    if (in.batch) axes.batch = *in.batch + shift;
    if (in.channel) axes.channel = *in.channel + shift;
    return Result::SUCCESS;
}
Result merge_broadcast_axes(const Tensor& ta, const Tensor& tb, Tensor& out) {
    SignalAxes aa, ab, ao;
    JST_CHECK(map_axes_right_aligned(ta, out.rank(), aa));
    JST_CHECK(map_axes_right_aligned(tb, out.rank(), ab));
    auto merge = [](const std::optional<Index>& x, const std::optional<Index>& y,
                    std::optional<Index>& o) {
        if (x && y && *x != *y) {
            JST_ERROR("[MEMORY:AXIS] Signal roles map to conflicting output axes.");
            return Result::ERROR;
        }
        o = x ? x : y;
        return Result::SUCCESS;
    };
    JST_CHECK(merge(aa.sample, ab.sample, ao.sample));
    JST_CHECK(merge(aa.batch, ab.batch, ao.batch));
    JST_CHECK(merge(aa.channel, ab.channel, ao.channel));
    return SetSignalAxes(out, ao);
}
}  // namespace

Result Multiply::validate() {
    a = Tensor();
    b = Tensor();
    outputShape.clear();
    if (!inputs_.count("a") || !inputs_.count("b")) return Result::SUCCESS;
    const Tensor& ta = inputs_.at("a");
    const Tensor& tb = inputs_.at("b");
    if (!ta.validShape() || !tb.validShape() || ta.size() == 0 || tb.size() == 0)
        return Result::SUCCESS;
    if (ta.dtype() != tb.dtype() || (ta.dtype() != DataType::F32 && ta.dtype() != DataType::CF32)) {
        JST_ERROR("[MODULE_%s_NATIVE_HIP] Unsupported data types '%s' x '%s'.", tag(),
                  DataTypeName(ta.dtype()), DataTypeName(tb.dtype()));
        return Result::ERROR;
    }
    const U64 ra = ta.rank(), rb = tb.rank(), mr = std::max(ra, rb);
    Shape os(mr == 0 ? 1 : mr, 1);
    for (U64 i = 0; i < mr; ++i) {
        const U64 da = ra > i ? ta.shape(ra - 1 - i) : 1;
        const U64 db = rb > i ? tb.shape(rb - 1 - i) : 1;
        if (da != db && da != 1 && db != 1) {
            JST_ERROR("[MODULE_%s] Input shapes %s and %s are not broadcastable.", tag(),
                      ShapeToString(ta.shape()).c_str(), ShapeToString(tb.shape()).c_str());
            return Result::ERROR;
        }
        os[os.size() - 1 - i] = std::max(da, db);
    }
    Tensor ba = ta.clone(), bb = tb.clone();
    if (ba.broadcastTo(os) != Result::SUCCESS || bb.broadcastTo(os) != Result::SUCCESS) {
        JST_ERROR("[MODULE_%s] Failed to construct validated broadcast views.", tag());
        return Result::ERROR;
    }
    a = ba;
    b = bb;
    outputShape = os;
    return Result::SUCCESS;
}
Result Multiply::define() {
    JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
    JST_CHECK(defineInterfaceOutput("product"));
    JST_CHECK(defineInterfaceInput("a"));
    return defineInterfaceInput("b");
}
Result Multiply::create() {
    JST_CHECK(c.create(device(), a.dtype(), outputShape));
    JST_CHECK(c.propagateAttributes(inputs_.at("a")));
    JST_CHECK(merge_broadcast_axes(inputs_.at("a"), inputs_.at("b"), c));
    produced(outputPort(), c);
    return Result::SUCCESS;
}
Result Multiply::computeSubmit(hipStream_t stream) {
    EwLayout L;
    if (!MakeEwLayout(c, &a, &b, L)) {
        JST_ERROR("[MODULE_MULTIPLY] Unsupported tensor rank.");
        return Result::ERROR;
    }
    if (a.dtype() == DataType::CF32)
        return hip_result(kernels::launch_multiply_cf32(L, ptr<float2>(c), ptr<const float2>(a),
                                                        ptr<const float2>(b), stream),
                          "multiply kernel");
    return hip_result(kernels::launch_multiply_f32(L, ptr<float>(c), ptr<const float>(a),
                                                   ptr<const float>(b), stream),
                      "multiply kernel");
}

// ---- Add (core/add/{module_impl.cc:9-146, module_impl_native_cpu.cc:24-98}) ---------------------
// Same broadcast planning as Multiply; DISCONTIGUOUS only (not STATELESS), output port "sum", and
// the sampleRate / frequency attributes follow the "a unless a is unset" rule of :108-135.
class Add : public Multiply {
 public:
    const char* type() const override { return "add"; }
    const char* tag() const override { return "ADD"; }
    const char* outputPort() const override { return "sum"; }
    Result define() override {
        JST_CHECK(defineTaint(DISCONTIGUOUS));
        JST_CHECK(defineInterfaceOutput("sum"));
        JST_CHECK(defineInterfaceInput("a"));
        return defineInterfaceInput("b");
    }
    Result create() override {
        JST_CHECK(Multiply::create());
        for (const char* key : {"sampleRate", "frequency"}) {
            const auto read = [&](const Tensor& t) -> F64 {
                const AttrValue* v = t.attribute(key);
                if (!v) return 0.0;
                if (const F64* f = std::get_if<F64>(v)) return (F64)(F32)*f;
                if (const U64* u = std::get_if<U64>(v)) return (F64)(F32)*u;
                return 0.0;
            };
            const F64 va = read(inputs_.at("a")), vb = read(inputs_.at("b"));
            const F64 merged = (va == vb || vb == 0.0) ? va : (va == 0.0 ? vb : va);
            if (inputs_.at("a").hasAttribute(key) || inputs_.at("b").hasAttribute(key))
                c.setAttribute(key, AttrValue{merged});
        }
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t stream) override {
        EwLayout L;
        if (!MakeEwLayout(c, &a, &b, L)) {
            JST_ERROR("[MODULE_ADD] Unsupported tensor rank.");
            return Result::ERROR;
        }
        if (a.dtype() == DataType::CF32)
            return hip_result(kernels::launch_add_cf32(L, ptr<float2>(c), ptr<const float2>(a),
                                                       ptr<const float2>(b), stream),
                              "add kernel");
        return hip_result(kernels::launch_add_f32(L, ptr<float>(c), ptr<const float>(a),
                                                  ptr<const float>(b), stream),
                          "add kernel");
    }
};

// ---- MultiplyConstant --------------------------------------------------------------------------
Result MultiplyConstant::validate() {
    bool ok = true;
    constant = (F32)ConfigF64(config_, "constant", 1.0, &ok);
    if (!ok) {
        JST_ERROR("[MODULE_MULTIPLY_CONSTANT] Invalid constant.");
        return Result::ERROR;
    }
    if (!inputs_.count("factor")) return Result::SUCCESS;
    const DataType dt = inputs_.at("factor").dtype();
    if (dt != DataType::F32 && dt != DataType::CF32) {
        JST_ERROR("[MODULE_MULTIPLY_CONSTANT_NATIVE_HIP] Unsupported data type '%s'.",
                  DataTypeName(dt));
        return Result::ERROR;
    }
    return Result::SUCCESS;
}
Result MultiplyConstant::reconfigureImpl(const Config& previous) {  // multiply_constant/module_impl.cc:26-35
    return (F32)ConfigF64(previous, "constant", 1.0) != constant ? Result::SUCCESS : Result::RECREATE;
}
Result MultiplyConstant::define() {
    JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
    JST_CHECK(defineInterfaceInput("factor"));
    return defineInterfaceOutput("product");
}
Result MultiplyConstant::create() {
    input = inputs_.at("factor");
    JST_CHECK(output.create(device(), input.dtype(), input.shape()));
    JST_CHECK(output.propagateAttributes(input));
    produced("product", output);
    return Result::SUCCESS;
}
Result MultiplyConstant::computeSubmit(hipStream_t stream) {
    EwLayout L;
    if (!MakeEwLayout(output, &input, nullptr, L)) return Result::ERROR;
    if (input.dtype() == DataType::CF32)
        return hip_result(kernels::launch_multiply_constant_cf32(
                              L, ptr<float2>(output), ptr<const float2>(input), constant, stream),
                          "multiply_constant kernel");
    return hip_result(kernels::launch_multiply_constant_f32(L, ptr<float>(output),
                                                            ptr<const float>(input), constant,
                                                            stream),
                      "multiply_constant kernel");
}

// ---- FFT ---------------------------------------------------------------------------------------
Result Fft::validate() {
    bool ok1 = true, ok2 = true;
    forward = ConfigBool(config_, "forward", true, &ok1);
    complexOutput = ConfigBool(config_, "complexOutput", false, &ok2);
    if (!ok1 || !ok2) {
        JST_ERROR("[MODULE_FFT] Invalid boolean configuration value.");
        return Result::ERROR;
    }
    if (!inputs_.count("signal")) return Result::SUCCESS;
    const Tensor& in = inputs_.at("signal");
    if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
    SignalAxes axes;
    if (ResolveSignalAxes(in, axes) != Result::SUCCESS) {
        JST_ERROR("[MODULE_FFT] Input must contain valid signal axis metadata.");
        return Result::ERROR;
    }
    if (in.dtype() != DataType::CF32 && in.dtype() != DataType::F32) {
        JST_ERROR("[MODULE_FFT_NATIVE_HIP] Data type '%s' is not implemented on the HIP device.",
                  DataTypeName(in.dtype()));
        return Result::ERROR;
    }
    const U64 n = in.shape(*axes.sample);
    realInput = in.dtype() == DataType::F32;
    complexOut = realInput && forward && complexOutput;  // fft/module_impl.cc:33-38
    if (realInput && !kernels::rfft_supported(n) && kernels::rfft_bluestein_size(n) == 0) {
        JST_ERROR("[MODULE_FFT_NATIVE_HIP] Real transform length %llu has a prime factor above the "
                  "generic radix limit of the HIP device.",
                  (unsigned long long)n);
        return Result::ERROR;
    }
    if (!kernels::fft_lds_supported(n) && !kernels::fft_global_supported(n)) {
        JST_ERROR("[MODULE_FFT_NATIVE_HIP] Transform length %llu exceeds the supported range.",
                  (unsigned long long)n);
        return Result::ERROR;
    }
    if (in.rank() - 1 > (Index)dev::kMaxOuterRank) {
        JST_ERROR("[MODULE_FFT] Output shape exceeds the supported layout range.");
        return Result::ERROR;
    }
    resolvedAxis = *axes.sample;
    return Result::SUCCESS;
}
Result Fft::define() {
    JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
    JST_CHECK(defineInterfaceInput("signal"));
    return defineInterfaceOutput("signal");
}
Result Fft::create() {
    input = inputs_.at("signal");
    if (complexOut) {  // r2c: n/2 + 1 complex bins along the transform axis
        Shape os = input.shape();
        os[resolvedAxis] = input.shape(resolvedAxis) / 2 + 1;
        JST_CHECK(output.create(device(), DataType::CF32, os));
        JST_CHECK(output.propagateAttributes(input));
        produced("signal", output);
        return Result::SUCCESS;
    }
    JST_CHECK(output.create(device(), input.dtype(), input.shape()));
    JST_CHECK(output.propagateAttributes(input));
    produced("signal", output);
    return Result::SUCCESS;
}
Result Fft::computeInitialize() {
    const U64 n = input.shape(resolvedAxis);
    const U64 transforms = input.size() / n;
    bluesteinSize = realInput ? kernels::rfft_bluestein_size(n) : kernels::fft_bluestein_size(n);
    if (realInput) {
        JST_CHECK(realA.create(device(), DataType::F32, {transforms * n}));
        JST_CHECK(realB.create(device(), DataType::F32, {transforms * n}));
        if (bluesteinSize) {
            JST_CHECK(realLine.create(device(), DataType::CF32, {transforms * n}));
        } else {
            std::vector<float> w(2 * n), tw(kernels::rfft_twiddle_count(n) + 1, 0.0f);
            ComputeTwiddles(n, w.data());
            kernels::rfft_twiddle_fill(n, w.data(), tw.data());
            JST_CHECK(realTw.create(device(), DataType::F32, {(U64)tw.size()}));
            JST_CHECK(realTw.copyFromHost(tw.data(), tw.size() * sizeof(float), nullptr));
            JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize(rfft twiddles)");
            return Result::SUCCESS;  // no complex machinery needed
        }
    }
    // the length the pass kernels actually run at: n, or the Bluestein convolution length
    const U64 m = bluesteinSize ? bluesteinSize : n;
    JST_CHECK(GetTwiddles(m, &twiddles));
    useTiled = !kernels::fft_lds_supported(m) && kernels::fft_tiled_supported(m);
    if (useTiled) JST_CHECK(GetPassTwiddles(m, &twiddles));  // the tiled kernels' table layout
    useGlobalPasses = !kernels::fft_lds_supported(m) && !useTiled;
    if (useTiled && kernels::fft_tiled_may_use_scratch(m))
        JST_CHECK(scratchA.create(device(), DataType::CF32, {transforms * m}));
    if (useGlobalPasses) {  // ping-pong scratch for the pass-per-launch path
        uint32_t fact[64];
        const int nf = kernels::fft_plan_factors(m, fact);
        if (nf >= 2) JST_CHECK(scratchA.create(device(), DataType::CF32, {transforms * m}));
        if (nf >= 3) JST_CHECK(scratchB.create(device(), DataType::CF32, {transforms * m}));
        if (kernels::fft_plan_has_generic_radix(m))
            JST_CHECK(scratchH.create(device(), DataType::CF32, {transforms * m}));
    }
    if (!bluesteinSize) return Result::SUCCESS;

    // fftblue's constructor (pocketfft.hh:2402-2429): chirp b_k from the 2n-point table, and the
    // transform of the zero-padded, 1/n2-scaled chirp -- computed with the same device passes.
    const U64 n2 = bluesteinSize;
    JST_CHECK(akf.create(device(), DataType::CF32, {transforms * n2}));
    JST_CHECK(bk.create(device(), DataType::CF32, {n}));
    JST_CHECK(bkf.create(device(), DataType::CF32, {n2 / 2 + 1}));
    std::vector<float> table(4 * n), chirp(2 * n), padded(2 * n2, 0.0f);
    ComputeTwiddles(2 * n, table.data());
    chirp[0] = 1.0f;
    chirp[1] = 0.0f;
    U64 coeff = 0;
    for (U64 k = 1; k < n; ++k) {
        coeff += 2 * k - 1;
        if (coeff >= 2 * n) coeff -= 2 * n;
        chirp[2 * k] = table[2 * coeff];
        chirp[2 * k + 1] = table[2 * coeff + 1];
    }
    const float xn2 = 1.0f / (float)n2;
    padded[0] = chirp[0] * xn2;
    padded[1] = chirp[1] * xn2;
    for (U64 k = 1; k < n; ++k) {
        const float re = chirp[2 * k] * xn2, im = chirp[2 * k + 1] * xn2;
        padded[2 * k] = padded[2 * (n2 - k)] = re;
        padded[2 * k + 1] = padded[2 * (n2 - k) + 1] = im;
    }
    JST_HIP_CHECK(hipMemcpy(bk.data(), chirp.data(), n * sizeof(float2), hipMemcpyHostToDevice),
                  "hipMemcpy(bluestein chirp)");
    JST_HIP_CHECK(hipMemcpy(akf.data(), padded.data(), n2 * sizeof(float2), hipMemcpyHostToDevice),
                  "hipMemcpy(bluestein padded chirp)");
    JST_CHECK(innerTransform(ptr<float2>(akf), n2, 1, true, nullptr));
    JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize(bluestein setup)");
    JST_HIP_CHECK(hipMemcpy(bkf.data(), akf.data(), (n2 / 2 + 1) * sizeof(float2),
                            hipMemcpyDeviceToDevice),
                  "hipMemcpy(bluestein kernel)");
    return Result::SUCCESS;
}
Result Fft::computeDeinitialize() {
    scratchA = Tensor();
    scratchB = Tensor();
    scratchH = Tensor();
    realA = Tensor();
    realB = Tensor();
    realTw = Tensor();
    realLine = Tensor();
    akf = Tensor();
    bk = Tensor();
    bkf = Tensor();
    return Result::SUCCESS;
}
Result Fft::innerTransform(float2* data, U64 length, U64 transforms, bool fwd, hipStream_t stream) {
    FftLayout L;
    std::memset(&L, 0, sizeof(L));
    L.transforms = transforms;
    L.outer_rank = 1;
    L.outer_shape[0] = transforms;
    L.in_outer_stride[0] = L.out_outer_stride[0] = (int64_t)length;
    L.in_axis_stride = L.out_axis_stride = 1;
    if (useTiled)
        return hip_result(kernels::launch_fft_c2c_tiled(length, fwd, L, twiddles, data, data,
                                                        ptr<float2>(scratchA), stream),
                          "fft (tiled) kernel");
    if (useGlobalPasses)
        return hip_result(kernels::launch_fft_c2c_global(length, fwd, L, twiddles, data, data,
                                                         ptr<float2>(scratchA), ptr<float2>(scratchB),
                                                         ptr<float2>(scratchH), stream),
                          "fft (global passes) kernel");
    return hip_result(kernels::launch_fft_c2c(length, fwd, L, twiddles, data, data, stream),
                      "fft kernel");
}
Result Fft::layout(FftLayout& L) const {
    std::memset(&L, 0, sizeof(L));
    L.transforms = 1;
    int r = 0;
    for (Index ax = 0; ax < input.rank(); ++ax) {
        if (ax == resolvedAxis) continue;
        L.outer_shape[r] = input.shape(ax);
        L.in_outer_stride[r] = (int64_t)input.stride(ax);
        L.out_outer_stride[r] = (int64_t)output.stride(ax);
        L.transforms *= input.shape(ax);
        ++r;
    }
    L.outer_rank = r;
    L.in_axis_stride = (int64_t)input.stride(resolvedAxis);
    L.out_axis_stride = (int64_t)output.stride(resolvedAxis);
    L.in_offset = input.offset();
    L.out_offset = output.offset();
    return Result::SUCCESS;
}
Result Fft::submitComplex(const FftLayout& L, const float2* in, float2* out, bool fwd,
                          hipStream_t stream) {
    const U64 n = input.shape(resolvedAxis);
    if (bluesteinSize) {  // fftblue::fft (pocketfft.hh:2370-2399)
        const U64 n2 = bluesteinSize;
        JST_CHECK(hip_result(kernels::launch_bluestein_pre(fwd, L, ptr<float2>(akf), in,
                                                           ptr<const float2>(bk), n, n2, stream),
                             "bluestein chirp kernel"));
        JST_CHECK(innerTransform(ptr<float2>(akf), n2, L.transforms, true, stream));
        JST_CHECK(hip_result(kernels::launch_bluestein_mul(fwd, ptr<float2>(akf),
                                                           ptr<const float2>(bkf), L.transforms,
                                                           n2, stream),
                             "bluestein convolution kernel"));
        JST_CHECK(innerTransform(ptr<float2>(akf), n2, L.transforms, false, stream));
        return hip_result(kernels::launch_bluestein_post(fwd, L, out, ptr<const float2>(akf),
                                                         ptr<const float2>(bk), n, n2, stream),
                          "bluestein output kernel");
    }
    if (useTiled)
        return hip_result(kernels::launch_fft_c2c_tiled(n, fwd, L, twiddles, in, out,
                                                        ptr<float2>(scratchA), stream),
                          "fft (tiled) kernel");
    if (useGlobalPasses)
        return hip_result(kernels::launch_fft_c2c_global(n, fwd, L, twiddles, in, out,
                                                         ptr<float2>(scratchA), ptr<float2>(scratchB),
                                                         ptr<float2>(scratchH), stream),
                          "fft (global passes) kernel");
    return hip_result(kernels::launch_fft_c2c(n, fwd, L, twiddles, in, out, stream), "fft kernel");
}
Result Fft::computeSubmit(hipStream_t stream) {
    FftLayout L;
    JST_CHECK(layout(L));
    if (!realInput)
        return submitComplex(L, ptr<const float2>(input), ptr<float2>(output), forward, stream);
    // F32 input: gather rows, rfftp passes (or Bluestein on a complex line), scatter / re-pack
    const U64 n = input.shape(resolvedAxis);
    const bool r2hc = forward;  // r2r_fftpack(real2hermitian = forward, forward), and r2c
    JST_CHECK(hip_result(kernels::launch_rfft_gather(L, ptr<float>(realA), ptr<const float>(input), n,
                                                     stream),
                         "rfft gather kernel"));
    float* result = ptr<float>(realA);
    if (bluesteinSize) {  // fftblue::exec_r (pocketfft.hh:2434-2457)
        JST_CHECK(hip_result(kernels::launch_rfft_blue_in(ptr<float2>(realLine), ptr<const float>(realA),
                                                          L.transforms, n, r2hc, stream),
                             "rfft bluestein pack kernel"));
        FftLayout D;
        std::memset(&D, 0, sizeof(D));
        D.transforms = L.transforms;
        D.outer_rank = 1;
        D.outer_shape[0] = L.transforms;
        D.in_outer_stride[0] = D.out_outer_stride[0] = (int64_t)n;
        D.in_axis_stride = D.out_axis_stride = 1;
        JST_CHECK(submitComplex(D, ptr<const float2>(realLine), ptr<float2>(realLine), r2hc, stream));
        JST_CHECK(hip_result(kernels::launch_rfft_blue_out(ptr<float>(realA), ptr<const float2>(realLine),
                                                           L.transforms, n, r2hc, stream),
                             "rfft bluestein unpack kernel"));
    } else {
        JST_CHECK(hip_result(kernels::launch_rfft_passes(n, r2hc, L.transforms, ptr<float>(realA),
                                                         ptr<float>(realB), ptr<const float>(realTw),
                                                         &result, stream),
                             "rfft pass kernels"));
    }
    return hip_result(kernels::launch_rfft_scatter(L, ptr<float>(output), result, n, complexOut, stream),
                      "rfft scatter kernel");
}

// ---- Amplitude ---------------------------------------------------------------------------------
Result Amplitude::validate() {
    normalizationSize = 1;
    if (!inputs_.count("signal")) return Result::SUCCESS;
    const Tensor& in = inputs_.at("signal");
    SignalAxes axes;
    if (MapSignalAxes(in, axes) != Result::SUCCESS) {
        JST_ERROR("[MODULE_AMPLITUDE] Input must contain valid signal axis metadata.");
        return Result::ERROR;
    }
    if (!axes.sample && !axes.channel) {
        JST_ERROR("[MODULE_AMPLITUDE] Input must contain sampleAxis or channelAxis metadata.");
        return Result::ERROR;
    }
    if (in.dtype() != DataType::F32 && in.dtype() != DataType::CF32) {
        JST_ERROR("[MODULE_AMPLITUDE_NATIVE_HIP] Unsupported data type '%s'.",
                  DataTypeName(in.dtype()));
        return Result::ERROR;
    }
    if (axes.sample) normalizationSize = in.shape(*axes.sample);
    return Result::SUCCESS;
}
Result Amplitude::define() {
    JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
    JST_CHECK(defineInterfaceInput("signal"));
    return defineInterfaceOutput("signal");
}
Result Amplitude::create() {
    input = inputs_.at("signal");
    scalingCoeff = 20.0f * std::log10(1.0f / static_cast<F32>(normalizationSize));
    JST_CHECK(output.create(device(), DataType::F32, input.shape()));
    JST_CHECK(output.propagateAttributes(input));
    produced("signal", output);
    return Result::SUCCESS;
}
Result Amplitude::computeSubmit(hipStream_t stream) {
    EwLayout L;
    if (!MakeEwLayout(output, &input, nullptr, L)) return Result::ERROR;
    if (input.dtype() == DataType::CF32)
        return hip_result(kernels::launch_amplitude_cf32(L, ptr<float>(output),
                                                         ptr<const float2>(input), scalingCoeff,
                                                         provider() == "fast", stream),
                          "amplitude kernel");
    return hip_result(kernels::launch_amplitude_f32(L, ptr<float>(output), ptr<const float>(input),
                                                    scalingCoeff, stream),
                      "amplitude kernel");
}

// ---- Range -------------------------------------------------------------------------------------
Result Range::validate() {
    bool ok1 = true, ok2 = true;
    min = (F32)ConfigF64(config_, "min", -1.0, &ok1);
    max = (F32)ConfigF64(config_, "max", 1.0, &ok2);
    if (!ok1 || !ok2) {
        JST_ERROR("[MODULE_RANGE] Invalid min/max.");
        return Result::ERROR;
    }
    if (inputs_.count("signal") && inputs_.at("signal").dtype() != DataType::F32) {
        JST_ERROR("[MODULE_RANGE_NATIVE_HIP] Unsupported data type '%s'.",
                  DataTypeName(inputs_.at("signal").dtype()));
        return Result::ERROR;
    }
    return Result::SUCCESS;
}
Result Range::define() {
    JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
    JST_CHECK(defineInterfaceOutput("signal"));
    return defineInterfaceInput("signal");
}
void Range::updateCoefficients() {  // range/module_impl.cc:51-62
    const F32 lower = std::min(min, max), upper = std::max(min, max);
    if (lower == upper) {
        scalingCoeff = 0.0f;
        offsetCoeff = 0.5f;
    } else {
        scalingCoeff = 1.0f / (upper - lower);
        offsetCoeff = -lower * scalingCoeff;
    }
}
Result Range::reconfigureImpl(const Config&) {  // range/module_impl.cc:40-49: min / max move in place
    updateCoefficients();
    return Result::SUCCESS;
}
Result Range::create() {
    input = inputs_.at("signal");
    updateCoefficients();
    JST_CHECK(output.create(device(), input.dtype(), input.shape()));
    JST_CHECK(output.propagateAttributes(input));
    produced("signal", output);
    return Result::SUCCESS;
}
Result Range::computeSubmit(hipStream_t stream) {
    EwLayout L;
    if (!MakeEwLayout(output, &input, nullptr, L)) return Result::ERROR;
    return hip_result(kernels::launch_range_f32(L, ptr<float>(output), ptr<const float>(input),
                                                scalingCoeff, offsetCoeff, provider() == "fast",
                                                stream),
                      "range kernel");
}

// ---- Spectrogram / Waterfall -------------------------------------------------------------------
namespace {
Result validate_surface_input(const char* tag, const Config& cfg, U64 default_height,
                              const std::map<std::string, Tensor>& inputs, U64& height, U64& width,
                              U64& batches, U64& estride, U64& bstride) {
    bool ok = true;
    height = ConfigU64(cfg, "height", default_height, &ok);
    if (!ok || height == 0 || height > 2048) {
        JST_ERROR("[%s] Invalid height value '%s', must be between 1 and 2048.", tag,
                  ConfigStr(cfg, "height", "?").c_str());
        return Result::ERROR;
    }
    width = batches = estride = bstride = 0;
    if (!inputs.count("signal")) return Result::SUCCESS;
    const Tensor& in = inputs.at("signal");
    if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
    SignalAxes axes;
    if (MapSignalAxes(in, axes) != Result::SUCCESS) {
        JST_ERROR("[%s] Input must contain valid signal axis metadata.", tag);
        return Result::ERROR;
    }
    if (axes.sample && axes.channel) {
        JST_ERROR("[%s] Input cannot contain both sampleAxis and channelAxis.", tag);
        return Result::ERROR;
    }
    const auto element = axes.sample ? axes.sample : axes.channel;
    if (!element) {
        JST_ERROR("[%s] Input must contain sampleAxis or channelAxis.", tag);
        return Result::ERROR;
    }
    for (Index ax = 0; ax < in.rank(); ++ax) {
        if (ax != *element && (!axes.batch || ax != *axes.batch)) {
            JST_ERROR("[%s] Unsupported auxiliary input axis %llu. Every dimension must be the "
                      "element axis or batchAxis.",
                      tag, (unsigned long long)ax);
            return Result::ERROR;
        }
    }
    if (in.dtype() != DataType::F32) {
        JST_ERROR("[%s] Input must be F32.", tag);
        return Result::ERROR;
    }
    width = in.shape(*element);
    if (width > 0xffffffffull / height) {
        JST_ERROR("[%s] Render bin count exceeds the supported range.", tag);
        return Result::ERROR;
    }
    batches = axes.batch ? in.shape(*axes.batch) : 1;
    estride = in.stride(*element);
    bstride = axes.batch ? in.stride(*axes.batch) : 0;
    return Result::SUCCESS;
}
}  // namespace

Result Spectrogram::validate() {
    const std::string merge = ConfigStr(config_, "merge", "local");
    if (merge != "local" && merge != "counts") {
        JST_ERROR("[MODULE_SPECTROGRAM] Invalid merge mode '%s', must be 'local' or 'counts'.", merge.c_str());
        return Result::ERROR;
    }
    countsOnly = merge == "counts";
    return validate_surface_input("MODULE_SPECTROGRAM", config_, 256, inputs_, height,
                                  numberOfElements, numberOfBatches, inputElementStride,
                                  inputBatchStride);
}
Result Spectrogram::define() {
    JST_CHECK(defineTaint(SURFACE));
    JST_CHECK(defineInterfaceInput("signal"));
    if (ConfigStr(config_, "merge", "local") == "counts") return defineInterfaceOutput("counts");
    return Result::SUCCESS;
}
Result Spectrogram::create() {
    input = inputs_.at("signal");
    decayFactor = std::pow(0.999f, static_cast<F32>(numberOfBatches));  // module_impl.cc:104
    JST_CHECK(frequencyBins.create(device(), DataType::F32, {numberOfElements, height}));
    if (countsOnly) {
        JST_CHECK(hitCounts.create(device(), DataType::U32, {numberOfElements, height}));
        produced("counts", hitCounts);
    }
    return Result::SUCCESS;
}
Result Spectrogram::computeSubmit(hipStream_t stream) {
    if (countsOnly)
        return hip_result(
            kernels::launch_spectrogram_counts(ptr<uint32_t>(hitCounts), ptr<const float>(input), input.offset(),
                                               numberOfBatches, numberOfElements, height, (int64_t)inputBatchStride,
                                               (int64_t)inputElementStride, stream),
            "spectrogram counts kernel");
    if (indexFed)
        return hip_result(kernels::launch_spectrogram_index(ptr<float>(frequencyBins), static_cast<const uint8_t*>(rowIndices.data()),
                                                            numberOfBatches, rowIndices.shape(0), numberOfElements, height,
                                                            decayFactor, stream),
                          "spectrogram kernel (row indices)");
    return hip_result(
        kernels::launch_spectrogram(ptr<float>(frequencyBins), ptr<const float>(input),
                                    input.offset(), numberOfBatches, numberOfElements, height,
                                    (int64_t)inputBatchStride, (int64_t)inputElementStride,
                                    decayFactor, stream),
        "spectrogram kernel");
}

Result Spectrogram::computeSubmitSpan(hipStream_t stream, U64 first_slot, U64 n) {
    const U64 ring = rowIndices.ringSlots();
    if (!spanCapable() || ring < 2 || first_slot >= ring) {
        JST_ERROR("[MODULE_SPECTROGRAM] A cycle-batched span needs the row-index ring of a batched spectrum unit.");
        return Result::ERROR;
    }
    // ONE launch whatever the span: cycle c reads ring slot (first_slot + c) mod ring (a span that wraps the ring, or laps it)
    JST_CHECK(hip_result(kernels::launch_spectrogram_index_span(ptr<float>(frequencyBins),
                                                                static_cast<const uint8_t*>(rowIndices.ringSlotData(0)),
                                                                numberOfBatches, rowIndices.shape(0), numberOfElements, height,
                                                                decayFactor, n, first_slot, ring, stream),
                         "spectrogram kernel (row indices, cycle-batched span)"));
    return rowIndices.ringSelect((first_slot + n - 1) % ring);
}

Result SpectrogramMerge::validate() {
    bool ok = true;
    totalBatches = ConfigU64(config_, "batches", 0, &ok);
    if (!ok || totalBatches == 0) {
        JST_ERROR("[MODULE_SPECTROGRAM_MERGE] Invalid batches value '%s': the batch count of ALL merged ranks.",
                  ConfigStr(config_, "batches", "?").c_str());
        return Result::ERROR;
    }
    if (!inputs_.count("counts")) return Result::SUCCESS;
    const Tensor& in = inputs_.at("counts");
    if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
    if (in.dtype() != DataType::U32 || in.rank() != 2 || !in.contiguous()) {
        JST_ERROR("[MODULE_SPECTROGRAM_MERGE] Input must be a contiguous U32 {width, height} hit-count tensor.");
        return Result::ERROR;
    }
    return Result::SUCCESS;
}
Result SpectrogramMerge::define() {
    JST_CHECK(defineTaint(SURFACE));
    return defineInterfaceInput("counts");
}
Result SpectrogramMerge::create() {
    counts = inputs_.at("counts");
    decayFactor = std::pow(0.999f, static_cast<F32>(totalBatches));  // module_impl.cc:104 on the merged batch count
    JST_CHECK(frequencyBins.create(device(), DataType::F32, {counts.shape(0), counts.shape(1)}));
    return Result::SUCCESS;
}
Result SpectrogramMerge::computeSubmit(hipStream_t stream) {
    return hip_result(kernels::launch_spectrogram_apply_counts(ptr<float>(frequencyBins), ptr<const uint32_t>(counts) + counts.offset(),
                                                               counts.size(), decayFactor, stream),
                      "spectrogram merge kernel");
}

Result Waterfall::validate() {
    return validate_surface_input("MODULE_WATERFALL", config_, 512, inputs_, height,
                                  numberOfElements, numberOfBatches, inputElementStride,
                                  inputBatchStride);
}
Result Waterfall::define() {
    JST_CHECK(defineTaint(SURFACE));
    return defineInterfaceInput("signal");
}
Result Waterfall::create() {
    input = inputs_.at("signal");
    JST_CHECK(frequencyBins.create(device(), DataType::F32, {height, numberOfElements}));
    JST_CHECK(ringState.create(device(), DataType::U64, {4}));  // zeroed: ringState = {}
    return Result::SUCCESS;
}
Result Waterfall::computeSubmit(hipStream_t stream) {
    return hip_result(
        kernels::launch_waterfall(ptr<float>(frequencyBins), ptr<uint64_t>(ringState),
                                  ptr<const float>(input), input.offset(), numberOfBatches,
                                  numberOfElements, height, (int64_t)inputBatchStride,
                                  (int64_t)inputElementStride, stream),
        "waterfall kernel");
}

// ---- RingSource --------------------------------------------------------------------------------
Result RingSource::validate() {
    bool o1 = true, o2 = true, o3 = true;
    batches = ConfigU64(config_, "batches", 8, &o1);
    samples = ConfigU64(config_, "samples", 2048, &o2);
    slots = ConfigU64(config_, "slots", 1, &o3);
    if (!o1 || !o2 || !o3 || batches == 0 || samples == 0 || slots == 0) {
        JST_ERROR("[MODULE_RING_SOURCE] batches, samples and slots must be positive integers.");
        return Result::ERROR;
    }
    bool o4 = true, o5 = true;
    live = ConfigBool(config_, "live", false, &o4);
    publishedConfig = ConfigU64(config_, "published", 0, &o5);
    if (!o4 || !o5) {
        JST_ERROR("[MODULE_RING_SOURCE] Invalid live / published value.");
        return Result::ERROR;
    }
    const std::string dt = ConfigStr(config_, "dtype", "CF32");
    if (dt == "CF32") sampleType = DataType::CF32;
    else if (dt == "CI16") sampleType = DataType::CI16;
    else if (dt == "CI8") sampleType = DataType::CI8;
    else if (dt == "CU8") sampleType = DataType::CU8;
    else {
        JST_ERROR("[MODULE_RING_SOURCE] Unsupported sample format '%s' (CF32, CI16, CI8, CU8).", dt.c_str());
        return Result::ERROR;
    }
    const std::string policy = ConfigStr(config_, "overflow", "overwrite");
    if (policy != "overwrite" && policy != "reject") {
        JST_ERROR("[MODULE_RING_SOURCE] Invalid overflow policy '%s' (overwrite, reject).", policy.c_str());
        return Result::ERROR;
    }
    rejectOnOverflow = policy == "reject";
    return Result::SUCCESS;
}
Result RingSource::reconfigureImpl(const Config& previous) {  // the counters move in place, the geometry does not
    if (ConfigU64(previous, "batches", 8) != batches || ConfigU64(previous, "samples", 2048) != samples ||
        ConfigU64(previous, "slots", 1) != slots || ConfigBool(previous, "live", false) != live ||
        ConfigStr(previous, "dtype", "CF32") != ConfigStr(config_, "dtype", "CF32") ||
        ConfigStr(previous, "overflow", "overwrite") != ConfigStr(config_, "overflow", "overwrite"))
        return Result::RECREATE;
    return Result::SUCCESS;
}
Result RingSource::define() { return defineInterfaceOutput("buffer"); }
Result RingSource::create() {
    JST_CHECK(output.createRing(device(), sampleType, {batches, samples}, slots));
    JST_CHECK(SetSignalAxes(output, {.sample = Index{1}, .batch = Index{0}}));
    output.setAttribute("sampleRate", AttrValue{ConfigF64(config_, "sampleRate", 2.0e6)});
    output.setAttribute("frequency", AttrValue{ConfigF64(config_, "frequency", 96.9e6)});
    elementBytes = DataTypeSize(sampleType);
    cursor = 0;
    consumed = 0;
    pushed = 0;
    first = true;
    overflowCount = 0;
    stagingIndex = stagingFill = 0;
    pendingFreeSlot = -1;
    lastComputeStream = nullptr;
    produced("buffer", output);
    return Result::SUCCESS;
}
RingSource::~RingSource() { (void)destroy(); }
Result RingSource::destroy() {
    std::lock_guard<std::mutex> lock(mu);
    if (uploadStream) (void)hipStreamSynchronize(uploadStream);
    for (U64 i = 0; i < kStaging; ++i) {
        if (staging[i]) (void)hipHostFree(staging[i]);
        if (stagingFree[i]) (void)hipEventDestroy(stagingFree[i]);
        staging[i] = nullptr;
        stagingFree[i] = nullptr;
        stagingBusy[i] = false;
    }
    for (hipEvent_t e : slotUploaded) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : slotFree) if (e) (void)hipEventDestroy(e);
    slotUploaded.clear();
    slotFree.clear();
    slotUploadValid.clear();
    slotFreeValid.clear();
    if (uploadStream) (void)hipStreamDestroy(uploadStream);
    uploadStream = nullptr;
    return Result::SUCCESS;
}
Result RingSource::computeSubmit(hipStream_t stream) {
    if (live) {
        std::lock_guard<std::mutex> lock(mu);
        lastComputeStream = stream;
        if (pendingFreeSlot >= 0 && !slotFree.empty()) {
            // everything the previous cycle enqueued precedes this point of the stream: its slot is free behind it
            JST_HIP_CHECK(hipEventRecord(slotFree[(size_t)pendingFreeSlot], stream), "hipEventRecord");
            slotFreeValid[(size_t)pendingFreeSlot] = 1;
            pendingFreeSlot = -1;
        }
        if (consumed >= published()) return Result::YIELD;  // nothing new from the host: no cycle
        const U64 slot = consumed % slots;
        if (!slotUploaded.empty() && slotUploadValid[slot])  // the batch's H2D copy (upload stream) before its readers
            JST_HIP_CHECK(hipStreamWaitEvent(stream, slotUploaded[slot], 0), "hipStreamWaitEvent");
        ++consumed;
        if (!slotFree.empty()) {
            slotFreeValid[slot] = 0;
            pendingFreeSlot = (I64)slot;
        }
        return output.ringSelect(slot);
    }
    // First cycle exposes slot 0, then round-robin; no data moves.
    if (first) first = false;
    else cursor = (cursor + 1) % slots;
    return output.ringSelect(cursor);
}

// Every unit of the cycle that consumed `pendingFreeSlot` is on the stream now: an event recorded here is behind all of
// the slot's readers (recorded from the producer thread it could land in front of kernels the compute thread had not
// enqueued yet -- the upload then overwrote the slot under its own cycle).
void RingSource::cycleSubmitted(hipStream_t stream) {
    if (!live) return;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (pendingFreeSlot < 0 || slotFree.empty()) return;
        if (hipEventRecord(slotFree[(size_t)pendingFreeSlot], stream) != hipSuccess) return;  // the next cycle records it
        slotFreeValid[(size_t)pendingFreeSlot] = 1;
        pendingFreeSlot = -1;
    }
    cycleClosed.notify_all();
}
Result RingSource::computeDeinitialize() {  // the runtime (and its stream) go away; producers may keep pushing
    std::lock_guard<std::mutex> lock(mu);
    lastComputeStream = nullptr;
    if (pendingFreeSlot >= 0 && !slotFree.empty()) {
        // the runtime synchronised its stream before tearing down: the slot's last readers are done
        slotFreeValid[(size_t)pendingFreeSlot] = 0;
        pendingFreeSlot = -1;
    }
    cycleClosed.notify_all();
    return Result::SUCCESS;
}

void RingSource::advanceHostState(U64 cycles) {
    if (live || cycles == 0) return;
    if (first) {  // the first cycle exposes slot 0 without moving
        first = false;
        --cycles;
    }
    cursor = (cursor + cycles) % slots;
    (void)output.ringSelect(cursor);
}

// ---- RingSource: producer side -------------------------------------------------------------------
Result RingSource::ensureProducer() {  // mu held
    if (!live) {
        JST_ERROR("[MODULE_RING_SOURCE] The producer interface needs a live source (config live = true).");
        return Result::ERROR;
    }
    if (uploadStream) return Result::SUCCESS;
    JST_HIP_CHECK(hipStreamCreateWithFlags(&uploadStream, hipStreamNonBlocking), "hipStreamCreate");
    const size_t batch_bytes = (size_t)(batches * samples) * elementBytes;
    for (U64 i = 0; i < kStaging; ++i) {
        JST_HIP_CHECK(hipHostMalloc(&staging[i], batch_bytes, hipHostMallocDefault), "hipHostMalloc");
        JST_HIP_CHECK(hipEventCreateWithFlags(&stagingFree[i], hipEventDisableTiming), "hipEventCreate");
        stagingBusy[i] = false;
    }
    slotUploaded.assign(slots, nullptr);
    slotFree.assign(slots, nullptr);
    slotUploadValid.assign(slots, 0);
    slotFreeValid.assign(slots, 0);
    for (U64 s = 0; s < slots; ++s) {
        JST_HIP_CHECK(hipEventCreateWithFlags(&slotUploaded[s], hipEventDisableTiming), "hipEventCreate");
        JST_HIP_CHECK(hipEventCreateWithFlags(&slotFree[s], hipEventDisableTiming), "hipEventCreate");
    }
    return Result::SUCCESS;
}

Result RingSource::ringAcquire(void** ptr, U64* max_elements) {
    if (!ptr || !max_elements) return Result::ERROR;
    std::unique_lock<std::mutex> lock(mu);
    JST_CHECK(ensureProducer());
    if (stagingFill == 0 && stagingBusy[stagingIndex]) {  // the copy that last left this staging buffer must be done
        // wait WITHOUT the lock: computeSubmit takes it, and a compute thread must not stall behind a producer that
        // is waiting for PCIe (one producer thread is assumed, like the reference's Soapy thread)
        const U64 index = stagingIndex;
        hipEvent_t done = stagingFree[index];
        lock.unlock();
        JST_HIP_CHECK(hipEventSynchronize(done), "hipEventSynchronize");
        lock.lock();
        if (index == stagingIndex) stagingBusy[index] = false;
    }
    *ptr = static_cast<char*>(staging[stagingIndex]) + stagingFill * elementBytes;
    *max_elements = batches * samples - stagingFill;
    return Result::SUCCESS;
}

Result RingSource::publishStagedBatch(std::unique_lock<std::mutex>& lock) {  // mu held; the staging buffer holds one whole batch
    bool overflowed = false, dropped = false;
    if (published() - consumed >= slots) {
        // every slot holds a published batch no cycle has consumed
        ++overflowCount;
        overflowed = true;
        if (rejectOnOverflow) return Result::INCOMPLETE;
        ++consumed;  // OverwriteOldest (circular_buffer.cc:151-161): the oldest unconsumed batch is dropped
        dropped = true;
    }
    U64 slot = published() % slots;
    // The cycle that consumed this slot last must have finished before the copy lands.  If that cycle is still being
    // enqueued (its free event is pending), the compute thread records it in cycleSubmitted -- microseconds away; wait
    // for that (without the lock).  Only a cycle that died half way leaves the slot pending: after the timeout its
    // completion is recorded here, behind whatever it did enqueue.
    while (pendingFreeSlot == (I64)slot) {
        const U64 epoch = clearEpoch, fill = stagingFill, index = stagingIndex, before = published();
        cycleClosed.wait_for(lock, std::chrono::milliseconds(200), [&] { return pendingFreeSlot != (I64)slot; });
        // `mu` was released while waiting: a ringClear() has dropped the staged batch with everything else, a second producer
        // thread may have published this staging buffer itself -- in both cases there is nothing left to upload from here,
        // and what this call counted for a batch it does not publish is taken back.
        if (clearEpoch != epoch || stagingFill != fill || stagingIndex != index) {
            if (clearEpoch == epoch) {  // (a clear reset the counters itself)
                if (dropped) --consumed;
                if (overflowed) --overflowCount;
            }
            return Result::SUCCESS;
        }
        // Only the published count moved (a host raised `published` through reconfigure while this batch was waiting): the
        // batch is still ours to upload -- into the slot the count names NOW (returning here left stagingFill at a
        // whole batch, and the producer's next acquire handed out zero elements).
        if (published() != before) {
            slot = published() % slots;
            continue;
        }
        if (pendingFreeSlot == (I64)slot && lastComputeStream) {
            JST_HIP_CHECK(hipEventRecord(slotFree[slot], lastComputeStream), "hipEventRecord");
            slotFreeValid[slot] = 1;
            pendingFreeSlot = -1;
        }
        break;
    }
    if (slotFreeValid[slot]) JST_HIP_CHECK(hipStreamWaitEvent(uploadStream, slotFree[slot], 0), "hipStreamWaitEvent");
    const size_t batch_bytes = (size_t)(batches * samples) * elementBytes;
    JST_HIP_CHECK(hipMemcpyAsync(output.ringSlotData(slot), staging[stagingIndex], batch_bytes, hipMemcpyHostToDevice,
                                 uploadStream),
                  "hipMemcpyAsync(H2D)");
    JST_HIP_CHECK(hipEventRecord(stagingFree[stagingIndex], uploadStream), "hipEventRecord");
    JST_HIP_CHECK(hipEventRecord(slotUploaded[slot], uploadStream), "hipEventRecord");
    stagingBusy[stagingIndex] = true;
    slotUploadValid[slot] = 1;
    ++pushed;
    stagingIndex = (stagingIndex + 1) % kStaging;
    stagingFill = 0;
    return Result::SUCCESS;
}

Result RingSource::ringCommit(U64 elements) {
    Result r = Result::SUCCESS;
    {
        std::unique_lock<std::mutex> lock(mu);
        JST_CHECK(ensureProducer());
        const U64 batch = batches * samples;
        if (elements > batch - stagingFill) {
            JST_ERROR("[MODULE_RING_SOURCE] commit of %llu elements exceeds the %llu acquired.",
                      (unsigned long long)elements, (unsigned long long)(batch - stagingFill));
            return Result::ERROR;
        }
        stagingFill += elements;
        if (stagingFill == batch) {
            r = publishStagedBatch(lock);
            if (r == Result::INCOMPLETE) stagingFill = batch - elements;  // rejected: the chunk was not taken
        }
    }
    dataAvailable.notify_all();
    return r;
}

namespace {
// The producer's copy into PINNED staging memory: the bytes are written once by the CPU and read once by the DMA engine,
// so they should not pass through the cache -- non-temporal 16-byte stores (SSE2: baseline x86-64), four per 64-byte line,
// skip the read-for-ownership of every destination line that a plain memcpy of a 64 KiB chunk pays.  The sfence orders
// them before the commit that lets the upload start.  Small or misaligned pieces go through memcpy.
void staging_copy(void* dst, const void* src, size_t bytes) {
#if defined(__SSE2__)
    if (bytes >= 4096 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        auto* d = static_cast<__m128i*>(dst);
        const auto* s = static_cast<const __m128i*>(src);
        const size_t lines = bytes / 64;
        for (size_t i = 0; i < lines; ++i) {
            const __m128i a = _mm_loadu_si128(s + 4 * i), b = _mm_loadu_si128(s + 4 * i + 1);
            const __m128i c = _mm_loadu_si128(s + 4 * i + 2), e = _mm_loadu_si128(s + 4 * i + 3);
            _mm_stream_si128(d + 4 * i, a);
            _mm_stream_si128(d + 4 * i + 1, b);
            _mm_stream_si128(d + 4 * i + 2, c);
            _mm_stream_si128(d + 4 * i + 3, e);
        }
        _mm_sfence();
        const size_t done = lines * 64;
        if (done < bytes) std::memcpy(static_cast<char*>(dst) + done, static_cast<const char*>(src) + done, bytes - done);
        return;
    }
#endif
    std::memcpy(dst, src, bytes);
}
}  // namespace

Result RingSource::ringPush(const void* data, U64 elements) {
    if (elements > 0 && !data) return Result::ERROR;
    const U64 batch = batches * samples;
    if (rejectOnOverflow) {  // Reject takes all of a push or nothing (circular_buffer.cc:143-149)
        std::lock_guard<std::mutex> lock(mu);
        JST_CHECK(ensureProducer());
        const U64 completing = (stagingFill + elements) / batch;
        if (published() - consumed + completing > slots) {
            ++overflowCount;
            return elements > ringCapacity() ? Result::ERROR : Result::INCOMPLETE;
        }
    }
    const char* src = static_cast<const char*>(data);
    while (elements > 0) {
        void* dst = nullptr;
        U64 room = 0;
        JST_CHECK(ringAcquire(&dst, &room));
        const U64 n = elements < room ? elements : room;
        staging_copy(dst, src, (size_t)n * elementBytes);
        const Result r = ringCommit(n);
        if (r != Result::SUCCESS) return r;
        src += (size_t)n * elementBytes;
        elements -= n;
    }
    return Result::SUCCESS;
}

U64 RingSource::ringSize() {
    std::lock_guard<std::mutex> lock(mu);
    return (published() - consumed) * batches * samples + stagingFill;
}
U64 RingSource::ringOverflows() {
    std::lock_guard<std::mutex> lock(mu);
    return overflowCount;
}
Result RingSource::ringWait(U64 elements, U32 timeout_ms) {  // circular_buffer.cc waitForSize
    std::unique_lock<std::mutex> lock(mu);
    if (elements > ringCapacity() + batches * samples) return Result::ERROR;
    const bool ok = dataAvailable.wait_for(lock, std::chrono::milliseconds(timeout_ms), [&] {
        return (published() - consumed) * batches * samples + stagingFill >= elements;
    });
    return ok ? Result::SUCCESS : Result::TIMEOUT;
}
Result RingSource::ringClear() {
    std::lock_guard<std::mutex> lock(mu);
    if (uploadStream) JST_HIP_CHECK(hipStreamSynchronize(uploadStream), "hipStreamSynchronize");
    consumed = published();
    stagingFill = 0;
    overflowCount = 0;
    ++clearEpoch;  // a producer that waits inside publishStagedBatch (lock released) sees that its batch is gone
    for (U64 i = 0; i < kStaging; ++i) stagingBusy[i] = false;
    // nothing is published any more: no slot is waiting for a consumer's completion, no upload is outstanding
    pendingFreeSlot = -1;
    std::fill(slotFreeValid.begin(), slotFreeValid.end(), 0);
    std::fill(slotUploadValid.begin(), slotUploadValid.end(), 0);
    return Result::SUCCESS;
}

// ---- fusion ------------------------------------------------------------------------------------
bool TryFuseSpectrum(const std::vector<Module*>& ordered, size_t at, std::string& name,
                     std::vector<Module*>& members, std::function<Result(hipStream_t)>& submit,
                     size_t& consumed, bool allow_combine, std::function<Result(hipStream_t)>* flush, bool allow_side,
                     SpanSupport* batch, const std::set<const void*>* static_storage) {
    if (at + 2 >= ordered.size()) return false;
    auto* mul = dynamic_cast<Multiply*>(ordered[at]);
    auto* fft = dynamic_cast<Fft*>(ordered[at + 1]);
    auto* amp = dynamic_cast<Amplitude*>(ordered[at + 2]);
    if (!mul || !fft || !amp || std::string(mul->type()) != "multiply") return false;
    Range* rng = at + 3 < ordered.size() ? dynamic_cast<Range*>(ordered[at + 3]) : nullptr;

    // dataflow: mul.c -> fft.input, fft.output -> amp.input [, amp.output -> rng.input]
    if (fft->input.storageId() != mul->c.storageId() ||
        amp->input.storageId() != fft->output.storageId())
        return false;
    if (rng && rng->input.storageId() != amp->output.storageId()) rng = nullptr;
    if (mul->c.dtype() != DataType::CF32 || !fft->forward) return false;
    // intermediates must be plain dense tensors nobody else reads
    auto sole_consumer = [&](const Tensor& t, const Module* consumer) {
        for (const Module* m : ordered) {
            if (m == consumer) continue;
            if (const auto* c = dynamic_cast<const Cast*>(m); c && c->bypass) continue;  // a pure alias reads nothing
            for (const auto& kv : m->inputs())
                if (kv.second.storageId() == t.storageId()) return false;
        }
        return true;
    };
    if (!sole_consumer(mul->c, fft) || !sole_consumer(fft->output, amp)) return false;
    if (rng && !sole_consumer(amp->output, rng)) rng = nullptr;

    // geometry: transform along the LAST axis of dense tensors; window broadcast over all outer
    // axes (stride 0) and equal to the signal only along the transform axis.
    const Tensor& sig = mul->a;
    const Tensor& win = mul->b;
    const Index axis = fft->resolvedAxis;
    const U64 n = sig.shape(axis);
    const bool tiled = !kernels::fft_fused_supported(n);
    if (tiled && (kernels::fft_lds_supported(n) || !kernels::fft_tiled_supported(n) ||
                  kernels::fft_bluestein_size(n) != 0))
        return false;
    if (!fft->input.contiguous() || !fft->output.contiguous() || !amp->output.contiguous())
        return false;
    for (Index ax = 0; ax < win.rank(); ++ax)
        if (ax != axis && win.shape(ax) != 1 && win.stride(ax) != 0) return false;
    if (amp->normalizationSize != n) return false;
    if (sig.rank() - 1 > (Index)dev::kMaxOuterRank) return false;

    // one arithmetic flavour per fused kernel: both "generic" (libm-exact) or both "fast"
    const bool fast = amp->provider() == "fast";
    if (rng && (rng->provider() == "fast") != fast) return false;
    Tensor final_out = rng ? rng->output : amp->output;
    if (!final_out.contiguous()) return false;

    // provider "fast" with Spectrogram consumers: their heights guard the bin edges (device_math.hh)
    float guard[2] = {0.0f, 0.0f};
    if (fast && rng) {
        int found = 0;
        for (const Module* m : ordered) {
            const auto* spec = dynamic_cast<const Spectrogram*>(m);
            if (!spec) continue;
            for (const auto& kv : m->inputs()) {
                if (kv.second.storageId() != rng->output.storageId()) continue;
                const float h = (float)spec->height;
                if (h == guard[0] || h == guard[1]) continue;
                if (found == 2) return false;  // three different quantisers: leave the chain unfused
                guard[found++] = h;
            }
        }
    }
    const float guard0 = guard[0], guard1 = guard[1];

    // Raw SDR samples: a Cast (CI16 / CI8 / CU8 -> CF32, cast/module_impl.cc:49-70) whose output feeds ONLY this Multiply
    // folds into the transform's first load -- the unit reads the cast's input (4 or 2 bytes per sample instead of 8)
    // and the cast module launches nothing.  It need not be adjacent in the order (the window chain usually sits in
    // between): its own unit stays, as a no-op.  LDS kernels only (n <= 16384), not with the combined spectrogram.
    Cast* cast = nullptr;
    if (!tiled) {
        for (size_t i = 0; i < at; ++i) {
            auto* c = dynamic_cast<Cast*>(ordered[i]);
            if (!c || c->bypass || c->output.storageId() != sig.storageId()) continue;
            const DataType it = c->input.dtype();
            const bool raw_format = it == DataType::CI16 || it == DataType::CI8 || it == DataType::CU8;
            if (raw_format && c->outputDtype == DataType::CF32 && c->input.contiguous() && c->output.contiguous() &&
                sig.contiguous() && sig.offset() == c->output.offset() && sig.shape() == c->input.shape() &&
                sole_consumer(c->output, mul))
                cast = c;
        }
    }

    members = {mul, fft, amp};
    if (rng) members.push_back(rng);
    consumed = members.size();
    name = "spectrum_fused(" + mul->name() + "+" + fft->name() + "+" + amp->name() +
           (rng ? "+" + rng->name() : "") + ")";

    // A Spectrogram that follows in the order and is the ONLY reader of the fused output rides on the next cycle's
    // launch (kernels::launch_spectrum_spectrogram_fused): the output becomes a ring of two slots, this unit takes the
    // module in, and `flush` runs the spectrogram that is still waiting when a compute call ends.
    static const bool no_combine = std::getenv("JST_NO_SPECTROGRAM_COMBINE") != nullptr;
    if (cast) {
        cast->fusedIntoSpectrum = true;
        name = "spectrum_fused(" + cast->name() + "+" + mul->name() + "+" + fft->name() + "+" + amp->name() +
               (rng ? "+" + rng->name() : "") + ")";
    }
    if (allow_combine && flush && !tiled && !no_combine && !cast && at + consumed < ordered.size()) {
        auto* spec = dynamic_cast<Spectrogram*>(ordered[at + consumed]);
        Tensor& out = rng ? rng->output : amp->output;
        FftLayout L;
        std::memset(&L, 0, sizeof(L));
        if (spec && sig.rank() == 2 && axis == 1) {
            L.transforms = sig.shape(0);
            L.outer_rank = 1;
            L.outer_shape[0] = sig.shape(0);
            L.in_outer_stride[0] = (int64_t)sig.stride(0);
            L.out_outer_stride[0] = (int64_t)out.stride(0);
            L.in_axis_stride = (int64_t)sig.stride(1);
            L.out_axis_stride = (int64_t)out.stride(1);
        }
        if (spec && sig.rank() == 2 && axis == 1 && spec->input.storageId() == out.storageId() &&
            sole_consumer(out, spec) && out.offset() == 0 && spec->input.offset() == 0 &&
            spec->numberOfElements == n && spec->numberOfBatches == sig.shape(0) && spec->inputElementStride == 1 &&
            spec->inputBatchStride == n && (out.ringSlots() == 1 || out.ringSlots() == 2) &&
            kernels::spectrum_spectrogram_supported(n, L, (int64_t)win.stride(axis), spec->height)) {
            bool ok = out.ringSlots() == 2 || out.promoteToRing(2) == Result::SUCCESS;
            ok = ok && spec->combineCtrl.create(DeviceType::HIP, DataType::U64, {1}) == Result::SUCCESS;
            if (ok) {
                spec->combined = true;
                spec->combinedPending = false;
                spec->combinedCycle = 0;
                members.push_back(spec);
                consumed = members.size();
                name = "spectrum_fused_spectrogram(" + mul->name() + "+" + fft->name() + "+" + amp->name() +
                       (rng ? "+" + rng->name() : "") + "+" + spec->name() + ")";
                submit = [mul, fft, amp, rng, spec, n, fast, guard0, guard1](hipStream_t stream) -> Result {
                    const Tensor& sig = mul->a;
                    const Tensor& win = mul->b;
                    Tensor& out = rng ? rng->output : amp->output;
                    const U64 k = spec->combinedCycle;  // cycle k writes slot k & 1, the spectrogram part reads the other
                    JST_CHECK(out.ringSelect(0));
                    float* const slot0 = static_cast<float*>(out.data());
                    JST_CHECK(out.ringSelect(1));
                    float* const slot1 = static_cast<float*>(out.data());
                    JST_CHECK(out.ringSelect(k & 1));
                    FftLayout L;
                    std::memset(&L, 0, sizeof(L));
                    L.transforms = sig.shape(0);
                    L.outer_rank = 1;
                    L.outer_shape[0] = sig.shape(0);
                    L.in_outer_stride[0] = (int64_t)sig.stride(0);
                    L.out_outer_stride[0] = (int64_t)out.stride(0);
                    L.in_axis_stride = (int64_t)sig.stride(1);
                    L.out_axis_stride = (int64_t)out.stride(1);
                    L.in_offset = sig.offset();
                    L.out_offset = 0;
                    const Result r = hip_result(
                        kernels::launch_spectrum_spectrogram_fused(
                            L, fft->twiddles, static_cast<const float2*>(sig.data()),
                            static_cast<const float2*>(win.data()) + win.offset(), (k & 1) ? slot1 : slot0,
                            amp->scalingCoeff, rng != nullptr, rng ? rng->scalingCoeff : 0.0f,
                            rng ? rng->offsetCoeff : 0.0f, fast, guard0, guard1, ptr<float>(spec->frequencyBins),
                            (k & 1) ? slot0 : slot1, spec->height, spec->decayFactor,
                            static_cast<uint32_t*>(spec->combineCtrl.data()), stream),
                        "fused spectrum + spectrogram kernel");
                    spec->combinedCycle = k + 1;
                    spec->combinedPending = true;
                    return r;
                };
                *flush = [rng, amp, spec, n](hipStream_t stream) -> Result {
                    if (!spec->combinedPending) return Result::SUCCESS;
                    Tensor& out = rng ? rng->output : amp->output;
                    JST_CHECK(out.ringSelect((spec->combinedCycle - 1) & 1));  // the handle shows the latest cycle
                    JST_CHECK(hip_result(kernels::launch_spectrogram(ptr<float>(spec->frequencyBins),
                                                                     static_cast<const float*>(out.data()), 0,
                                                                     spec->numberOfBatches, n, spec->height, (int64_t)n, 1,
                                                                     spec->decayFactor, stream),
                                         "spectrogram kernel (flush)"));
                    JST_HIP_CHECK(hipMemsetAsync(spec->combineCtrl.data(), 0, sizeof(uint32_t), stream), "hipMemsetAsync");
                    spec->combinedPending = false;
                    return Result::SUCCESS;
                };
                return true;
            }
        }
    }

    // Row indices for the Spectrogram: when exactly ONE Spectrogram quantises the whole range output (dense {batches, n}
    // rows, height <= 256, its own display -- not the counts of a multi-rank merge), the kernel writes the index that
    // module would derive from every value as one byte beside it (fft_side.hip) and the module reads those instead of
    // the values: 1 byte instead of 4 per sample on its side, a quarter of the wavefronts.  Not for the surfaces of a
    // pipelined runtime (they read one cycle behind).  JST_NO_SPECTROGRAM_SIDE=1 is the A/B switch.
    Spectrogram* fed = nullptr;
    static const bool no_side = std::getenv("JST_NO_SPECTROGRAM_SIDE") != nullptr;
    // (an output that is already a ring of the source's size was promoted by an earlier cycle-batched runtime over these
    // very modules: a runtime re-created on them plans the same way again)
    const U64 source_slots = (cast ? cast->input : sig).ringSlots();
    const bool out_plain_or_batched = rng && (rng->output.ringSlots() == 1 || (batch && source_slots > 1 && rng->output.ringSlots() == source_slots));
    if (allow_side && !no_side && rng && !tiled && sig.rank() == 2 && axis == 1 && out_plain_or_batched) {
        int readers = 0;
        Spectrogram* only = nullptr;
        for (Module* m : ordered) {
            auto* spec = dynamic_cast<Spectrogram*>(m);
            if (!spec) continue;
            for (const auto& kv : m->inputs())
                if (kv.second.storageId() == rng->output.storageId()) {
                    ++readers;
                    only = spec;
                }
        }
        FftLayout L;
        std::memset(&L, 0, sizeof(L));
        L.transforms = sig.shape(0);
        L.outer_rank = 1;
        L.outer_shape[0] = sig.shape(0);
        L.in_outer_stride[0] = (int64_t)sig.stride(0);
        L.out_outer_stride[0] = (int64_t)rng->output.stride(0);
        L.in_axis_stride = (int64_t)sig.stride(1);
        L.out_axis_stride = (int64_t)rng->output.stride(1);
        if (readers == 1 && !only->countsOnly && !only->combined && only->input.offset() == rng->output.offset() &&
            only->numberOfElements == n && only->numberOfBatches == sig.shape(0) && only->inputElementStride == 1 &&
            only->inputBatchStride == n &&
            kernels::spectrum_side_supported(n, L, (int64_t)win.stride(axis), only->height) &&
            kernels::spectrogram_index_supported(only->numberOfBatches, n, only->height) &&
            (!cast || sig.stride(1) == 1) &&
            ((only->rowIndices.valid() && only->rowIndices.shape() == Shape{kernels::spectrum_side_pitch(sig.shape(0)), n} &&
              only->rowIndices.ringSlots() == rng->output.ringSlots()) ||
             only->rowIndices.create(DeviceType::HIP, DataType::U8, {kernels::spectrum_side_pitch(sig.shape(0)), n}) == Result::SUCCESS) &&
            (only->schedWords.valid() ||
             only->schedWords.create(DeviceType::HIP, DataType::U32, {kernels::spectrum_sched_words()}) == Result::SUCCESS)) {
            fed = only;
            fed->indexFed = true;
            name += "+indices";
        }
    }

    // Cycle batching: the signal is a dense, offset-free view of a RESIDENT ring (slots one behind the other in one
    // allocation, tensor.cc: createRing), so the transforms of n consecutive slots are one dense {n * batches, samples}
    // problem for the persistent kernel -- its ramp, cold start and tail are paid once per launch instead of once per
    // cycle.  `prepare` (called by Runtime::planBatch once every dynamic unit of the runtime can be batched) turns the
    // range output and the Spectrogram's row indices into rings of as many slots; from then on the per-cycle submit
    // below writes the slot the source exposes, and submit_span covers runs of consecutive slots.
    // Is the window REAL (every imaginary part +-0)?  Provider "fast" then multiplies it as two products per sample
    // (kernels.hh: launch_spectrum_fused_side, real_window).  The table is a STATIC tensor that the window chain fills in
    // the first, eager cycle, on this very stream in front of this unit: the first submission outside a capture waits for
    // it, reads it back once (n complex values) and remembers the answer; until then the full product is used.
    // Only an operand that a statically SETTLED unit produced keeps its first answer for the lifetime of this unit: any
    // other producer may hand over imaginary parts in a later cycle, and the real-operand kernel would drop them silently.
    const bool operand_static = static_storage && static_storage->count(mul->b.storageId()) != 0;
    auto window_real = std::make_shared<int>(operand_static ? -1 : 0);
    auto know_window = [mul, n, axis, window_real](hipStream_t stream) -> bool {
        if (*window_real >= 0) return *window_real == 1;
        const Tensor& win = mul->b;
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
        if (win.stride(axis) != 1 || hipStreamIsCapturing(stream, &status) != hipSuccess ||
            status != hipStreamCaptureStatusNone)
            return false;
        std::vector<float2> host(n);
        if (hipStreamSynchronize(stream) != hipSuccess ||
            hipMemcpy(host.data(), static_cast<const float2*>(win.data()) + win.offset(), n * sizeof(float2),
                      hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        bool real = true;
        for (const float2& w : host) {
            uint32_t bits;
            std::memcpy(&bits, &w.y, sizeof(bits));
            real &= (bits & 0x7fffffffu) == 0u;
        }
        *window_real = real ? 1 : 0;
        return real;
    };

    auto batched = std::make_shared<bool>(false);
    if (batch && fed) {
        const Tensor in_t = cast ? cast->input : sig;
        const U64 slot_elems = sig.shape(0) * n;
        const bool dense_ring = in_t.ringSlots() > 1 && in_t.contiguous() && in_t.offset() == 0 && in_t.size() == slot_elems &&
                                rng->output.offset() == 0 && rng->output.size() == slot_elems;
        // per launch: < 2^31 bytes of input (buffer-descriptor offsets), i.e. at most max_run slots
        const U64 in_bytes = slot_elems * DataTypeSize(in_t.dtype());
        const U64 max_run = in_bytes ? ((1ull << 31) - 1) / in_bytes : 0;
        // (the span launch addresses the whole ring: kernels::spectrum_side_supported bounds it at 2^28 elements)
        if (dense_ring && max_run >= 2 && in_t.ringSlots() * slot_elems < (1ull << 28)) {
            batch->phase = in_t;
            batch->prepare = [rng, fed, batched](U64 slots) -> Result {
                if (rng->output.ringSlots() != slots) JST_CHECK(rng->output.promoteToRing(slots));
                if (fed->rowIndices.ringSlots() != slots) JST_CHECK(fed->rowIndices.promoteToRing(slots));
                if (fed->input.ringSlots() != slots) {
                    JST_ERROR("[RUNTIME] The Spectrogram's input did not follow the spectrum output's ring.");
                    return Result::ERROR;
                }
                *batched = true;
                return Result::SUCCESS;
            };
            batch->rings = {rng->output, fed->rowIndices};
            batch->submit_span = [mul, fft, amp, rng, cast, fed, n, fast, guard0, guard1, max_run, know_window](hipStream_t stream, U64 first,
                                                                                                  U64 cycles) -> Result {
                const Tensor& sig = mul->a;
                const Tensor& win = mul->b;
                const Tensor in_t = cast ? cast->input : sig;
                Tensor& out = rng->output;
                const U64 ring = in_t.ringSlots();
                if (first >= ring || out.ringSlots() != ring || fed->rowIndices.ringSlots() != ring) {
                    JST_ERROR("[RUNTIME] Batched spectrum span: the output rings do not match the source ring.");
                    return Result::ERROR;
                }
                const DataType it = in_t.dtype();
                // ONE launch whatever the span: transform t is row (first * batches + t) mod (ring * batches) of the ring
                // tensors (FftLayout::ring_*) -- a span that wraps the ring, or laps it, included
                (void)max_run;
                FftLayout L;
                std::memset(&L, 0, sizeof(L));
                L.transforms = sig.shape(0) * cycles;
                L.outer_rank = 1;
                L.outer_shape[0] = L.transforms;
                L.in_outer_stride[0] = (int64_t)n;
                L.out_outer_stride[0] = (int64_t)n;
                L.in_axis_stride = 1;
                L.out_axis_stride = 1;
                L.ring_first = first * sig.shape(0);
                L.ring_transforms = ring * sig.shape(0);
                JST_CHECK(hip_result(
                    kernels::launch_spectrum_fused_side(
                        n, L, fft->twiddles, in_t.ringSlotData(0),
                        !cast ? 0 : (it == DataType::CI16 ? 1 : (it == DataType::CI8 ? 2 : 3)), cast ? cast->scaler : 1.0f,
                        static_cast<const float2*>(win.data()) + win.offset(), static_cast<float*>(out.ringSlotData(0)),
                        amp->scalingCoeff, rng->scalingCoeff, rng->offsetCoeff, fast, guard0, guard1,
                        static_cast<uint8_t*>(fed->rowIndices.ringSlotData(0)), fed->height, sig.shape(0),
                        fed->rowIndices.shape(0), fast && know_window(stream), stream,
                        static_cast<uint32_t*>(fed->schedWords.data())),
                    "fused spectrum kernel (+ row indices, cycle-batched span)"));
                const U64 slot = (first + cycles) % ring;
                const U64 last = (slot + ring - 1) % ring;
                JST_CHECK(out.ringSelect(last));
                return fed->rowIndices.ringSelect(last);
            };
        }
    }

    // Cycle batching for the TILED unit (beyond 16384 points / mixed radix: config 5's 65536-point transforms, whose three
    // launches per cycle of 1 Mi samples are launch bound): the transforms of a run of consecutive ring slots are one dense
    // problem for the columns / blocks kernel pair too.  No ring arithmetic in those kernels: a span that wraps is two runs;
    // the scratch image grows to a whole ring's transforms.  Surfaces behind it (lineplot, waterfall) ride as sinks.
    if (batch && !fed && tiled && !cast && sig.rank() == 2 && axis == 1) {
        const Tensor in_t = sig;
        Tensor& out_t = rng ? rng->output : amp->output;
        const U64 slot_elems = sig.shape(0) * n;
        const bool dense_ring = in_t.ringSlots() > 1 && in_t.contiguous() && in_t.offset() == 0 && in_t.size() == slot_elems &&
                                out_t.offset() == 0 && out_t.size() == slot_elems && win.stride(axis) <= 1 &&
                                (out_t.ringSlots() == 1 || out_t.ringSlots() == in_t.ringSlots()) &&
                                in_t.ringSlots() * slot_elems < (1ull << 28);
        if (dense_ring) {
            batch->phase = in_t;
            batch->prepare = [fft, amp, rng, batched, n, slot_elems](U64 slots) -> Result {
                Tensor& out = rng ? rng->output : amp->output;
                if (out.ringSlots() != slots) JST_CHECK(out.promoteToRing(slots));
                if (fft->scratchA.valid() && fft->scratchA.size() < slots * slot_elems)
                    JST_CHECK(fft->scratchA.create(DeviceType::HIP, DataType::CF32, {slots * slot_elems}));
                *batched = true;
                return Result::SUCCESS;
            };
            batch->rings = {out_t};
            batch->submit_span = [mul, fft, amp, rng, n, fast, guard0, guard1, axis](hipStream_t stream, U64 first, U64 cycles) -> Result {
                const Tensor& sig = mul->a;
                const Tensor& win = mul->b;
                Tensor& out = rng ? rng->output : amp->output;
                const U64 ring = sig.ringSlots();
                if (first >= ring || out.ringSlots() != ring) {
                    JST_ERROR("[RUNTIME] Batched spectrum span (tiled): the output ring does not match the source ring.");
                    return Result::ERROR;
                }
                U64 slot = first;
                while (cycles > 0) {
                    const U64 run = std::min<U64>(cycles, ring - slot);
                    FftLayout L;
                    std::memset(&L, 0, sizeof(L));
                    L.transforms = sig.shape(0) * run;
                    L.outer_rank = 1;
                    L.outer_shape[0] = L.transforms;
                    L.in_outer_stride[0] = (int64_t)n;
                    L.out_outer_stride[0] = (int64_t)n;
                    L.in_axis_stride = 1;
                    L.out_axis_stride = 1;
                    JST_CHECK(hip_result(
                        kernels::launch_spectrum_fused_tiled(
                            n, L, fft->twiddles, static_cast<const float2*>(sig.ringSlotData(slot)),
                            static_cast<const float2*>(win.data()) + win.offset(), (int64_t)win.stride(axis),
                            static_cast<float*>(out.ringSlotData(slot)), amp->scalingCoeff, rng != nullptr,
                            rng ? rng->scalingCoeff : 0.0f, rng ? rng->offsetCoeff : 0.0f, fast, guard0, guard1,
                            static_cast<float2*>(fft->scratchA.data()), stream),
                        "fused spectrum (tiled) kernel, cycle-batched span"));
                    slot = (slot + run) % ring;
                    cycles -= run;
                }
                return out.ringSelect((slot + ring - 1) % ring);
            };
        }
    }

    submit = [mul, fft, amp, rng, cast, fed, axis, n, fast, tiled, guard0, guard1, batched, know_window](hipStream_t stream) -> Result {
        const Tensor& sig = mul->a;
        const Tensor& win = mul->b;
        if (*batched) {  // cycle-batched runtime: this cycle writes the ring slot the source exposes
            const U64 slot = (cast ? cast->input : sig).ringSlot();
            JST_CHECK((rng ? rng->output : amp->output).ringSelect(slot));
            if (fed) JST_CHECK(fed->rowIndices.ringSelect(slot));
        }
        const Tensor& out = rng ? rng->output : amp->output;
        FftLayout L;
        std::memset(&L, 0, sizeof(L));
        L.transforms = 1;
        int r = 0;
        for (Index ax = 0; ax < sig.rank(); ++ax) {
            if (ax == axis) continue;
            L.outer_shape[r] = sig.shape(ax);
            L.in_outer_stride[r] = (int64_t)sig.stride(ax);
            L.out_outer_stride[r] = (int64_t)out.stride(ax);
            L.transforms *= sig.shape(ax);
            ++r;
        }
        L.outer_rank = r;
        L.in_axis_stride = (int64_t)sig.stride(axis);
        L.out_axis_stride = (int64_t)out.stride(axis);
        L.in_offset = sig.offset();
        L.out_offset = out.offset();
        if (tiled)  // mixed radix / beyond 16384 points: LDS-tiled kernels, same functors
            return hip_result(
                kernels::launch_spectrum_fused_tiled(
                    n, L, fft->twiddles, static_cast<const float2*>(sig.data()),
                    static_cast<const float2*>(win.data()) + win.offset(),
                    (int64_t)win.stride(axis), static_cast<float*>(out.data()), amp->scalingCoeff,
                    rng != nullptr, rng ? rng->scalingCoeff : 0.0f, rng ? rng->offsetCoeff : 0.0f,
                    fast, guard0, guard1, static_cast<float2*>(fft->scratchA.data()), stream),
                "fused spectrum (tiled) kernel");
        if (fed) {  // + the Spectrogram's row indices as a side output
            const DataType it = cast ? cast->input.dtype() : DataType::CF32;
            if (cast) L.in_offset = cast->input.offset();
            return hip_result(
                kernels::launch_spectrum_fused_side(
                    n, L, fft->twiddles, cast ? cast->input.data() : sig.data(),
                    !cast ? 0 : (it == DataType::CI16 ? 1 : (it == DataType::CI8 ? 2 : 3)), cast ? cast->scaler : 1.0f,
                    static_cast<const float2*>(win.data()) + win.offset(), static_cast<float*>(out.data()),
                    amp->scalingCoeff, rng->scalingCoeff, rng->offsetCoeff, fast, guard0, guard1,
                    static_cast<uint8_t*>(fed->rowIndices.data()), fed->height, sig.shape(0), fed->rowIndices.shape(0),
                    fast && know_window(stream), stream),
                "fused spectrum kernel (+ row indices)");
        }
        if (cast) {  // raw samples: same dense shape as the cast's output, element strides therefore equal
            const DataType it = cast->input.dtype();
            L.in_offset = cast->input.offset();
            return hip_result(
                kernels::launch_spectrum_fused_cast(
                    n, L, fft->twiddles, cast->input.data(), it == DataType::CI16 ? 1 : (it == DataType::CI8 ? 2 : 3),
                    cast->scaler, static_cast<const float2*>(win.data()) + win.offset(), (int64_t)win.stride(axis),
                    static_cast<float*>(out.data()), amp->scalingCoeff, rng != nullptr, rng ? rng->scalingCoeff : 0.0f,
                    rng ? rng->offsetCoeff : 0.0f, fast, guard0, guard1, stream),
                "fused spectrum kernel (raw sample input)");
        }
        return hip_result(
            kernels::launch_spectrum_fused(
                n, L, fft->twiddles, static_cast<const float2*>(sig.data()),
                static_cast<const float2*>(win.data()) + win.offset(), (int64_t)win.stride(axis),
                static_cast<float*>(out.data()), amp->scalingCoeff, rng != nullptr,
                rng ? rng->scalingCoeff : 0.0f, rng ? rng->offsetCoeff : 0.0f, fast, guard0, guard1, stream),
            "fused spectrum kernel");
    };
    return true;
}

// ---- registration ------------------------------------------------------------------------------
JST_REGISTER_MODULE(Window, "window", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Invert, "invert", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Reshape, "reshape", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Cast, "cast", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Add, "add", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Multiply, "multiply", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(MultiplyConstant, "multiply_constant", DeviceType::HIP, RuntimeType::NATIVE,
                    "generic");
JST_REGISTER_MODULE(Fft, "fft", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Amplitude, "amplitude", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Range, "range", DeviceType::HIP, RuntimeType::NATIVE, "generic");
// provider "fast": same modules, hardware-transcendental arithmetic (device_math.hh)
using AmplitudeFast = Amplitude;
using RangeFast = Range;
JST_REGISTER_MODULE(AmplitudeFast, "amplitude", DeviceType::HIP, RuntimeType::NATIVE, "fast");
JST_REGISTER_MODULE(RangeFast, "range", DeviceType::HIP, RuntimeType::NATIVE, "fast");
JST_REGISTER_MODULE(Spectrogram, "spectrogram", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(SpectrogramMerge, "spectrogram_merge", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Waterfall, "waterfall", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(RingSource, "ring_source", DeviceType::HIP, RuntimeType::NATIVE, "generic");

}  // namespace jst::modules
