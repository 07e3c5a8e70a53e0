// filter_modules.cc -- host side of the Filter (FFT overlap-add) and FM side-chain modules:
// pad, unpad, fold, overlap_add, phase_correction, filter_taps, arithmetic, expand_dims,
// squeeze_dims, duplicate, fm.  Same contract as modules.cc; kernels in kernels/filter_kernels.hip.
#include <cmath>
#include <cstring>

#include "modules.hh"

namespace jst::modules {

using dev::EwLayout;

namespace {

template <class T>
T* ptr(const Tensor& t) {
    return static_cast<T*>(t.data());
}
// src/memory/axis.cc:196-212 (ResolveAxis): negative axes count from the end.
std::optional<Index> resolve_axis(I64 axis, Index rank) {
    if (rank == 0) return std::nullopt;
    const I64 r = (I64)rank, a = axis < 0 ? r + axis : axis;
    if (a < 0 || a >= r) return std::nullopt;
    return (Index)a;
}
I64 config_i64(const Config& c, const std::string& key, I64 fallback, bool* ok) {
    *ok = true;
    auto it = c.find(key);
    if (it == c.end()) return fallback;
    char* end = nullptr;
    const long long v = std::strtoll(it->second.c_str(), &end, 10);
    if (end == it->second.c_str() || *end != '\0') *ok = false;
    return v;
}
bool parse_f64_list(const std::string& s, std::vector<F64>& out) {
    out.clear();
    std::string body = s;
    if (!body.empty() && body.front() == '[') {
        if (body.back() != ']') return false;
        body = body.substr(1, body.size() - 2);
    }
    size_t pos = 0;
    while (pos < body.size()) {
        while (pos < body.size() && (std::isspace((unsigned char)body[pos]) || body[pos] == ',')) ++pos;
        if (pos >= body.size()) break;
        char* end = nullptr;
        const double v = std::strtod(body.c_str() + pos, &end);
        if (end == body.c_str() + pos) return false;
        out.push_back(v);
        pos = (size_t)(end - body.c_str());
    }
    return true;
}
void split_axis(const Tensor& t, Index axis, U64& outer, U64& inner) {
    outer = inner = 1;
    for (Index i = 0; i < axis; ++i) outer *= t.shape(i);
    for (Index i = axis + 1; i < t.rank(); ++i) inner *= t.shape(i);
}
bool is_f32_or_cf32(const Tensor& t) { return t.dtype() == DataType::F32 || t.dtype() == DataType::CF32; }

}  // namespace

// ---- Pad (core/pad/module_impl.cc, module_impl_native_cpu.cc:75-140) ---------------------------
class Pad : public Module {
 public:
    const char* type() const override { return "pad"; }
    Result validate() override {
        bool ok1, ok2;
        size = ConfigU64(config_, "size", 0, &ok1);
        const I64 axis = config_i64(config_, "axis", -1, &ok2);
        if (!ok1 || !ok2) {
            JST_ERROR("[MODULE_PAD] Invalid size/axis.");
            return Result::ERROR;
        }
        if (!inputs_.count("unpadded")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("unpadded");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        const auto a = resolve_axis(axis, in.rank());
        if (!a) {
            JST_ERROR("[MODULE_PAD] Axis %lld out of range for tensor with %llu dimensions.",
                      (long long)axis, (unsigned long long)in.rank());
            return Result::ERROR;
        }
        if (!is_f32_or_cf32(in)) {
            JST_ERROR("[MODULE_PAD_NATIVE_HIP] Unsupported data type '%s'.", DataTypeName(in.dtype()));
            return Result::ERROR;
        }
        resolvedAxis = *a;
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(STATELESS));
        JST_CHECK(defineInterfaceInput("unpadded"));
        return defineInterfaceOutput("padded");
    }
    Result create() override {
        input = inputs_.at("unpadded");
        Shape os = input.shape();
        os[resolvedAxis] += size;
        JST_CHECK(output.create(device(), input.dtype(), os));
        JST_CHECK(output.propagateAttributes(input));
        produced("padded", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        U64 outer, inner;
        split_axis(input, resolvedAxis, outer, inner);
        const bool cx = input.dtype() == DataType::CF32;
        return hip_result(kernels::launch_pad(ptr<char>(output) + output.offsetBytes(),
                                              ptr<char>(input) + input.offsetBytes(), cx, outer,
                                              input.shape(resolvedAxis), output.shape(resolvedAxis),
                                              inner, s),
                          "pad kernel");
    }
    Tensor input, output;
    U64 size = 0;
    Index resolvedAxis = 0;
};

// ---- Unpad (core/unpad/module_impl.cc, module_impl_native_cpu.cc:66-135) -----------------------
class Unpad : public Module {
 public:
    const char* type() const override { return "unpad"; }
    Result validate() override {
        bool ok1, ok2;
        size = ConfigU64(config_, "size", 0, &ok1);
        const I64 axis = config_i64(config_, "axis", -1, &ok2);
        if (!ok1 || !ok2) {
            JST_ERROR("[MODULE_UNPAD] Invalid size/axis.");
            return Result::ERROR;
        }
        if (!inputs_.count("padded")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("padded");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        const auto a = resolve_axis(axis, in.rank());
        if (!a) {
            JST_ERROR("[MODULE_UNPAD] Axis %lld out of range for tensor with %llu dimensions.",
                      (long long)axis, (unsigned long long)in.rank());
            return Result::ERROR;
        }
        if (size > in.shape(*a)) {
            JST_ERROR("[MODULE_UNPAD] Size %llu exceeds axis dimension %llu.",
                      (unsigned long long)size, (unsigned long long)in.shape(*a));
            return Result::ERROR;
        }
        if (!is_f32_or_cf32(in)) {
            JST_ERROR("[MODULE_UNPAD_NATIVE_HIP] Unsupported data type '%s'.", DataTypeName(in.dtype()));
            return Result::ERROR;
        }
        resolvedAxis = *a;
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(STATELESS));
        JST_CHECK(defineInterfaceInput("padded"));
        JST_CHECK(defineInterfaceOutput("unpadded"));
        return defineInterfaceOutput("pad");
    }
    Result create() override {
        input = inputs_.at("padded");
        Shape bs = input.shape(), ts = input.shape();
        bs[resolvedAxis] = input.shape(resolvedAxis) - size;
        ts[resolvedAxis] = size;
        JST_CHECK(body.create(device(), input.dtype(), bs));
        JST_CHECK(tail.create(device(), input.dtype(), ts));
        JST_CHECK(body.propagateAttributes(input));
        JST_CHECK(tail.propagateAttributes(input));
        produced("unpadded", body);
        produced("pad", tail);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        U64 outer, inner;
        split_axis(input, resolvedAxis, outer, inner);
        return hip_result(kernels::launch_unpad(ptr<char>(body) + body.offsetBytes(),
                                                ptr<char>(tail) + tail.offsetBytes(),
                                                ptr<char>(input) + input.offsetBytes(),
                                                input.dtype() == DataType::CF32, outer,
                                                input.shape(resolvedAxis), body.shape(resolvedAxis),
                                                inner, s),
                          "unpad kernel");
    }
    Tensor input, body, tail;
    U64 size = 0;
    Index resolvedAxis = 0;
};

// ---- Fold (dsp/fold/module_impl.cc:16-172 region, module_impl_native_cpu.cc:103-172) -----------
class Fold : public Module {
 public:
    const char* type() const override { return "fold"; }
    Result validate() override {
        bool ok1, ok2;
        offset = ConfigU64(config_, "offset", 0, &ok1);
        size = ConfigU64(config_, "size", 0, &ok2);
        if (!ok1 || !ok2 || size == 0) {
            JST_ERROR("[MODULE_FOLD] Size cannot be zero.");
            return Result::ERROR;
        }
        channelOffsets.clear();
        channelAxis.reset();
        if (!inputs_.count("buffer")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("buffer");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        SignalAxes axes;
        if (ResolveSignalAxes(in, axes) != Result::SUCCESS) {
            JST_ERROR("[MODULE_FOLD] Input must contain valid signal axis metadata.");
            return Result::ERROR;
        }
        const U64 axisSize = in.shape(*axes.sample);
        if (axisSize % size != 0) {
            JST_ERROR("[MODULE_FOLD] Size (%llu) is not a divisor of the input shape (%llu) along "
                      "axis (%llu).",
                      (unsigned long long)size, (unsigned long long)axisSize,
                      (unsigned long long)*axes.sample);
            return Result::ERROR;
        }
        if (const AttrValue* v = in.attribute("channelOffsets")) {
            const auto* offs = std::get_if<std::vector<U64>>(v);
            if (!offs) {
                JST_ERROR("[MODULE_FOLD] Input channelOffsets metadata must have type vector<U64>.");
                return Result::ERROR;
            }
            if (offs->empty()) {
                JST_ERROR("[MODULE_FOLD] Input channelOffsets metadata cannot be empty.");
                return Result::ERROR;
            }
            channelOffsets = *offs;
        }
        if (channelOffsets.empty()) {
            if (axisSize < offset) {
                JST_ERROR("[MODULE_FOLD] Offset (%llu) is greater than the input shape (%llu) along "
                          "axis (%llu).",
                          (unsigned long long)offset, (unsigned long long)axisSize,
                          (unsigned long long)*axes.sample);
                return Result::ERROR;
            }
        } else {
            if (!axes.channel || channelOffsets.size() != in.shape(*axes.channel)) {
                JST_ERROR("[MODULE_FOLD] Channel offsets must match channelAxis extent.");
                return Result::ERROR;
            }
            for (size_t c = 0; c < channelOffsets.size(); ++c)
                if (axisSize < channelOffsets[c]) {
                    JST_ERROR("[MODULE_FOLD] Channel offset #%zu (%llu) is greater than the input "
                              "shape (%llu) along axis (%llu).",
                              c, (unsigned long long)channelOffsets[c],
                              (unsigned long long)axisSize, (unsigned long long)*axes.sample);
                    return Result::ERROR;
                }
            channelAxis = axes.channel;
        }
        if (!is_f32_or_cf32(in)) {
            JST_ERROR("[MODULE_FOLD_NATIVE_HIP] Unsupported data type '%s'.", DataTypeName(in.dtype()));
            return Result::ERROR;
        }
        resolvedAxis = *axes.sample;
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(STATELESS));
        JST_CHECK(defineInterfaceInput("buffer"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        input = inputs_.at("buffer");
        Shape os = input.shape();
        os[resolvedAxis] = size;
        JST_CHECK(output.create(device(), input.dtype(), os));
        JST_CHECK(output.propagateAttributes(input));
        output.removeAttribute("channelOffsets");
        if (!channelOffsets.empty()) {
            JST_CHECK(devOffsets.create(device(), DataType::U64, {(U64)channelOffsets.size()}));
            JST_CHECK(devOffsets.copyFromHost(channelOffsets.data(),
                                              channelOffsets.size() * sizeof(U64), nullptr));
            JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
        }
        produced("buffer", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        U64 outer, inner;
        split_axis(input, resolvedAxis, outer, inner);
        U64 chanCount = 1, chanInner = 1;
        if (channelAxis) {
            chanCount = output.shape(*channelAxis);
            for (Index i = *channelAxis + 1; i < output.rank(); ++i) chanInner *= output.shape(i);
        }
        return hip_result(
            kernels::launch_fold(ptr<float>(output), ptr<const float>(input),
                                 input.dtype() == DataType::CF32, outer, input.shape(resolvedAxis),
                                 size, inner, offset % input.shape(resolvedAxis),
                                 channelAxis ? ptr<const uint64_t>(devOffsets) : nullptr, chanCount,
                                 chanInner, s),
            "fold kernel");
    }
    Tensor input, output, devOffsets;
    U64 offset = 0, size = 0;
    Index resolvedAxis = 0;
    std::optional<Index> channelAxis;
    std::vector<U64> channelOffsets;
};

// ---- OverlapAdd (dsp/overlap_add/module_impl.cc:15-140, module_impl_native_cpu.cc:121-202) -----
class OverlapAdd : public Module {
 public:
    const char* type() const override { return "overlap_add"; }
    Result validate() override {
        if (!inputs_.count("buffer") || !inputs_.count("overlap")) return Result::SUCCESS;
        const Tensor& b = inputs_.at("buffer");
        const Tensor& o = inputs_.at("overlap");
        if (!b.validShape() || !o.validShape() || b.size() == 0 || o.size() == 0)
            return Result::SUCCESS;
        if (b.rank() != o.rank()) {
            JST_ERROR("[MODULE_OVERLAP_ADD] Buffer rank (%llu) does not match overlap rank (%llu).",
                      (unsigned long long)b.rank(), (unsigned long long)o.rank());
            return Result::ERROR;
        }
        SignalAxes ba, oa;
        if (ResolveSignalAxes(b, ba) != Result::SUCCESS || ResolveSignalAxes(o, oa) != Result::SUCCESS) {
            JST_ERROR("[MODULE_OVERLAP_ADD] Input signal axis metadata is invalid.");
            return Result::ERROR;
        }
        if (ba.sample != oa.sample || ba.batch != oa.batch || ba.channel != oa.channel) {
            JST_ERROR("[MODULE_OVERLAP_ADD] Buffer and overlap sample, batch, and channel axes must "
                      "match.");
            return Result::ERROR;
        }
        if (b.shape(*ba.sample) < o.shape(*oa.sample)) {
            JST_ERROR("[MODULE_OVERLAP_ADD] Overlap size (%llu) is larger than buffer size (%llu) "
                      "along axis (%llu).",
                      (unsigned long long)o.shape(*oa.sample), (unsigned long long)b.shape(*ba.sample),
                      (unsigned long long)*ba.sample);
            return Result::ERROR;
        }
        for (Index d = 0; d < b.rank(); ++d) {
            if (d == *ba.sample) continue;
            if (b.shape(d) != o.shape(d)) {
                JST_ERROR("[MODULE_OVERLAP_ADD] Shape mismatch on axis (%llu): buffer has %llu, "
                          "overlap has %llu. Non-overlap axes must match exactly.",
                          (unsigned long long)d, (unsigned long long)b.shape(d),
                          (unsigned long long)o.shape(d));
                return Result::ERROR;
            }
        }
        if (b.dtype() != o.dtype() || !is_f32_or_cf32(b)) {
            JST_ERROR("[MODULE_OVERLAP_ADD_NATIVE_HIP] Unsupported data types.");
            return Result::ERROR;
        }
        if (b.rank() > (Index)dev::kMaxRank) {
            JST_ERROR("[MODULE_OVERLAP_ADD] Rank exceeds the supported layout range.");
            return Result::ERROR;
        }
        batchAxis = ba.batch;
        return Result::SUCCESS;
    }
    Result define() override {  // no taint: contiguous inputs, stateful
        JST_CHECK(defineInterfaceInput("buffer"));
        JST_CHECK(defineInterfaceInput("overlap"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        buffer = inputs_.at("buffer");
        overlap = inputs_.at("overlap");
        JST_CHECK(output.create(device(), buffer.dtype(), buffer.shape()));
        JST_CHECK(output.propagateAttributes(buffer));
        Shape ps = overlap.shape();
        if (batchAxis) ps[*batchAxis] = 1;
        JST_CHECK(previousOverlap.create(device(), buffer.dtype(), ps));  // zeroed
        produced("buffer", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        return hip_result(
            kernels::launch_overlap_add(
                ptr<char>(output), ptr<char>(buffer) + buffer.offsetBytes(),
                ptr<char>(overlap) + overlap.offsetBytes(), ptr<char>(previousOverlap),
                buffer.dtype() == DataType::CF32, (uint32_t)buffer.rank(),
                batchAxis ? (int32_t)*batchAxis : -1, buffer.shape().data(), overlap.shape().data(), s),
            "overlap_add kernel");
    }
    const Tensor* state(const std::string& key) const override {
        return key == "previousOverlap" ? &previousOverlap : nullptr;
    }
    Tensor buffer, overlap, output, previousOverlap;
    void planStorage(std::set<const void*>&, std::set<const void*>& writes) const override {
        if (previousOverlap.valid()) writes.insert(previousOverlap.storageId());
    }
    std::optional<Index> batchAxis;
};

// ---- FirTaps + FirDecimate: the Filter block as ONE time-domain kernel (provider "fast") ---------
// No reference counterpart as modules: together they replace the block's pad -> fft -> multiply ->
// fold -> ifft -> normalize -> unpad -> overlap_add chain (filter/block_impl.cc:350-582) when every
// head is centred on 0 Hz (real taps, no fold offset, no phase correction).  Same ports as the block
// sees: signal CF32 [S] or [B, S] in, coeffs CF32 [heads, T] from filter_taps, buffer CF32
// [heads, S/r] or [B, heads, S/r] out, stream continuity across rows and cycles through a history
// tensor.  fir_taps re-lays the (static) coefficients out for the kernel's scalar loads: stateless on
// a static input, so the scheduler settles it once like filter_taps itself.
class FirTaps : public Module {
 public:
    const char* type() const override { return "fir_taps"; }
    Result validate() override {
        bool ok;
        decimation = ConfigU64(config_, "decimation", 1, &ok);
        if (!ok || decimation == 0) {
            JST_ERROR("[MODULE_FIR_TAPS] Decimation must be a positive integer.");
            return Result::ERROR;
        }
        if (!inputs_.count("coeffs")) return Result::SUCCESS;
        const Tensor& h = inputs_.at("coeffs");
        if (h.dtype() != DataType::CF32 || h.rank() != 2 || !h.contiguous() || h.shape(1) == 0 ||
            decimation > h.shape(1) || decimation > 32 || h.shape(1) > 16384) {
            JST_ERROR("[MODULE_FIR_TAPS] Expected contiguous CF32 coeffs [heads, T] with decimation <= min(T, 32).");
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(STATELESS));
        JST_CHECK(defineInterfaceInput("coeffs"));
        return defineInterfaceOutput("table");
    }
    Result create() override {
        coeffs = inputs_.at("coeffs");
        const U64 heads = coeffs.shape(0), taps = coeffs.shape(1);
        JST_CHECK(table.create(device(), DataType::F32, {(U64)kernels::fir_table_floats(taps, decimation, heads)}));
        table.setAttribute("taps", AttrValue{(U64)taps});
        table.setAttribute("heads", AttrValue{(U64)heads});
        table.setAttribute("decimation", AttrValue{(U64)decimation});
        produced("table", table);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        return hip_result(
            kernels::launch_fir_table(ptr<float>(table),
                                      reinterpret_cast<const float2*>(ptr<char>(coeffs) + coeffs.offsetBytes()),
                                      coeffs.shape(1), decimation, coeffs.shape(0), s),
            "fir_taps kernel");
    }
    Tensor coeffs, table;
    U64 decimation = 1;
};

class FirDecimate : public Module {
 public:
    const char* type() const override { return "fir_decimate"; }
    Result validate() override {
        if (!inputs_.count("signal") || !inputs_.count("table")) return Result::SUCCESS;
        const Tensor& x = inputs_.at("signal");
        const Tensor& t = inputs_.at("table");
        auto u64_attr = [&](const char* key, U64& out) {
            const AttrValue* a = t.attribute(key);
            if (!a || !std::holds_alternative<U64>(*a)) return false;
            out = std::get<U64>(*a);
            return true;
        };
        if (t.dtype() != DataType::F32 || !u64_attr("taps", taps) || !u64_attr("heads", heads) ||
            !u64_attr("decimation", decimation) || decimation == 0 ||
            t.size() != kernels::fir_table_floats(taps, decimation, heads)) {
            JST_ERROR("[MODULE_FIR_DECIMATE] The table input must come from a fir_taps module.");
            return Result::ERROR;
        }
        if (x.dtype() != DataType::CF32) {
            JST_ERROR("[MODULE_FIR_DECIMATE_NATIVE_HIP] The signal must be CF32.");
            return Result::ERROR;
        }
        if (x.rank() < 1 || x.rank() > 2 || !x.contiguous()) {
            JST_ERROR("[MODULE_FIR_DECIMATE] Expected a contiguous signal [S] or [B, S].");
            return Result::ERROR;
        }
        SignalAxes axes;
        if (ResolveSignalAxes(x, axes) != Result::SUCCESS || *axes.sample != x.rank() - 1) {
            JST_ERROR("[MODULE_FIR_DECIMATE] The sample axis must be the last axis of the signal.");
            return Result::ERROR;
        }
        if (!kernels::fir_decimate_supported(x.shape(x.rank() - 1), taps, decimation)) {
            JST_ERROR("[MODULE_FIR_DECIMATE_NATIVE_HIP] Unsupported plan: %llu samples per row, %llu taps, "
                      "decimation %llu.", (unsigned long long)x.shape(x.rank() - 1),
                      (unsigned long long)taps, (unsigned long long)decimation);
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineInterfaceInput("signal"));
        JST_CHECK(defineInterfaceInput("table"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        signal = inputs_.at("signal");
        table = inputs_.at("table");
        const bool batched = signal.rank() == 2;
        rows = batched ? signal.shape(0) : 1;
        rowSamples = signal.shape(signal.rank() - 1);
        const U64 m = rowSamples / decimation;
        if (batched) JST_CHECK(output.create(device(), DataType::CF32, {rows, heads, m}));
        else JST_CHECK(output.create(device(), DataType::CF32, {heads, m}));
        SignalAxes axes;
        axes.sample = Index{batched ? 2u : 1u};
        axes.channel = Index{batched ? 1u : 0u};
        if (batched) axes.batch = Index{0};
        JST_CHECK(SetSignalAxes(output, axes));
        JST_CHECK(history.create(device(), DataType::CF32, {std::max<U64>(taps - 1, 1)}));  // zeroed
        produced("buffer", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        return hip_result(
            kernels::launch_fir_decimate(ptr<float2>(output),
                                         reinterpret_cast<const float2*>(ptr<char>(signal) + signal.offsetBytes()),
                                         reinterpret_cast<const float*>(ptr<char>(table) + table.offsetBytes()),
                                         ptr<float2>(history), rows, rowSamples, taps, decimation, heads, s),
            "fir_decimate kernel");
    }
    const Tensor* state(const std::string& key) const override { return key == "history" ? &history : nullptr; }
    Tensor signal, table, output, history;
    U64 decimation = 1, rows = 1, rowSamples = 0, heads = 1, taps = 1;
};

// ---- PhaseCorrection (dsp/phase_correction/module_impl.cc, module_impl_native_cpu.cc:36-115) ---
class PhaseCorrection : public Module {
 public:
    const char* type() const override { return "phase_correction"; }
    Result validate() override {
        bool ok;
        phaseIncrement = ConfigF64(config_, "phaseIncrement", 0.0, &ok);
        if (!ok || !std::isfinite(phaseIncrement)) {
            JST_ERROR("[MODULE_PHASE_CORRECTION] Phase increment must be finite.");
            return Result::ERROR;
        }
        increments.clear();
        channelAxis.reset();
        if (!inputs_.count("signal")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("signal");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        if (const AttrValue* v = in.attribute("channelPhaseIncrements")) {
            const auto* inc = std::get_if<std::vector<F64>>(v);
            if (!inc) {
                JST_ERROR("[MODULE_PHASE_CORRECTION] Input channelPhaseIncrements metadata must "
                          "have type vector<F64>.");
                return Result::ERROR;
            }
            if (inc->empty()) {
                JST_ERROR("[MODULE_PHASE_CORRECTION] Input channelPhaseIncrements metadata cannot "
                          "be empty.");
                return Result::ERROR;
            }
            increments = *inc;
        }
        SignalAxes axes;
        if (ResolveSignalAxes(in, axes) != Result::SUCCESS) {
            JST_ERROR("[MODULE_PHASE_CORRECTION] Input signal axis metadata is invalid.");
            return Result::ERROR;
        }
        batchAxis = axes.batch;
        if (!increments.empty()) {
            if (!axes.channel || increments.size() != in.shape(*axes.channel)) {
                JST_ERROR("[MODULE_PHASE_CORRECTION] Channel phase increments must match "
                          "channelAxis extent.");
                return Result::ERROR;
            }
            for (size_t c = 0; c < increments.size(); ++c)
                if (!std::isfinite(increments[c])) {
                    JST_ERROR("[MODULE_PHASE_CORRECTION] Channel phase increment #%zu must be "
                              "finite.", c);
                    return Result::ERROR;
                }
            channelAxis = axes.channel;
        }
        if (in.dtype() != DataType::CF32) {
            JST_ERROR("[MODULE_PHASE_CORRECTION_NATIVE_HIP] Input must be CF32.");
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(DISCONTIGUOUS));
        JST_CHECK(defineInterfaceInput("signal"));
        return defineInterfaceOutput("signal");
    }
    Result create() override {
        input = inputs_.at("signal");
        JST_CHECK(output.create(device(), input.dtype(), input.shape()));
        JST_CHECK(output.propagateAttributes(input));
        output.removeAttribute("channelPhaseIncrements");
        // module_impl_native_cpu.cc:36-70: counts and inner sizes from the (row-major) shape
        batchCount = batchAxis ? input.shape(*batchAxis) : 1;
        channelCount = channelAxis ? input.shape(*channelAxis) : 1;
        batchInner = channelInner = 1;
        if (batchAxis)
            for (Index i = *batchAxis + 1; i < input.rank(); ++i) batchInner *= input.shape(i);
        if (channelAxis)
            for (Index i = *channelAxis + 1; i < input.rank(); ++i) channelInner *= input.shape(i);
        std::vector<F64> inc(channelCount, phaseIncrement);
        if (!increments.empty()) inc = increments;
        JST_CHECK(devIncrements.create(device(), DataType::F64, {channelCount}));
        JST_CHECK(devIncrements.copyFromHost(inc.data(), inc.size() * sizeof(F64), nullptr));
        JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
        JST_CHECK(phases.create(device(), DataType::F64, {channelCount}));  // zero
        JST_CHECK(corrections.create(device(), DataType::CF32, {channelCount, batchCount}));
        produced("signal", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        EwLayout L;
        if (!MakeEwLayout(output, &input, nullptr, L)) return Result::ERROR;
        return hip_result(
            kernels::launch_phase_correction(L, ptr<float2>(output), ptr<const float2>(input),
                                             ptr<float2>(corrections), ptr<double>(phases),
                                             ptr<const double>(devIncrements), batchCount, batchInner,
                                             channelCount, channelInner, s),
            "phase_correction kernel");
    }
    const Tensor* state(const std::string& key) const override {
        return key == "phases" ? &phases : nullptr;
    }
    Tensor input, output, phases, corrections, devIncrements;
    void planStorage(std::set<const void*>&, std::set<const void*>& writes) const override {  // state + the table a fused tail leaves for the next cycle
        if (phases.valid()) writes.insert(phases.storageId());
        if (corrections.valid()) writes.insert(corrections.storageId());
    }
    F64 phaseIncrement = 0.0;
    std::vector<F64> increments;
    std::optional<Index> batchAxis, channelAxis;
    U64 batchCount = 1, channelCount = 1, batchInner = 1, channelInner = 1;
};

// ---- FilterTaps (dsp/filter_taps/module_impl.cc:15-175, module_impl_native_cpu.cc:46-80) -------
class FilterTaps : public Module {
 public:
    const char* type() const override { return "filter_taps"; }
    Result validate() override {
        bool ok1, ok2, ok3;
        sampleRate = ConfigF64(config_, "sampleRate", 2.0e6, &ok1);
        bandwidth = ConfigF64(config_, "bandwidth", 1.0e6, &ok2);
        taps = ConfigU64(config_, "taps", 101, &ok3);
        if (!parse_f64_list(ConfigStr(config_, "center", "[0.0]"), center)) ok1 = false;
        if (!ok1 || !std::isfinite(sampleRate) || sampleRate <= 0.0) {
            JST_ERROR("[MODULE_FILTER_TAPS] Sample rate must be positive (%g).", sampleRate);
            return Result::ERROR;
        }
        if (!ok2 || !std::isfinite(bandwidth) || bandwidth <= 0.0 || bandwidth > sampleRate) {
            JST_ERROR("[MODULE_FILTER_TAPS] Bandwidth (%.2f MHz) must be between 0 and sample rate "
                      "(%.2f MHz).", bandwidth / 1e6, sampleRate / 1e6);
            return Result::ERROR;
        }
        if (!ok3 || taps == 0) {
            JST_ERROR("[MODULE_FILTER_TAPS] Number of taps cannot be zero.");
            return Result::ERROR;
        }
        if ((taps % 2) == 0) {
            JST_ERROR("[MODULE_FILTER_TAPS] Number of taps must be odd (%llu).",
                      (unsigned long long)taps);
            return Result::ERROR;
        }
        if (center.empty()) {
            JST_ERROR("[MODULE_FILTER_TAPS] At least one center frequency is required.");
            return Result::ERROR;
        }
        const F64 half = sampleRate / 2.0;
        for (size_t i = 0; i < center.size(); ++i)
            if (!std::isfinite(center[i]) || center[i] > half || center[i] < -half) {
                JST_ERROR("[MODULE_FILTER_TAPS] Center frequency #%zu (%.2f MHz) must be between "
                          "%.2f MHz and %.2f MHz.", i, center[i] / 1e6, -half / 1e6, half / 1e6);
                return Result::ERROR;
            }
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(STATIC_OUTPUT));
        return defineInterfaceOutput("coeffs");
    }
    Result create() override {
        const U64 heads = center.size();
        JST_CHECK(coeffs.create(device(), DataType::CF32, {heads, taps}));
        JST_CHECK(SetSignalAxes(coeffs, {.sample = Index{1}, .channel = Index{0}}));
        coeffs.setAttribute("sampleRate", AttrValue{(F64)(F32)sampleRate});
        coeffs.setAttribute("bandwidth", AttrValue{(F64)(F32)bandwidth});
        std::vector<F64> narrowed(center.size());
        for (size_t i = 0; i < center.size(); ++i) narrowed[i] = (F64)(F32)center[i];
        if (narrowed.size() == 1) coeffs.setAttribute("center", AttrValue{narrowed[0]});
        else coeffs.setAttribute("center", AttrValue{narrowed});
        JST_CHECK(devCenter.create(device(), DataType::F64, {heads}));
        JST_CHECK(devCenter.copyFromHost(center.data(), heads * sizeof(F64), nullptr));
        JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
        produced("coeffs", coeffs);
        return Result::SUCCESS;
    }
    // STATIC table: evaluated on the host with the host libm, the sin()/cos() the reference's CPU module calls
    // (dsp/filter_taps/module_impl_native_cpu.cc:46-80), and uploaded -- see Window::computeSubmit.
    Result computeSubmit(hipStream_t s) override {
        const U64 heads = center.size();
        hostCoeffs.assign(2 * heads * taps, 0.0f);
        const double pi = 3.14159265358979323846;
        const double filter_width = (bandwidth / sampleRate) / 2.0;
        for (U64 c = 0; c < heads; ++c) {
            const double filter_offset = center[c] / sampleRate;
            for (U64 i = 0; i < taps; ++i) {
                const double fi = (double)i, half = (double)(taps - 1) / 2.0, n = fi - half;
                const double sinc =
                    (n == 0.0) ? (2.0 * filter_width) : std::sin(2.0 * pi * filter_width * n) / (pi * n);
                const double win = (taps == 1) ? 1.0
                                               : 0.42 - 0.50 * std::cos(2.0 * pi * fi / (double)(taps - 1)) +
                                                     0.08 * std::cos(4.0 * pi * fi / (double)(taps - 1));
                const double theta = ((2.0 * pi) * n) * filter_offset;
                const double sw = sinc * win;
                hostCoeffs[2 * (c * taps + i)] = (float)(sw * std::cos(theta));
                hostCoeffs[2 * (c * taps + i) + 1] = (float)(sw * std::sin(theta));
            }
        }
        JST_HIP_CHECK(hipMemcpyAsync(ptr<float2>(coeffs), hostCoeffs.data(), hostCoeffs.size() * sizeof(float),
                                     hipMemcpyHostToDevice, s),
                      "filter_taps upload");
        JST_HIP_CHECK(hipStreamSynchronize(s), "hipStreamSynchronize");
        return Result::SUCCESS;
    }
    bool capturable() const override { return false; }
    std::vector<float> hostCoeffs;
    Tensor coeffs, devCenter;
    F64 sampleRate = 2.0e6, bandwidth = 1.0e6;
    std::vector<F64> center;
    U64 taps = 101;
};

// ---- Arithmetic (core/arithmetic/module_impl.cc:10-120, module_impl_native_cpu.cc:98-146) ------
class Arithmetic : public Module {
 public:
    const char* type() const override { return "arithmetic"; }
    Result validate() override {
        operation = ConfigStr(config_, "operation", "add");
        if (operation != "add" && operation != "sub" && operation != "mul" && operation != "div") {
            JST_ERROR("[MODULE_ARITHMETIC] Invalid operation '%s'.", operation.c_str());
            return Result::ERROR;
        }
        bool ok1, ok2;
        const I64 axis = config_i64(config_, "axis", -1, &ok1);
        squeeze = ConfigBool(config_, "squeeze", false, &ok2);
        if (!ok1 || !ok2) {
            JST_ERROR("[MODULE_ARITHMETIC] Invalid axis/squeeze.");
            return Result::ERROR;
        }
        if (!inputs_.count("buffer")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("buffer");
        SignalAxes axes;
        JST_CHECK(MapSignalAxes(in, axes));
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        const auto a = resolve_axis(axis, in.rank());
        if (!a) {
            JST_ERROR("[MODULE_ARITHMETIC] Axis %lld out of range for input buffer rank %llu.",
                      (long long)axis, (unsigned long long)in.rank());
            return Result::ERROR;
        }
        if (!is_f32_or_cf32(in) || (in.dtype() == DataType::CF32 && operation == "div")) {
            JST_ERROR("[MODULE_ARITHMETIC_NATIVE_HIP] Unsupported data type / operation.");
            return Result::ERROR;
        }
        resolvedAxis = *a;
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
        JST_CHECK(defineInterfaceInput("buffer"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        input = inputs_.at("buffer");
        Shape os = input.shape();
        os[resolvedAxis] = 1;
        JST_CHECK(output.create(device(), input.dtype(), os));
        unsqueezed = output.clone();
        if (squeeze) JST_CHECK(output.squeezeDims(resolvedAxis));
        JST_CHECK(output.propagateAttributes(input));
        SignalAxes in_axes, out_axes;
        JST_CHECK(MapSignalAxes(input, in_axes));
        auto remap = [&](const std::optional<Index>& a) -> std::optional<Index> {
            if (!a) return std::nullopt;
            if (!squeeze || *a < resolvedAxis) return *a;
            if (*a > resolvedAxis) return *a - 1;
            return std::nullopt;  // the reduced axis disappears
        };
        out_axes.sample = remap(in_axes.sample);
        out_axes.batch = remap(in_axes.batch);
        out_axes.channel = remap(in_axes.channel);
        JST_CHECK(SetSignalAxes(output, out_axes));
        produced("buffer", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        // index space = output shape with the reduced axis kept at extent 1
        Tensor in_view = input.clone();
        in_view.slice(resolvedAxis, 0, 1, 1);
        EwLayout L;
        if (!MakeEwLayout(unsqueezed, &in_view, nullptr, L)) return Result::ERROR;
        L.contiguous = 0;
        const int op = operation == "add" ? 0 : operation == "sub" ? 1 : operation == "mul" ? 2 : 3;
        return hip_result(kernels::launch_arithmetic(L, unsqueezed.data(), input.data(),
                                                     input.dtype() == DataType::CF32, op,
                                                     input.shape(resolvedAxis),
                                                     (int64_t)input.stride(resolvedAxis), s),
                          "arithmetic kernel");
    }
    Tensor input, output, unsqueezed;
    std::string operation = "add";
    bool squeeze = false;
    Index resolvedAxis = 0;
};

// ---- view modules: expand_dims, squeeze_dims (core/{expand_dims,squeeze_dims}) -----------------
class ExpandDims : public Module {
 public:
    const char* type() const override { return "expand_dims"; }
    Result validate() override {
        bool ok;
        axis = config_i64(config_, "axis", 0, &ok);
        if (!ok) {
            JST_ERROR("[MODULE_EXPAND_DIMS] Invalid axis.");
            return Result::ERROR;
        }
        if (!inputs_.count("buffer")) return Result::SUCCESS;
        const I64 r = (I64)inputs_.at("buffer").rank();
        const I64 a = axis < 0 ? r + 1 + axis : axis;  // axis.cc ResolveInsertionAxis
        if (a < 0 || a > r) {
            JST_ERROR("[MODULE_EXPAND_DIMS] Axis %lld out of range.", (long long)axis);
            return Result::ERROR;
        }
        resolved = (Index)a;
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
        JST_CHECK(defineInterfaceInput("buffer"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        Tensor view = inputs_.at("buffer").clone();
        SignalAxes axes;
        JST_CHECK(MapSignalAxes(view, axes));
        JST_CHECK(view.expandDims(resolved));
        auto shift = [&](std::optional<Index>& a) { if (a && *a >= resolved) a = *a + 1; };
        shift(axes.sample);
        shift(axes.batch);
        shift(axes.channel);
        JST_CHECK(SetSignalAxes(view, axes));
        produced("buffer", view);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t) override { return Result::SUCCESS; }
    bool launchesKernels() const override { return false; }
    I64 axis = 0;
    Index resolved = 0;
};

class SqueezeDims : public Module {
 public:
    const char* type() const override { return "squeeze_dims"; }
    Result validate() override {
        bool ok;
        axis = config_i64(config_, "axis", 0, &ok);
        if (!ok) {
            JST_ERROR("[MODULE_SQUEEZE_DIMS] Invalid axis.");
            return Result::ERROR;
        }
        if (!inputs_.count("buffer")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("buffer");
        const auto a = resolve_axis(axis, in.rank());
        if (!a || in.shape(*a) != 1) {
            JST_ERROR("[MODULE_SQUEEZE_DIMS] Axis %lld is not a size-1 axis.", (long long)axis);
            return Result::ERROR;
        }
        resolved = *a;
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
        JST_CHECK(defineInterfaceInput("buffer"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        Tensor view = inputs_.at("buffer").clone();
        SignalAxes axes;
        JST_CHECK(MapSignalAxes(view, axes));
        JST_CHECK(view.squeezeDims(resolved));
        auto shift = [&](std::optional<Index>& a) {
            if (!a) return;
            if (*a == resolved) a.reset();
            else if (*a > resolved) a = *a - 1;
        };
        shift(axes.sample);
        shift(axes.batch);
        shift(axes.channel);
        JST_CHECK(SetSignalAxes(view, axes));
        produced("buffer", view);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t) override { return Result::SUCCESS; }
    bool launchesKernels() const override { return false; }
    I64 axis = 0;
    Index resolved = 0;
};

// ---- FM (dsp/fm/module_impl.cc:20-172, module_impl_native_cpu.cc:30-174) -----------------------
class Fm : public Module {
 public:
    const char* type() const override { return "fm"; }
    Result validate() override {
        mode = ConfigStr(config_, "mode", "narrow");
        deemphasis = ConfigStr(config_, "deemphasis", "none");
        bool ok;
        sampleRate = (F32)ConfigF64(config_, "sampleRate", 240e3, &ok);
        if (mode != "narrow" && mode != "wide") {
            JST_ERROR("[MODULE_FM] Mode must be 'narrow' or 'wide'.");
            return Result::ERROR;
        }
        if (deemphasis != "none" && deemphasis != "50us" && deemphasis != "75us") {
            JST_ERROR("[MODULE_FM] De-emphasis must be 'none', '50us', or '75us'.");
            return Result::ERROR;
        }
        if (!ok || !std::isfinite(sampleRate) || sampleRate <= 0.0f) {
            JST_ERROR("[MODULE_FM] Sample rate must be finite and positive.");
            return Result::ERROR;
        }
        if (sampleRate > 20e6f) {
            JST_ERROR("[MODULE_FM] Sample rate must not exceed 20 MHz.");
            return Result::ERROR;
        }
        if (mode == "wide" && sampleRate < 200e3f) {
            JST_ERROR("[MODULE_FM] Wideband mode requires a sample rate of at least 200 kHz.");
            return Result::ERROR;
        }
        if (!inputs_.count("signal")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("signal");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        if (ResolveSignalAxes(in, axes) != Result::SUCCESS) {
            JST_ERROR("[MODULE_FM] Input must contain valid signal axis metadata.");
            return Result::ERROR;
        }
        if (mode == "wide" && axes.channel) {
            JST_ERROR("[MODULE_FM] Wideband mode does not support channelized input.");
            return Result::ERROR;
        }
        if (in.dtype() != DataType::CF32) {
            JST_ERROR("[MODULE_FM_NATIVE_HIP] Input must be complex (CF32).");
            return Result::ERROR;
        }
        laneCount = in.size() / in.shape(*axes.sample);
        if (axes.batch) laneCount /= in.shape(*axes.batch);
        return Result::SUCCESS;
    }
    Result define() override {  // no taint: contiguous input, stateful
        JST_CHECK(defineInterfaceInput("signal"));
        return defineInterfaceOutput("signal");
    }
    Result create() override {
        input = inputs_.at("signal");
        const bool wide = mode == "wide";
        // FmImpl::updateCoefficients (fm/module_impl.cc:108-157)
        const double pi = 3.14159265358979323846;
        const F32 deviation = wide ? 75e3f : 100e3f;
        const F32 kf = deviation / sampleRate;
        std::memset(&k, 0, sizeof(k));
        k.wide = wide;
        k.deemph_enabled = deemphasis != "none";
        k.ref = (F32)(1.0f / (2.0f * pi * kf));
        k.pilot_inc = (F32)(2.0f * pi * 19e3f / sampleRate);
        const F64 sr = sampleRate;
        k.pilot_alpha = (F32)(1.0 - std::exp(-2.0 * pi * 200.0 / sr));
        k.deemph_alpha = deemphasis == "none"
                             ? 1.0f
                             : (F32)(1.0 - std::exp(-1.0 / (sr * (deemphasis == "50us" ? 50e-6 : 75e-6))));
        const F64 pw = 2.0 * pi * 19e3 / sr, pc = std::cos(pw), ps = std::sin(pw);
        const F64 na = ps / (2.0 * 20.0), na0 = 1.0 + na;
        k.notch[0] = (F32)(1.0 / na0);
        k.notch[1] = (F32)(-2.0 * pc / na0);
        k.notch[2] = k.notch[0];
        k.notch[3] = k.notch[1];
        k.notch[4] = (F32)((1.0 - na) / na0);
        const F64 q[3] = {0.51763809, 0.70710678, 1.93185165};
        const F64 w = 2.0 * pi * 15e3 / sr, co = std::cos(w), si = std::sin(w);
        for (int s = 0; s < 3; ++s) {
            const F64 al = si / (2.0 * q[s]), a0 = 1.0 + al;
            k.lp[s][0] = (F32)((1.0 - co) * 0.5 / a0);
            k.lp[s][1] = (F32)((1.0 - co) / a0);
            k.lp[s][2] = k.lp[s][0];
            k.lp[s][3] = (F32)(-2.0 * co / a0);
            k.lp[s][4] = (F32)((1.0 - al) / a0);
        }
        Shape os = input.shape();
        SignalAxes out_axes = axes;
        if (wide) {
            out_axes.channel = os.size();
            os.push_back(2);
        }
        JST_CHECK(output.create(device(), DataType::F32, os));
        JST_CHECK(output.propagateAttributes(input));
        JST_CHECK(SetSignalAxes(output, out_axes));
        output.setAttribute("frequency", AttrValue{F64{0.0}});
        JST_CHECK(states.create(device(), DataType::U8, {laneCount * (U64)kernels::fm_state_bytes()}));
        produced("signal", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        if (laneCount == 0) return Result::SUCCESS;  // empty input: no axes were resolved
        dev::FmLayout L;
        std::memset(&L, 0, sizeof(L));
        L.lanes = laneCount;
        L.samples = input.shape(*axes.sample);
        L.batches = axes.batch ? input.shape(*axes.batch) : 1;
        L.in_sample_stride = (int64_t)input.stride(*axes.sample);
        L.out_sample_stride = (int64_t)output.stride(*axes.sample);
        L.in_batch_stride = axes.batch ? (int64_t)input.stride(*axes.batch) : 0;
        L.out_batch_stride = axes.batch ? (int64_t)output.stride(*axes.batch) : 0;
        L.out_channel_stride = k.wide ? (int64_t)output.stride(input.rank()) : 0;
        int r = 0;
        for (Index ax = 0; ax < input.rank(); ++ax) {
            if (ax == *axes.sample || (axes.batch && ax == *axes.batch)) continue;
            L.lane_shape[r] = input.shape(ax);
            L.in_lane_stride[r] = (int64_t)input.stride(ax);
            L.out_lane_stride[r] = (int64_t)output.stride(ax);
            ++r;
        }
        L.lane_rank = r;
        L.in_offset = input.offset();
        L.out_offset = output.offset();
        return hip_result(kernels::launch_fm(ptr<float>(output), ptr<const float2>(input), states.data(), k, L, s),
                          "fm kernel");
    }
    Tensor input, output, states;
    void planStorage(std::set<const void*>& reads, std::set<const void*>& writes) const override {  // `input` may have been pointed at a duplicate's source
        reads.insert(input.storageId());
        if (states.valid()) writes.insert(states.storageId());
    }
    std::string mode = "narrow", deemphasis = "none";
    F32 sampleRate = 240e3f;
    SignalAxes axes;
    U64 laneCount = 0;
    dev::FmCoeffs k;
};

// ---- Duplicate (core/duplicate/module_impl_native_cuda.cc:15-151): dense device copy -----------
class Duplicate : public Module {
 public:
    const char* type() const override { return "duplicate"; }
    Result define() override {
        JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
        JST_CHECK(defineInterfaceInput("buffer"));
        return defineInterfaceOutput("buffer");
    }
    Result validate() override {
        if (inputs_.count("buffer") && !is_f32_or_cf32(inputs_.at("buffer"))) {
            JST_ERROR("[MODULE_DUPLICATE_NATIVE_HIP] Unsupported data type.");
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    // TryElideDuplicate points this module's readers at its source; the rewiring is undone whenever a plan ends
    std::vector<std::pair<Tensor*, Tensor>> rewired;
    void resetPlan() override {
        for (auto& [field, original] : rewired) *field = original;
        rewired.clear();
    }
    Result create() override {
        input = inputs_.at("buffer");
        JST_CHECK(output.create(device(), input.dtype(), input.shape()));
        JST_CHECK(output.propagateAttributes(input));
        produced("buffer", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {  // strided gather = multiply by 1 in the
        EwLayout L;                                 // elementwise engine (exact)
        if (!MakeEwLayout(output, &input, nullptr, L)) return Result::ERROR;
        if (input.dtype() == DataType::CF32)
            return hip_result(kernels::launch_multiply_constant_cf32(
                                  L, ptr<float2>(output), ptr<const float2>(input), 1.0f, s),
                              "duplicate kernel");
        return hip_result(kernels::launch_multiply_constant_f32(L, ptr<float>(output),
                                                                ptr<const float>(input), 1.0f, s),
                          "duplicate kernel");
    }
    Tensor input, output;
};

// ---- Lineplot, compute part (visualization/lineplot/module_impl.cc:30-215,
// module_impl_native_cpu.cc:80-118) ------------------------------------------------------------
class Lineplot : public Module {
 public:
    const char* type() const override { return "lineplot"; }
    Result validate() override {
        bool ok1, ok2;
        averaging = ConfigU64(config_, "averaging", 1, &ok1);
        decimation = ConfigU64(config_, "decimation", 1, &ok2);
        if (!ok1 || averaging == 0) {
            JST_ERROR("[MODULE_LINEPLOT] Averaging must be greater than zero.");
            return Result::ERROR;
        }
        if (!ok2 || decimation == 0) {
            JST_ERROR("[MODULE_LINEPLOT] Decimation must be greater than zero.");
            return Result::ERROR;
        }
        if (!inputs_.count("signal")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("signal");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        SignalAxes axes;
        if (MapSignalAxes(in, axes) != Result::SUCCESS) {
            JST_ERROR("[MODULE_LINEPLOT] Input must contain valid signal axis metadata.");
            return Result::ERROR;
        }
        if (axes.sample && axes.channel) {
            JST_ERROR("[MODULE_LINEPLOT] Input cannot contain both sampleAxis and channelAxis.");
            return Result::ERROR;
        }
        const auto element = axes.sample ? axes.sample : axes.channel;
        if (!element) {
            JST_ERROR("[MODULE_LINEPLOT] Input must contain sampleAxis or channelAxis.");
            return Result::ERROR;
        }
        for (Index ax = 0; ax < in.rank(); ++ax)
            if (ax != *element && (!axes.batch || ax != *axes.batch)) {
                JST_ERROR("[MODULE_LINEPLOT] Unsupported auxiliary input axis %llu.",
                          (unsigned long long)ax);
                return Result::ERROR;
            }
        if (in.dtype() != DataType::F32) {
            JST_ERROR("[MODULE_LINEPLOT] Input must be F32.");
            return Result::ERROR;
        }
        numberOfElements = in.shape(*element) / decimation;
        if (numberOfElements < 2) {
            JST_ERROR("[MODULE_LINEPLOT] Decimated input must keep at least two elements.");
            return Result::ERROR;
        }
        numberOfBatches = axes.batch ? in.shape(*axes.batch) : 1;
        elementStride = in.stride(*element);
        batchStride = axes.batch ? in.stride(*axes.batch) : 0;
        normalizationFactor = 1.0f / (0.5f * static_cast<F32>(numberOfBatches));
        return Result::SUCCESS;
    }
    // lineplot/module_impl.cc:234-246: the averaging moves in place, the geometry does not
    Result reconfigureImpl(const Config& previous) override {
        return ConfigU64(previous, "decimation", 1) == decimation ? Result::SUCCESS : Result::RECREATE;
    }
    Result define() override {
        JST_CHECK(defineTaint(SURFACE));
        return defineInterfaceInput("signal");
    }
    Result create() override {
        input = inputs_.at("signal");
        JST_CHECK(signalPoints.create(device(), DataType::F32, {numberOfElements, 2}));
        JST_CHECK(averagingBuffer.create(device(), DataType::F32, {numberOfElements}));
        // the X coordinates of the points, once, with the reference's F32 expression
        // (lineplot/module_impl_native_cpu.cc:64-69: i * 2.0f / (numberOfElements - 1) - 1.0f); Y starts at 0
        std::vector<F32> xy(2 * numberOfElements, 0.0f);
        for (U64 i = 0; i < numberOfElements; ++i) xy[2 * i] = i * 2.0f / (numberOfElements - 1) - 1.0f;
        JST_CHECK(hip_result(hipMemcpy(ptr<float>(signalPoints), xy.data(), xy.size() * sizeof(F32), hipMemcpyHostToDevice),
                             "lineplot X coordinates"));
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        return hip_result(
            kernels::launch_lineplot(ptr<float>(signalPoints), ptr<float>(averagingBuffer),
                                     ptr<const float>(input), input.offset(), numberOfBatches,
                                     numberOfElements, (int64_t)batchStride, (int64_t)elementStride,
                                     decimation, normalizationFactor, static_cast<F32>(averaging), s),
            "lineplot kernel");
    }
    // Cycle batching (Runtime::planBatch): the n cycles of a span as ONE launch when the input is a ring the batched
    // spectrum unit fills (the moving average is a per-bin recursion: one thread's loop over the cycles); any other input
    // -- the same tensor every cycle -- is n per-cycle launches.
    bool spanCapable() const override { return true; }
    Result computeSubmitSpan(hipStream_t s, U64 first_slot, U64 n) override {
        const U64 ring = input.ringSlots();
        if (ring < 2 || first_slot >= ring) {
            for (U64 c = 0; c < n; ++c) JST_CHECK(computeSubmit(s));
            return Result::SUCCESS;
        }
        const U64 slot_elems = (U64)((const char*)input.ringSlotData(1) - (const char*)input.ringSlotData(0)) / sizeof(float);
        return hip_result(
            kernels::launch_lineplot_span(ptr<float>(signalPoints), ptr<float>(averagingBuffer),
                                          static_cast<const float*>(input.ringSlotData(0)), input.offset(), slot_elems,
                                          first_slot, ring, n, numberOfBatches, numberOfElements, (int64_t)batchStride,
                                          (int64_t)elementStride, decimation, normalizationFactor,
                                          static_cast<F32>(averaging), s),
            "lineplot kernel (cycle-batched span)");
    }
    const Tensor* state(const std::string& key) const override {
        if (key == "signalPoints") return &signalPoints;
        if (key == "averagingBuffer") return &averagingBuffer;
        return nullptr;
    }
    Tensor input, signalPoints, averagingBuffer;
    void planStorage(std::set<const void*>&, std::set<const void*>& writes) const override {
        if (signalPoints.valid()) writes.insert(signalPoints.storageId());
        if (averagingBuffer.valid()) writes.insert(averagingBuffer.storageId());
    }
    U64 averaging = 1, decimation = 1, numberOfElements = 0, numberOfBatches = 0;
    U64 elementStride = 0, batchStride = 0;
    F32 normalizationFactor = 1.0f;
};

// ---- SignalGenerator (dsp/signal_generator/module_impl.cc:16-170, module_impl_native_cpu.cc) ---
// Every waveform of the reference.  The periodic ones and the chirp share a serial F64 phase walk
// (bit-exact); noise is distribution-equivalent only (the reference seeds from random_device).
class SignalGenerator : public Module {
 public:
    const char* type() const override { return "signal_generator"; }
    Result validate() override {
        constexpr F64 maxF32 = 3.40282346638528859811704183484516925e+38;
        constexpr F64 minF32 = 1.17549435082228750796873653722224568e-38;
        signalType = ConfigStr(config_, "signalType", "cosine");
        dataType = ConfigStr(config_, "signalDataType", "F32");
        bool ok[10];
        sampleRate = ConfigF64(config_, "sampleRate", 1.0e6, &ok[0]);
        frequency = ConfigF64(config_, "frequency", 1000.0, &ok[1]);
        amplitude = ConfigF64(config_, "amplitude", 1.0, &ok[2]);
        phase = ConfigF64(config_, "phase", 0.0, &ok[3]);
        dcOffset = ConfigF64(config_, "dcOffset", 0.0, &ok[4]);
        bufferSize = ConfigU64(config_, "bufferSize", 8192, &ok[5]);
        noiseVariance = ConfigF64(config_, "noiseVariance", 1.0, &ok[6]);
        chirpStartFreq = ConfigF64(config_, "chirpStartFreq", 1000.0, &ok[7]);
        chirpEndFreq = ConfigF64(config_, "chirpEndFreq", 10000.0, &ok[8]);
        chirpDuration = ConfigF64(config_, "chirpDuration", 1.0, &ok[9]);
        static const std::pair<const char*, kernels::SignalShape> kTypes[] = {
            {"sine", kernels::SignalShape::Sine},         {"cosine", kernels::SignalShape::Cosine},
            {"square", kernels::SignalShape::Square},     {"triangle", kernels::SignalShape::Triangle},
            {"sawtooth", kernels::SignalShape::Sawtooth}, {"noise", kernels::SignalShape::Noise},
            {"dc", kernels::SignalShape::Dc},             {"chirp", kernels::SignalShape::Chirp}};
        bool known = false;
        for (const auto& t : kTypes)
            if (signalType == t.first) {
                known = true;
                shape = t.second;
            }
        if (!known) {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] Invalid signal type '%s'.", signalType.c_str());
            return Result::ERROR;
        }
        if (dataType != "F32" && dataType != "CF32") {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] Invalid data type '%s'.", dataType.c_str());
            return Result::ERROR;
        }
        for (bool o : ok)
            if (!o) {
                JST_ERROR("[MODULE_SIGNAL_GENERATOR] Invalid numeric configuration value.");
                return Result::ERROR;
            }
        const bool cx = dataType == "CF32";
        const bool sinusoid = signalType == "sine" || signalType == "cosine";
        const bool periodic = sinusoid || signalType == "square" || signalType == "triangle" ||
                              signalType == "sawtooth";
        const bool noise = signalType == "noise", dc = signalType == "dc", chirp = signalType == "chirp";
        if (!std::isfinite(sampleRate) || sampleRate < minF32 || sampleRate > maxF32) {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] Sample rate must be positive and within the F32 "
                      "range (%g).", sampleRate);
            return Result::ERROR;
        }
        const F64 nyquist = sampleRate * 0.5;
        if (periodic) {
            const F64 lo = cx && sinusoid ? -nyquist : 0.0;
            if (!std::isfinite(frequency) || frequency < lo || frequency > nyquist) {
                JST_ERROR("[MODULE_SIGNAL_GENERATOR] Frequency (%g) must be within the supported "
                          "range [%g, %g].", frequency, lo, nyquist);
                return Result::ERROR;
            }
        }
        if (!std::isfinite(amplitude) || amplitude < 0.0 || amplitude > maxF32) {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] Amplitude must be non-negative and within the F32 "
                      "range (%g).", amplitude);
            return Result::ERROR;
        }
        if ((periodic || chirp) && !std::isfinite(phase)) {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] Phase must be finite (%g).", phase);
            return Result::ERROR;
        }
        if (!std::isfinite(dcOffset) || std::abs(dcOffset) > maxF32) {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] DC offset must be within the F32 range (%g).", dcOffset);
            return Result::ERROR;
        }
        if (dc) {
            const F64 v = amplitude + dcOffset;
            if (!std::isfinite(v) || std::abs(v) > maxF32) {
                JST_ERROR("[MODULE_SIGNAL_GENERATOR] DC value exceeds the F32 output range.");
                return Result::ERROR;
            }
        } else if (!noise && amplitude > maxF32 - std::abs(dcOffset)) {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] Amplitude and DC offset exceed the F32 output range.");
            return Result::ERROR;
        }
        if (bufferSize == 0) {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] Buffer size cannot be zero.");
            return Result::ERROR;
        }
        if (noise && (!std::isfinite(noiseVariance) || noiseVariance < 0.0 || noiseVariance > maxF32)) {
            JST_ERROR("[MODULE_SIGNAL_GENERATOR] Noise variance must be non-negative and within the "
                      "F32 range (%g).", noiseVariance);
            return Result::ERROR;
        }
        if (chirp) {
            const F64 lo = cx ? -nyquist : 0.0;  // module_impl.cc:117-135
            if (!std::isfinite(chirpStartFreq) || chirpStartFreq < lo || chirpStartFreq > nyquist) {
                JST_ERROR("[MODULE_SIGNAL_GENERATOR] Chirp start frequency (%g) must be within "
                          "[%g, %g].", chirpStartFreq, lo, nyquist);
                return Result::ERROR;
            }
            if (!std::isfinite(chirpEndFreq) || chirpEndFreq < lo || chirpEndFreq > nyquist) {
                JST_ERROR("[MODULE_SIGNAL_GENERATOR] Chirp end frequency (%g) must be within "
                          "[%g, %g].", chirpEndFreq, lo, nyquist);
                return Result::ERROR;
            }
            if (!std::isfinite(chirpDuration) || chirpDuration <= 0.0) {
                JST_ERROR("[MODULE_SIGNAL_GENERATOR] Chirp duration must be positive (%g).", chirpDuration);
                return Result::ERROR;
            }
        }
        return Result::SUCCESS;
    }
    // signal_generator/module_impl.cc:185-211: waveform parameters move in place (the running phase is
    // state and stays); type, sample format, rate and buffer size need a rebuild
    Result reconfigureImpl(const Config& previous) override {
        if (ConfigStr(previous, "signalType", "cosine") != signalType ||
            ConfigStr(previous, "signalDataType", "F32") != dataType ||
            ConfigF64(previous, "sampleRate", 1.0e6) != sampleRate ||
            ConfigU64(previous, "bufferSize", 8192) != bufferSize)
            return Result::RECREATE;
        // The reference folds a new `phase` into the running oscillator phase and rescales the chirp clock by a
        // new `chirpDuration` (signal_generator/module_impl_native_cpu.cc:128-152).  That state lives on the
        // device here and is not patched in place: ask for a rebuild rather than report a phase the waveform
        // does not have.
        if (ConfigF64(previous, "phase", 0.0) != phase ||
            ConfigF64(previous, "chirpDuration", 1.0) != chirpDuration)
            return Result::RECREATE;
        return Result::SUCCESS;
    }
    Result define() override { return defineInterfaceOutput("signal"); }
    Result create() override {
        const bool cx = dataType == "CF32";
        JST_CHECK(signal.create(device(), cx ? DataType::CF32 : DataType::F32, {bufferSize}));
        JST_CHECK(SetSignalAxes(signal, {.sample = Index{0}}));
        signal.setAttribute("sampleRate", AttrValue{(F64)(F32)sampleRate});
        JST_CHECK(phases.create(device(), DataType::F64, {bufferSize}));
        JST_CHECK(oscillator.create(device(), DataType::F64, {4}));
        const F64 period = 2.0 * 3.14159265358979323846;
        F64 init[4] = {std::fmod(phase, period), 0.0, 0.0, 0.0};
        if (init[0] < 0.0) init[0] += period;
        const uint64_t seed = 0x243F6A8885A308D3ull ^ (uint64_t)reinterpret_cast<uintptr_t>(this);
        std::memcpy(&init[2], &seed, sizeof(seed));  // noise counter base (any value: see kernel)
        JST_CHECK(oscillator.copyFromHost(init, sizeof(init), nullptr));
        JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
        produced("signal", signal);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t s) override {
        kernels::SignalParams p{shape,          amplitude,      frequency,    sampleRate,   dcOffset,
                                noiseVariance, chirpStartFreq, chirpEndFreq, chirpDuration};
        return hip_result(kernels::launch_signal_generator(ptr<float>(signal), ptr<double>(phases),
                                                           ptr<double>(oscillator), bufferSize,
                                                           dataType == "CF32", p, s),
                          "signal_generator kernel");
    }
    Tensor signal, phases, oscillator;
    std::string signalType = "cosine", dataType = "F32";
    kernels::SignalShape shape = kernels::SignalShape::Cosine;
    F64 sampleRate = 1.0e6, frequency = 1000.0, amplitude = 1.0, phase = 0.0, dcOffset = 0.0;
    F64 noiseVariance = 1.0, chirpStartFreq = 1000.0, chirpEndFreq = 10000.0, chirpDuration = 1.0;
    U64 bufferSize = 8192;
};

// ---- the Filter block's plan (filter/block_impl.cc:40-168) -------------------------------------------
Result CalculateFilterPlan(F32 sampleRate, F32 bandwidth, const std::vector<F32>& center, U64 taps, U64 heads,
                           U64 signalSize, FilterPlan& plan) {
    plan = FilterPlan{};
    plan.padSize = taps - 1;
    // taps + (signalSize - 1) must fit U64 (the block refuses the configuration otherwise)
    if (signalSize == 0 || taps > ~0ull - (signalSize - 1)) {
        JST_ERROR("[BLOCK_FILTER] Combined signal and filter extent exceeds the supported range.");
        return Result::ERROR;
    }
    const U64 conv = taps + (signalSize - 1);
    plan.convolutionSize = conv;
    // resampling needs an integer ratio that divides both the pad and the convolution; anything else is a bypass
    const F64 sr = sampleRate, bw = bandwidth;
    const F64 ratio = sr / bw;
    const F64 two64 = std::ldexp(1.0, 64);
    if (!std::isfinite(ratio) || !(ratio > 0.0) || ratio >= two64 || ratio != std::floor(ratio)) return Result::SUCCESS;
    const U64 r = (U64)ratio;
    if (plan.padSize % r != 0 || conv % r != 0) return Result::SUCCESS;
    // per head: the fold starts at minus the centre bin, wrapped into [0, conv)
    plan.resamplerOffsets.assign(heads, 0);
    const F64 binWidth = sr / (F64)conv;
    for (U64 h = 0; h < heads; ++h) {
        const F64 ct = h < center.size() ? (F64)center[h] : 0.0;
        if (ct == 0.0) continue;
        const F64 bin = ct / binWidth;
        if (!std::isfinite(bin)) {
            JST_ERROR("[BLOCK_FILTER] Center frequency (%g) cannot be mapped to a finite resampler bin.", ct);
            return Result::ERROR;
        }
        const F64 start = -std::round(bin);  // half away from zero
        if (std::fabs(start) >= two64) {
            JST_ERROR("[BLOCK_FILTER] Center frequency (%g) cannot be mapped to a representable resampler bin.", ct);
            return Result::ERROR;
        }
        if (start < 0.0) {
            const U64 rem = (U64)(-start) % conv;
            plan.resamplerOffsets[h] = rem ? conv - rem : 0;
        } else {
            plan.resamplerOffsets[h] = (U64)std::fmod((long double)start, (long double)conv);
        }
    }
    plan.resamplerSize = conv / r;
    plan.padSize /= r;
    plan.resampledSampleRate = (F32)(sr / (F64)r);
    plan.resample = true;
    return Result::SUCCESS;
}

// ---- fusion hooks for the Filter block's chain (filter/block_impl.cc:350-582) --------------------
namespace {
bool sole_consumer(const std::vector<Module*>& ordered, const Tensor& t, const Module* consumer) {
    for (const Module* m : ordered) {
        if (m == consumer) continue;
        for (const auto& kv : m->inputs())
            if (kv.second.storageId() == t.storageId()) return false;
    }
    return true;
}
}  // namespace

bool TryFuseFilter(const std::vector<Module*>& ordered, size_t at, std::string& name,
                   std::vector<Module*>& members, std::function<Result(hipStream_t)>& submit,
                   size_t& consumed) {
    // A/B switches (read once): keep the two-kernel forms of the forward / inverse side
    static const bool no_fold_epilogue = std::getenv("JST_NO_FOLD_EPILOGUE") != nullptr;
    static const bool no_unpad_epilogue = std::getenv("JST_NO_UNPAD_EPILOGUE") != nullptr;
    if (at + 1 >= ordered.size()) return false;
    // pad(last axis) -> fft(same axis): the padded tensor is only ever read by the transform
    if (auto* pad = dynamic_cast<Pad*>(ordered[at])) {
        auto* fft = dynamic_cast<Fft*>(ordered[at + 1]);
        if (!fft || fft->input.storageId() != pad->output.storageId()) return false;
        if (!sole_consumer(ordered, pad->output, fft)) return false;
        const Index axis = fft->resolvedAxis;
        if (axis != pad->resolvedAxis || axis + 1 != pad->input.rank()) return false;
        if (pad->input.dtype() != DataType::CF32 || !pad->input.contiguous() ||
            !fft->input.contiguous() || fft->input.offset() != 0)
            return false;
        if (!fft->useTiled || fft->bluesteinSize != 0) return false;  // decided in computeInitialize
        // ... -> multiply(spectrum, operand broadcast over the transforms) -> fold(last axis): the aliases of a bin
        // meet in one workgroup of the transform's second kernel (kernels::FoldProductArgs); the spectrum and the
        // product are never written
        if (at + 3 < ordered.size()) {
            auto* mul = dynamic_cast<Multiply*>(ordered[at + 2]);
            auto* fold = dynamic_cast<Fold*>(ordered[at + 3]);
            // heads: the spectrum [.., 1, n] broadcast along the axis in front of the transform axis against an operand
            // [1.., heads, n] (filter/block_impl.cc:350-582 with more than one head); 1 = the plain same-shape product
            const Index rank = fft->output.rank();
            const Index head_axis = rank >= 2 ? rank - 2 : 0;
            U64 heads = 1;
            const auto whole_spectrum = [&](const Tensor& v) {
                if (v.storageId() != fft->output.storageId() || v.offset() != 0 || v.rank() != rank) return false;
                if (v.contiguous() && v.shape() == fft->output.shape()) return true;
                if (rank < 2 || fft->output.shape(head_axis) != 1 || v.stride(head_axis) != 0) return false;
                for (Index ax = 0; ax < rank; ++ax) {
                    if (ax == head_axis) continue;
                    if (v.shape(ax) != fft->output.shape(ax) || v.stride(ax) != fft->output.stride(ax)) return false;
                }
                heads = v.shape(head_axis);
                return true;
            };
            const auto broadcast_row = [&](const Tensor& v) {
                if (v.dtype() != DataType::CF32 || v.rank() != fft->output.rank()) return false;
                for (Index ax = 0; ax + 1 < v.rank(); ++ax) {
                    if (heads > 1 && ax == head_axis) continue;  // one operand row per head
                    if (v.stride(ax) != 0 && v.shape(ax) != 1) return false;
                }
                return true;
            };
            bool ok = mul && fold && std::string(mul->type()) == "multiply" && fft->forward &&
                      fft->output.contiguous() && fft->output.offset() == 0 &&
                      sole_consumer(ordered, fft->output, mul) &&
                      fold->input.storageId() == mul->c.storageId() && sole_consumer(ordered, mul->c, fold) &&
                      mul->c.dtype() == DataType::CF32 &&
                      fold->resolvedAxis + 1 == mul->c.rank() && fold->output.contiguous() &&
                      !no_fold_epilogue;
            bool spectrum_first = true;
            if (ok) {
                if (whole_spectrum(mul->a) && broadcast_row(mul->b)) spectrum_first = true;
                else if (whole_spectrum(mul->b) && broadcast_row(mul->a)) spectrum_first = false;
                else ok = false;
            }
            if (ok) {  // the product's shape: the spectrum's, with `heads` on the head axis
                Shape want = fft->output.shape();
                if (heads > 1) want[head_axis] = heads;
                ok = mul->c.shape() == want;
            }
            U64 chanCount = 1, chanDiv = 1;
            if (ok && heads > 1) {
                // one fold offset per head: the channel axis of the fold is the head axis
                const Tensor& hop = spectrum_first ? mul->b : mul->a;
                ok = fold->channelAxis && *fold->channelAxis == head_axis && hop.shape(head_axis) == heads &&
                     hop.stride(head_axis) >= 0 && fold->output.shape(head_axis) == heads;
                chanCount = heads;
            } else if (ok && fold->channelAxis) {
                U64 chanInner = 1;
                chanCount = fold->output.shape(*fold->channelAxis);
                for (Index i = *fold->channelAxis + 1; i < fold->output.rank(); ++i) chanInner *= fold->output.shape(i);
                ok = chanInner % fold->size == 0;  // the channel of an output element depends on its transform only
                chanDiv = ok ? chanInner / fold->size : 1;
            }
            const U64 n = fft->input.shape(axis);
            ok = ok && kernels::fft_tiled_fold_supported(n, fft->input.size() / n, fold->size);
            if (ok) {
                members = {pad, fft, mul, fold};
                consumed = 4;
                name = "fft_padded_fold(" + pad->name() + "+" + fft->name() + "+" + mul->name() + "+" + fold->name() + ")";
                submit = [pad, fft, mul, fold, axis, spectrum_first, chanCount, chanDiv, n, heads, head_axis](hipStream_t stream) -> Result {
                    dev::FftLayout L;
                    JST_CHECK(fft->layout(L));
                    int r = 0;
                    for (Index ax = 0; ax < pad->input.rank(); ++ax) {
                        if (ax == axis) continue;
                        L.in_outer_stride[r++] = (int64_t)pad->input.stride(ax);
                    }
                    L.in_axis_stride = (int64_t)pad->input.stride(axis);
                    L.in_offset = pad->input.offset();
                    const Tensor& h = spectrum_first ? mul->b : mul->a;
                    kernels::FoldProductArgs f{};
                    f.out = ptr<float2>(fold->output) + fold->output.offset();
                    f.h = ptr<const float2>(h) + h.offset();
                    f.h_stride = (int64_t)h.stride(axis);
                    f.fold = fold->size;
                    f.offset = fold->offset % n;
                    f.chan_offsets = fold->channelAxis ? ptr<const uint64_t>(fold->devOffsets) : nullptr;
                    f.chan_count = chanCount;
                    f.chan_div = chanDiv;
                    f.spectrum_first = spectrum_first;
                    f.heads = heads;
                    f.h_head_stride = heads > 1 ? (int64_t)h.stride(head_axis) : 0;
                    return hip_result(
                        kernels::launch_fft_c2c_tiled_padded_fold(n, pad->input.shape(axis), true, L, fft->twiddles,
                                                                  ptr<const float2>(pad->input),
                                                                  ptr<float2>(fft->scratchA), f, stream),
                        "fft (tiled, padded, multiply + fold epilogue) kernel");
                };
                return true;
            }
        }
        members = {pad, fft};
        consumed = 2;
        name = "fft_padded(" + pad->name() + "+" + fft->name() + ")";
        submit = [pad, fft, axis](hipStream_t stream) -> Result {
            dev::FftLayout L;
            JST_CHECK(fft->layout(L));
            int r = 0;  // input side of the layout: the UNPADDED tensor
            for (Index ax = 0; ax < pad->input.rank(); ++ax) {
                if (ax == axis) continue;
                L.in_outer_stride[r++] = (int64_t)pad->input.stride(ax);
            }
            L.in_axis_stride = (int64_t)pad->input.stride(axis);
            L.in_offset = pad->input.offset();
            return hip_result(
                kernels::launch_fft_c2c_tiled_padded(
                    fft->input.shape(axis), pad->input.shape(axis), fft->forward, L, fft->twiddles,
                    ptr<const float2>(pad->input), ptr<float2>(fft->output),
                    ptr<float2>(fft->scratchA), stream),
                "fft (tiled, padded) kernel");
        };
        return true;
    }
    // fft(inverse, tiled) -> multiply_constant [-> phase_correction] -> unpad(same axis) -> overlap_add: the scale, the
    // phase correction and the body / tail split ride on the transform's last store (the body lands in overlap_add's
    // output), one small kernel then patches the overlap region, rolls the state and -- with a phase_correction -- advances
    // its phases and writes the next cycle's correction table (filter/block_impl.cc:499-582)
    if (auto* fft = dynamic_cast<Fft*>(ordered[at])) {
        if (at + 3 >= ordered.size()) return false;
        auto* norm = dynamic_cast<MultiplyConstant*>(ordered[at + 1]);
        auto* phase = dynamic_cast<PhaseCorrection*>(ordered[at + 2]);
        static const bool no_phase_epilogue = std::getenv("JST_NO_PHASE_EPILOGUE") != nullptr;
        if (phase && (no_phase_epilogue || at + 4 >= ordered.size())) return false;
        const size_t tail_at = at + (phase ? 3 : 2);
        auto* unpad = dynamic_cast<Unpad*>(ordered[tail_at]);
        auto* ola = dynamic_cast<OverlapAdd*>(ordered[tail_at + 1]);
        if (!norm || !unpad || !ola || no_unpad_epilogue) return false;
        const Index axis = fft->resolvedAxis;
        if (fft->realInput || !fft->useTiled || fft->bluesteinSize != 0 || axis + 1 != fft->input.rank()) return false;
        if (fft->input.dtype() != DataType::CF32 || fft->output.dtype() != DataType::CF32) return false;
        if (norm->input.storageId() != fft->output.storageId() || !sole_consumer(ordered, fft->output, norm)) return false;
        if (norm->input.shape() != fft->output.shape() || norm->input.offset() != fft->output.offset() ||
            !norm->input.contiguous() || norm->output.dtype() != DataType::CF32)
            return false;
        const Tensor* scaled = &norm->output;
        Module* scaled_reader = unpad;
        U64 batchDiv = 1, chanDiv = 1;
        if (phase) {
            if (fft->forward) return false;  // the epilogue is compiled for the Filter's inverse transform
            if (phase->input.storageId() != norm->output.storageId() || !sole_consumer(ordered, norm->output, phase)) return false;
            if (phase->input.shape() != fft->output.shape() || phase->input.offset() != 0 || !phase->input.contiguous() ||
                !phase->output.contiguous())
                return false;
            // the (batch, channel) cell of a TRANSFORM: both axes in front of the transform axis
            const U64 n = fft->input.shape(axis);
            if ((phase->batchAxis && *phase->batchAxis == axis) || (phase->channelAxis && *phase->channelAxis == axis)) return false;
            if ((phase->batchAxis && phase->batchInner % n) || (phase->channelAxis && phase->channelInner % n)) return false;
            batchDiv = phase->batchAxis ? phase->batchInner / n : 1;
            chanDiv = phase->channelAxis ? phase->channelInner / n : 1;
            if ((fft->input.size() / n) >> 32) return false;
            scaled = &phase->output;
            scaled_reader = phase;
        }
        (void)scaled_reader;
        if (unpad->input.storageId() != scaled->storageId() || !sole_consumer(ordered, *scaled, unpad)) return false;
        if (unpad->resolvedAxis != axis || unpad->input.shape() != fft->output.shape() || !unpad->input.contiguous()) return false;
        if (ola->buffer.storageId() != unpad->body.storageId() || ola->overlap.storageId() != unpad->tail.storageId()) return false;
        if (!sole_consumer(ordered, unpad->body, ola) || !sole_consumer(ordered, unpad->tail, ola)) return false;
        if (ola->buffer.shape() != unpad->body.shape() || ola->overlap.shape() != unpad->tail.shape() ||
            ola->buffer.offset() != 0 || ola->overlap.offset() != 0 || !ola->buffer.contiguous() ||
            !ola->overlap.contiguous() || !ola->output.contiguous() || ola->output.offset() != 0)
            return false;
        // the table cycle k's epilogue reads is left by cycle k - 1's overlap kernel; the FIRST submission of this unit (and the
        // first one after a cycle that died between its two launches) builds it from `phases` as it stands then, on the
        // unit's own stream -- no device work in this predicate.  A host that rewrites the `phases` state of a
        // running chain re-creates the runtime (the standalone module rebuilds the table every cycle; the fused unit does not).
        auto table_ready = std::make_shared<bool>(false);
        if (phase) {
            members = {fft, norm, phase, unpad, ola};
            consumed = 5;
            name = "ifft_phase_unpad_overlap(" + fft->name() + "+" + norm->name() + "+" + phase->name() + "+" + unpad->name() +
                   "+" + ola->name() + ")";
        } else {
            members = {fft, norm, unpad, ola};
            consumed = 4;
            name = "ifft_unpad_overlap(" + fft->name() + "+" + norm->name() + "+" + unpad->name() + "+" + ola->name() + ")";
        }
        submit = [fft, norm, phase, unpad, ola, axis, batchDiv, chanDiv, table_ready](hipStream_t stream) -> Result {
            dev::FftLayout L;
            JST_CHECK(fft->layout(L));
            const U64 n = fft->input.shape(axis);
            if (phase) {
                if (!*table_ready)
                    JST_CHECK(hip_result(kernels::launch_phase_table_prime(ptr<float2>(phase->corrections), ptr<double>(phase->phases),
                                                                           ptr<const double>(phase->devIncrements), phase->channelCount,
                                                                           phase->batchCount, stream),
                                         "phase_correction (table) kernel"));
                *table_ready = false;  // until the overlap kernel below has been enqueued, the next cycle's table is not on its way
                JST_CHECK(hip_result(kernels::launch_fft_c2c_tiled_scaled_phase_unpad(
                                         n, L, fft->twiddles, ptr<const float2>(fft->input), ptr<float2>(fft->scratchA),
                                         ptr<float2>(ola->output), ptr<float2>(unpad->tail), norm->constant,
                                         unpad->body.shape(axis), ptr<const float2>(phase->corrections), phase->batchCount,
                                         batchDiv, phase->channelCount, chanDiv, stream),
                                     "fft (tiled, multiply_constant + phase_correction + unpad epilogue) kernel"));
                JST_CHECK(hip_result(kernels::launch_overlap_heads_phase(
                                      ptr<char>(ola->output), ptr<char>(ola->overlap), ptr<char>(ola->previousOverlap), true,
                                      (uint32_t)ola->buffer.rank(), ola->batchAxis ? (int32_t)*ola->batchAxis : -1,
                                      ola->buffer.shape().data(), ola->overlap.shape().data(),
                                      ptr<float2>(phase->corrections), ptr<double>(phase->phases),
                                      ptr<const double>(phase->devIncrements), phase->channelCount, phase->batchCount, stream),
                                  "overlap_add (overlap region) + phase_correction (state) kernel"));
                *table_ready = true;
                return Result::SUCCESS;
            }
            JST_CHECK(hip_result(kernels::launch_fft_c2c_tiled_scaled_unpad(
                                     n, fft->forward, L, fft->twiddles, ptr<const float2>(fft->input),
                                     ptr<float2>(fft->scratchA), ptr<float2>(ola->output), ptr<float2>(unpad->tail),
                                     norm->constant, unpad->body.shape(axis), stream),
                                 "fft (tiled, multiply_constant + unpad epilogue) kernel"));
            return hip_result(kernels::launch_overlap_heads(
                                  ptr<char>(ola->output), ptr<char>(ola->overlap), ptr<char>(ola->previousOverlap), true,
                                  (uint32_t)ola->buffer.rank(), ola->batchAxis ? (int32_t)*ola->batchAxis : -1,
                                  ola->buffer.shape().data(), ola->overlap.shape().data(), stream),
                              "overlap_add (overlap region) kernel");
        };
        return true;
    }
    // multiply(CF32, broadcast) -> fold(last axis): fold reads the operands and forms the product
    if (auto* mul = dynamic_cast<Multiply*>(ordered[at])) {
        auto* fold = dynamic_cast<Fold*>(ordered[at + 1]);
        if (!fold || std::string(mul->type()) != "multiply") return false;
        if (fold->input.storageId() != mul->c.storageId() || !sole_consumer(ordered, mul->c, fold))
            return false;
        if (mul->c.dtype() != DataType::CF32 || !mul->c.contiguous() || mul->c.offset() != 0) return false;
        if (fold->resolvedAxis + 1 != mul->c.rank() || !fold->output.contiguous()) return false;
        members = {mul, fold};
        consumed = 2;
        name = "fold_product(" + mul->name() + "+" + fold->name() + ")";
        submit = [mul, fold](hipStream_t stream) -> Result {
            EwLayout P;
            if (!MakeEwLayout(mul->c, &mul->a, &mul->b, P)) {
                JST_ERROR("[MODULE_MULTIPLY] Unsupported tensor rank.");
                return Result::ERROR;
            }
            P.contiguous = 0;  // operands are addressed through their (broadcast) strides
            const Index axis = fold->resolvedAxis;
            U64 chanCount = 1, chanInner = 1;
            if (fold->channelAxis) {
                chanCount = fold->output.shape(*fold->channelAxis);
                for (Index i = *fold->channelAxis + 1; i < fold->output.rank(); ++i)
                    chanInner *= fold->output.shape(i);
            }
            return hip_result(
                kernels::launch_fold_product_cf32(
                    ptr<float2>(fold->output) + fold->output.offset(), ptr<const float2>(mul->a),
                    ptr<const float2>(mul->b), P, mul->c.shape(axis), fold->size,
                    fold->offset % mul->c.shape(axis),
                    fold->channelAxis ? ptr<const uint64_t>(fold->devOffsets) : nullptr, chanCount,
                    chanInner, stream),
                "fold (of product) kernel");
        };
        return true;
    }
    return false;
}

// duplicate (the dense copy a `slice` block puts behind its view, core/slice/block_impl.cc with `contiguous: true`) whose EVERY
// reader addresses its input through strides anyway -- a Multiply that fuses into `fft_windowed` (the transform's first load
// walks the operand's strides) or an Fm (lane / batch / sample strides): the readers are pointed at the view itself and the copy
// is not made.  The duplicate's own output tensor is then an intermediate nobody reads (the contract of every fusion here);
// pointing a reader at the source is also right when a later, unfused runtime runs the copy again -- same bytes either way.
// JST_NO_CHAIN_FUSION=1 keeps the copy.
bool TryElideDuplicate(const std::vector<Module*>& ordered, size_t at, std::string& name, std::vector<Module*>& members,
                       std::function<Result(hipStream_t)>& submit, size_t& consumed) {
    static const bool off = std::getenv("JST_NO_CHAIN_FUSION") != nullptr;
    auto* dup = dynamic_cast<Duplicate*>(ordered[at]);
    if (off || !dup || dup->input.dtype() != DataType::CF32 || dup->input.shape() != dup->output.shape()) return false;
    if (dup->input.storageId() == dup->output.storageId()) return false;
    std::vector<Multiply*> muls;
    std::vector<Fm*> fms;
    for (size_t i = 0; i < ordered.size(); ++i) {
        Module* m = ordered[i];
        if (m == dup) continue;
        if (const auto* c = dynamic_cast<const Cast*>(m); c && c->bypass) continue;  // a pure alias reads nothing
        bool reads = false;
        for (const auto& kv : m->inputs()) reads |= kv.second.storageId() == dup->output.storageId();
        if (!reads) continue;
        if (i < at) return false;
        if (auto* mul = dynamic_cast<Multiply*>(m)) {
            if (std::string(mul->type()) != "multiply" || mul->a.storageId() != dup->output.storageId() ||
                mul->b.storageId() == dup->output.storageId() || mul->a.shape() != dup->output.shape() || mul->a.offset() != 0 ||
                !mul->a.contiguous())
                return false;
            // (a chain multiply -> fft -> amplitude belongs to TryFuseSpectrum, which wants its dense operand: leave it alone)
            if (i + 2 < ordered.size() && dynamic_cast<Amplitude*>(ordered[i + 2])) return false;
            std::string n2;
            std::vector<Module*> mem2;
            std::function<Result(hipStream_t)> sub2;
            size_t cons2 = 0;
            if (!TryFuseMultiplyFft(ordered, i, n2, mem2, sub2, cons2)) return false;  // only the fused unit walks strides in its load
            muls.push_back(mul);
        } else if (auto* fm = dynamic_cast<Fm*>(m)) {
            if (fm->input.storageId() != dup->output.storageId() || fm->input.shape() != dup->output.shape() || fm->input.offset() != 0)
                return false;
            fms.push_back(fm);
        } else {
            return false;
        }
    }
    if (muls.empty() && fms.empty()) return false;  // nobody inside the runtime reads it: somebody outside may
    // The copy also protected its readers from a module that WRITES the source's storage between the duplicate and its last
    // reader: with such a module in between, the copy stays.
    size_t last_reader = at;
    for (size_t i = at + 1; i < ordered.size(); ++i) {
        for (Multiply* mul : muls) if (ordered[i] == mul) last_reader = i;
        for (Fm* fm : fms) if (ordered[i] == fm) last_reader = i;
    }
    for (size_t i = at + 1; i < last_reader; ++i)
        for (const auto& kv : ordered[i]->outputs())
            if (kv.second.storageId() == dup->input.storageId()) return false;
    for (Multiply* mul : muls) {
        dup->rewired.emplace_back(&mul->a, mul->a);
        mul->a = dup->input;
    }
    for (Fm* fm : fms) {
        dup->rewired.emplace_back(&fm->input, fm->input);
        fm->input = dup->input;
    }
    members = {dup};
    consumed = 1;
    name = dup->name() + "(elided)";
    submit = [](hipStream_t) -> Result { return Result::SUCCESS; };
    return true;
}

JST_REGISTER_MODULE(SignalGenerator, "signal_generator", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Lineplot, "lineplot", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Pad, "pad", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Unpad, "unpad", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Fold, "fold", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(OverlapAdd, "overlap_add", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(FirTaps, "fir_taps", DeviceType::HIP, RuntimeType::NATIVE, "fast");
JST_REGISTER_MODULE(FirDecimate, "fir_decimate", DeviceType::HIP, RuntimeType::NATIVE, "fast");
JST_REGISTER_MODULE(PhaseCorrection, "phase_correction", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(FilterTaps, "filter_taps", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Arithmetic, "arithmetic", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(ExpandDims, "expand_dims", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(SqueezeDims, "squeeze_dims", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Fm, "fm", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Duplicate, "duplicate", DeviceType::HIP, RuntimeType::NATIVE, "generic");

}  // namespace jst::modules
