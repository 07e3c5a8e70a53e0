// view_modules.cc -- the small core modules the example flowgraphs put between the blocks of the
// hot path: flatten, permutation and signal_axes (metadata-only views of their
// input storage), ones_tensor (constant source, e.g. the unit taps of a multiply) and the AM
// envelope demodulator that shares the FM chain's lane layout.  Validation rules and error texts
// follow the reference modules cited at each class.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "modules.hh"

namespace jst::modules {

namespace {

std::string trim(const std::string& s) {
    size_t b = 0, e = s.size();
    while (b < e && std::isspace((unsigned char)s[b])) ++b;
    while (e > b && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(b, e - b);
}

// "[a, b, c]" (or a bare "a, b, c") -> unsigned entries; zero is a legal entry.
bool parse_u64_list(const std::string& text, std::vector<U64>& out) {
    out.clear();
    std::string body = trim(text);
    if (!body.empty() && body.front() == '[') {
        if (body.back() != ']') return false;
        body = trim(body.substr(1, body.size() - 2));
    }
    if (body.empty()) return true;
    size_t pos = 0;
    while (true) {
        const size_t comma = body.find(',', pos);
        const std::string tok = trim(body.substr(pos, comma == std::string::npos ? comma : comma - pos));
        if (tok.empty() || tok.size() > 19 ||
            !std::all_of(tok.begin(), tok.end(), [](char c) { return c >= '0' && c <= '9'; }))
            return false;
        out.push_back(std::strtoull(tok.c_str(), nullptr, 10));
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return true;
}

}  // namespace

// ---- Flatten (core/flatten/module_impl.cc:7-42) -------------------------------------------------
// Contiguous input -> rank-1 view of the same storage.  Axis roles survive only when the geometry
// did not change (a rank-1 input); otherwise they are cleared.
class Flatten : public Module {
 public:
    const char* type() const override { return "flatten"; }
    Result validate() override { return Result::SUCCESS; }
    Result define() override {
        JST_CHECK(defineInterfaceInput("buffer"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        const Tensor& in = inputs_.at("buffer");
        SignalAxes axes;
        JST_CHECK(MapSignalAxes(in, axes));
        if (!in.contiguous()) {
            JST_ERROR("[MODULE_FLATTEN] Cannot flatten non-contiguous tensor. "
                      "Use the contiguous option or duplicate the tensor first.");
            return Result::ERROR;
        }
        Tensor view = in.clone();
        JST_CHECK(view.reshape({in.size()}));
        JST_CHECK(SetSignalAxes(view, in.shape() == view.shape() ? axes : SignalAxes{}));
        produced("buffer", view);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t) override { return Result::SUCCESS; }
    bool launchesKernels() const override { return false; }
};

// ---- Permutation (core/permutation/module_impl.cc:8-81) -----------------------------------------
// output axis k = input axis permutation[k]; a strided view, the axis roles follow their axes.
class Permutation : public Module {
 public:
    const char* type() const override { return "permutation"; }
    Result validate() override {
        std::vector<U64> p;
        if (!parse_u64_list(ConfigStr(config_, "permutation", "[0]"), p)) {
            JST_ERROR("[MODULE_PERMUTATION] Invalid permutation syntax.");
            return Result::ERROR;
        }
        if (p.empty()) {
            JST_ERROR("[MODULE_PERMUTATION] Permutation cannot be empty.");
            return Result::ERROR;
        }
        std::vector<bool> seen(p.size(), false);
        for (const U64 axis : p) {
            if (axis >= p.size()) {
                JST_ERROR("[MODULE_PERMUTATION] Axis %llu is out of range for permutation size %zu.",
                          (unsigned long long)axis, p.size());
                return Result::ERROR;
            }
            if (seen[axis]) {
                JST_ERROR("[MODULE_PERMUTATION] Axis %llu appears more than once.", (unsigned long long)axis);
                return Result::ERROR;
            }
            seen[axis] = true;
        }
        permutation.assign(p.begin(), p.end());
        if (!inputs_.count("buffer")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("buffer");
        SignalAxes axes;
        JST_CHECK(MapSignalAxes(in, axes));
        if (in.validShape() && in.size() > 0 && in.rank() != permutation.size()) {
            JST_ERROR("[MODULE_PERMUTATION] Input tensor rank %zu does not match permutation size %zu.",
                      (size_t)in.rank(), permutation.size());
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(DISCONTIGUOUS));
        JST_CHECK(defineInterfaceInput("buffer"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        const Tensor& in = inputs_.at("buffer");
        SignalAxes axes, moved;
        JST_CHECK(MapSignalAxes(in, axes));
        Tensor view = in.clone();
        JST_CHECK(view.permute(permutation));
        auto follow = [&](const std::optional<Index>& from, std::optional<Index>& to) {
            if (!from) return;
            for (Index k = 0; k < permutation.size(); ++k)
                if (permutation[k] == *from) to = k;
        };
        follow(axes.sample, moved.sample);
        follow(axes.batch, moved.batch);
        follow(axes.channel, moved.channel);
        JST_CHECK(SetSignalAxes(view, moved));
        produced("buffer", view);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t) override { return Result::SUCCESS; }
    bool launchesKernels() const override { return false; }
    std::vector<Index> permutation;
};

// ---- SignalAxes (core/signal_axes/module_impl.cc:8-107; layout grammar src/memory/axis.cc:101-196)
// axes = "[B, S]" re-tags the roles of the view; '_' clears an axis, '*' keeps whatever role the
// input had on that axis, "" leaves the metadata alone.  Storage is shared, nothing is launched.
class SignalAxesModule : public Module {
 public:
    const char* type() const override { return "signal_axes"; }
    struct Layout {
        bool specified = false;
        SignalAxes axes;
        std::vector<Index> inherited;
    };
    static Result parse(const std::string& value, Index rank, Layout& parsed) {
        parsed = {};
        Layout cand;
        std::string layout = trim(value);
        if (layout.empty()) return Result::SUCCESS;
        if (layout.front() != '[' || layout.back() != ']') {
            JST_ERROR("[MEMORY:AXIS] Signal axes '%s' must use bracketed notation.", value.c_str());
            return Result::ERROR;
        }
        layout = trim(layout.substr(1, layout.size() - 2));
        if (layout.empty()) {
            JST_ERROR("[MEMORY:AXIS] Signal axes cannot be empty.");
            return Result::ERROR;
        }
        if (layout.back() == ',') {
            JST_ERROR("[MEMORY:AXIS] Signal axes contain an empty entry.");
            return Result::ERROR;
        }
        Index axis = 0;
        size_t pos = 0;
        while (true) {
            const size_t comma = layout.find(',', pos);
            const std::string tok = trim(layout.substr(pos, comma == std::string::npos ? comma : comma - pos));
            if (tok.size() != 1 || !std::strchr("BCS_*", tok[0])) {
                JST_ERROR("[MEMORY:AXIS] Signal axes entry '%s' must be one of B, C, S, _, or *.", tok.c_str());
                return Result::ERROR;
            }
            if (axis >= rank) {
                JST_ERROR("[MEMORY:AXIS] Signal axes describe more than %zu dimensions.", (size_t)rank);
                return Result::ERROR;
            }
            std::optional<Index>* role = tok[0] == 'B'   ? &cand.axes.batch
                                         : tok[0] == 'C' ? &cand.axes.channel
                                         : tok[0] == 'S' ? &cand.axes.sample
                                                         : nullptr;
            if (tok[0] == '*') cand.inherited.push_back(axis);
            if (role) {
                if (*role) {
                    JST_ERROR("[MEMORY:AXIS] Signal axes use role '%c' more than once.", tok[0]);
                    return Result::ERROR;
                }
                *role = axis;
            }
            ++axis;
            if (comma == std::string::npos) break;
            pos = comma + 1;
        }
        cand.specified = true;
        parsed = cand;
        return Result::SUCCESS;
    }
    Result validate() override {
        overrideAxes = false;
        target = {};
        if (!inputs_.count("buffer")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("buffer");
        if (!in.validShape()) return Result::SUCCESS;
        Layout layout;
        if (parse(ConfigStr(config_, "axes", ""), in.rank(), layout) != Result::SUCCESS) {
            JST_ERROR("[MODULE_SIGNAL_AXES] Invalid axes layout.");
            return Result::ERROR;
        }
        SignalAxes inherited;
        if (!layout.specified) {
            if (MapSignalAxes(in, inherited) != Result::SUCCESS) {
                JST_ERROR("[MODULE_SIGNAL_AXES] Input contains invalid signal axis metadata.");
                return Result::ERROR;
            }
            return Result::SUCCESS;
        }
        target = layout.axes;
        if (!layout.inherited.empty()) {
            if (MapSignalAxes(in, inherited) != Result::SUCCESS) {
                JST_ERROR("[MODULE_SIGNAL_AXES] Cannot inherit invalid input signal axis metadata.");
                return Result::ERROR;
            }
            if (!in.hasAttribute(SampleAxisAttribute)) inherited.sample.reset();  // implicit rank-1 role
            auto inherit = [&](char role, const std::optional<Index>& from, std::optional<Index>& to) {
                if (!from || std::find(layout.inherited.begin(), layout.inherited.end(), *from) ==
                                 layout.inherited.end())
                    return Result::SUCCESS;
                if (to) {
                    JST_ERROR("[MODULE_SIGNAL_AXES] Role '%c' is assigned to axis %zu and inherited from axis %zu.",
                              role, (size_t)*to, (size_t)*from);
                    return Result::ERROR;
                }
                to = from;
                return Result::SUCCESS;
            };
            JST_CHECK(inherit('B', inherited.batch, target.batch));
            JST_CHECK(inherit('C', inherited.channel, target.channel));
            JST_CHECK(inherit('S', inherited.sample, target.sample));
        }
        overrideAxes = true;
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(DISCONTIGUOUS | STATELESS));
        JST_CHECK(defineInterfaceInput("buffer"));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        Tensor view = inputs_.at("buffer").clone();
        if (overrideAxes) JST_CHECK(SetSignalAxes(view, target));
        produced("buffer", view);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t) override { return Result::SUCCESS; }
    bool launchesKernels() const override { return false; }
    bool overrideAxes = false;
    SignalAxes target;
};

// ---- OnesTensor (core/ones_tensor/module_impl.cc:19-103) ----------------------------------------
// STATIC_OUTPUT source: a dense tensor of ones (complex: 1 + 0j) in HBM, filled at create and
// re-filled by every compute like the reference's computeSubmit.
class OnesTensor : public Module {
 public:
    const char* type() const override { return "ones_tensor"; }
    Result validate() override {
        std::vector<U64> dims;
        if (!parse_u64_list(ConfigStr(config_, "shape", "[1]"), dims)) {
            JST_ERROR("[MODULE_ONES_TENSOR] Invalid shape syntax.");
            return Result::ERROR;
        }
        if (dims.empty()) {
            JST_ERROR("[MODULE_ONES_TENSOR] Shape cannot be empty.");
            return Result::ERROR;
        }
        for (size_t axis = 0; axis < dims.size(); ++axis) {
            if (dims[axis] == 0) {
                JST_ERROR("[MODULE_ONES_TENSOR] Shape dimension %zu cannot be zero.", axis);
                return Result::ERROR;
            }
        }
        const std::string name = ConfigStr(config_, "dataType", "F32");
        if (name != "F32" && name != "CF32" && name != "F64" && name != "CF64") {
            JST_ERROR("[MODULE_ONES_TENSOR] Invalid data type '%s'.", name.c_str());
            return Result::ERROR;
        }
        dtype = NameToDataType(name);
        U64 count = 1;
        for (const U64 d : dims) {
            if (__builtin_mul_overflow(count, d, &count)) {
                JST_ERROR("[MODULE_ONES_TENSOR] Shape exceeds the supported layout range.");
                return Result::ERROR;
            }
        }
        U64 bytes = 0;
        if (__builtin_mul_overflow(count, (U64)DataTypeSize(dtype), &bytes)) {
            JST_ERROR("[MODULE_ONES_TENSOR] Tensor exceeds the supported byte range.");
            return Result::ERROR;
        }
        if (bytes > (288ull << 30)) {  // one MI355X: 288 GB of HBM3E
            JST_ERROR("[MODULE_ONES_TENSOR_NATIVE_HIP] Output allocation size is too large.");
            return Result::ERROR;
        }
        shape.assign(dims.begin(), dims.end());
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineTaint(STATIC_OUTPUT));
        return defineInterfaceOutput("buffer");
    }
    Result create() override {
        JST_CHECK(output.create(device(), dtype, shape));
        JST_CHECK(fill(nullptr));
        JST_HIP_CHECK(hipStreamSynchronize(nullptr), "ones_tensor fill");
        produced("buffer", output);
        return Result::SUCCESS;
    }
    Result fill(hipStream_t s) {
        const bool pair = dtype == DataType::CF32 || dtype == DataType::CF64;
        return hip_result(kernels::launch_fill_ones(output.data(), output.size(), (int)DataTypeSize(dtype), pair, s),
                          "ones_tensor kernel");
    }
    Result computeSubmit(hipStream_t s) override { return fill(s); }
    Tensor output;
    Shape shape;
    DataType dtype = DataType::F32;
};

// ---- AM (dsp/am/module_impl.cc:8-75, module_impl_native_cpu.cc:20-101) --------------------------
class Am : public Module {
 public:
    const char* type() const override { return "am"; }
    Result validate() override {
        bool okRate, okAlpha;
        sampleRate = (F32)ConfigF64(config_, "sampleRate", 240e3, &okRate);
        dcAlpha = (F32)ConfigF64(config_, "dcAlpha", 0.995, &okAlpha);
        if (!okRate || !std::isfinite(sampleRate) || sampleRate <= 0.0f) {
            JST_ERROR("[MODULE_AM] Sample rate must be finite and positive.");
            return Result::ERROR;
        }
        if (!okAlpha || !std::isfinite(dcAlpha) || dcAlpha < 0.0f || dcAlpha >= 1.0f) {
            JST_ERROR("[MODULE_AM] DC alpha must be in range [0, 1).");
            return Result::ERROR;
        }
        laneCount = 0;
        if (!inputs_.count("signal")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("signal");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        if (ResolveSignalAxes(in, axes) != Result::SUCCESS) {
            JST_ERROR("[MODULE_AM] Input must contain valid signal axis metadata.");
            return Result::ERROR;
        }
        if (in.dtype() != DataType::CF32) {
            JST_ERROR("[MODULE_AM_NATIVE_HIP] Input must be complex (CF32).");
            return Result::ERROR;
        }
        if (in.rank() - 1 - (axes.batch ? 1 : 0) > (Index)dev::kMaxRank) {
            JST_ERROR("[MODULE_AM_NATIVE_HIP] Too many lane axes.");
            return Result::ERROR;
        }
        laneCount = in.size() / in.shape(*axes.sample);
        if (axes.batch) laneCount /= in.shape(*axes.batch);
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineInterfaceInput("signal"));
        return defineInterfaceOutput("signal");
    }
    Result create() override {
        input = inputs_.at("signal");
        JST_CHECK(output.create(device(), DataType::F32, input.shape()));
        JST_CHECK(output.propagateAttributes(input));
        JST_CHECK(SetSignalAxes(output, axes));
        output.setAttribute("frequency", AttrValue{F64{0.0}});
        // zero state = the reference's prevEnvelope / prevOutput reset
        JST_CHECK(states.create(device(), DataType::U8, {std::max<U64>(laneCount, 1) * (U64)kernels::am_state_bytes()}));
        produced("signal", output);
        return Result::SUCCESS;
    }
    Result reconfigureImpl(const Config& previous) override {
        // dcAlpha is read by every submission; a new sample rate changes nothing the kernel uses
        (void)previous;
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
        return hip_result(kernels::launch_am(static_cast<float*>(output.data()),
                                             static_cast<const float2*>(input.data()), states.data(), dcAlpha, L, s),
                          "am kernel");
    }
    Tensor input, output, states;
    F32 sampleRate = 240e3f, dcAlpha = 0.995f;
    SignalAxes axes;
    U64 laneCount = 0;
};

JST_REGISTER_MODULE(Flatten, "flatten", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Permutation, "permutation", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(SignalAxesModule, "signal_axes", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(OnesTensor, "ones_tensor", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Am, "am", DeviceType::HIP, RuntimeType::NATIVE, "generic");

}  // namespace jst::modules
