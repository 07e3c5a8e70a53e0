// ingest_modules.cc -- the modules either side of the hot path that the example flowgraphs use
// (SURVEY §8f rows 3 and 4): slice (view selection between the Filter block and its consumers)
// and agc (the optional stage of the spectrum_engine block, spectrum_engine/block_impl.cc:185-196).
// Integer-format cast and add live next to their float siblings in modules.cc.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>

#include "modules.hh"

namespace jst::modules {

namespace {

const char* kSpace = " \t\n\r\f\v";

bool is_unsigned(const std::string& v) {
    return !v.empty() && std::all_of(v.begin(), v.end(), [](char c) { return c >= '0' && c <= '9'; });
}
bool parse_u64(const std::string& v, U64& out) {
    errno = 0;
    char* end = nullptr;
    const unsigned long long r = std::strtoull(v.c_str(), &end, 10);
    if (errno == ERANGE || end != v.c_str() + v.size()) return false;
    out = r;
    return true;
}

}  // namespace

// ---- Slice (core/slice/module_impl.cc:9-257; layout rules src/memory/tensor.cc:308-440) --------
// "[a, b:c:d, ..., :]": a number drops the axis, a colon token keeps it (end defaults to the
// extent), one ellipsis expands to the untouched axes.  Pure view: no kernel.
class Slice : public Module {
 public:
    struct Token {
        enum Kind { NUMBER, COLON, ELLIPSIS } kind;
        U64 a = 0, b = 0, c = 1;
        bool hasEnd = false;
    };
    const char* type() const override { return "slice"; }
    Result parse(const std::string& text) {
        tokens.clear();
        if (text.empty()) {
            JST_ERROR("[MODULE_SLICE] Slice string cannot be empty.");
            return Result::ERROR;
        }
        if (text.front() != '[' || text.back() != ']') {
            JST_ERROR("[MODULE_SLICE] Invalid slice syntax: Missing brackets.");
            return Result::ERROR;
        }
        std::string inner = text.substr(1, text.size() - 2);
        const auto first = inner.find_first_not_of(kSpace);
        if (first == std::string::npos) {
            tokens.push_back({Token::ELLIPSIS});
            return Result::SUCCESS;
        }
        inner = inner.substr(first, inner.find_last_not_of(kSpace) - first + 1);
        size_t at = 0;
        while (at <= inner.size()) {
            const auto comma = inner.find(',', at);
            std::string el = inner.substr(at, comma == std::string::npos ? std::string::npos : comma - at);
            const auto s0 = el.find_first_not_of(kSpace);
            if (s0 == std::string::npos) {
                JST_ERROR("[MODULE_SLICE] Invalid slice syntax: Empty token.");
                return Result::ERROR;
            }
            el = el.substr(s0, el.find_last_not_of(kSpace) - s0 + 1);
            JST_CHECK(parseElement(el));
            if (comma == std::string::npos) break;
            at = comma + 1;
        }
        if (std::count_if(tokens.begin(), tokens.end(),
                          [](const Token& t) { return t.kind == Token::ELLIPSIS; }) > 1) {
            JST_ERROR("[MODULE_SLICE] Ellipsis can only appear once in a slice.");
            return Result::ERROR;
        }
        for (const Token& t : tokens)
            if (t.kind == Token::COLON && t.c == 0) {
                JST_ERROR("[MODULE_SLICE] Slice step cannot be zero.");
                return Result::ERROR;
            }
        return Result::SUCCESS;
    }
    Result parseElement(const std::string& el) {
        const auto bad = [&]() {
            JST_ERROR("[MODULE_SLICE] Invalid slice syntax: Invalid token '%s'.", el.c_str());
            return Result::ERROR;
        };
        const auto num = [&](const std::string& v, U64& out) {
            if (parse_u64(v, out)) return Result::SUCCESS;
            JST_ERROR("[MODULE_SLICE] Invalid numeric value in token '%s'.", el.c_str());
            return Result::ERROR;
        };
        if (el == "...") {
            tokens.push_back({Token::ELLIPSIS});
            return Result::SUCCESS;
        }
        const auto c1 = el.find(':');
        if (c1 != std::string::npos) {
            const auto c2 = el.find(':', c1 + 1);
            if ((c2 != std::string::npos && el.find(':', c2 + 1) != std::string::npos) ||
                (c2 != std::string::npos && c2 + 1 == el.size()))
                return bad();
            const std::string st = el.substr(0, c1);
            const std::string en = c2 == std::string::npos ? el.substr(c1 + 1) : el.substr(c1 + 1, c2 - c1 - 1);
            const std::string sp = c2 == std::string::npos ? std::string() : el.substr(c2 + 1);
            if ((!st.empty() && !is_unsigned(st)) || (!en.empty() && !is_unsigned(en)) ||
                (c2 != std::string::npos && !is_unsigned(sp)))
                return bad();
            Token t{Token::COLON};
            if (!st.empty()) JST_CHECK(num(st, t.a));
            if (!en.empty()) JST_CHECK(num(en, t.b));
            if (!sp.empty()) JST_CHECK(num(sp, t.c));
            t.hasEnd = !en.empty();
            tokens.push_back(t);
            return Result::SUCCESS;
        }
        if (is_unsigned(el)) {
            Token t{Token::NUMBER};
            JST_CHECK(num(el, t.a));
            tokens.push_back(t);
            return Result::SUCCESS;
        }
        return bad();
    }
    // Applies the tokens to `view` and fills axisMap[input axis] = output axis (or nullopt).
    Result apply(Tensor& view, std::vector<std::optional<Index>>& axisMap) const {
        const Index rank = view.rank();
        const size_t consuming = (size_t)std::count_if(
            tokens.begin(), tokens.end(), [](const Token& t) { return t.kind != Token::ELLIPSIS; });
        if (consuming > rank) {
            JST_ERROR("[MEMORY:TENSOR] Slice index exceeds dimensions.");
            return Result::ERROR;
        }
        axisMap.assign(rank, std::nullopt);
        std::vector<Index> dropped;
        Index in_axis = 0, out_axis = 0;
        for (const Token& t : tokens) {
            switch (t.kind) {
                case Token::NUMBER:
                    if (t.a >= view.shape(in_axis)) {
                        JST_ERROR("[MEMORY:TENSOR] Slice index %llu out of range %llu.",
                                  (unsigned long long)t.a, (unsigned long long)view.shape(in_axis));
                        return Result::ERROR;
                    }
                    JST_CHECK(view.slice(in_axis, t.a, t.a + 1, 1));
                    dropped.push_back(in_axis);
                    ++in_axis;
                    break;
                case Token::COLON: {
                    const U64 extent = view.shape(in_axis);
                    const U64 end = t.hasEnd ? t.b : extent;
                    if (t.a > extent || end > extent) {
                        JST_ERROR("[MEMORY:TENSOR] Slice range [%llu:%llu] exceeds dimension %llu.",
                                  (unsigned long long)t.a, (unsigned long long)end,
                                  (unsigned long long)extent);
                        return Result::ERROR;
                    }
                    if (end <= t.a) {
                        JST_ERROR("[MODULE_SLICE_NATIVE_HIP] Empty slices are not implemented on "
                                  "the HIP device.");
                        return Result::ERROR;
                    }
                    JST_CHECK(view.slice(in_axis, t.a, end, t.c));
                    axisMap[in_axis] = out_axis;
                    ++in_axis;
                    ++out_axis;
                    break;
                }
                case Token::ELLIPSIS: {
                    const Index expanded = rank - (Index)consuming;
                    for (Index i = 0; i < expanded; ++i) axisMap[in_axis + i] = out_axis + i;
                    in_axis += expanded;
                    out_axis += expanded;
                    break;
                }
            }
        }
        while (in_axis < rank) axisMap[in_axis++] = out_axis++;
        for (auto it = dropped.rbegin(); it != dropped.rend(); ++it) JST_CHECK(view.squeezeDims(*it));
        return Result::SUCCESS;
    }
    Result validate() override {
        JST_CHECK(parse(ConfigStr(config_, "slice", "[...]")));
        if (!inputs_.count("buffer")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("buffer");
        SignalAxes axes;
        JST_CHECK(MapSignalAxes(in, axes));
        if (in.validShape() && in.size() > 0) {
            Tensor probe = in.clone();
            std::vector<std::optional<Index>> map;
            JST_CHECK(apply(probe, map));
            if (probe.rank() == 0) {
                JST_ERROR("[MODULE_SLICE_NATIVE_HIP] Rank-zero results are not implemented on the "
                          "HIP device.");
                return Result::ERROR;
            }
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
        output = in.clone();
        std::vector<std::optional<Index>> map;
        JST_CHECK(apply(output, map));
        SignalAxes in_axes, out_axes;
        JST_CHECK(MapSignalAxes(in, in_axes));
        if (in_axes.sample) out_axes.sample = map[*in_axes.sample];
        if (in_axes.batch) out_axes.batch = map[*in_axes.batch];
        if (in_axes.channel) out_axes.channel = map[*in_axes.channel];
        JST_CHECK(SetSignalAxes(output, out_axes));
        produced("buffer", output);
        return Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t) override { return Result::SUCCESS; }
    bool launchesKernels() const override { return false; }
    std::vector<Token> tokens;
    Tensor output;
};

// ---- Agc (dsp/agc/{module_impl.cc:7-96, module_impl_native_cpu.cc:20-218}) ---------------------
class Agc : public Module {
 public:
    const char* type() const override { return "agc"; }
    Result validate() override {
        bool o0, o1, o2, o3, o4, o5;
        tileSize = ConfigU64(config_, "tileSize", 1024, &o0);
        reference = ConfigF64(config_, "reference", 1.0, &o1);
        epsilon = ConfigF64(config_, "epsilon", 1e-12, &o2);
        minGain = ConfigF64(config_, "minGain", 0.01, &o3);
        maxGain = ConfigF64(config_, "maxGain", 100.0, &o4);
        maxGainChange = ConfigF64(config_, "maxGainChange", 4.0, &o5);
        if (!o0 || tileSize == 0) {
            JST_ERROR("[MODULE_AGC] Tile size must be greater than zero.");
            return Result::ERROR;
        }
        if (!o1 || !std::isfinite(reference) || reference <= 0.0) {
            JST_ERROR("[MODULE_AGC] Reference must be finite and positive.");
            return Result::ERROR;
        }
        if (!o2 || !std::isfinite(epsilon) || epsilon <= 0.0) {
            JST_ERROR("[MODULE_AGC] Epsilon must be finite and positive.");
            return Result::ERROR;
        }
        if (!o3 || !std::isfinite(minGain) || minGain <= 0.0) {
            JST_ERROR("[MODULE_AGC] Minimum gain must be finite and positive.");
            return Result::ERROR;
        }
        if (!o4 || !std::isfinite(maxGain) || maxGain < minGain) {
            JST_ERROR("[MODULE_AGC] Maximum gain must be finite and no less than minimum gain.");
            return Result::ERROR;
        }
        if (!o5 || !std::isfinite(maxGainChange) || maxGainChange < 1.0) {
            JST_ERROR("[MODULE_AGC] Maximum gain change must be finite and at least one.");
            return Result::ERROR;
        }
        if (!inputs_.count("signal")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("signal");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        if (ResolveSignalAxes(in, axes) != Result::SUCCESS) {
            JST_ERROR("[MODULE_AGC] Input must contain valid signal axis metadata.");
            return Result::ERROR;
        }
        if (in.dtype() != DataType::CF32 && in.dtype() != DataType::F32) {
            JST_ERROR("[MODULE_AGC_NATIVE_HIP] Unsupported data type '%s'.", DataTypeName(in.dtype()));
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    // agc/module_impl.cc:84-94: every parameter moves in place -- except that the tile count sizes this
    // implementation's device buffers, so a new tileSize asks for a rebuild
    Result reconfigureImpl(const Config& previous) override {
        return ConfigU64(previous, "tileSize", 1024) == tileSize ? Result::SUCCESS : Result::RECREATE;
    }
    Result define() override {
        JST_CHECK(defineTaint(STATELESS));
        JST_CHECK(defineInterfaceInput("signal"));
        return defineInterfaceOutput("signal");
    }
    Result create() override {
        input = inputs_.at("signal");
        sampleAxis = *axes.sample;
        laneCount = input.size() / input.shape(sampleAxis);
        tiles = 1 + (input.shape(sampleAxis) - 1) / tileSize;
        JST_CHECK(output.create(device(), input.dtype(), input.shape()));
        JST_CHECK(output.propagateAttributes(input));
        JST_CHECK(gains.create(device(), DataType::F64, {laneCount, tiles, 2}));
        produced("signal", output);
        return Result::SUCCESS;
    }
    void params(kernels::AgcParams& p) const {
        std::memset(&p, 0, sizeof(p));
        p.lanes = laneCount;
        p.samples = input.shape(sampleAxis);
        p.tile = tileSize;
        p.tiles = tiles;
        int r = 0;
        for (Index ax = 0; ax < input.rank(); ++ax) {
            if (ax == sampleAxis) continue;
            p.lane_shape[r] = input.shape(ax);
            p.in_lane_stride[r] = (int64_t)input.stride(ax);
            p.out_lane_stride[r] = (int64_t)output.stride(ax);
            ++r;
        }
        p.lane_rank = r;
        p.in_sample_stride = (int64_t)input.stride(sampleAxis);
        p.out_sample_stride = (int64_t)output.stride(sampleAxis);
        p.in_offset = input.offset();
        p.out_offset = output.offset();
        p.reference = reference;
        p.epsilon = epsilon;
        p.min_gain = minGain;
        p.max_gain = maxGain;
        p.max_gain_change = maxGainChange;
    }
    Result computeSubmit(hipStream_t s) override {
        kernels::AgcParams p;
        params(p);
        return hip_result(kernels::launch_agc(output.data(), input.data(),
                                              input.dtype() == DataType::CF32,
                                              static_cast<double*>(gains.data()), p, s),
                          "agc kernel");
    }
    const Tensor* state(const std::string& key) const override {
        return key == "gains" ? &gains : nullptr;
    }
    Tensor input, output, gains;
    SignalAxes axes;
    Index sampleAxis = 0;
    U64 tileSize = 1024, laneCount = 0, tiles = 0;
    F64 reference = 1.0, epsilon = 1e-12, minGain = 0.01, maxGain = 100.0, maxGainChange = 4.0;
};

// ---- Squelch (dsp/squelch/module_impl.cc:8-62, module_impl_native_cpu.cc:28-98) -----------------
// Passes its input on (the output is a view of it) when the peak amplitude of the buffer exceeds the
// threshold, answers SKIP otherwise: the runtime then skips everything downstream for this cycle.  The
// decision is the host's, so the cycle waits for one device scalar -- not graph-capturable.
class Squelch : public Module {
 public:
    const char* type() const override { return "squelch"; }
    Result validate() override {
        bool ok = true;
        threshold = (F32)ConfigF64(config_, "threshold", 0.1, &ok);
        if (!ok || !std::isfinite(threshold) || threshold < 0.0f) {
            JST_ERROR("[MODULE_SQUELCH] Invalid threshold '%s', must be non-negative.",
                      ConfigStr(config_, "threshold", "?").c_str());
            return Result::ERROR;
        }
        if (!inputs_.count("signal")) return Result::SUCCESS;
        const Tensor& in = inputs_.at("signal");
        if (!in.validShape() || in.size() == 0) return Result::SUCCESS;
        if (in.dtype() != DataType::CF32 && in.dtype() != DataType::F32) {
            JST_ERROR("[MODULE_SQUELCH_NATIVE_HIP] Unsupported data type '%s'.", DataTypeName(in.dtype()));
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    Result define() override {
        JST_CHECK(defineInterfaceInput("signal"));
        return defineInterfaceOutput("signal");
    }
    Result create() override {
        input = inputs_.at("signal");
        output = input.clone();
        JST_CHECK(peak.create(device(), DataType::F32, {1}));
        JST_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&hostPeak), sizeof(float), hipHostMallocDefault),
                      "hipHostMalloc");
        *hostPeak = 0.0f;
        passing = false;
        produced("signal", output);
        return Result::SUCCESS;
    }
    Result destroy() override {
        if (hostPeak) (void)hipHostFree(hostPeak);
        hostPeak = nullptr;
        passing = false;
        return Result::SUCCESS;
    }
    Result reconfigureImpl(const Config&) override { return Result::SUCCESS; }  // the threshold moves in place
    Result computeSubmit(hipStream_t s) override {
        JST_HIP_CHECK(kernels::launch_peak_abs(static_cast<float*>(peak.data()),
                                               static_cast<const char*>(input.data()) + input.offsetBytes(),
                                               input.size(), input.dtype() == DataType::CF32, s),
                      "squelch peak kernel");
        JST_HIP_CHECK(hipMemcpyAsync(hostPeak, peak.data(), sizeof(float), hipMemcpyDeviceToHost, s), "hipMemcpyAsync");
        JST_HIP_CHECK(hipStreamSynchronize(s), "hipStreamSynchronize");
        passing = *hostPeak > threshold;
        return passing ? Result::SUCCESS : Result::SKIP;
    }
    bool capturable() const override { return false; }
    const Tensor* state(const std::string& key) const override { return key == "amplitude" ? &peak : nullptr; }
    Tensor input, output, peak;
    float* hostPeak = nullptr;
    F32 threshold = 0.1f;
    bool passing = false;
};

// agc (one tile per lane, CF32) -> amplitude -> range [-> waterfall]: the spectrum_engine block behind its transform when the
// AGC is enabled (spectrum_engine/block_impl.cc:183-217), and the Waterfall that usually reads the block's output.  The
// workgroup that found a lane's gain also forms the level of every sample it scales and -- with a Waterfall -- puts the row
// into its ring (kernels/agc.hip agc_single_tile_kernel<.., TAIL>): three (four) launch-floor kernels become one.  Same
// contract as the other fusions: the amplitude has no other reader and is not written; the scaled signal and the Range's
// output are.  JST_NO_CHAIN_FUSION=1 keeps the modules apart.
bool TryFuseAgcChain(const std::vector<Module*>& ordered, size_t at, std::string& name, std::vector<Module*>& members,
                     std::function<Result(hipStream_t)>& submit, size_t& consumed) {
    static const bool off = std::getenv("JST_NO_CHAIN_FUSION") != nullptr;
    if (off || at + 2 >= ordered.size()) return false;
    auto* agc = dynamic_cast<Agc*>(ordered[at]);
    auto* amp = dynamic_cast<Amplitude*>(ordered[at + 1]);
    auto* rng = dynamic_cast<Range*>(ordered[at + 2]);
    if (!agc || !amp || !rng || agc->tiles != 1 || agc->input.dtype() != DataType::CF32) return false;
    auto sole_reader = [&](const Tensor& t, const Module* consumer) {
        for (const Module* m : ordered) {
            if (m == consumer) continue;
            if (const auto* c = dynamic_cast<const Cast*>(m); c && c->bypass) continue;
            for (const auto& kv : m->inputs())
                if (kv.second.storageId() == t.storageId()) return false;
        }
        return true;
    };
    if (amp->input.storageId() != agc->output.storageId() || rng->input.storageId() != amp->output.storageId()) return false;
    if (!sole_reader(amp->output, rng)) return false;  // (the scaled signal is still written: the AGC's output port stays valid)
    // dense tensors of one shape all the way, the sample axis last: a lane of the AGC is a row of the Range's output
    if (agc->sampleAxis + 1 != agc->input.rank() || !agc->output.contiguous() || agc->output.offset() != 0) return false;
    if (amp->input.shape() != agc->output.shape() || !amp->input.contiguous() || amp->input.offset() != 0) return false;
    if (!amp->output.contiguous() || rng->input.shape() != amp->output.shape() || !rng->input.contiguous() ||
        rng->input.offset() != amp->output.offset() || !rng->output.contiguous() || rng->output.offset() != 0 ||
        rng->output.shape() != agc->output.shape())
        return false;
    const bool fast = amp->provider() == "fast";
    if ((rng->provider() == "fast") != fast) return false;
    Waterfall* wf = at + 3 < ordered.size() ? dynamic_cast<Waterfall*>(ordered[at + 3]) : nullptr;
    if (wf && (wf->input.storageId() != rng->output.storageId() || wf->input.offset() != 0 || !wf->input.contiguous() ||
               wf->input.rank() != 2 || wf->numberOfBatches != agc->laneCount ||
               wf->numberOfElements != agc->input.shape(agc->sampleAxis) || wf->frequencyBins.offset() != 0))
        wf = nullptr;
    members = {agc, amp, rng};
    name = "agc_amplitude_range(" + agc->name() + "+" + amp->name() + "+" + rng->name();
    if (wf) {
        members.push_back(wf);
        name = "agc_amplitude_range_waterfall(" + agc->name() + "+" + amp->name() + "+" + rng->name() + "+" + wf->name();
    }
    name += ")";
    consumed = members.size();
    submit = [agc, amp, rng, wf, fast](hipStream_t stream) -> Result {
        kernels::AgcParams p;
        agc->params(p);
        kernels::AgcTail t;
        t.level = static_cast<float*>(rng->output.data());
        t.coeff = amp->scalingCoeff;
        t.scale = rng->scalingCoeff;
        t.offset = rng->offsetCoeff;
        t.fast = fast ? 1 : 0;
        if (wf) {
            t.ring = static_cast<float*>(wf->frequencyBins.data());
            t.ring_state = static_cast<uint64_t*>(wf->ringState.data());
            t.height = wf->height;
        }
        return hip_result(kernels::launch_agc_tail(agc->output.data(), agc->input.data(), static_cast<double*>(agc->gains.data()), p, t, stream),
                          "agc + amplitude + range kernel");
    };
    return true;
}

JST_REGISTER_MODULE(Slice, "slice", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Agc, "agc", DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(Squelch, "squelch", DeviceType::HIP, RuntimeType::NATIVE, "generic");

}  // namespace jst::modules
