// abi.cc -- extern "C" surface of libjetstream_hip.so (include/jetstream_hip.h).
#pragma GCC visibility push(default)  // the C ABI is the only exported surface
#include "../../include/jetstream_hip.h"
#pragma GCC visibility pop

#include <cstring>
#include <mutex>

#include "jst/comm.hh"
#include "jst/switches.hh"
#include "jst/module.hh"
#include "modules/modules.hh"

using namespace jst;

struct jst_tensor_s {
    Tensor t;
};
struct jst_module_s {
    std::shared_ptr<Module> m;  // shared with every runtime the module was handed to
    bool initialized = false;
};
struct jst_comm_s {
    Comm c;
};
struct jst_runtime_s {
    Runtime rt;
    std::vector<std::shared_ptr<Module>> keep;  // modules outlive the runtime that runs them
    ~jst_runtime_s() { (void)rt.destroy(); }    // before 'keep' lets go of them
};

extern "C" {

__attribute__((visibility("default"))) const JetstreamPluginAbi jetstream_plugin_abi = {
    0x4a535450u, (uint32_t)sizeof(JetstreamPluginAbi), 1u};

}  // extern "C"

namespace {

inline jst_result R(Result r) { return static_cast<jst_result>(r); }

#undef JST_HIP_CHECK
#define JST_HIP_CHECK(call, what)                                                       \
    do {                                                                                \
        const hipError_t jst_hip_err_ = (call);                                         \
        if (jst_hip_err_ != hipSuccess) {                                               \
            JST_ERROR("[HIP] %s failed: %s", what, hipGetErrorString(jst_hip_err_));    \
            return static_cast<jst_result>(::jst::Result::ERROR);                       \
        }                                                                               \
    } while (0)

#define JST_ARG(cond, msg)                 \
    do {                                   \
        if (!(cond)) {                     \
            JST_ERROR("[ABI] %s", msg);    \
            return R(Result::ERROR);       \
        }                                  \
    } while (0)

DataType to_dtype(uint8_t d) {
    switch (d) {
        case JST_DTYPE_F32: return DataType::F32;
        case JST_DTYPE_CF32: return DataType::CF32;
        case JST_DTYPE_F64: return DataType::F64;
        case JST_DTYPE_U64: return DataType::U64;
        default:
            // the integer sample formats share their numeric value with jst::DataType
            if (d >= JST_DTYPE_I8 && d <= JST_DTYPE_CF64) return static_cast<DataType>(d);
            return DataType::None;
    }
}
DeviceType to_device(uint8_t d) {
    if (d == JST_DEVICE_HIP) return DeviceType::HIP;
    if (d == JST_DEVICE_CPU) return DeviceType::CPU;
    return DeviceType::None;
}

// Side stream for async H2D feeds (src/memory/buffer_cuda.cc:284-308 copies on the context's
// stream; here uploads overlap compute and are fenced by an event).
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    bool pending = false;
    std::mutex mu;
    jst_result ensure() {
        if (stream) return R(Result::SUCCESS);
        JST_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "hipStreamCreate");
        JST_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming), "hipEventCreate");
        return R(Result::SUCCESS);
    }
};
SideStream& side() {
    static SideStream s;
    return s;
}

size_t join_lines(const std::vector<std::string>& v, char* buffer, size_t capacity) {
    std::string s;
    for (const auto& e : v) {
        s += e;
        s += '\n';
    }
    if (buffer && capacity) {
        const size_t n = s.size() < capacity - 1 ? s.size() : capacity - 1;
        std::memcpy(buffer, s.data(), n);
        buffer[n] = '\0';
    }
    return v.size();
}

}  // namespace

extern "C" {

const char* jst_version(void) { return "jetstream-hip 0.1.0 (gfx950)"; }
const char* jst_last_error(void) { return last_error(); }
const char* jst_result_name(jst_result r) { return ResultName(static_cast<Result>(r)); }

int jst_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
jst_result jst_device_set(int ordinal) {
    JST_HIP_CHECK(hipSetDevice(ordinal), "hipSetDevice");
    return R(Result::SUCCESS);
}
size_t jst_registry_list(char* buffer, size_t capacity) {
    return join_lines(Registry::instance().listAvailableModules(), buffer, capacity);
}

// ---- tensors -----------------------------------------------------------------------------------
jst_result jst_tensor_create(uint8_t device, uint8_t dtype, uint32_t rank, const uint64_t* shape,
                             jst_tensor* out) {
    return jst_tensor_create_ring(device, dtype, rank, shape, 1, out);
}
jst_result jst_tensor_create_ring(uint8_t device, uint8_t dtype, uint32_t rank,
                                  const uint64_t* shape, uint64_t slots, jst_tensor* out) {
    JST_ARG(out && (shape || rank == 0) && rank <= JST_MAX_RANK, "invalid tensor arguments");
    auto h = std::make_unique<jst_tensor_s>();
    const Result r =
        h->t.createRing(to_device(device), to_dtype(dtype), Shape(shape, shape + rank), slots);
    if (r != Result::SUCCESS) return R(r);
    *out = h.release();
    return R(Result::SUCCESS);
}
jst_result jst_tensor_wrap(void* ptr, size_t bytes, uint8_t device, uint8_t dtype, uint32_t rank,
                           const uint64_t* shape, const uint64_t* stride, uint64_t offset,
                           jst_tensor* out) {
    JST_ARG(out && ptr && shape && rank <= JST_MAX_RANK, "invalid tensor arguments");
    auto h = std::make_unique<jst_tensor_s>();
    std::vector<U64> st;
    if (stride) st.assign(stride, stride + rank);
    const Result r = h->t.wrap(ptr, bytes, to_device(device), to_dtype(dtype),
                               Shape(shape, shape + rank), st, offset);
    if (r != Result::SUCCESS) return R(r);
    *out = h.release();
    return R(Result::SUCCESS);
}
jst_result jst_tensor_view(jst_tensor base, uint32_t rank, const uint64_t* shape, const uint64_t* stride, uint64_t offset,
                           jst_tensor* out) {
    JST_ARG(base && out && (shape || rank == 0) && rank <= JST_MAX_RANK, "invalid tensor arguments");
    auto h = std::make_unique<jst_tensor_s>();
    std::vector<U64> st;
    if (stride) st.assign(stride, stride + rank);
    const Result r = h->t.view(base->t, Shape(shape, shape + rank), st, offset);
    if (r != Result::SUCCESS) return R(r);
    *out = h.release();
    return R(Result::SUCCESS);
}
jst_result jst_tensor_rebind(jst_tensor t, void* ptr, size_t bytes) {
    JST_ARG(t && ptr, "null tensor or pointer");
    return R(t->t.rebind(ptr, bytes));
}
jst_result jst_tensor_clone(jst_tensor t, jst_tensor* out) {
    JST_ARG(t && out, "null tensor");
    auto h = std::make_unique<jst_tensor_s>();
    h->t = t->t.clone();
    *out = h.release();
    return R(Result::SUCCESS);
}
jst_result jst_tensor_copy(jst_tensor dst, jst_tensor src, void* hip_stream) {
    JST_ARG(dst && src, "null tensor");
    return R(dst->t.copyFrom(src->t, static_cast<hipStream_t>(hip_stream)));
}
jst_result jst_tensor_destroy(jst_tensor t) {
    delete t;
    return R(Result::SUCCESS);
}
jst_result jst_tensor_describe(jst_tensor t, jst_tensor_desc* d) {
    JST_ARG(t && d, "null tensor");
    std::memset(d, 0, sizeof(*d));
    d->data = t->t.data();
    d->offset = t->t.offset();
    d->dtype = static_cast<uint8_t>(t->t.dtype());
    d->device = static_cast<uint8_t>(t->t.device());
    d->rank = (uint32_t)t->t.rank();
    JST_ARG(d->rank <= JST_MAX_RANK, "rank exceeds JST_MAX_RANK");
    for (uint32_t i = 0; i < d->rank; ++i) {
        d->shape[i] = t->t.shape(i);
        d->stride[i] = t->t.stride(i);
    }
    auto axis = [&](const char* key) -> int64_t {
        const AttrValue* v = t->t.attribute(key);
        const U64* idx = v ? std::get_if<U64>(v) : nullptr;
        return idx ? (int64_t)*idx : -1;
    };
    d->sample_axis = axis(SampleAxisAttribute);
    d->batch_axis = axis(BatchAxisAttribute);
    d->channel_axis = axis(ChannelAxisAttribute);
    return R(Result::SUCCESS);
}
jst_result jst_tensor_ring_select(jst_tensor t, uint64_t slot) {
    JST_ARG(t, "null tensor");
    return R(t->t.ringSelect(slot));
}
jst_result jst_tensor_reshape(jst_tensor t, uint32_t rank, const uint64_t* shape) {
    JST_ARG(t && shape, "null argument");
    return R(t->t.reshape(Shape(shape, shape + rank)));
}
jst_result jst_tensor_expand_dims(jst_tensor t, uint64_t axis) {
    JST_ARG(t, "null tensor");
    return R(t->t.expandDims(axis));
}
jst_result jst_tensor_squeeze_dims(jst_tensor t, uint64_t axis) {
    JST_ARG(t, "null tensor");
    return R(t->t.squeezeDims(axis));
}
jst_result jst_tensor_slice(jst_tensor t, uint64_t axis, uint64_t begin, uint64_t end,
                            uint64_t step) {
    JST_ARG(t, "null tensor");
    return R(t->t.slice(axis, begin, end, step));
}
jst_result jst_tensor_permute(jst_tensor t, uint32_t rank, const uint64_t* axes) {
    JST_ARG(t && axes, "null argument");
    return R(t->t.permute(std::vector<Index>(axes, axes + rank)));
}
jst_result jst_tensor_broadcast_to(jst_tensor t, uint32_t rank, const uint64_t* shape) {
    JST_ARG(t && shape, "null argument");
    return R(t->t.broadcastTo(Shape(shape, shape + rank)));
}
jst_result jst_tensor_set_attribute_u64(jst_tensor t, const char* key, uint64_t value) {
    JST_ARG(t && key, "null argument");
    return R(t->t.setAttribute(key, AttrValue{U64{value}}));
}
jst_result jst_tensor_set_attribute_f64(jst_tensor t, const char* key, double value) {
    JST_ARG(t && key, "null argument");
    return R(t->t.setAttribute(key, AttrValue{F64{value}}));
}
jst_result jst_tensor_set_attribute_u64v(jst_tensor t, const char* key, const uint64_t* values,
                                         uint64_t count) {
    JST_ARG(t && key && (values || count == 0), "null argument");
    return R(t->t.setAttribute(key, AttrValue{std::vector<U64>(values, values + count)}));
}
jst_result jst_tensor_set_attribute_f64v(jst_tensor t, const char* key, const double* values,
                                         uint64_t count) {
    JST_ARG(t && key && (values || count == 0), "null argument");
    return R(t->t.setAttribute(key, AttrValue{std::vector<F64>(values, values + count)}));
}
jst_result jst_tensor_remove_attribute(jst_tensor t, const char* key) {
    JST_ARG(t && key, "null argument");
    return R(t->t.removeAttribute(key));
}
jst_result jst_tensor_get_attribute_f64v(jst_tensor t, const char* key, double* values, uint64_t* count) {
    JST_ARG(t && key && count && (values || *count == 0), "null argument");
    const AttrValue* v = t->t.attribute(key);
    if (!v) {
        JST_ERROR("[ABI] tensor has no attribute '%s'", key);
        return R(Result::ERROR);
    }
    std::vector<double> flat;
    if (const U64* u = std::get_if<U64>(v)) flat = {(double)*u};
    else if (const F64* f = std::get_if<F64>(v)) flat = {*f};
    else if (const auto* fv = std::get_if<std::vector<F64>>(v)) flat = *fv;
    else if (const auto* uv = std::get_if<std::vector<U64>>(v)) flat.assign(uv->begin(), uv->end());
    const uint64_t room = *count;
    *count = flat.size();
    for (uint64_t i = 0; i < flat.size() && i < room; ++i) values[i] = flat[i];
    return R(Result::SUCCESS);
}
jst_result jst_tensor_copy_from_host(jst_tensor t, const void* src, size_t bytes) {
    JST_ARG(t && (src || bytes == 0), "null argument");
    const Result r = t->t.copyFromHost(src, bytes, nullptr);
    if (r != Result::SUCCESS) return R(r);
    JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
    return R(Result::SUCCESS);
}
jst_result jst_tensor_copy_to_host(jst_tensor t, void* dst, size_t bytes) {
    JST_ARG(t && (dst || bytes == 0), "null argument");
    JST_HIP_CHECK(hipDeviceSynchronize(), "hipDeviceSynchronize");
    const Result r = t->t.copyToHost(dst, bytes, nullptr);
    if (r != Result::SUCCESS) return R(r);
    JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
    return R(Result::SUCCESS);
}
jst_result jst_tensor_copy_from_host_async(jst_tensor t, const void* src, size_t bytes) {
    JST_ARG(t && src, "null argument");
    SideStream& s = side();
    std::lock_guard<std::mutex> lock(s.mu);
    if (const jst_result e = s.ensure(); e != 0) return e;
    const Result r = t->t.copyFromHost(src, bytes, s.stream);
    if (r != Result::SUCCESS) return R(r);
    JST_HIP_CHECK(hipEventRecord(s.done, s.stream), "hipEventRecord");
    s.pending = true;
    return R(Result::SUCCESS);
}

// ---- modules -----------------------------------------------------------------------------------
jst_result jst_module_create(const char* type, uint8_t device, const char* provider,
                             const char* name, const char* const* config, uint32_t n_config,
                             const char* const* input_ports, const jst_tensor* input_tensors,
                             uint32_t n_inputs, jst_module* out) {
    JST_ARG(type && provider && name && out, "null argument");
    JST_ARG(n_config == 0 || config, "null config");
    JST_ARG(n_inputs == 0 || (input_ports && input_tensors), "null inputs");
    auto m = Registry::instance().build(type, to_device(device), RuntimeType::NATIVE, provider);
    if (!m) return R(Result::ERROR);
    Config cfg;
    for (uint32_t i = 0; i < n_config; ++i) {
        const char* eq = config[i] ? std::strchr(config[i], '=') : nullptr;
        if (!eq) {
            JST_ERROR("[ABI] Config entry %u is not of the form key=value.", i);
            return R(Result::ERROR);
        }
        cfg[std::string(config[i], eq - config[i])] = std::string(eq + 1);
    }
    std::map<std::string, Tensor> inputs;
    for (uint32_t i = 0; i < n_inputs; ++i) {
        JST_ARG(input_ports[i] && input_tensors[i], "null input entry");
        inputs[input_ports[i]] = input_tensors[i]->t;
    }
    const Result r = m->construct(name, cfg, inputs);
    if (r != Result::SUCCESS) return R(r);
    auto h = std::make_unique<jst_module_s>();
    h->m = std::shared_ptr<Module>(m.release(), [](Module* p) {
        (void)p->teardown();  // Module::destroy() runs when the last owner lets go
        delete p;
    });
    *out = h.release();
    return R(Result::SUCCESS);
}
jst_result jst_module_destroy(jst_module m) {
    if (!m) return R(Result::SUCCESS);
    if (m->initialized) (void)m->m->computeDeinitialize();
    delete m;
    return R(Result::SUCCESS);
}
jst_result jst_module_reconfigure(jst_module m, const char* const* config, uint32_t n_config, int validate_only) {
    JST_ARG(m, "null argument");
    JST_ARG(n_config == 0 || config, "null config");
    Config cfg;
    for (uint32_t i = 0; i < n_config; ++i) {
        const char* eq = config[i] ? std::strchr(config[i], '=') : nullptr;
        if (!eq) {
            JST_ERROR("[ABI] Config entry %u is not of the form key=value.", i);
            return R(Result::ERROR);
        }
        cfg[std::string(config[i], eq - config[i])] = std::string(eq + 1);
    }
    return R(m->m->reconfigure(cfg, validate_only != 0));
}
jst_result jst_module_output(jst_module m, const char* port, jst_tensor* out) {
    JST_ARG(m && port && out, "null argument");
    auto it = m->m->outputs().find(port);
    if (it == m->m->outputs().end()) {
        JST_ERROR("[MODULE] Module '%s' has no output '%s'.", m->m->name().c_str(), port);
        return R(Result::ERROR);
    }
    auto h = std::make_unique<jst_tensor_s>();
    h->t = it->second;
    *out = h.release();
    return R(Result::SUCCESS);
}
jst_result jst_module_state(jst_module m, const char* key, jst_tensor* out) {
    JST_ARG(m && key && out, "null argument");
    const Tensor* t = m->m->state(key);
    if (!t) {
        JST_ERROR("[MODULE] Module '%s' has no state '%s'.", m->m->name().c_str(), key);
        return R(Result::ERROR);
    }
    auto h = std::make_unique<jst_tensor_s>();
    h->t = *t;
    *out = h.release();
    return R(Result::SUCCESS);
}
uint64_t jst_module_taint(jst_module m) { return m ? m->m->taint() : 0; }
jst_result jst_module_timing(jst_module m, uint64_t* cycles, double* ms) {
    JST_ARG(m, "null module");
    if (cycles) *cycles = m->m->timing.cycles;
    if (ms) *ms = m->m->timing.computeTimeMs;
    return R(Result::SUCCESS);
}
// ---- live ring source: producer side ---------------------------------------------------------------
namespace {
modules::RingSource* ring_of(jst_module m) {
    auto* r = m ? dynamic_cast<modules::RingSource*>(m->m.get()) : nullptr;
    if (!r) JST_ERROR("[MODULE_RING_SOURCE] Not a ring_source module.");
    return r;
}
}  // namespace
jst_result jst_ring_push(jst_module source, const void* samples, uint64_t count) {
    auto* r = ring_of(source);
    return r ? R(r->ringPush(samples, count)) : R(Result::ERROR);
}
// The reference's producer loop as native code (soapy/module_impl.cc:375-399: read <= chunk samples, push, repeat): `count`
// elements of `samples` go in as consecutive pushes of at most `chunk` elements.  What bench.py's host_fed times as
// push_8192 -- a Python loop of 512 ctypes calls per batch measured the interpreter, not the library.
jst_result jst_probe_ring_push_chunks(jst_module source, const void* samples, uint64_t count, uint64_t chunk) {
    auto* r = ring_of(source);
    if (!r || chunk == 0 || (count > 0 && !samples)) return R(Result::ERROR);
    const size_t eb = r->ringElementBytes();
    const char* p = static_cast<const char*>(samples);
    while (count > 0) {
        const uint64_t n = count < chunk ? count : chunk;
        const Result res = r->ringPush(p, n);
        if (res != Result::SUCCESS) return R(res);
        p += (size_t)n * eb;
        count -= n;
    }
    return R(Result::SUCCESS);
}
jst_result jst_ring_acquire(jst_module source, void** ptr, uint64_t* max_count) {
    auto* r = ring_of(source);
    return r ? R(r->ringAcquire(ptr, max_count)) : R(Result::ERROR);
}
jst_result jst_ring_commit(jst_module source, uint64_t count) {
    auto* r = ring_of(source);
    return r ? R(r->ringCommit(count)) : R(Result::ERROR);
}
jst_result jst_ring_wait(jst_module source, uint64_t size, uint32_t timeout_ms) {
    auto* r = ring_of(source);
    return r ? R(r->ringWait(size, timeout_ms)) : R(Result::ERROR);
}
jst_result jst_ring_clear(jst_module source) {
    auto* r = ring_of(source);
    return r ? R(r->ringClear()) : R(Result::ERROR);
}
uint64_t jst_ring_size(jst_module source) {
    auto* r = ring_of(source);
    return r ? r->ringSize() : 0;
}
uint64_t jst_ring_capacity(jst_module source) {
    auto* r = ring_of(source);
    return r ? r->ringCapacity() : 0;
}
uint64_t jst_ring_overflows(jst_module source) {
    auto* r = ring_of(source);
    return r ? r->ringOverflows() : 0;
}

jst_result jst_module_compute_initialize(jst_module m) {
    JST_ARG(m, "null module");
    const Result r = m->m->computeInitialize();
    m->initialized = (r == Result::SUCCESS);
    return R(r);
}
jst_result jst_module_compute_submit(jst_module m, void* stream) {
    JST_ARG(m, "null module");
    return R(m->m->computeSubmit(static_cast<hipStream_t>(stream)));
}
jst_result jst_module_compute_deinitialize(jst_module m) {
    JST_ARG(m, "null module");
    m->initialized = false;
    return R(m->m->computeDeinitialize());
}

// ---- runtime -----------------------------------------------------------------------------------
jst_result jst_runtime_create(const jst_module* modules, uint32_t n, uint32_t flags,
                              jst_runtime* out) {
    JST_ARG(out && (modules || n == 0), "null argument");
    std::vector<Module*> ms;
    for (uint32_t i = 0; i < n; ++i) {
        JST_ARG(modules[i], "null module");
        ms.push_back(modules[i]->m.get());
    }
    auto h = std::make_unique<jst_runtime_s>();
    for (uint32_t i = 0; i < n; ++i) h->keep.push_back(modules[i]->m);
    const Result r = h->rt.create(ms, flags);
    if (r != Result::SUCCESS) return R(r);
    *out = h.release();
    return R(Result::SUCCESS);
}
jst_result jst_runtime_destroy(jst_runtime r) {
    delete r;
    return R(Result::SUCCESS);
}
jst_result jst_runtime_compute(jst_runtime r, uint64_t cycles, int sync) {
    JST_ARG(r, "null runtime");
    {
        SideStream& s = side();
        std::lock_guard<std::mutex> lock(s.mu);
        if (s.pending) {  // order pending async uploads before this segment's work
            JST_HIP_CHECK(hipStreamWaitEvent(r->rt.stream(), s.done, 0), "hipStreamWaitEvent");
            s.pending = false;
        }
    }
    return R(r->rt.compute(cycles, sync != 0));
}
jst_result jst_runtime_synchronize(jst_runtime r) {
    JST_ARG(r, "null runtime");
    return R(r->rt.synchronize());
}
void* jst_runtime_stream(jst_runtime r) { return r ? r->rt.stream() : nullptr; }
uint64_t jst_runtime_period(jst_runtime r) { return r ? r->rt.period() : 0; }
int jst_runtime_graph_active(jst_runtime r) { return r && r->rt.graphActive() ? 1 : 0; }
jst_result jst_filter_plan(float sample_rate, float bandwidth, const float* center, uint64_t centers,
                           uint64_t taps, uint64_t heads, uint64_t signal_size, jst_filter_plan_desc* plan,
                           uint64_t* offsets) {
    JST_ARG(plan && (center || centers == 0) && (offsets || heads == 0) && taps > 0, "invalid filter plan arguments");
    jst::modules::FilterPlan p;
    const Result r = jst::modules::CalculateFilterPlan(sample_rate, bandwidth,
                                                       std::vector<float>(center, center + centers), taps, heads,
                                                       signal_size, p);
    if (r != Result::SUCCESS) return R(r);
    plan->pad_size = p.padSize;
    plan->convolution_size = p.convolutionSize;
    plan->resampler_size = p.resamplerSize;
    plan->resample = p.resample ? 1 : 0;
    plan->resampled_sample_rate = p.resampledSampleRate;
    for (uint64_t h = 0; h < heads; ++h) offsets[h] = h < p.resamplerOffsets.size() ? p.resamplerOffsets[h] : 0;
    return R(Result::SUCCESS);
}

jst_result jst_debug_set(const char* name, const char* value) {
    JST_ARG(name && switch_set(name, value), "unknown switch");
    return R(Result::SUCCESS);
}

// ---- collectives (jst/comm.cc: RCCL over xGMI, dlopen'ed) -------------------------------------------------------
int jst_comm_available(void) { return Comm::available() ? 1 : 0; }
jst_result jst_comm_unique_id(uint8_t* id128) {
    JST_ARG(id128, "null argument");
    return R(Comm::uniqueId(id128));
}
jst_result jst_comm_init(uint32_t rank, uint32_t world, const uint8_t* id128, jst_comm* out) {
    JST_ARG(out, "null argument");
    auto* c = new jst_comm_s();
    const Result r = c->c.create(rank, world, id128);
    if (r != Result::SUCCESS) {
        delete c;
        return R(r);
    }
    *out = c;
    return R(Result::SUCCESS);
}
jst_result jst_comm_destroy(jst_comm c) {
    delete c;
    return R(Result::SUCCESS);
}
uint32_t jst_comm_rank(jst_comm c) { return c ? c->c.rank() : 0; }
uint32_t jst_comm_world(jst_comm c) { return c ? c->c.world() : 0; }
uint64_t jst_comm_calls(jst_comm c) { return c ? c->c.calls() : 0; }
int jst_comm_uses_rccl(jst_comm c) { return c && c->c.usesRccl() ? 1 : 0; }
jst_result jst_comm_allreduce(jst_comm c, jst_tensor t, int op, int average, void* hip_stream) {
    JST_ARG(c && t && (op == JST_COMM_SUM || op == JST_COMM_MAX), "invalid all-reduce arguments");
    return R(c->c.allReduce(t->t, op == JST_COMM_SUM ? Comm::Op::SUM : Comm::Op::MAX, average != 0,
                            static_cast<hipStream_t>(hip_stream)));
}

size_t jst_runtime_order(jst_runtime r, char* buffer, size_t capacity) {
    return r ? join_lines(r->rt.order(), buffer, capacity) : 0;
}
size_t jst_runtime_units(jst_runtime r, char* buffer, size_t capacity) {
    return r ? join_lines(r->rt.units(), buffer, capacity) : 0;
}
double jst_runtime_unit_mean_ms(jst_runtime r, const char* prefix) {
    if (!r || !prefix) return -1.0;
    for (const auto& name : r->rt.units())
        if (name.compare(0, std::strlen(prefix), prefix) == 0) return r->rt.unitMeanMs(name);
    return -1.0;
}
double jst_runtime_event_overhead_ms(jst_runtime r) { return r ? r->rt.eventPairOverheadMs() : -1.0; }
double jst_runtime_unit_mean_cycles(jst_runtime r, const char* prefix) {
    if (!r || !prefix) return -1.0;
    for (const auto& name : r->rt.units())
        if (name.compare(0, std::strlen(prefix), prefix) == 0) return r->rt.unitMeanCycles(name);
    return -1.0;
}
int jst_runtime_batched(jst_runtime r) { return r && r->rt.batched() ? 1 : 0; }
int jst_runtime_branches(jst_runtime r) { return r ? (int)r->rt.branches() : 0; }
jst_result jst_runtime_reset_timing(jst_runtime r) {
    JST_ARG(r, "null runtime");
    r->rt.resetTiming();
    return R(Result::SUCCESS);
}

// ---- probes ------------------------------------------------------------------------------------
jst_result jst_fft_twiddles(uint64_t n, float* out) {
    JST_ARG(n > 0 && out, "invalid argument");
    modules::ComputeTwiddles(n, out);
    return R(Result::SUCCESS);
}
int jst_probe_fft_path(uint64_t n) {
    if (kernels::fft_lds_supported(n)) return JST_FFT_PATH_REGISTER;
    if (!kernels::fft_tiled_supported(n)) return JST_FFT_PATH_PASSES;
    return kernels::fft_tiled_needs_scratch(n) ? JST_FFT_PATH_TILE_PAIR : JST_FFT_PATH_TILE;
}
jst_result jst_probe_amplitude_range(const float* in, float* out_exact, float* out_fast, uint64_t count,
                                     float amplitude_coeff, float range_scale, float range_offset,
                                     float guard_h0, float guard_h1) {
    JST_ARG(in && out_exact && out_fast, "null argument");
    JST_HIP_CHECK(kernels::launch_amplitude_range_probe(out_exact, out_fast, reinterpret_cast<const float2*>(in),
                                                        count, amplitude_coeff, range_scale, range_offset,
                                                        guard_h0, guard_h1, nullptr),
                  "amplitude/range probe");
    JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
    return R(Result::SUCCESS);
}
jst_result jst_probe_tanhf(const float* in, float* out, uint64_t count) {
    JST_ARG(in && out, "null argument");
    JST_HIP_CHECK(kernels::launch_tanhf_probe(out, in, count, nullptr), "tanhf probe");
    JST_HIP_CHECK(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
    return R(Result::SUCCESS);
}

jst_result jst_probe_exact_sweep(int which, float amplitude_coeff, float range_scale, float range_offset,
                                 float height, uint64_t* mismatches, uint64_t* visited, uint32_t* first_bad) {
    JST_ARG(which >= 0 && which <= 6, "unknown sweep");
    JST_HIP_CHECK(kernels::launch_exact_sweep(which, amplitude_coeff, range_scale, range_offset, height,
                                              mismatches, visited, first_bad),
                  "exact sweep");
    return R(Result::SUCCESS);
}

}  // extern "C"
