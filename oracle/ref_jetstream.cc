// ref_jetstream.cc -- extern "C" harness around the REFERENCE's own libjetstream core, compiled IN PLACE from
// /root/reference by oracle/ref_jetstream_build.sh into oracle/_ref/libref_jetstream.so.  TEST INFRASTRUCTURE ONLY:
// tests/ and tools/make_reference_vectors.py drive it (ctypes) to pin oracle/jst_oracle.c and to generate golden
// vectors; the product never loads it.  Everything numeric that runs behind these entry points is the reference's
// code (Registry::BuildModule / Module::create / Runtime::compute for single modules -- what TestContext does,
// src/testing.cc:123-186 -- and Flowgraph::blockCreate / Flowgraph::compute for blocks, as tests/support/
// flowgraph_fixture.hh does).  What is mine: this glue, a non-static source module + block ("oracle_source": the
// counterpart of the reference's test-only flowgraph_test_source, with a dtype and a shape), and three symbols the
// core references from subsystems that are not built (python runtime, YAML parser).
#include <any>
#include <chrono>
#include <complex>
#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE
#include <hip/hip_runtime_api.h>
#include "hip_library_module.hh"  // integration/device_hip: the library tensors behind the reference's HIP modules
#endif

#include "jetstream/block.hh"
#include "jetstream/detail/block_impl.hh"
#include "jetstream/detail/module_impl.hh"
#include "jetstream/flowgraph.hh"
#include "jetstream/flowgraph_view.hh"
#include "jetstream/logger.hh"
#include "jetstream/memory/token.hh"
#include "jetstream/module.hh"
#include "jetstream/module_context.hh"
#include "jetstream/parser.hh"
#include "jetstream/registry.hh"
#include "jetstream/runtime.hh"
#include "jetstream/runtime_context_native_cpu.hh"
#include "jetstream/scheduler_context.hh"
// the visualization modules' state tensors are protected members: read as the reference's own tests read them
// (spectrogram/module_tests.cc:22-41, waterfall/module_tests.cc:24-47, lineplot/module_tests.cc:25-44)
#include "domains/visualization/lineplot/module_impl.hh"
#include "domains/visualization/spectrogram/module_impl.hh"
#include "domains/visualization/waterfall/module_impl.hh"

namespace Jetstream {

// ---- symbols of subsystems that are not part of this build ----------------------------------------------------
std::shared_ptr<Runtime::Impl> PythonRuntimeFactory() { return nullptr; }
Result Parser::YamlDecode(const std::string&, Map&) { return Result::ERROR; }
Result Parser::YamlEncode(const Map&, std::string&) { return Result::ERROR; }

// ---- oracle_source: a source whose tensor the test fills between computes ---------------------------------------
namespace Modules {
struct OracleSource : public Module::Config {
    std::vector<U64> shape = {1};
    std::string dataType = "CF32";
    JST_MODULE_TYPE(oracle_source);
    JST_MODULE_PARAMS(shape, dataType);
};
struct OracleSourceImpl : public Module::Impl,
                          public DynamicConfig<OracleSource>,
                          public NativeCpuRuntimeContext,
                          public Scheduler::Context {
    Result validate() override {
        const auto& c = *candidate();
        if (c.shape.empty() || NameToDataType(c.dataType) == DataType::None) return Result::ERROR;
        for (const U64 d : c.shape) if (d == 0) return Result::ERROR;
        return Result::SUCCESS;
    }
    Result define() override { return defineInterfaceOutput("signal"); }
    Result create() override {
        JST_CHECK(signal.create(device(), NameToDataType(dataType), shape));
        outputs()["signal"].produced(name(), "signal", signal);
        return Result::SUCCESS;
    }
    Result computeSubmit() override { return Result::SUCCESS; }
    Tensor signal;
};
JST_REGISTER_MODULE(OracleSourceImpl, DeviceType::CPU, RuntimeType::NATIVE, "generic");
}  // namespace Modules

namespace Blocks {
struct OracleSource : public Block::Config {
    std::vector<U64> shape = {1};
    std::string dataType = "CF32";
    JST_BLOCK_TYPE(oracle_source);
    JST_BLOCK_DOMAIN("Test");
    JST_BLOCK_PARAMS(shape, dataType);
    JST_BLOCK_DESCRIPTION("Oracle Source", "Test-owned tensor.", "Source block of the oracle harness.");
};
struct OracleSourceBlockImpl : public Block::Impl, public DynamicConfig<Blocks::OracleSource> {
    Result configure() override {
        moduleConfig->shape = shape;
        moduleConfig->dataType = dataType;
        return Result::SUCCESS;
    }
    Result define() override { return defineInterfaceOutput("signal", "Output", "Test-owned tensor."); }
    Result create() override {
        JST_CHECK(moduleCreate("source", moduleConfig, {}));
        return moduleExposeOutput("signal", {"source", "signal"});
    }
    std::shared_ptr<Modules::OracleSource> moduleConfig = std::make_shared<Modules::OracleSource>();
};
JST_REGISTER_BLOCK(OracleSourceBlockImpl, {"oracle_source"});
}  // namespace Blocks

}  // namespace Jetstream

using namespace Jetstream;

namespace {

// "key=value" lines -> Parser::Map of strings (the decoder turns them into the typed config fields,
// include/jetstream/parser.hh Decode: a std::string entry goes through StringToTyped)
Parser::Map parse_config(const char* lines) {
    Parser::Map m;
    if (!lines) return m;
    std::istringstream in(lines);
    std::string line;
    while (std::getline(in, line)) {
        const auto eq = line.find('=');
        if (eq == std::string::npos) continue;
        m[line.substr(0, eq)] = std::string(line.substr(eq + 1));
    }
    return m;
}

struct Desc {  // what the Python side reads back: element strides / offset like Tensor::stride() / offset()
    void* data;
    uint64_t offset;
    uint32_t dtype;
    uint32_t rank;
    uint64_t shape[8];
    uint64_t stride[8];
    int64_t sample_axis, batch_axis, channel_axis;
    uint64_t device;      // DeviceType value: anything but CPU is read / written through ref_dev_read / ref_dev_write
    uint64_t buffer_bytes;
};

int64_t axis_attr(const Tensor& t, const char* key) {
    if (!t.hasAttribute(key)) return -1;
    const std::any a = t.attribute(key);
    if (const auto* v = std::any_cast<Index>(&a)) return (int64_t)*v;
    return -1;
}

int fill_desc(const Tensor& t, Desc* d) {
    if (t.rank() > 8) return 1;
    d->data = const_cast<void*>(t.data());
    d->offset = t.offset();
    d->dtype = (uint32_t)t.dtype();
    d->rank = (uint32_t)t.rank();
    for (Index a = 0; a < t.rank(); ++a) {
        d->shape[a] = t.shape(a);
        d->stride[a] = t.stride(a);
    }
    d->sample_axis = axis_attr(t, "sampleAxis");
    d->batch_axis = axis_attr(t, "batchAxis");
    d->channel_axis = axis_attr(t, "channelAxis");
    d->device = (uint64_t)t.device();
    d->buffer_bytes = d->data ? t.buffer().sizeBytes() : 0;
    return 0;
}

// attribute kinds the path's modules read: 0 Index, 1 F32, 2 vector<F32>, 3 vector<U64>, 4 vector<F64>, 5 F64
int set_attr(Tensor& t, const char* key, int kind, const double* v, uint64_t n) {
    Result r = Result::ERROR;
    switch (kind) {
        case 0: r = t.setAttribute(key, Index{(Index)v[0]}); break;
        case 1: r = t.setAttribute(key, F32{(F32)v[0]}); break;
        case 2: { std::vector<F32> x(n); for (uint64_t i = 0; i < n; ++i) x[i] = (F32)v[i]; r = t.setAttribute(key, x); break; }
        case 3: { std::vector<U64> x(n); for (uint64_t i = 0; i < n; ++i) x[i] = (U64)v[i]; r = t.setAttribute(key, x); break; }
        case 4: { std::vector<F64> x(v, v + n); r = t.setAttribute(key, x); break; }
        case 5: r = t.setAttribute(key, F64{v[0]}); break;
        default: break;
    }
    return r == Result::SUCCESS ? 0 : 1;
}

// ---- one module + one runtime, alive across computes (TestContext::start / compute / stop) --------------------
struct ModSession {
    std::string provider = "generic";  // the registry's 4th key (src/registry.cc:605-618): "mi355x" selects integration/mi355x_provider
    DeviceType device = DeviceType::CPU;  // the registry's 2nd key: "hip" in the libref_jetstream_devhip.so build (integration/device_hip)
    std::string type;
    Parser::Map config;
    std::unordered_map<std::string, Tensor> inputs;
    std::shared_ptr<Module> module;
    std::unique_ptr<Runtime> runtime;
    ~ModSession() {
        if (runtime) { (void)runtime->destroy(); runtime.reset(); }
        if (module) { (void)module->destroy(); module.reset(); }
    }
};

struct SpectrogramAccess : Modules::SpectrogramImpl {
    static auto bins() { return &SpectrogramAccess::frequencyBins; }
};
struct WaterfallAccess : Modules::WaterfallImpl {
    static auto bins() { return &WaterfallAccess::frequencyBins; }
    static auto ring() { return &WaterfallAccess::ringState; }
};
struct LineplotAccess : Modules::LineplotImpl {
    static auto points() { return &LineplotAccess::signalPoints; }
};

struct FgSession {
    std::unique_ptr<Flowgraph> fg;
    ~FgSession() {
        if (!fg) return;
        std::vector<std::string> names;
        if (fg->view().keys(names) == Result::SUCCESS)
            for (const auto& n : names) (void)fg->blockDestroy(n, false);
        (void)fg->destroy();
    }
};

}  // namespace

extern "C" {

int ref_jst_probe() { return 1; }
uint64_t ref_jst_desc_size() { return sizeof(Desc); }

// quiet by default: the reference logs every created module at INFO
void ref_jst_log_level(int level) { JST_LOG_SET_DEBUG_LEVEL(level); }

// ---- modules ---------------------------------------------------------------------------------------------------
void* ref_mod_new(const char* type, const char* config_lines) {
    auto* s = new ModSession();
    s->type = type;
    s->config = parse_config(config_lines);
    return s;
}
// allocates the input tensor (CPU, zero-filled) and returns its description; the caller writes the data in place
int ref_mod_input(void* h, const char* port, const char* dtype, uint32_t rank, const uint64_t* shape, Desc* out) {
    auto* s = static_cast<ModSession*>(h);
    Shape sh(shape, shape + rank);
    Tensor t;
    if (t.create(s->device, NameToDataType(dtype), sh) != Result::SUCCESS) return 1;
    s->inputs[port] = t;
    return fill_desc(t, out);
}
int ref_mod_input_attr(void* h, const char* port, const char* key, int kind, const double* v, uint64_t n) {
    auto* s = static_cast<ModSession*>(h);
    const auto it = s->inputs.find(port);
    if (it == s->inputs.end()) return 1;
    return set_attr(it->second, key, kind, v, n);
}
// layout edits of an input before start(): op 0 permute(axes), 1 reshape(shape), 2 expandDims(axis), 3 broadcastTo(shape),
// 4 slice(tokens): four numbers per axis {kind, a, b, c}: kind 0 = ':' (all), 1 = index a (the axis goes), 2 = a:b:c
int ref_mod_input_view(void* h, const char* port, int op, const uint64_t* v, uint64_t n, Desc* out) {
    auto* s = static_cast<ModSession*>(h);
    const auto it = s->inputs.find(port);
    if (it == s->inputs.end()) return 1;
    Result r = Result::ERROR;
    const Shape sh(v, v + n);
    if (op == 0) r = it->second.permute(sh);
    else if (op == 1) r = it->second.reshape(sh);
    else if (op == 2) r = it->second.expandDims((Index)v[0]);
    else if (op == 3) r = it->second.broadcastTo(sh);
    else if (op == 4 && n % 4 == 0) {
        std::vector<Token> tokens;
        for (uint64_t a = 0; a < n; a += 4) {
            if (v[a] == 0) tokens.emplace_back();
            else if (v[a] == 1) tokens.emplace_back((U64)v[a + 1]);
            else tokens.emplace_back((U64)v[a + 1], (U64)v[a + 2], (U64)v[a + 3], true);
        }
        r = it->second.slice(tokens);
    }
    if (r != Result::SUCCESS) return 1;
    return fill_desc(it->second, out);
}
// Registry::BuildModule + Module::create + Runtime::create; returns the reference's Result value
int ref_mod_start(void* h) {
    auto* s = static_cast<ModSession*>(h);
    if (s->module || s->runtime) return (int)Result::ERROR;
    Result r = Registry::BuildModule(s->type, s->device, RuntimeType::NATIVE, s->provider, s->module);
    if (r != Result::SUCCESS) return (int)r;
    TensorMap in;
    for (auto& [name, t] : s->inputs) {
        in[name].requested("test", name);
        in[name].tensor = t;
    }
    r = s->module->create("test", s->config, in);
    if (r != Result::SUCCESS) { s->module.reset(); return (int)r; }
    s->runtime = std::make_unique<Runtime>("test", s->device, RuntimeType::NATIVE);
    r = s->runtime->create({{"test", s->module}});
    if (r != Result::SUCCESS) { (void)s->module->destroy(); s->module.reset(); s->runtime.reset(); }
    return (int)r;
}
int ref_mod_set_provider(void* h, const char* provider) {
    static_cast<ModSession*>(h)->provider = provider;
    return 0;
}
// "cpu" | "hip" (the latter only where the core was built with core_hip_device.patch); before the inputs are allocated
int ref_mod_set_device(void* h, const char* device) {
    if (!IsDeviceName(device)) return 1;
    static_cast<ModSession*>(h)->device = StringToDevice(device);
    return 0;
}
int ref_device_known(const char* device) { return IsDeviceName(device) ? 1 : 0; }
// device tensors: the bytes [0, bytes) of the buffer at `device_ptr`, through the HIP runtime's copy engine (synchronous)
int ref_dev_read(const void* device_ptr, void* host, uint64_t bytes) {
#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE
    return hipMemcpy(host, device_ptr, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
#else
    (void)device_ptr; (void)host; (void)bytes;
    return 1;
#endif
}
int ref_dev_write(void* device_ptr, const void* host, uint64_t bytes) {
#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE
    return hipMemcpy(device_ptr, host, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
#else
    (void)device_ptr; (void)host; (void)bytes;
    return 1;
#endif
}
// The library tensor standing behind output `port` (or "state:<key>") of the reference's HIP module `module` (flowgraph
// modules are named "<block>-<module>", src/block_impl.cc:64): where it lives NOW -- for a ring, the selected slot.
int ref_hip_directory(const char* module, const char* port, Desc* out) {
#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE
    jst_tensor t = Hip::TensorDirectory::Get().find(module, port);
    jst_tensor_desc d{};
    if (!t || jst_tensor_describe(t, &d) != JST_SUCCESS || d.rank > 8) return 1;
    static const DataType kinds[] = {DataType::None, DataType::F32, DataType::CF32, DataType::F64, DataType::U64, DataType::I8, DataType::CI8,
                                     DataType::I16, DataType::CI16, DataType::U8, DataType::CU8, DataType::U16, DataType::CU16, DataType::I32,
                                     DataType::CI32, DataType::U32, DataType::CU32, DataType::CF64};
    std::memset(out, 0, sizeof(*out));
    out->data = d.data;
    out->offset = d.offset;
    out->dtype = d.dtype < sizeof(kinds) / sizeof(kinds[0]) ? (uint32_t)kinds[d.dtype] : 0;
    out->rank = d.rank;
    uint64_t last = d.offset;
    for (uint32_t a = 0; a < d.rank; ++a) {
        out->shape[a] = d.shape[a];
        out->stride[a] = d.stride[a];
        if (d.shape[a]) last += (d.shape[a] - 1) * d.stride[a];
    }
    out->sample_axis = d.sample_axis;
    out->batch_axis = d.batch_axis;
    out->channel_axis = d.channel_axis;
    out->device = (uint64_t)DeviceType::HIP;
    out->buffer_bytes = (last + 1) * DataTypeSize(out->dtype ? (DataType)out->dtype : DataType::F32);
    return 0;
#else
    (void)module; (void)port; (void)out;
    return 1;
#endif
}
// 2 = device memory (HBM), 1 = host memory registered with / allocated by the HIP runtime, 0 = unknown to it, -1 = not a HIP build
int ref_dev_pointer_kind(const void* ptr) {
#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return a.type == hipMemoryTypeDevice ? 2 : (a.type == hipMemoryTypeHost ? 1 : 0);
#else
    (void)ptr;
    return -1;
#endif
}
// 1 when the registry holds (type, CPU, NATIVE, provider)
int ref_registry_has(const char* type, const char* provider) {
    std::shared_ptr<Module> probe;
    return Registry::BuildModule(type, DeviceType::CPU, RuntimeType::NATIVE, provider, probe) == Result::SUCCESS ? 1 : 0;
}
int ref_registry_has_on(const char* type, const char* provider, const char* device) {
    if (!IsDeviceName(device)) return 0;
    std::shared_ptr<Module> probe;
    return Registry::BuildModule(type, StringToDevice(device), RuntimeType::NATIVE, provider, probe) == Result::SUCCESS ? 1 : 0;
}
int ref_mod_compute(void* h) {
    auto* s = static_cast<ModSession*>(h);
    if (!s->module || !s->runtime) return (int)Result::ERROR;
    std::unordered_set<std::string> skipped, failed;
    return (int)s->runtime->compute({}, skipped, failed);
}
int ref_mod_output(void* h, const char* port, Desc* out) {
    auto* s = static_cast<ModSession*>(h);
    if (!s->module) return 1;
    const auto& outs = s->module->outputs();
    const auto it = outs.find(port);
    if (it == outs.end()) return 1;
    return fill_desc(it->second.tensor, out);
}
// reads an output attribute back: kind as in set_attr; returns the element count written (0 = absent / other type)
uint64_t ref_mod_output_attr(void* h, const char* port, const char* key, int kind, double* v, uint64_t cap) {
    auto* s = static_cast<ModSession*>(h);
    if (!s->module) return 0;
    const auto& outs = s->module->outputs();
    const auto it = outs.find(port);
    if (it == outs.end() || !it->second.tensor.hasAttribute(key)) return 0;
    const std::any a = it->second.tensor.attribute(key);
    if (kind == 0) { if (const auto* p = std::any_cast<Index>(&a)) { v[0] = (double)*p; return 1; } }
    if (kind == 1) { if (const auto* p = std::any_cast<F32>(&a)) { v[0] = (double)*p; return 1; } }
    if (kind == 2) { if (const auto* p = std::any_cast<std::vector<F32>>(&a)) { uint64_t n = p->size() < cap ? p->size() : cap; for (uint64_t i = 0; i < n; ++i) v[i] = (*p)[i]; return n; } }
    if (kind == 3) { if (const auto* p = std::any_cast<std::vector<U64>>(&a)) { uint64_t n = p->size() < cap ? p->size() : cap; for (uint64_t i = 0; i < n; ++i) v[i] = (double)(*p)[i]; return n; } }
    return 0;
}
// state of a visualization module: "frequencyBins" (spectrogram F32[width, height] as created -- row-major words
// bins[x + idx * width] --, waterfall F32[height, width]) or "signalPoints" (lineplot F32[width, 2]); waterfall also
// "writeIndex" through ref_mod_state_scalar
int ref_mod_state(void* h, const char* name, Desc* out) {
    auto* s = static_cast<ModSession*>(h);
    if (!s->module) return 1;
    const std::string n(name);
    if (s->type == "spectrogram" && n == "frequencyBins") {
        const auto* impl = s->module->getImpl<Modules::SpectrogramImpl>();
        return impl ? fill_desc(impl->*SpectrogramAccess::bins(), out) : 1;
    }
    if (s->type == "waterfall" && n == "frequencyBins") {
        const auto* impl = s->module->getImpl<Modules::WaterfallImpl>();
        return impl ? fill_desc(impl->*WaterfallAccess::bins(), out) : 1;
    }
    if (s->type == "lineplot" && n == "signalPoints") {
        const auto* impl = s->module->getImpl<Modules::LineplotImpl>();
        return impl ? fill_desc(impl->*LineplotAccess::points(), out) : 1;
    }
    return 1;
}
int64_t ref_mod_state_scalar(void* h, const char* name) {
    auto* s = static_cast<ModSession*>(h);
    if (!s->module) return -1;
    if (s->type == "waterfall" && std::string(name) == "writeIndex") {
        const auto* impl = s->module->getImpl<Modules::WaterfallImpl>();
        return impl ? (int64_t)(impl->*WaterfallAccess::ring()).writeIndex : -1;
    }
    return -1;
}
void ref_mod_free(void* h) { delete static_cast<ModSession*>(h); }

// ---- flowgraphs of blocks --------------------------------------------------------------------------------------
void* ref_fg_new() {
    auto* s = new FgSession();
    s->fg = std::make_unique<Flowgraph>();
    if (s->fg->create({}, nullptr, nullptr, nullptr) != Result::SUCCESS) { s->fg.reset(); delete s; return nullptr; }
    return s;
}
// inputs_lines: "port=block:port" per line
int ref_fg_block(void* h, const char* name, const char* type, const char* config_lines, const char* inputs_lines) {
    auto* s = static_cast<FgSession*>(h);
    TensorMap in;
    if (inputs_lines) {
        std::istringstream is(inputs_lines);
        std::string line;
        while (std::getline(is, line)) {
            const auto eq = line.find('=');
            const auto colon = line.find(':', eq == std::string::npos ? 0 : eq);
            if (eq == std::string::npos || colon == std::string::npos) continue;
            in[line.substr(0, eq)].requested(line.substr(eq + 1, colon - eq - 1), line.substr(colon + 1));
        }
    }
    return (int)s->fg->blockCreate(name, std::string(type), parse_config(config_lines), in);
}
// the same with the registry's provider key: every module the block creates is built with it (src/block_impl.cc:46-53)
int ref_fg_block_provider(void* h, const char* name, const char* type, const char* config_lines, const char* inputs_lines,
                          const char* provider) {
    auto* s = static_cast<FgSession*>(h);
    TensorMap in;
    if (inputs_lines) {
        std::istringstream is(inputs_lines);
        std::string line;
        while (std::getline(is, line)) {
            const auto eq = line.find('=');
            const auto colon = line.find(':', eq == std::string::npos ? 0 : eq);
            if (eq == std::string::npos || colon == std::string::npos) continue;
            in[line.substr(0, eq)].requested(line.substr(eq + 1, colon - eq - 1), line.substr(colon + 1));
        }
    }
    return (int)s->fg->blockCreate(name, std::string(type), parse_config(config_lines), in, DeviceType::CPU, RuntimeType::NATIVE,
                                   std::string(provider));
}
// ... and with the registry's device key ("cpu" | "hip"): every module of the block is built for that device, so the
// scheduler puts them into a runtime segment of that device (src/scheduler_synchronous.cc:698-757)
int ref_fg_block_on(void* h, const char* name, const char* type, const char* config_lines, const char* inputs_lines,
                    const char* provider, const char* device) {
    auto* s = static_cast<FgSession*>(h);
    if (!IsDeviceName(device)) return (int)Result::ERROR;
    TensorMap in;
    if (inputs_lines) {
        std::istringstream is(inputs_lines);
        std::string line;
        while (std::getline(is, line)) {
            const auto eq = line.find('=');
            const auto colon = line.find(':', eq == std::string::npos ? 0 : eq);
            if (eq == std::string::npos || colon == std::string::npos) continue;
            in[line.substr(0, eq)].requested(line.substr(eq + 1, colon - eq - 1), line.substr(colon + 1));
        }
    }
    return (int)s->fg->blockCreate(name, std::string(type), parse_config(config_lines), in, StringToDevice(device), RuntimeType::NATIVE,
                                   std::string(provider));
}
// state of a block: Block::State value, or -1 when the block does not exist
int ref_fg_block_state(void* h, const char* name) {
    auto* s = static_cast<FgSession*>(h);
    Flowgraph::View::BlockInfo info;
    if (s->fg->view().info(name, info) != Result::SUCCESS) return -1;
    return (int)info.state;
}
int ref_fg_tensor(void* h, const char* block, const char* port, Desc* out) {
    auto* s = static_cast<FgSession*>(h);
    TensorMap outs;
    if (s->fg->view().outputs(block, outs) != Result::SUCCESS) return 1;
    const auto it = outs.find(port);
    if (it == outs.end()) return 1;
    return fill_desc(it->second.tensor, out);
}
int ref_fg_tensor_attr(void* h, const char* block, const char* port, const char* key, int kind, const double* v, uint64_t n) {
    auto* s = static_cast<FgSession*>(h);
    TensorMap outs;
    if (s->fg->view().outputs(block, outs) != Result::SUCCESS) return 1;
    const auto it = outs.find(port);
    if (it == outs.end()) return 1;
    Tensor t = it->second.tensor;  // copies share the attribute store (the reference's own tests set them this way)
    return set_attr(t, key, kind, v, n);
}
// op as in ref_mod_input_view, applied to a block's output tensor (e.g. reshape a source to [heads, taps])
int ref_fg_tensor_view(void* h, const char* block, const char* port, int op, const uint64_t* v, uint64_t n) {
    auto* s = static_cast<FgSession*>(h);
    TensorMap outs;
    if (s->fg->view().outputs(block, outs) != Result::SUCCESS) return 1;
    const auto it = outs.find(port);
    if (it == outs.end()) return 1;
    Tensor t = it->second.tensor;
    const Shape sh(v, v + n);
    Result r = Result::ERROR;
    if (op == 0) r = t.permute(sh);
    else if (op == 1) r = t.reshape(sh);
    else if (op == 2) r = t.expandDims((Index)v[0]);
    return r == Result::SUCCESS ? 0 : 1;
}
uint64_t ref_fg_tensor_attr_get(void* h, const char* block, const char* port, const char* key, int kind, double* v, uint64_t cap) {
    auto* s = static_cast<FgSession*>(h);
    TensorMap outs;
    if (s->fg->view().outputs(block, outs) != Result::SUCCESS) return 0;
    const auto it = outs.find(port);
    if (it == outs.end() || !it->second.tensor.hasAttribute(key)) return 0;
    const std::any a = it->second.tensor.attribute(key);
    if (kind == 0) { if (const auto* p = std::any_cast<Index>(&a)) { v[0] = (double)*p; return 1; } }
    if (kind == 1) { if (const auto* p = std::any_cast<F32>(&a)) { v[0] = (double)*p; return 1; } }
    if (kind == 2) { if (const auto* p = std::any_cast<std::vector<F32>>(&a)) { uint64_t n = p->size() < cap ? p->size() : cap; for (uint64_t i = 0; i < n; ++i) v[i] = (*p)[i]; return n; } }
    if (kind == 3) { if (const auto* p = std::any_cast<std::vector<U64>>(&a)) { uint64_t n = p->size() < cap ? p->size() : cap; for (uint64_t i = 0; i < n; ++i) v[i] = (double)(*p)[i]; return n; } }
    return 0;
}
int ref_fg_compute(void* h) { return (int)static_cast<FgSession*>(h)->fg->compute(); }
// nanobench-style timing of Flowgraph::compute() (src/benchmark.cc:100-106,175-186: warm-up, epochs of a minimum duration,
// the caller takes the median): rates[e] = computes per second of epoch e; returns the Result of the last compute
int ref_fg_compute_timed(void* h, double epoch_s, uint32_t epochs, double* rates) {
    auto* s = static_cast<FgSession*>(h);
    Result r = s->fg->compute();
    if (r != Result::SUCCESS) return (int)r;
    for (uint32_t e = 0; e < epochs; ++e) {
        const auto t0 = std::chrono::steady_clock::now();
        uint64_t n = 0;
        double dt = 0.0;
        do {
            r = s->fg->compute();
            ++n;
            dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } while (r == Result::SUCCESS && dt < epoch_s);
        if (r != Result::SUCCESS) return (int)r;
        rates[e] = (double)n / dt;
    }
    return (int)r;
}
// n x Flowgraph::compute() in a C loop (the caller times it); returns the Result of the last one
int ref_fg_compute_n(void* h, uint64_t n) {
    auto* s = static_cast<FgSession*>(h);
    Result r = Result::SUCCESS;
    for (uint64_t i = 0; i < n && r == Result::SUCCESS; ++i) r = s->fg->compute();
    return (int)r;
}
void ref_fg_free(void* h) { delete static_cast<FgSession*>(h); }

}  // extern "C"
