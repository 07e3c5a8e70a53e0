// integration/device_hip/native_hip_module.hh -- REFERENCE-SIDE code: would live at include/jetstream/backend/devices/hip/
// native_module.hh.  The shape every src/domains/<domain>/<module>/module_impl_native_hip.cc has:
// the reference's own Impl (validate / define / create: shapes, attribute propagation, error strings, the output tensors
// allocated on the device) + the runtime hooks of (DeviceType::HIP, RuntimeType::NATIVE) forwarded to ONE library module
// that works in place on those device tensors (hip_library_module.hh).  A unit derives from NativeHipModule<XImpl>, and
// writes create() -- the library module's type, its "key=value" configuration and which tensors are its ports.
#pragma once

#include <jetstream/module_context.hh>
#include <jetstream/parser.hh>
#include <jetstream/registry.hh>
#include <jetstream/runtime_context_native_hip.hh>
#include <jetstream/scheduler_context.hh>

#include "hip_library_module.hh"

namespace Jetstream::Modules {

template <class Impl>
struct NativeHipModule : public Impl, public NativeHipRuntimeContext, public Scheduler::Context {
    Result computeInitialize() override { return library.computeInitialize(); }
    Result computeSubmit(void* hipStream) override { return library.computeSubmit(hipStream); }
    Result computeDeinitialize() override { return library.computeDeinitialize(); }
    void* libraryModule() const override { return library.handle(); }
    Result publishOutputs(void* hipStream) override { return library.publishLatest(hipStream); }

 protected:
    // the link the block wired to `port` (the tensor as the PRODUCER published it: a broadcastTo / slice a base Impl applies
    // to its private copy is the library module's own business, as it is the CPU module's)
    Hip::LibraryModule::Input in(const char* libraryPort, const char* port) { return {libraryPort, &this->inputs().at(port)}; }
    Hip::LibraryModule::Input in(const char* port) { return in(port, port); }
    static Hip::LibraryModule::Output out(const char* libraryPort, const char* port, Tensor& tensor) { return {libraryPort, port, &tensor}; }
    static Hip::LibraryModule::Output out(const char* port, Tensor& tensor) { return {port, port, &tensor}; }

    Hip::LibraryModule library;
};

// The whole unit for a module whose library counterpart has the same type name, the same port names and the same
// configuration keys (every module of the path but the ones with state the reference reads back or an alias to publish):
//   * configuration: the staged Config serialised by the reference's own JST_MODULE_PARAMS (Parser::Map) and written out by
//     Parser::TypedToString ("{}": the shortest text that round-trips the value, so F32 / F64 parameters arrive bit for bit);
//   * inputs: every link the block wired; outputs: every tensor Impl::create() published.
// Provider "fast" exists in the library for amplitude and range only; every other module of a block built with it is generic.
template <class Impl>
struct LibraryBackedModule : public NativeHipModule<Impl> {
    Result create() override {
        JST_CHECK(Impl::create());
        Parser::Map fields;
        JST_CHECK(this->serialize(fields));
        std::vector<std::string> config;
        for (const auto& [key, value] : fields) {
            std::string text;
            JST_CHECK(Parser::TypedToString(value, text));
            config.push_back(key + "=" + text);
        }
        std::vector<Hip::LibraryModule::Input> ins;
        for (const auto& [port, link] : this->inputs()) ins.push_back({port, &link});
        std::vector<Hip::LibraryModule::Output> outs;
        for (auto& [port, link] : this->outputs()) outs.push_back({port, port, &link.tensor});
        const std::string type = this->type();
        const std::string tag = "MODULE_" + type + "_NATIVE_HIP";
        const bool hasFast = type == "amplitude" || type == "range";
        JST_CHECK(this->library.create(tag, type.c_str(), hasFast ? this->provider() : "generic", this->name(), config, ins, outs));
        return bindStates();
    }
    Result destroy() override {
        (void)this->library.destroy();
        return Impl::destroy();
    }

 protected:
    virtual Result bindStates() { return Result::SUCCESS; }
};

#define JST_REGISTER_HIP_LIBRARY_MODULE(impl_type)                                     \
    JST_REGISTER_MODULE(impl_type, DeviceType::HIP, RuntimeType::NATIVE, "generic");   \
    JST_REGISTER_MODULE(impl_type, DeviceType::HIP, RuntimeType::NATIVE, "fast")

}  // namespace Jetstream::Modules
