// integration/device_hip/native_hip_module.hh -- REFERENCE-SIDE code: would live at include/jetstream/backend/devices/hip/
// native_module.hh.  The shape every src/domains/<domain>/<module>/module_impl_native_hip.cc has:
// the reference's own Impl (validate / define / create: shapes, attribute propagation, error strings, the output tensors
// allocated on the device) + the runtime hooks of (DeviceType::HIP, RuntimeType::NATIVE) forwarded to ONE library module
// that works in place on those device tensors (hip_library_module.hh).  A unit derives from NativeHipModule<XImpl>, and
// writes create() -- the library module's type, its "key=value" configuration and which tensors are its ports.
#pragma once

#include <jetstream/module_context.hh>
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

}  // namespace Jetstream::Modules
