// integration/mi355x_provider/window.cc -- REFERENCE-SIDE code: would live at src/domains/dsp/window/module_impl_native_cpu_mi355x.cc
// (INTEGRATION.md section 2).  The reference's own WindowImpl (validate / define / create) with computeSubmit() forwarded to
// libjetstream_hip.so through the host-staging bridge; registered under provider "mi355x".
#include <jetstream/runtime_context_native_cpu.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/registry.hh>

#include "module_impl.hh"
#include "mi355x_bridge.hh"

namespace Jetstream::Modules {

struct WindowImplMi355x : public WindowImpl, public NativeCpuRuntimeContext, public Scheduler::Context {
    Result create() override {
        JST_CHECK(WindowImpl::create());
        return bridge.create("MODULE_WINDOW_MI355X", "window", name(), {"size=" + std::to_string(size)}, {}, "window");
    }
    Result computeSubmit() override { return bridge.run(output); }
    Result destroy() override { return bridge.destroy(); }
    Mi355x::Bridge bridge;
};

JST_REGISTER_MODULE(WindowImplMi355x, DeviceType::CPU, RuntimeType::NATIVE, "mi355x");

}  // namespace Jetstream::Modules
