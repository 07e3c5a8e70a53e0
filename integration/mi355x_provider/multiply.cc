// integration/mi355x_provider/multiply.cc -- REFERENCE-SIDE code: would live at src/domains/core/multiply/module_impl_native_cpu_mi355x.cc
// (INTEGRATION.md section 2).  The reference's own MultiplyImpl (validate / define / create) with computeSubmit() forwarded to
// libjetstream_hip.so through the host-staging bridge; registered under provider "mi355x".
#include <jetstream/runtime_context_native_cpu.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/registry.hh>

#include "module_impl.hh"
#include "mi355x_bridge.hh"

namespace Jetstream::Modules {

struct MultiplyImplMi355x : public MultiplyImpl, public NativeCpuRuntimeContext, public Scheduler::Context {
    Result create() override {
        JST_CHECK(MultiplyImpl::create());  // the broadcast rules and the output shape are the reference's
        return bridge.create("MODULE_MULTIPLY_MI355X", "multiply", name(), {}, {{"a", &a}, {"b", &b}}, "product");
    }
    Result computeSubmit() override { return bridge.run(c); }
    Result destroy() override {
        (void)bridge.destroy();
        return MultiplyImpl::destroy();
    }
    Mi355x::Bridge bridge;
};

JST_REGISTER_MODULE(MultiplyImplMi355x, DeviceType::CPU, RuntimeType::NATIVE, "mi355x");

}  // namespace Jetstream::Modules
