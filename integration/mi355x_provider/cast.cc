// integration/mi355x_provider/cast.cc -- REFERENCE-SIDE code: would live at src/domains/core/cast/module_impl_native_cpu_mi355x.cc
// (INTEGRATION.md section 2).  The reference's own CastImpl (validate / define / create) with computeSubmit() forwarded to
// libjetstream_hip.so through the host-staging bridge; registered under provider "mi355x".
#include <jetstream/runtime_context_native_cpu.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/registry.hh>

#include "module_impl.hh"
#include "mi355x_bridge.hh"

namespace Jetstream::Modules {

struct CastImplMi355x : public CastImpl, public NativeCpuRuntimeContext, public Scheduler::Context {
    Result create() override {
        JST_CHECK(CastImpl::create());
        if (bypass) return Result::SUCCESS;  // same type in and out: the output IS the input (core/cast/module_impl.cc:96-104)
        return bridge.create("MODULE_CAST_MI355X", "cast", name(), {"outputType=" + outputType}, {{"buffer", &input}}, "buffer");
    }
    Result computeSubmit() override { return bypass ? Result::SUCCESS : bridge.run(output); }
    Result destroy() override { return bridge.destroy(); }
    Mi355x::Bridge bridge;
};

JST_REGISTER_MODULE(CastImplMi355x, DeviceType::CPU, RuntimeType::NATIVE, "mi355x");

}  // namespace Jetstream::Modules
