// integration/mi355x_provider/range.cc -- REFERENCE-SIDE code: would live at src/domains/core/range/module_impl_native_cpu_mi355x.cc
// (INTEGRATION.md section 2).  The reference's own RangeImpl (validate / define / create) with computeSubmit() forwarded to
// libjetstream_hip.so through the host-staging bridge; registered under provider "mi355x".
#include <jetstream/runtime_context_native_cpu.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/registry.hh>

#include "module_impl.hh"
#include "mi355x_bridge.hh"

namespace Jetstream::Modules {

struct RangeImplMi355x : public RangeImpl, public NativeCpuRuntimeContext, public Scheduler::Context {
    Result create() override {
        JST_CHECK(RangeImpl::create());
        return bridge.create("MODULE_RANGE_MI355X", "range", name(), {"min=" + Mi355x::Number(min), "max=" + Mi355x::Number(max)},
                             {{"signal", &input}}, "signal");
    }
    Result computeSubmit() override { return bridge.run(output); }
    Result destroy() override {
        (void)bridge.destroy();
        return RangeImpl::destroy();
    }
    Mi355x::Bridge bridge;
};

JST_REGISTER_MODULE(RangeImplMi355x, DeviceType::CPU, RuntimeType::NATIVE, "mi355x");

}  // namespace Jetstream::Modules
