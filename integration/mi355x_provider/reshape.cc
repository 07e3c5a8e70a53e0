// integration/mi355x_provider/reshape.cc -- REFERENCE-SIDE code: would live at src/domains/core/reshape/module_impl_native_cpu_mi355x.cc
// (INTEGRATION.md section 2).  The reference's own ReshapeImpl (validate / define / create) with computeSubmit() forwarded to
// libjetstream_hip.so through the host-staging bridge; registered under provider "mi355x".
#include <jetstream/runtime_context_native_cpu.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/registry.hh>

#include "module_impl.hh"
#include "mi355x_bridge.hh"

namespace Jetstream::Modules {

// a view: nothing to compute on either side (core/reshape/module_impl_native_cpu.cc:17-19)
struct ReshapeImplMi355x : public ReshapeImpl, public NativeCpuRuntimeContext, public Scheduler::Context {
    Result computeSubmit() override { return Result::SUCCESS; }
};

JST_REGISTER_MODULE(ReshapeImplMi355x, DeviceType::CPU, RuntimeType::NATIVE, "mi355x");

}  // namespace Jetstream::Modules
