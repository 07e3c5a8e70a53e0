// integration/mi355x_provider/fft.cc -- REFERENCE-SIDE code: would live at src/domains/dsp/fft/module_impl_native_cpu_mi355x.cc
// (INTEGRATION.md section 2).  The reference's own FftImpl (validate / define / create) with computeSubmit() forwarded to
// libjetstream_hip.so through the host-staging bridge; registered under provider "mi355x".
#include <jetstream/runtime_context_native_cpu.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/registry.hh>

#include "module_impl.hh"
#include "mi355x_bridge.hh"

namespace Jetstream::Modules {

struct FftImplMi355x : public FftImpl, public NativeCpuRuntimeContext, public Scheduler::Context {
    Result create() override {
        JST_CHECK(FftImpl::create());  // validates, allocates the CPU `output`, propagates the attributes
        return bridge.create("MODULE_FFT_MI355X", "fft", name(),
                             {forward ? "forward=true" : "forward=false", complexOutput ? "complexOutput=true" : "complexOutput=false"},
                             {{"signal", &input}}, "signal");
    }
    Result computeSubmit() override { return bridge.run(output); }
    Result destroy() override {
        (void)bridge.destroy();
        return FftImpl::destroy();
    }
    Mi355x::Bridge bridge;
};

JST_REGISTER_MODULE(FftImplMi355x, DeviceType::CPU, RuntimeType::NATIVE, "mi355x");

}  // namespace Jetstream::Modules
