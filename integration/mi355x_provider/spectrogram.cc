// integration/mi355x_provider/spectrogram.cc -- REFERENCE-SIDE code: would live at src/domains/visualization/spectrogram/module_impl_native_cpu_mi355x.cc
// (INTEGRATION.md section 2).  The reference's own SpectrogramImpl (validate / define / create) with computeSubmit() forwarded to
// libjetstream_hip.so through the host-staging bridge; registered under provider "mi355x".
#include <jetstream/runtime_context_native_cpu.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/registry.hh>

#include "module_impl.hh"
#include "mi355x_bridge.hh"

namespace Jetstream::Modules {

// a module with STATE: the bins live in HBM between cycles (decay + saturating hits on the device) and come back when a frame --
// or a test reading `frequencyBins` -- needs them
struct SpectrogramImplMi355x : public SpectrogramImpl, public NativeCpuRuntimeContext, public Scheduler::Context {
    Result create() override {
        JST_CHECK(SpectrogramImpl::create());  // validates, allocates the CPU `frequencyBins`
        JST_CHECK(bridge.create("MODULE_SPECTROGRAM_MI355X", "spectrogram", name(), {"height=" + std::to_string(height)},
                                {{"signal", &input}}, nullptr));
        if (jst_module_state(bridge.handle(), "frequencyBins", &devBins) != JST_SUCCESS) {
            JST_ERROR("[MODULE_SPECTROGRAM_MI355X] {}", jst_last_error());
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    Result computeSubmit() override {
        JST_CHECK(bridge.upload());
        JST_CHECK(bridge.compute(true));
        return bridge.download(devBins, frequencyBins);  // kept current for the present half and for the tests
    }
    Result presentInitialize() override { return createPresent(); }
    Result presentSubmit() override { return present(); }
    Result destroy() override {
        if (devBins) (void)jst_tensor_destroy(devBins);
        devBins = {};
        (void)bridge.destroy();
        return SpectrogramImpl::destroy();
    }
    Mi355x::Bridge bridge;
    jst_tensor devBins{};
};

JST_REGISTER_MODULE(SpectrogramImplMi355x, DeviceType::CPU, RuntimeType::NATIVE, "mi355x");

}  // namespace Jetstream::Modules
