// integration/device_hip/modules/spectrogram.cc -- REFERENCE-SIDE code: would live at
// src/domains/visualization/spectrogram/module_impl_native_hip.cc.  SpectrogramImpl validates and allocates
// `frequencyBins` on the device; the library's spectrogram keeps ITS state in that very tensor (decay + saturating hits, bins
// bit-exact), so the present half reads it where it always did.  No output port: a SURFACE module.
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct SpectrogramImplNativeHip : public NativeHipModule<SpectrogramImpl> {
    Result create() override {
        JST_CHECK(SpectrogramImpl::create());
        JST_CHECK(library.create("MODULE_SPECTROGRAM_NATIVE_HIP", "spectrogram", "generic", name(), {"height=" + std::to_string(height)},
                                 {in("signal")}, {}));
        return library.bindState("frequencyBins", frequencyBins);
    }
    Result presentInitialize() override { return createPresent(); }
    Result presentSubmit() override { return present(); }
    Result destroy() override {
        (void)library.destroy();
        return SpectrogramImpl::destroy();
    }
};

JST_REGISTER_MODULE(SpectrogramImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(SpectrogramImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
