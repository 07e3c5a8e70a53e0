// integration/device_hip/modules/amplitude.cc -- REFERENCE-SIDE code: would live at src/domains/dsp/amplitude/module_impl_native_hip.cc
// (INTEGRATION.md section 3).  The reference's own AmplitudeImpl (validate / define / create: the output allocated ON THE DEVICE,
// attributes propagated) with the runtime hooks forwarded to the library's `amplitude` module, in place on the device tensors.
// Provider "fast" (hardware transcendentals behind the exact power; Spectrogram bins stay bit-exact) is the library's own second
// registration of this module; every other module of a block built with provider "fast" is the generic one.
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct AmplitudeImplNativeHip : public NativeHipModule<AmplitudeImpl> {
    Result create() override {
        JST_CHECK(AmplitudeImpl::create());
        return library.create("MODULE_AMPLITUDE_NATIVE_HIP", "amplitude", provider(), name(), {}, {in("signal")}, {out("signal", output)});
    }
    Result destroy() override {
        (void)library.destroy();
        return AmplitudeImpl::destroy();
    }
};

JST_REGISTER_MODULE(AmplitudeImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(AmplitudeImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
