// integration/device_hip/modules/range.cc -- REFERENCE-SIDE code: would live at src/domains/core/range/module_impl_native_hip.cc
// (INTEGRATION.md section 3).  The reference's own RangeImpl (validate / define / create: the output allocated ON THE DEVICE,
// attributes propagated) with the runtime hooks forwarded to the library's `range` module, in place on the device tensors.
// Provider "fast" (hardware transcendentals behind the exact power; Spectrogram bins stay bit-exact) is the library's own second
// registration of this module; every other module of a block built with provider "fast" is the generic one.
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct RangeImplNativeHip : public NativeHipModule<RangeImpl> {
    Result create() override {
        JST_CHECK(RangeImpl::create());
        return library.create("MODULE_RANGE_NATIVE_HIP", "range", provider(), name(), {"min=" + Hip::Number(min), "max=" + Hip::Number(max)}, {in("signal")}, {out("signal", output)});
    }
    Result destroy() override {
        (void)library.destroy();
        return RangeImpl::destroy();
    }
};

JST_REGISTER_MODULE(RangeImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(RangeImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
