// integration/device_hip/modules/invert.cc -- REFERENCE-SIDE code: would live at src/domains/dsp/invert/module_impl_native_hip.cc
// (INTEGRATION.md section 3).  The reference's own InvertImpl (validate / define / create: the output allocated ON THE DEVICE,
// attributes propagated) with the runtime hooks forwarded to the library's `invert` module, in place on the device tensors.
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct InvertImplNativeHip : public NativeHipModule<InvertImpl> {
    Result create() override {
        JST_CHECK(InvertImpl::create());
        return library.create("MODULE_INVERT_NATIVE_HIP", "invert", "generic", name(), {}, {in("signal")}, {out("signal", output)});
    }
    Result destroy() override {
        (void)library.destroy();
        return InvertImpl::destroy();
    }
};

JST_REGISTER_MODULE(InvertImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(InvertImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
