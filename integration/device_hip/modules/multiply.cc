// integration/device_hip/modules/multiply.cc -- REFERENCE-SIDE code: would live at src/domains/core/multiply/module_impl_native_hip.cc
// (INTEGRATION.md section 3).  The reference's own MultiplyImpl (validate / define / create: the output allocated ON THE DEVICE,
// attributes propagated) with the runtime hooks forwarded to the library's `multiply` module, in place on the device tensors. The operands go to the library as the block wired them (b: the window reshaped to [1, .., N]): the library module broadcasts by shape as MultiplyImpl::create does (core/multiply/module_impl.cc:10-84), and a window that stays [1, .., N] is what lets the runtime fuse multiply + fft + amplitude + range into one kernel.
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct MultiplyImplNativeHip : public NativeHipModule<MultiplyImpl> {
    Result create() override {
        JST_CHECK(MultiplyImpl::create());
        return library.create("MODULE_MULTIPLY_NATIVE_HIP", "multiply", "generic", name(), {}, {in("a"), in("b")}, {out("product", c)});
    }
    Result destroy() override {
        (void)library.destroy();
        return MultiplyImpl::destroy();
    }
};

JST_REGISTER_MODULE(MultiplyImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(MultiplyImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
