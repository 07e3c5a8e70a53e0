// integration/device_hip/modules/reshape.cc -- REFERENCE-SIDE code: would live at src/domains/core/reshape/module_impl_native_hip.cc
// (INTEGRATION.md section 3).  The reference's own ReshapeImpl (validate / define / create: the output allocated ON THE DEVICE,
// attributes propagated) with the runtime hooks forwarded to the library's `reshape` module, in place on the device tensors. A view: the library's reshape makes the same view of the same (already adopted) storage; nothing is launched.
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct ReshapeImplNativeHip : public NativeHipModule<ReshapeImpl> {
    Result create() override {
        JST_CHECK(ReshapeImpl::create());
        return library.create("MODULE_RESHAPE_NATIVE_HIP", "reshape", "generic", name(), {"shape=" + shape}, {in("buffer")}, {out("buffer", output)});
    }
    Result destroy() override { return library.destroy(); }
};

JST_REGISTER_MODULE(ReshapeImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(ReshapeImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
