// integration/device_hip/modules/cast.cc -- REFERENCE-SIDE code: would live at src/domains/core/cast/module_impl_native_hip.cc
// (INTEGRATION.md section 3).  The reference's own CastImpl (validate / define / create: the output allocated ON THE DEVICE,
// attributes propagated) with the runtime hooks forwarded to the library's `cast` module, in place on the device tensors. CastImpl allocates `output` on the device -- or, for equal types, aliases the input (core/cast/module_impl.cc:96-104); the library's cast does the same, so the adopted storage is the producer's and nothing is launched.
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct CastImplNativeHip : public NativeHipModule<CastImpl> {
    Result create() override {
        JST_CHECK(CastImpl::create());
        return library.create("MODULE_CAST_NATIVE_HIP", "cast", "generic", name(), {"outputType=" + outputType}, {in("buffer")},
                              {out("buffer", bypass ? input : output)});   // equal types: the published tensor IS the input (:96-100)
    }
    Result destroy() override { return library.destroy(); }
};

JST_REGISTER_MODULE(CastImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(CastImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
