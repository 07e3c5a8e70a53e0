// integration/device_hip/modules/window.cc -- REFERENCE-SIDE code: would live at src/domains/dsp/window/module_impl_native_hip.cc
// (INTEGRATION.md section 3).  The reference's own WindowImpl (validate / define / create: the output allocated ON THE DEVICE,
// attributes propagated) with the runtime hooks forwarded to the library's `window` module, in place on the device tensors. The taps are evaluated in F64 on the host and uploaded once (STATIC_OUTPUT: the scheduler settles the module after its first cycle, src/scheduler_synchronous.cc:534-546).
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct WindowImplNativeHip : public NativeHipModule<WindowImpl> {
    Result create() override {
        JST_CHECK(WindowImpl::create());
        return library.create("MODULE_WINDOW_NATIVE_HIP", "window", "generic", name(), {"size=" + std::to_string(size)}, {}, {out("window", output)});
    }
    Result destroy() override { return library.destroy(); }
};

JST_REGISTER_MODULE(WindowImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(WindowImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
