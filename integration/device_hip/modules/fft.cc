// integration/device_hip/modules/fft.cc -- REFERENCE-SIDE code: would live at src/domains/dsp/fft/module_impl_native_hip.cc
// (the CUDA peer is src/domains/dsp/fft/module_impl_native_cuda.cc).  FftImpl validates, allocates
// `output` ON THE DEVICE and propagates the attributes; the library's fft module transforms in place on those buffers.
#include "module_impl.hh"
#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct FftImplNativeHip : public NativeHipModule<FftImpl> {
    Result create() override {
        JST_CHECK(FftImpl::create());
        return library.create("MODULE_FFT_NATIVE_HIP", "fft", "generic", name(),
                              {"forward=" + Hip::Flag(forward), "complexOutput=" + Hip::Flag(complexOutput)},
                              {in("signal")}, {out("signal", output)});
    }
    Result destroy() override {
        (void)library.destroy();
        return FftImpl::destroy();
    }
};

JST_REGISTER_MODULE(FftImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(FftImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules
