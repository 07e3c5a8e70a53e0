// integration/device_hip/fft_module_impl_native_hip.cc -- REFERENCE-SIDE code: would live at
// src/domains/dsp/fft/module_impl_native_hip.cc: the FFT module of (DeviceType::HIP, NATIVE).
// The reference's own FftImpl validates, allocates `output` ON THE DEVICE (Tensor::create(device(), ..) through
// buffer_hip.cc) and propagates the attributes; the library's fft module works directly on those two device buffers
// (jst_tensor_wrap / jst_tensor_rebind: borrowed pointers, no copy) and computeSubmit only enqueues on the segment's stream -- a chain of
// such modules never leaves HBM.  Every other hot-path module has the same shape (type name, config lines and ports differ).
#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE

#include <vector>

#include <jetstream/runtime_context_native_hip.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/registry.hh>
#include <jetstream_hip.h>  // this repo's include/

#include "module_impl.hh"

namespace Jetstream::Modules {

struct FftImplNativeHip : public FftImpl, public NativeHipRuntimeContext, public Scheduler::Context {
    Result create() override {
        JST_CHECK(FftImpl::create());
        if (!input.contiguous() || !output.contiguous() || input.dtype() != DataType::CF32 || output.dtype() != DataType::CF32) {
            JST_ERROR("[MODULE_FFT_NATIVE_HIP] Dense CF32 tensors only.");
            return Result::ERROR;
        }
        const std::vector<uint64_t> inShape(input.shape().begin(), input.shape().end());
        if (jst_tensor_wrap(const_cast<void*>(input.data()), input.sizeBytes(), JST_DEVICE_HIP, JST_DTYPE_CF32,
                            (uint32_t)inShape.size(), inShape.data(), nullptr, 0, &devIn) != JST_SUCCESS)
            return fail();
        (void)jst_tensor_set_attribute_u64(devIn, "sampleAxis", resolvedAxis);
        const char* config[] = {forward ? "forward=true" : "forward=false"};
        const char* ports[] = {"signal"};
        if (jst_module_create("fft", JST_DEVICE_HIP, "generic", name().c_str(), config, 1, ports, &devIn, 1, &module) != JST_SUCCESS)
            return fail();
        // the library module allocated its own output; FftImpl::create() has already allocated -- and published -- `output`:
        // move the library's storage onto the reference's buffer, so that the kernel writes where the consumers read
        if (jst_module_output(module, "signal", &devOut) != JST_SUCCESS ||
            jst_tensor_rebind(devOut, output.data(), output.sizeBytes()) != JST_SUCCESS)
            return fail();
        return Result::SUCCESS;
    }

    Result fail() {
        JST_ERROR("[MODULE_FFT_NATIVE_HIP] {}", jst_last_error());
        return Result::ERROR;
    }

    Result destroy() override {
        if (module) (void)jst_module_destroy(module);
        if (devIn) (void)jst_tensor_destroy(devIn);
        if (devOut) (void)jst_tensor_destroy(devOut);
        module = {};
        devIn = devOut = {};
        return FftImpl::destroy();
    }

    Result computeInitialize() override { return static_cast<Result>(jst_module_compute_initialize(module)); }
    Result computeSubmit(void* hipStream) override { return static_cast<Result>(jst_module_compute_submit(module, hipStream)); }
    Result computeDeinitialize() override { return static_cast<Result>(jst_module_compute_deinitialize(module)); }

    jst_tensor devIn{}, devOut{};
    jst_module module{};
};

JST_REGISTER_MODULE(FftImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");

}  // namespace Jetstream::Modules

#endif  // JETSTREAM_BACKEND_HIP_AVAILABLE
