// integration/device_hip/runtime_context_native_hip.hh -- REFERENCE-SIDE code: would live at
// include/jetstream/runtime_context_native_hip.hh (the HIP counterpart of runtime_context_native_cuda.hh:32-34): the hooks a
// module of (DeviceType::HIP, RuntimeType::NATIVE) implements.  The stream is the segment's own (one per flowgraph
// runtime); computeSubmit must only ENQUEUE on it.
#ifndef JETSTREAM_RUNTIME_CONTEXT_NATIVE_HIP_HH
#define JETSTREAM_RUNTIME_CONTEXT_NATIVE_HIP_HH

#include "jetstream/runtime.hh"
#include "jetstream/runtime_context.hh"

namespace Jetstream {

struct NativeHipRuntimeContext : Runtime::Context {
 public:
    virtual Result computeInitialize() { return Result::SUCCESS; }
    virtual Result computeSubmit(void* hipStream) = 0;
    virtual Result computeDeinitialize() { return Result::SUCCESS; }
};

}  // namespace Jetstream

#endif  // JETSTREAM_RUNTIME_CONTEXT_NATIVE_HIP_HH
