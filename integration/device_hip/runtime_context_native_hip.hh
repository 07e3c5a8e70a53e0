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
    // The jst_module (include/jetstream_hip.h of libjetstream_hip.so) standing behind this module, or null for a module with
    // kernels of its own.  A runtime segment made ONLY of library modules is handed to one jst_runtime, which captures the
    // cycle into a hipGraph, fuses module chains into single kernels and batches cycles (src/runtime/native/hip/impl.cc);
    // a mixed segment runs module by module on the segment's stream.
    virtual void* libraryModule() const { return nullptr; }
    // Deferred cycles only: outputs the library keeps in rings of its own are copied (latest slot, on `hipStream`) into the
    // tensors this module published.
    virtual Result publishOutputs(void* /*hipStream*/) { return Result::SUCCESS; }
};

}  // namespace Jetstream

#endif  // JETSTREAM_RUNTIME_CONTEXT_NATIVE_HIP_HH
