// integration/device_hip/buffer_hip.cc -- REFERENCE-SIDE code: would live at src/memory/buffer_hip.cc.
// The HIP tensor store: a detail::Backend (src/memory/buffer_backend.hh:12-43) over hipMalloc'ed HBM, zero-initialised like
// every backend of the reference, with copies in every direction (host <-> HBM through the runtime's copy engine, HBM <-> HBM
// on the device).  `hostAccessible` buffers are pinned host memory mapped into the device (what the Soapy producer
// stages into).  MakeBackend (src/memory/buffer.cc:13-32) reaches it through
// core_hip_device.patch.
#include "buffer_backend.hh"

#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE

#include <hip/hip_runtime_api.h>

#include "jetstream/logger.hh"

namespace Jetstream::detail {

namespace {

#define JST_HIP_TRY(call, what)                                                                   \
    do {                                                                                          \
        const hipError_t err_ = (call);                                                           \
        if (err_ != hipSuccess) {                                                                 \
            JST_ERROR("[MEMORY:BUFFER:HIP] {}: {}.", what, hipGetErrorString(err_));              \
            return Result::ERROR;                                                                 \
        }                                                                                         \
    } while (0)

class HipBackend final : public Backend {
 public:
    HipBackend() = default;
    ~HipBackend() override { destroy(); }

    DeviceType device() const override { return DeviceType::HIP; }

    Result create(const U64& bytes, const Buffer::Config& config) override {
        destroy();
        sizeBytes = bytes;
        if (bytes == 0) return Result::SUCCESS;
        if (config.hostAccessible) {
            JST_HIP_TRY(hipHostMalloc(&pointer, bytes, hipHostMallocMapped), "Failed to allocate pinned host memory");
            std::memset(pointer, 0, bytes);
            locationState = Location::Unified;
        } else {
            JST_HIP_TRY(hipMalloc(&pointer, bytes), "Failed to allocate device memory");
            JST_HIP_TRY(hipMemset(pointer, 0, bytes), "Failed to clear device memory");
            locationState = Location::Device;
        }
        ownsMemory = true;
        return Result::SUCCESS;
    }

    Result create(void* external, const U64& bytes) override {  // borrowed device pointer (e.g. a ring slot of the library)
        destroy();
        pointer = external;
        sizeBytes = bytes;
        borrowed = true;
        ownsMemory = false;
        locationState = Location::Device;
        return Result::SUCCESS;
    }

    Result create(const Backend& source) override {  // the same bytes seen from this device
        if (source.device() == DeviceType::HIP) return create(const_cast<void*>(source.rawHandle()), source.size());
        JST_ERROR("[MEMORY:BUFFER:HIP] Cannot mirror a {} buffer without a copy.", source.device());
        return Result::ERROR;
    }

    void destroy() override {
        if (pointer && ownsMemory) {
            if (locationState == Location::Unified) (void)hipHostFree(pointer);
            else (void)hipFree(pointer);
        }
        pointer = nullptr;
        sizeBytes = 0;
        ownsMemory = true;
        borrowed = false;
        locationState = Location::Device;
    }

    Result copyFrom(const Backend& source, void* context) override {  // context: the hipStream_t of the calling runtime, or null
        if (source.size() > sizeBytes) {
            JST_ERROR("[MEMORY:BUFFER:HIP] Source ({} bytes) does not fit ({} bytes).", source.size(), sizeBytes);
            return Result::ERROR;
        }
        const hipMemcpyKind kind = source.device() == DeviceType::HIP ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        if (context) {
            JST_HIP_TRY(hipMemcpyAsync(pointer, source.rawHandle(), source.size(), kind, static_cast<hipStream_t>(context)),
                        "Failed to enqueue the copy");
        } else {
            JST_HIP_TRY(hipMemcpy(pointer, source.rawHandle(), source.size(), kind), "Failed to copy");
        }
        return Result::SUCCESS;
    }

    void* rawHandle() override { return pointer; }
    const void* rawHandle() const override { return pointer; }
    bool isBorrowed() const override { return borrowed; }
    Location location() const override { return locationState; }
    U64 size() const override { return sizeBytes; }

 private:
    void* pointer = nullptr;
    U64 sizeBytes = 0;
    bool ownsMemory = true;
    bool borrowed = false;
    Location locationState = Location::Device;
};

}  // namespace

std::unique_ptr<Backend> CreateHipBackend() { return std::make_unique<HipBackend>(); }

}  // namespace Jetstream::detail

#endif  // JETSTREAM_BACKEND_HIP_AVAILABLE
