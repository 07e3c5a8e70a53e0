// tensor.hh -- Tensor / Buffer of the HIP backend.
//
// Mirrors the contract of the reference's memory layer (include/jetstream/memory/tensor.hh:24-135,
// src/memory/tensor.cc): typed N-d view {shape, stride, offset} in ELEMENTS over a shared,
// ref-counted buffer, plus a string-keyed attribute map that carries the signal-axis roles.
// Re-designed for one MI355X per process: a buffer is either HBM (hipMalloc, zero-filled like
// src/memory/buffer_cuda.cc:31-124), pinned host memory (hipHostMalloc; the staging side of the
// async H2D feed) or a borrowed pointer (a torch tensor, a pinned SDR ring...).  As in the
// reference, data() of a device tensor is the buffer BASE: kernels add offset() themselves
// (src/memory/tensor.cc:1090-1095).
#pragma once

#include <hip/hip_runtime.h>

#include "types.hh"

namespace jst {

struct Buffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    DeviceType device = DeviceType::None;
    bool owned = false;
    // Ring storage: 'slots' equally sized slots, the views address the selected one.  This is
    // the HBM-resident replacement for the reference's host CircularBuffer + per-cycle memcpy
    // (src/domains/io/soapy/module_impl_native_cpu.cc:39-60): a source fills slots with async
    // H2D copies on a side stream and selects the slot a compute cycle reads.
    U64 slots = 1;
    size_t slot_bytes = 0;
    U64 slot = 0;
    // Number of live runtimes that planned over this storage: captured hipGraphs, transform plans and ring selections hold
    // its raw address from Runtime::create on, so the storage cannot move (Tensor::rebind) until they are gone.
    int bound = 0;
    void* base() const { return static_cast<char*>(ptr) + slot * slot_bytes; }
    ~Buffer();
};

using AttrValue = std::variant<U64, F64, std::vector<F64>, std::vector<U64>>;

class Tensor {
 public:
    Tensor() = default;

    // Allocating constructor-equivalents (tensor.cc:546-577).  Memory is zero-initialised.
    Result create(DeviceType device, DataType dtype, const Shape& shape);
    // Ring of 'slots' tensors of this shape in one allocation; ringSelect() moves every view.
    Result createRing(DeviceType device, DataType dtype, const Shape& shape, U64 slots);
    Result ringSelect(U64 slot);
    // Re-allocate an owned, single-slot HBM buffer as a ring of 'slots' (contents are dropped,
    // every view of the storage follows).  Used by the runtime to double-buffer an intermediate
    // tensor between a producer kernel and a consumer that runs concurrently with the NEXT cycle.
    Result promoteToRing(U64 slots);
    U64 ringSlots() const { return buffer_ ? buffer_->slots : 0; }
    // Address of ring slot 'slot' of this (dense, offset-free) tensor, independent of the selected slot: the
    // producer side of a live source uploads into slots the compute side is not reading.
    void* ringSlotData(U64 slot) const {
        return buffer_ && slot < buffer_->slots ? static_cast<char*>(buffer_->ptr) + slot * buffer_->slot_bytes : nullptr;
    }
    U64 ringSlot() const { return buffer_ ? buffer_->slot : 0; }
    // Borrow external memory (no ownership): stride empty = dense row-major.
    Result wrap(void* ptr, size_t bytes, DeviceType device, DataType dtype, const Shape& shape,
                const std::vector<U64>& stride = {}, U64 offset = 0);

    // Move this tensor's STORAGE -- and with it every view of that storage -- onto external memory of at least the same
    // size (no ownership; what was owned is freed, contents are dropped).  For a host framework that has already
    // allocated the buffer a module is to write (the reference's Impl::create() allocates `output` before a device
    // binding gets a say: integration/device_hip/fft_module_impl_native_hip.cc).  Single-slot storage only, and before
    // the first compute (a captured graph holds addresses).
    Result rebind(void* ptr, size_t bytes);
    void runtimeBound(int delta) const { if (buffer_) buffer_->bound += delta; }
    // This tensor becomes a view {shape, stride, offset} (elements; stride empty = dense) of `base`'s STORAGE: what the
    // reference's Tensor copies are once slice / permute / broadcastTo have edited them (src/memory/tensor.cc:196-306) --
    // for a host framework whose consumer sees a producer's tensor through a geometry of its own.  Storage identity (and
    // with it the runtime's data-flow edges) is kept; bounds are checked against one ring slot.
    Result view(const Tensor& base, const Shape& shape, const std::vector<U64>& stride, U64 offset);
    bool valid() const { return static_cast<bool>(buffer_); }
    bool validShape() const { return !shape_.empty(); }
    DeviceType device() const { return buffer_ ? buffer_->device : DeviceType::None; }
    DataType dtype() const { return dtype_; }
    const Shape& shape() const { return shape_; }
    U64 shape(Index axis) const { return shape_[axis]; }
    const std::vector<U64>& stride() const { return stride_; }
    U64 stride(Index axis) const { return stride_[axis]; }
    U64 offset() const { return offset_; }
    U64 offsetBytes() const { return offset_ * DataTypeSize(dtype_); }
    Index rank() const { return shape_.size(); }
    U64 size() const;
    U64 sizeBytes() const { return size() * DataTypeSize(dtype_); }
    bool contiguous() const;
    void* data() const { return buffer_ ? buffer_->base() : nullptr; }
    const void* storageId() const { return buffer_.get(); }

    // Views (mutating, like the reference: tensor.cc:196-306).
    Result reshape(const Shape& shape);
    Result expandDims(Index axis);
    Result squeezeDims(Index axis);
    Result slice(Index axis, U64 begin, U64 end, U64 step = 1);
    Result permute(const std::vector<Index>& axes);
    Result broadcastTo(const Shape& shape);
    Tensor clone() const { return *this; }  // shares storage and attributes

    // Attributes (axis.hh keys + sampleRate, frequency, ...).
    Result setAttribute(const std::string& key, AttrValue value);
    Result removeAttribute(const std::string& key);
    bool hasAttribute(const std::string& key) const { return attrs_.count(key) != 0; }
    const AttrValue* attribute(const std::string& key) const;
    const std::map<std::string, AttrValue>& attributes() const { return attrs_; }
    Result propagateAttributes(const Tensor& other);

    // Dense copies (tensor.cc:882-963 restricts copyFrom to contiguous, zero-offset tensors; here
    // the offset is honoured).  Asynchronous on 'stream'; host sides should be pinned.
    Result copyFromHost(const void* src, size_t bytes, hipStream_t stream);
    Result copyToHost(void* dst, size_t bytes, hipStream_t stream) const;
    Result copyFrom(const Tensor& other, hipStream_t stream);

 private:
    std::shared_ptr<Buffer> buffer_;
    DataType dtype_ = DataType::None;
    Shape shape_;
    std::vector<U64> stride_;
    U64 offset_ = 0;
    std::map<std::string, AttrValue> attrs_;
};

std::vector<U64> DenseStrides(const Shape& shape);
std::string ShapeToString(const Shape& shape);

// ---- signal axes (include/jetstream/memory/axis.hh:15-58, src/memory/axis.cc:231-313) ----------
constexpr const char* SampleAxisAttribute = "sampleAxis";
constexpr const char* BatchAxisAttribute = "batchAxis";
constexpr const char* ChannelAxisAttribute = "channelAxis";

struct SignalAxes {
    std::optional<Index> sample, batch, channel;
};

bool HasSignalAxes(const Tensor& tensor);
// Strict: a rank-1 tensor without attributes has sampleAxis 0; anything else must carry them.
Result ResolveSignalAxes(const Tensor& tensor, SignalAxes& axes);
// Lenient (identity axis map): a rank>=2 tensor without attributes yields EMPTY axes.
Result MapSignalAxes(const Tensor& tensor, SignalAxes& axes);
Result SetSignalAxes(Tensor& tensor, const SignalAxes& axes);

}  // namespace jst
