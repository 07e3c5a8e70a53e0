// tensor.cc -- Tensor/Buffer/axis implementation for the HIP backend (see tensor.hh).
#include "tensor.hh"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace jst {

// ---- names / logging ---------------------------------------------------------------------------
const char* DataTypeName(DataType t) {
    switch (t) {
        case DataType::F32: return "F32";
        case DataType::CF32: return "CF32";
        case DataType::F64: return "F64";
        case DataType::U64: return "U64";
        case DataType::I8: return "I8";
        case DataType::U8: return "U8";
        case DataType::CI8: return "CI8";
        case DataType::I16: return "I16";
        case DataType::CI16: return "CI16";
        case DataType::CU8: return "CU8";
        case DataType::U16: return "U16";
        case DataType::CU16: return "CU16";
        case DataType::I32: return "I32";
        case DataType::CI32: return "CI32";
        case DataType::U32: return "U32";
        case DataType::CU32: return "CU32";
        case DataType::CF64: return "CF64";
        default: return "None";
    }
}
DataType NameToDataType(const std::string& name) {
    for (uint8_t v = 1; v <= static_cast<uint8_t>(DataType::CF64); ++v)
        if (name == DataTypeName(static_cast<DataType>(v))) return static_cast<DataType>(v);
    return DataType::None;
}
const char* DeviceName(DeviceType d) {
    switch (d) {
        case DeviceType::CPU: return "cpu";
        case DeviceType::HIP: return "hip";
        default: return "none";
    }
}
DeviceType StringToDevice(const std::string& s) {
    std::string l(s);
    std::transform(l.begin(), l.end(), l.begin(), [](unsigned char c) { return std::tolower(c); });
    if (l == "cpu") return DeviceType::CPU;
    if (l == "hip" || l == "rocm" || l == "mi355x") return DeviceType::HIP;
    return DeviceType::None;
}
const char* ResultName(Result r) {
    static const char* names[] = {"SUCCESS", "ERROR", "WARNING", "FATAL", "SKIP",
                                  "YIELD", "RELOAD", "RECREATE", "TIMEOUT", "INCOMPLETE"};
    const auto i = static_cast<uint16_t>(r);
    return i < 10 ? names[i] : "UNKNOWN";
}

namespace {
thread_local char g_last_error[1024] = {0};
int log_level() {
    static const int level = [] {
        const char* e = std::getenv("JST_LOG");
        return e ? std::atoi(e) : 0;
    }();
    return level;
}
}  // namespace

void log_error(const char* fmt, ...) {
    char buf[sizeof(g_last_error)];  // arguments may point into g_last_error itself
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    std::memcpy(g_last_error, buf, sizeof(buf));
    if (log_level() >= 1) std::fprintf(stderr, "[jetstream-hip] ERROR %s\n", g_last_error);
}
void log_debug(const char* fmt, ...) {
    if (log_level() < 2) return;
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "[jetstream-hip] DEBUG %s\n", buf);
}
const char* last_error() { return g_last_error; }

// ---- Buffer ------------------------------------------------------------------------------------
Buffer::~Buffer() {
    if (!owned || !ptr) return;
    if (device == DeviceType::HIP) (void)hipFree(ptr);
    else if (device == DeviceType::CPU) (void)hipHostFree(ptr);
}

Result Tensor::rebind(void* ptr, size_t bytes) {
    if (!buffer_ || !ptr) {
        JST_ERROR("[MEMORY:TENSOR] Cannot rebind an unallocated tensor (or to a null pointer).");
        return Result::ERROR;
    }
    if (buffer_->slots != 1) {
        JST_ERROR("[MEMORY:TENSOR] Cannot rebind ring storage.");
        return Result::ERROR;
    }
    if (buffer_->bound > 0) {
        JST_ERROR("[MEMORY:TENSOR] Cannot rebind storage a live runtime has planned over (captured graphs hold its address): "
                  "rebind before Runtime::create, or destroy the runtime first.");
        return Result::ERROR;
    }
    if (buffer_->device == DeviceType::HIP) {  // memory the device can address: HBM, or host memory the HIP runtime pinned / mapped
        hipPointerAttribute_t attributes{};
        if (hipPointerGetAttributes(&attributes, ptr) != hipSuccess || attributes.type == hipMemoryTypeUnregistered) {
            (void)hipGetLastError();
            JST_ERROR("[MEMORY:TENSOR] Cannot rebind a HIP tensor to memory the device cannot address.");
            return Result::ERROR;
        }
    }
    if (bytes < buffer_->bytes) {
        JST_ERROR("[MEMORY:TENSOR] External buffer of %llu bytes is smaller than the storage it replaces (%llu bytes).",
                  (unsigned long long)bytes, (unsigned long long)buffer_->bytes);
        return Result::ERROR;
    }
    if (buffer_->owned && buffer_->ptr) {
        if (buffer_->device == DeviceType::HIP) (void)hipFree(buffer_->ptr);
        else if (buffer_->device == DeviceType::CPU) (void)hipHostFree(buffer_->ptr);
    }
    buffer_->ptr = ptr;
    buffer_->owned = false;
    return Result::SUCCESS;
}

std::vector<U64> DenseStrides(const Shape& shape) {
    std::vector<U64> s(shape.size(), 1);
    for (size_t i = shape.size(); i-- > 1;) s[i - 1] = s[i] * shape[i];
    return s;
}

std::string ShapeToString(const Shape& shape) {
    std::string s = "[";
    for (size_t i = 0; i < shape.size(); ++i) {
        if (i) s += ", ";
        s += std::to_string(shape[i]);
    }
    return s + "]";
}

// ---- Tensor ------------------------------------------------------------------------------------
U64 Tensor::size() const {
    if (shape_.empty()) return 0;
    U64 n = 1;
    for (U64 d : shape_) n *= d;
    return n;
}

bool Tensor::contiguous() const {
    U64 expect = 1;
    for (size_t i = shape_.size(); i-- > 0;) {
        if (shape_[i] != 1 && stride_[i] != expect) return false;
        expect *= shape_[i];
    }
    return true;
}

Result Tensor::create(DeviceType device, DataType dtype, const Shape& shape) {
    return createRing(device, dtype, shape, 1);
}

Result Tensor::ringSelect(U64 slot) {
    if (!buffer_ || slot >= buffer_->slots) {
        JST_ERROR("[MEMORY] Ring slot %llu out of range.", (unsigned long long)slot);
        return Result::ERROR;
    }
    buffer_->slot = slot;
    return Result::SUCCESS;
}

Result Tensor::promoteToRing(U64 slots) {
    if (!buffer_ || buffer_->device != DeviceType::HIP || buffer_->slots != 1 || slots < 2) {
        JST_ERROR("[MEMORY] Only a single-slot HBM buffer can be promoted to a ring.");
        return Result::ERROR;
    }
    // Borrowed storage (jst_tensor_rebind: a host framework's buffer) is left where it is and the ring becomes the
    // library's own allocation: the host's buffer no longer receives the cycles' results -- a host that batches cycles
    // publishes the latest slot itself (integration/device_hip/runtime_native_hip_impl.cc: publishLatest).
    const size_t slot_bytes = buffer_->bytes;
    void* fresh = nullptr;
    JST_HIP_CHECK(hipMalloc(&fresh, slot_bytes * slots), "hipMalloc");
    JST_HIP_CHECK(hipMemset(fresh, 0, slot_bytes * slots), "hipMemset");
    if (buffer_->owned) (void)hipFree(buffer_->ptr);
    buffer_->owned = true;
    buffer_->ptr = fresh;
    buffer_->bytes = slot_bytes * slots;
    buffer_->slot_bytes = slot_bytes;
    buffer_->slots = slots;
    buffer_->slot = 0;
    return Result::SUCCESS;
}

Result Tensor::createRing(DeviceType device, DataType dtype, const Shape& shape, U64 slots) {
    if (slots == 0) {
        JST_ERROR("[MEMORY] A ring needs at least one slot.");
        return Result::ERROR;
    }
    if (DataTypeSize(dtype) == 0) {
        JST_ERROR("[MEMORY] Cannot create a tensor of dtype %s.", DataTypeName(dtype));
        return Result::ERROR;
    }
    U64 n = shape.empty() ? 0 : 1;
    for (U64 d : shape) n *= d;
    const size_t bytes = n * DataTypeSize(dtype);
    auto buf = std::make_shared<Buffer>();
    buf->device = device;
    buf->bytes = bytes * slots;
    buf->owned = true;
    buf->slots = slots;
    buf->slot_bytes = bytes;
    const size_t alloc = bytes ? bytes * slots : 16;
    if (device == DeviceType::HIP) {
        JST_HIP_CHECK(hipMalloc(&buf->ptr, alloc), "hipMalloc");
        JST_HIP_CHECK(hipMemset(buf->ptr, 0, alloc), "hipMemset");
    } else if (device == DeviceType::CPU) {
        // pinned: these tensors are the host end of async H2D/D2H copies
        JST_HIP_CHECK(hipHostMalloc(&buf->ptr, alloc, hipHostMallocDefault), "hipHostMalloc");
        std::memset(buf->ptr, 0, alloc);
    } else {
        JST_ERROR("[MEMORY] Unsupported device for tensor allocation.");
        return Result::ERROR;
    }
    buffer_ = std::move(buf);
    dtype_ = dtype;
    shape_ = shape;
    stride_ = DenseStrides(shape);
    offset_ = 0;
    attrs_.clear();
    return Result::SUCCESS;
}

Result Tensor::wrap(void* ptr, size_t bytes, DeviceType device, DataType dtype, const Shape& shape,
                    const std::vector<U64>& stride, U64 offset) {
    if (!stride.empty() && stride.size() != shape.size()) {
        JST_ERROR("[MEMORY] Stride rank %zu does not match shape rank %zu.", stride.size(),
                  shape.size());
        return Result::ERROR;
    }
    auto buf = std::make_shared<Buffer>();
    buf->ptr = ptr;
    buf->bytes = bytes;
    buf->device = device;
    buf->owned = false;
    buffer_ = std::move(buf);
    dtype_ = dtype;
    shape_ = shape;
    stride_ = stride.empty() ? DenseStrides(shape) : stride;
    offset_ = offset;
    // the view must fit in the buffer
    U64 last = offset_;
    for (size_t i = 0; i < shape_.size(); ++i) {
        if (shape_[i] == 0) return Result::SUCCESS;
        last += (shape_[i] - 1) * stride_[i];
    }
    if (!shape_.empty() && (last + 1) * DataTypeSize(dtype_) > bytes) {
        JST_ERROR("[MEMORY] Wrapped view exceeds the %zu-byte buffer.", bytes);
        buffer_.reset();
        return Result::ERROR;
    }
    return Result::SUCCESS;
}

Result Tensor::view(const Tensor& base, const Shape& shape, const std::vector<U64>& stride, U64 offset) {
    if (!base.buffer_ || (!stride.empty() && stride.size() != shape.size())) {
        JST_ERROR("[MEMORY] A view needs an allocated tensor and one stride per axis.");
        return Result::ERROR;
    }
    const std::vector<U64> st = stride.empty() ? DenseStrides(shape) : stride;
    U64 last = offset;
    bool empty = shape.empty();
    for (size_t i = 0; i < shape.size(); ++i) {
        if (shape[i] == 0) empty = true;
        else last += (shape[i] - 1) * st[i];
    }
    const size_t slot_bytes = base.buffer_->slots > 1 ? base.buffer_->slot_bytes : base.buffer_->bytes;
    if (!empty && (last + 1) * DataTypeSize(base.dtype_) > slot_bytes) {
        JST_ERROR("[MEMORY] View exceeds the %zu-byte storage.", slot_bytes);
        return Result::ERROR;
    }
    *this = base;  // shares the storage; the attributes come along as a copy
    shape_ = shape;
    stride_ = st;
    offset_ = offset;
    return Result::SUCCESS;
}

Result Tensor::reshape(const Shape& shape) {
    if (!contiguous()) {  // tensor.cc:239-242
        JST_ERROR("[MEMORY] Cannot reshape a non-contiguous tensor.");
        return Result::ERROR;
    }
    U64 n = shape.empty() ? 0 : 1;
    for (U64 d : shape) n *= d;
    if (n != size()) {
        JST_ERROR("[MEMORY] Cannot reshape %s into %s.", ShapeToString(shape_).c_str(),
                  ShapeToString(shape).c_str());
        return Result::ERROR;
    }
    shape_ = shape;
    stride_ = DenseStrides(shape);
    return Result::SUCCESS;
}

Result Tensor::expandDims(Index axis) {
    if (axis > rank()) {
        JST_ERROR("[MEMORY] expandDims axis %llu out of range.", (unsigned long long)axis);
        return Result::ERROR;
    }
    const U64 s = axis < rank() ? stride_[axis] * shape_[axis] : 1;
    shape_.insert(shape_.begin() + axis, 1);
    stride_.insert(stride_.begin() + axis, s);
    return Result::SUCCESS;
}

Result Tensor::squeezeDims(Index axis) {
    if (axis >= rank() || shape_[axis] != 1) {
        JST_ERROR("[MEMORY] squeezeDims axis %llu is not a size-1 axis.", (unsigned long long)axis);
        return Result::ERROR;
    }
    shape_.erase(shape_.begin() + axis);
    stride_.erase(stride_.begin() + axis);
    return Result::SUCCESS;
}

Result Tensor::slice(Index axis, U64 begin, U64 end, U64 step) {
    if (axis >= rank() || begin > end || end > shape_[axis] || step == 0) {
        JST_ERROR("[MEMORY] Invalid slice.");
        return Result::ERROR;
    }
    offset_ += begin * stride_[axis];
    shape_[axis] = (end - begin + step - 1) / step;
    stride_[axis] *= step;
    return Result::SUCCESS;
}

Result Tensor::permute(const std::vector<Index>& axes) {
    if (axes.size() != rank()) {
        JST_ERROR("[MEMORY] Permutation rank mismatch.");
        return Result::ERROR;
    }
    std::vector<bool> seen(rank(), false);
    Shape ns(rank());
    std::vector<U64> nst(rank());
    for (size_t i = 0; i < axes.size(); ++i) {
        if (axes[i] >= rank() || seen[axes[i]]) {
            JST_ERROR("[MEMORY] Invalid permutation.");
            return Result::ERROR;
        }
        seen[axes[i]] = true;
        ns[i] = shape_[axes[i]];
        nst[i] = stride_[axes[i]];
    }
    shape_ = ns;
    stride_ = nst;
    return Result::SUCCESS;
}

Result Tensor::broadcastTo(const Shape& shape) {  // tensor.cc:268-306: right-aligned, stride 0
    if (shape.size() < rank()) {
        JST_ERROR("[MEMORY] Cannot broadcast to a lower rank.");
        return Result::ERROR;
    }
    const size_t lead = shape.size() - rank();
    std::vector<U64> nst(shape.size(), 0);
    for (size_t i = 0; i < rank(); ++i) {
        const U64 want = shape[lead + i];
        if (shape_[i] == want) nst[lead + i] = stride_[i];
        else if (shape_[i] == 1) nst[lead + i] = 0;
        else {
            JST_ERROR("[MEMORY] Shape %s is not broadcastable to %s.", ShapeToString(shape_).c_str(),
                      ShapeToString(shape).c_str());
            return Result::ERROR;
        }
    }
    shape_ = shape;
    stride_ = nst;
    return Result::SUCCESS;
}

Result Tensor::setAttribute(const std::string& key, AttrValue value) {
    attrs_[key] = std::move(value);
    return Result::SUCCESS;
}
Result Tensor::removeAttribute(const std::string& key) {
    attrs_.erase(key);
    return Result::SUCCESS;
}
const AttrValue* Tensor::attribute(const std::string& key) const {
    auto it = attrs_.find(key);
    return it == attrs_.end() ? nullptr : &it->second;
}
Result Tensor::propagateAttributes(const Tensor& other) {
    for (const auto& kv : other.attrs_) attrs_[kv.first] = kv.second;
    return Result::SUCCESS;
}

Result Tensor::copyFromHost(const void* src, size_t bytes, hipStream_t stream) {
    if (!contiguous() || bytes != sizeBytes()) {
        JST_ERROR("[MEMORY] copyFromHost needs a contiguous tensor of exactly %llu bytes.",
                  (unsigned long long)sizeBytes());
        return Result::ERROR;
    }
    if (bytes == 0) return Result::SUCCESS;
    char* dst = static_cast<char*>(data()) + offsetBytes();
    JST_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream), "hipMemcpyAsync(H2D)");
    return Result::SUCCESS;
}
Result Tensor::copyToHost(void* dst, size_t bytes, hipStream_t stream) const {
    if (!contiguous() || bytes != sizeBytes()) {
        JST_ERROR("[MEMORY] copyToHost needs a contiguous tensor of exactly %llu bytes.",
                  (unsigned long long)sizeBytes());
        return Result::ERROR;
    }
    if (bytes == 0) return Result::SUCCESS;
    const char* src = static_cast<const char*>(data()) + offsetBytes();
    JST_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream), "hipMemcpyAsync(D2H)");
    return Result::SUCCESS;
}
Result Tensor::copyFrom(const Tensor& other, hipStream_t stream) {
    if (!contiguous() || !other.contiguous() || sizeBytes() != other.sizeBytes()) {
        JST_ERROR("[MEMORY] copyFrom needs two contiguous tensors of equal byte size.");
        return Result::ERROR;
    }
    if (sizeBytes() == 0) return Result::SUCCESS;
    char* dst = static_cast<char*>(data()) + offsetBytes();
    const char* src = static_cast<const char*>(other.data()) + other.offsetBytes();
    JST_HIP_CHECK(hipMemcpyAsync(dst, src, sizeBytes(), hipMemcpyDefault, stream),
                  "hipMemcpyAsync");
    return Result::SUCCESS;
}

// ---- signal axes -------------------------------------------------------------------------------
namespace {
Result ReadAxis(const Tensor& t, const char* name, std::optional<Index>& axis) {
    axis.reset();
    const AttrValue* v = t.attribute(name);
    if (!v) return Result::SUCCESS;
    const U64* idx = std::get_if<U64>(v);
    if (!idx) {
        JST_ERROR("[MEMORY:AXIS] Attribute '%s' must have type Index.", name);
        return Result::ERROR;
    }
    if (*idx >= t.rank()) {
        JST_ERROR("[MEMORY:AXIS] Attribute '%s' axis %llu is out of range for rank %llu.", name,
                  (unsigned long long)*idx, (unsigned long long)t.rank());
        return Result::ERROR;
    }
    axis = *idx;
    return Result::SUCCESS;
}

Result ValidateAxes(const Tensor& t, const SignalAxes& axes, bool requireSample) {
    if (requireSample && !axes.sample) {
        JST_ERROR("[MEMORY:AXIS] Signal tensor is missing sampleAxis metadata.");
        return Result::ERROR;
    }
    const std::pair<const char*, std::optional<Index>> roles[3] = {
        {SampleAxisAttribute, axes.sample},
        {BatchAxisAttribute, axes.batch},
        {ChannelAxisAttribute, axes.channel}};
    for (int i = 0; i < 3; ++i) {
        if (!roles[i].second) continue;
        if (*roles[i].second >= t.rank()) {
            JST_ERROR("[MEMORY:AXIS] Attribute '%s' axis %llu is out of range for rank %llu.",
                      roles[i].first, (unsigned long long)*roles[i].second,
                      (unsigned long long)t.rank());
            return Result::ERROR;
        }
        for (int j = i + 1; j < 3; ++j) {
            if (roles[j].second && *roles[i].second == *roles[j].second) {
                JST_ERROR("[MEMORY:AXIS] Attributes '%s' and '%s' cannot use axis %llu.",
                          roles[i].first, roles[j].first, (unsigned long long)*roles[i].second);
                return Result::ERROR;
            }
        }
    }
    return Result::SUCCESS;
}
}  // namespace

bool HasSignalAxes(const Tensor& t) {
    return t.hasAttribute(SampleAxisAttribute) || t.hasAttribute(BatchAxisAttribute) ||
           t.hasAttribute(ChannelAxisAttribute);
}

Result ResolveSignalAxes(const Tensor& t, SignalAxes& axes) {  // axis.cc:231-245
    axes = {};
    JST_CHECK(ReadAxis(t, SampleAxisAttribute, axes.sample));
    JST_CHECK(ReadAxis(t, BatchAxisAttribute, axes.batch));
    JST_CHECK(ReadAxis(t, ChannelAxisAttribute, axes.channel));
    if (!axes.sample && t.rank() == 1) axes.sample = Index{0};
    JST_CHECK(ValidateAxes(t, axes, true));
    return Result::SUCCESS;
}

Result MapSignalAxes(const Tensor& t, SignalAxes& axes) {  // axis.cc:268-313, identity map
    axes = {};
    const bool has = HasSignalAxes(t);
    if (!has && t.rank() != 1) return Result::SUCCESS;
    JST_CHECK(ReadAxis(t, SampleAxisAttribute, axes.sample));
    JST_CHECK(ReadAxis(t, BatchAxisAttribute, axes.batch));
    JST_CHECK(ReadAxis(t, ChannelAxisAttribute, axes.channel));
    if (!has) axes.sample = Index{0};
    JST_CHECK(ValidateAxes(t, axes, false));
    return Result::SUCCESS;
}

Result SetSignalAxes(Tensor& t, const SignalAxes& axes) {  // axis.cc:247-266
    JST_CHECK(ValidateAxes(t, axes, false));
    auto set_or_remove = [&](const char* key, const std::optional<Index>& axis) {
        if (axis) t.setAttribute(key, AttrValue{U64{*axis}});
        else t.removeAttribute(key);
    };
    set_or_remove(SampleAxisAttribute, axes.sample);
    set_or_remove(BatchAxisAttribute, axes.batch);
    set_or_remove(ChannelAxisAttribute, axes.channel);
    return Result::SUCCESS;
}

}  // namespace jst
