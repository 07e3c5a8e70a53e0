// integration/device_hip/hip_library_module.hh -- REFERENCE-SIDE code: would live at include/jetstream/backend/devices/hip/
// library_module.hh.  What every module_impl_native_hip.cc shares: ONE library module of
// libjetstream_hip.so standing behind a reference module of (DeviceType::HIP, RuntimeType::NATIVE), DEVICE-RESIDENT --
//   * inputs are not copied: the library sees the very HBM the reference's producer wrote.  When that producer is itself a
//     library module, the consumer is handed the PRODUCER'S OWN library tensor (TensorDirectory: reference "module.port" ->
//     jst_tensor), so the library's runtime sees the same data-flow edges the reference's scheduler sees
//     (TensorLink::producer, include/jetstream/tensor_link.hh:22-34) and can fuse / graph-capture / batch the segment
//     (runtime_native_hip_impl.cc hands a segment of library modules to ONE jst_runtime).  A tensor with no library producer
//     (a test source, another vendor's HIP module) is borrowed by pointer (jst_tensor_wrap);
//   * outputs are not copied either: the reference's Impl::create() has already allocated and published `output` on the
//     device (Tensor::create(device(), ..) through buffer_hip.cc); the library module's output STORAGE is moved onto that
//     buffer (jst_tensor_rebind), so the kernels write where every reference-side reader looks;
//   * the geometry (shape / stride / offset, in elements) and the attributes (signal axes, sampleRate, centre lists ...) the
//     consumer sees on the reference's tensor are carried over to the library's handle (jst_tensor_view keeps the storage).
// Compiled against the reference's real headers into oracle/_ref/libref_jetstream_devhip.so by oracle/ref_jetstream_build.sh and
// run on the GPU by tests/test_gpu_reference_device_hip.py.
#pragma once

#include <any>
#include <cstdio>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <jetstream/logger.hh>
#include <jetstream/memory/tensor.hh>
#include <jetstream/tensor_link.hh>
#include <jetstream/types.hh>
#include <jetstream_hip.h>  // this repo's include/

namespace Jetstream::Hip {

inline uint8_t DtypeCode(const DataType dtype) {
    switch (dtype) {
        case DataType::F32: return JST_DTYPE_F32;
        case DataType::CF32: return JST_DTYPE_CF32;
        case DataType::F64: return JST_DTYPE_F64;
        case DataType::U64: return JST_DTYPE_U64;
        case DataType::I8: return JST_DTYPE_I8;
        case DataType::CI8: return JST_DTYPE_CI8;
        case DataType::I16: return JST_DTYPE_I16;
        case DataType::CI16: return JST_DTYPE_CI16;
        case DataType::U8: return JST_DTYPE_U8;
        case DataType::CU8: return JST_DTYPE_CU8;
        case DataType::U16: return JST_DTYPE_U16;
        case DataType::CU16: return JST_DTYPE_CU16;
        case DataType::I32: return JST_DTYPE_I32;
        case DataType::CI32: return JST_DTYPE_CI32;
        case DataType::U32: return JST_DTYPE_U32;
        case DataType::CU32: return JST_DTYPE_CU32;
        case DataType::CF64: return JST_DTYPE_CF64;
        default: return 0;
    }
}

inline std::string Number(const double v) {
    char text[64];
    std::snprintf(text, sizeof(text), "%.17g", v);
    return text;
}
inline std::string Flag(const bool v) { return v ? "true" : "false"; }

// reference "module\nport" -> the library tensor its producer published.  Process-wide like the registry; entries are
// clones (shared storage), retracted when the producing module is destroyed.
class TensorDirectory {
 public:
    static TensorDirectory& Get() {
        static TensorDirectory directory;
        return directory;
    }
    void publish(const std::string& module, const std::string& port, jst_tensor tensor) {
        jst_tensor copy{};
        if (jst_tensor_clone(tensor, &copy) != JST_SUCCESS) return;
        std::lock_guard<std::mutex> lock(mutex);
        auto& slot = entries[module + "\n" + port];
        if (slot) (void)jst_tensor_destroy(slot);
        slot = copy;
    }
    jst_tensor find(const std::string& module, const std::string& port) {
        std::lock_guard<std::mutex> lock(mutex);
        const auto it = entries.find(module + "\n" + port);
        return it == entries.end() ? jst_tensor{} : it->second;
    }
    void retract(const std::string& module) {
        std::lock_guard<std::mutex> lock(mutex);
        for (auto it = entries.begin(); it != entries.end();) {
            if (it->first.compare(0, module.size() + 1, module + "\n") == 0) {
                (void)jst_tensor_destroy(it->second);
                it = entries.erase(it);
            } else {
                ++it;
            }
        }
    }

 private:
    std::mutex mutex;
    std::unordered_map<std::string, jst_tensor> entries;
};

class LibraryModule {
 public:
    struct Input {
        std::string port;          // the LIBRARY module's port name
        const TensorLink* link;    // what the reference's module received on it (inputs().at(..))
    };
    struct Output {
        std::string port;          // the LIBRARY module's port name
        std::string published;     // the REFERENCE module's port name (outputs()[..])
        Tensor* tensor;            // the reference-side tensor the kernels must write (allocated by Impl::create())
    };

    ~LibraryModule() { (void)destroy(); }

    Result create(const std::string& tag, const char* type, const std::string& provider, const std::string& name,
                  const std::vector<std::string>& config, const std::vector<Input>& inputs, const std::vector<Output>& outputs) {
        (void)destroy();
        this->tag = tag;
        owner = name;
        std::vector<const char*> cfg, ports;
        for (const auto& line : config) cfg.push_back(line.c_str());
        for (const auto& input : inputs) {
            jst_tensor handle{};
            JST_CHECK(mirror(*input.link, handle));
            devIn.push_back(handle);
            ports.push_back(input.port.c_str());
        }
        if (jst_module_create(type, JST_DEVICE_HIP, provider.c_str(), name.c_str(), cfg.data(), (uint32_t)cfg.size(), ports.data(),
                              devIn.data(), (uint32_t)devIn.size(), &module) != JST_SUCCESS)
            return fail("jst_module_create");
        for (const auto& output : outputs) {
            jst_tensor handle{};
            if (jst_module_output(module, output.port.c_str(), &handle) != JST_SUCCESS) return fail("jst_module_output");
            devOut.push_back(handle);
            JST_CHECK(adopt(handle, *output.tensor, output.port));
            TensorDirectory::Get().publish(name, output.published, handle);
        }
        return Result::SUCCESS;
    }

    // a state tensor of the library module (spectrogram / waterfall "frequencyBins", lineplot "signalPoints" ...) lives in
    // the reference module's own state tensor: the present half and the tests read it there
    Result bindState(const char* key, Tensor& state) {
        jst_tensor handle{};
        if (jst_module_state(module, key, &handle) != JST_SUCCESS) return fail("jst_module_state");
        devOut.push_back(handle);
        // a state tensor has no consumer on the reference side: what must agree is the element type and the byte image
        // (the library's bins are the reference's row-major words, integration/mi355x_provider/spectrogram.cc downloads them 1:1)
        jst_tensor_desc d{};
        if (jst_tensor_describe(handle, &d) != JST_SUCCESS) return fail("jst_tensor_describe");
        uint64_t elements = d.rank ? 1 : 0;
        for (uint32_t axis = 0; axis < d.rank; ++axis) elements *= d.shape[axis];
        if (state.device() != DeviceType::HIP || d.dtype != DtypeCode(state.dtype()) || elements != state.size() || d.offset != 0 ||
            !state.contiguous() || state.offset() != 0) {
            JST_ERROR("[{}] The library's '{}' state does not match the reference module's.", tag, key);
            return Result::ERROR;
        }
        if (d.data != state.data()) {
            // what the library's create() put there (a lineplot's x coordinates, a waterfall's cleared rows) moves along
            if (hipMemcpy(state.data(), d.data, state.sizeBytes(), hipMemcpyDeviceToDevice) != hipSuccess) {
                JST_ERROR("[{}] Failed to move the '{}' state.", tag, key);
                return Result::ERROR;
            }
            if (jst_tensor_rebind(handle, state.data(), state.buffer().sizeBytes()) != JST_SUCCESS) return fail("jst_tensor_rebind");
        }
        TensorDirectory::Get().publish(owner, std::string("state:") + key, handle);  // for debuggers and tests
        return Result::SUCCESS;
    }

    jst_module handle() const { return module; }

    // Deferred cycles (runtime_native_hip_impl.cc): a cycle-batched library runtime turns some outputs into rings of its
    // own (one slot per cycle of a span); the reference's tensor then receives the LATEST slot, device to device.
    Result publishLatest(void* stream) {
        for (const auto& a : adopted) {
            jst_tensor_desc d{};
            if (jst_tensor_describe(a.dev, &d) != JST_SUCCESS) return fail("jst_tensor_describe");
            if (d.data == a.host) continue;
            if (hipMemcpyAsync(a.host, d.data, a.bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)) != hipSuccess) {
                JST_ERROR("[{}] Failed to publish the latest ring slot.", tag);
                return Result::ERROR;
            }
        }
        return Result::SUCCESS;
    }

    Result computeInitialize() { return check(jst_module_compute_initialize(module), "jst_module_compute_initialize"); }
    Result computeSubmit(void* stream) {
        const jst_result r = jst_module_compute_submit(module, stream);
        if (r == JST_SUCCESS || r == JST_SKIP || r == JST_YIELD || r == JST_TIMEOUT || r == JST_RELOAD) return static_cast<Result>(r);
        return fail("jst_module_compute_submit");
    }
    Result computeDeinitialize() { return module ? check(jst_module_compute_deinitialize(module), "jst_module_compute_deinitialize") : Result::SUCCESS; }

    Result destroy() {
        if (!owner.empty()) TensorDirectory::Get().retract(owner);
        if (module) (void)jst_module_destroy(module);
        for (auto& t : devIn) (void)jst_tensor_destroy(t);
        for (auto& t : devOut) (void)jst_tensor_destroy(t);
        module = {};
        devIn.clear();
        devOut.clear();
        adopted.clear();
        owner.clear();
        return Result::SUCCESS;
    }

 private:
    static bool sameGeometry(const jst_tensor_desc& d, const Tensor& t) {
        if (d.rank != t.rank() || d.offset != t.offset()) return false;
        for (Index axis = 0; axis < t.rank(); ++axis)
            if (d.shape[axis] != t.shape(axis) || (t.shape(axis) != 1 && d.stride[axis] != t.stride(axis))) return false;
        return true;
    }

    // the library's handle for what the reference's module sees on one input
    Result mirror(const TensorLink& link, jst_tensor& out) {
        const Tensor& host = link.tensor;
        if (host.device() != DeviceType::HIP || DtypeCode(host.dtype()) == 0 || host.rank() > JST_MAX_RANK) {
            JST_ERROR("[{}] Input must be a HIP tensor of at most {} axes and of a sample type the device path takes.", tag, JST_MAX_RANK);
            return Result::ERROR;
        }
        const std::vector<uint64_t> shape(host.shape().begin(), host.shape().end());
        const std::vector<uint64_t> stride(host.stride().begin(), host.stride().end());
        jst_tensor produced = link.producer ? TensorDirectory::Get().find(link.producer->module, link.producer->port) : jst_tensor{};
        if (produced) {
            jst_tensor_desc d{};
            if (jst_tensor_describe(produced, &d) != JST_SUCCESS) return fail("jst_tensor_describe");
            if (sameGeometry(d, host)) {
                if (jst_tensor_clone(produced, &out) != JST_SUCCESS) return fail("jst_tensor_clone");
            } else if (jst_tensor_view(produced, (uint32_t)shape.size(), shape.data(), stride.data(), host.offset(), &out) != JST_SUCCESS) {
                return fail("jst_tensor_view");
            }
        } else if (jst_tensor_wrap(const_cast<void*>(host.data()), host.buffer().sizeBytes(), JST_DEVICE_HIP, DtypeCode(host.dtype()),
                                   (uint32_t)shape.size(), shape.data(), stride.data(), host.offset(), &out) != JST_SUCCESS) {
            return fail("jst_tensor_wrap");
        }
        return attributes(host, out);
    }

    // every attribute kind the path's modules read (include/jetstream/memory/axis.hh:15-17, filter/block_impl.cc:498,542-543,
    // filter_taps/module_impl.cc:151-157)
    Result attributes(const Tensor& host, jst_tensor dev) {
        for (const auto& key : host.attributeKeys()) {
            const std::any a = host.attribute(key);
            jst_result r = JST_SUCCESS;
            if (const auto* v = std::any_cast<Index>(&a)) r = jst_tensor_set_attribute_u64(dev, key.c_str(), (uint64_t)*v);
            else if (const auto* v = std::any_cast<F32>(&a)) r = jst_tensor_set_attribute_f64(dev, key.c_str(), (double)*v);
            else if (const auto* v = std::any_cast<F64>(&a)) r = jst_tensor_set_attribute_f64(dev, key.c_str(), *v);
            else if (const auto* v = std::any_cast<std::vector<U64>>(&a)) r = jst_tensor_set_attribute_u64v(dev, key.c_str(), v->data(), v->size());
            else if (const auto* v = std::any_cast<std::vector<F64>>(&a)) r = jst_tensor_set_attribute_f64v(dev, key.c_str(), v->data(), v->size());
            else if (const auto* v = std::any_cast<std::vector<F32>>(&a)) {
                const std::vector<double> wide(v->begin(), v->end());
                r = jst_tensor_set_attribute_f64v(dev, key.c_str(), wide.data(), wide.size());
            }
            if (r != JST_SUCCESS) return fail("jst_tensor_set_attribute");
        }
        return Result::SUCCESS;
    }

    // the library's output (or state) tensor takes the reference tensor's buffer as its storage -- unless it already is a
    // view of it (reshape, a bypassing cast: the library made the same view of the same, earlier adopted, storage)
    Result adopt(jst_tensor dev, Tensor& host, const std::string& port) {
        jst_tensor_desc d{};
        if (jst_tensor_describe(dev, &d) != JST_SUCCESS) return fail("jst_tensor_describe");
        if (host.device() != DeviceType::HIP || !sameGeometry(d, host) || d.dtype != DtypeCode(host.dtype())) {
            JST_ERROR("[{}] The library's '{}' tensor does not have the layout the reference's module published.", tag, port);
            return Result::ERROR;
        }
        if (d.data == host.data() || host.size() == 0) return Result::SUCCESS;  // (an empty tensor has no buffer to write)
        if (jst_tensor_rebind(dev, host.data(), host.buffer().sizeBytes()) != JST_SUCCESS) return fail("jst_tensor_rebind");
        if (host.contiguous() && host.offset() == 0) adopted.push_back({dev, host.data(), host.sizeBytes()});
        return Result::SUCCESS;
    }

    Result check(const jst_result r, const char* what) { return r == JST_SUCCESS ? Result::SUCCESS : fail(what); }
    Result fail(const char* what) {
        JST_ERROR("[{}] {}: {}", tag, what, jst_last_error());
        return Result::ERROR;
    }

    struct Adopted {
        jst_tensor dev;
        void* host;
        size_t bytes;
    };
    std::string tag, owner;
    std::vector<jst_tensor> devIn, devOut;
    std::vector<Adopted> adopted;
    jst_module module{};
};

}  // namespace Jetstream::Hip
