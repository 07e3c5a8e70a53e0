// integration/mi355x_provider/mi355x_bridge.hh -- REFERENCE-SIDE code (what a CyberEther maintainer adds; see INTEGRATION.md
// section 2): the host-staging bridge every `provider: mi355x` module shares.  A module registered as
// (DeviceType::CPU, RuntimeType::NATIVE, "mi355x") keeps the reference's own Impl (validate / define / create: CPU tensors,
// attribute propagation, error strings) and replaces ONLY computeSubmit(): inputs go host -> HBM (pinned staging, the
// library's side stream), one library module runs on the device behind include/jetstream_hip.h, the output comes back.
// Every module boundary crosses PCIe: a bring-up / verification path (the reference's scheduler, registry and tests drive
// the kernels unmodified), not the throughput path -- that is DeviceType::HIP, integration/device_hip/.
// Compiled against the reference's real headers and linked into oracle/_ref/libref_jetstream_hip.so by
// oracle/ref_jetstream_build.sh HIP=1; exercised by tests/test_gpu_reference_drives_library.py.
#pragma once

#include <any>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include <jetstream/logger.hh>
#include <jetstream/memory/tensor.hh>
#include <jetstream/types.hh>
#include <jetstream_hip.h>  // this repo's include/

namespace Jetstream::Modules::Mi355x {

inline uint8_t DtypeCode(const DataType dtype) {
    switch (dtype) {
        case DataType::F32: return JST_DTYPE_F32;
        case DataType::CF32: return JST_DTYPE_CF32;
        case DataType::F64: return JST_DTYPE_F64;
        case DataType::U8: return JST_DTYPE_U8;
        case DataType::I8: return JST_DTYPE_I8;
        case DataType::I16: return JST_DTYPE_I16;
        case DataType::CI8: return JST_DTYPE_CI8;
        case DataType::CI16: return JST_DTYPE_CI16;
        case DataType::CU8: return JST_DTYPE_CU8;
        default: return 0;
    }
}

inline std::string Number(const double v) {
    char text[64];
    std::snprintf(text, sizeof(text), "%.17g", v);
    return text;
}

// One library module behind a reference module: device mirrors of the inputs, the module, its output, a runtime of one.
class Bridge {
 public:
    ~Bridge() { (void)destroy(); }

    // inputs: (port, the reference's CPU tensor); config: "key=value" lines of the library module
    Result create(const std::string& tag, const char* type, const std::string& name, const std::vector<std::string>& config,
                  const std::vector<std::pair<std::string, const Tensor*>>& inputs, const char* outputPort) {
        this->tag = tag;
        std::vector<const char*> cfg, ports;
        for (const auto& line : config) cfg.push_back(line.c_str());
        for (const auto& [port, host] : inputs) {
            // A broadcast view (stride 0 along the broadcast axes: what MultiplyImpl::create makes of its narrower operand)
            // is mirrored as the dense tensor underneath, extent 1 on those axes -- the library's modules broadcast by
            // shape as the reference's do (core/multiply/module_impl.cc:10-84).
            std::vector<uint64_t> shape(host->shape().begin(), host->shape().end());
            uint64_t dense = 1;
            bool ok = DtypeCode(host->dtype()) != 0;
            for (Index axis = host->rank(); ok && axis-- > 0;) {
                if (host->stride(axis) == 0) shape[axis] = 1;
                else if (shape[axis] != 1 && host->stride(axis) != dense) ok = false;
                dense *= shape[axis];
            }
            if (!ok) {
                JST_ERROR("[{}] Input '{}' must be a dense (or broadcast) tensor of a sample type the device path takes.", tag, port);
                return Result::ERROR;
            }
            jst_tensor dev{};
            if (jst_tensor_create(JST_DEVICE_HIP, DtypeCode(host->dtype()), (uint32_t)shape.size(), shape.data(), &dev) != JST_SUCCESS)
                return fail("jst_tensor_create");
            for (const char* key : {"sampleAxis", "batchAxis", "channelAxis"}) {
                if (!host->hasAttribute(key)) continue;
                const std::any a = host->attribute(key);
                if (const auto* index = std::any_cast<Index>(&a)) (void)jst_tensor_set_attribute_u64(dev, key, (uint64_t)*index);
            }
            devIn.push_back(dev);
            hostIn.push_back(host);
            hostBytes.push_back(dense * host->elementSize());
            ports.push_back(port.c_str());
        }
        if (jst_module_create(type, JST_DEVICE_HIP, "generic", name.c_str(), cfg.data(), (uint32_t)cfg.size(), ports.data(),
                              devIn.data(), (uint32_t)devIn.size(), &module) != JST_SUCCESS)
            return fail("jst_module_create");
        if (outputPort && jst_module_output(module, outputPort, &devOut) != JST_SUCCESS) return fail("jst_module_output");
        if (jst_runtime_create(&module, 1, 0, &runtime) != JST_SUCCESS) return fail("jst_runtime_create");
        return Result::SUCCESS;
    }

    jst_module handle() const { return module; }

    Result upload() {
        for (size_t i = 0; i < devIn.size(); ++i)
            if (jst_tensor_copy_from_host(devIn[i], hostIn[i]->data(), hostBytes[i]) != JST_SUCCESS)
                return fail("jst_tensor_copy_from_host");
        return Result::SUCCESS;
    }
    Result compute(const bool sync) {
        const jst_result r = jst_runtime_compute(runtime, 1, sync ? 1 : 0);
        return r == JST_SUCCESS ? Result::SUCCESS : fail("jst_runtime_compute");
    }
    Result download(jst_tensor from, Tensor& host) {
        if (jst_tensor_copy_to_host(from, host.data(), host.sizeBytes()) != JST_SUCCESS) return fail("jst_tensor_copy_to_host");
        return Result::SUCCESS;
    }
    // the whole cycle: inputs up, one compute, the output down into the reference's tensor
    Result run(Tensor& hostOut) {
        JST_CHECK(upload());
        JST_CHECK(compute(true));
        return download(devOut, hostOut);
    }
    Result synchronize() { return jst_runtime_synchronize(runtime) == JST_SUCCESS ? Result::SUCCESS : fail("jst_runtime_synchronize"); }

    Result destroy() {
        if (runtime) (void)jst_runtime_destroy(runtime);
        if (module) (void)jst_module_destroy(module);
        for (auto& t : devIn) (void)jst_tensor_destroy(t);
        if (devOut) (void)jst_tensor_destroy(devOut);
        runtime = {};
        module = {};
        devOut = {};
        devIn.clear();
        hostIn.clear();
        hostBytes.clear();
        return Result::SUCCESS;
    }

 private:
    Result fail(const char* what) {
        JST_ERROR("[{}] {}: {}", tag, what, jst_last_error());
        return Result::ERROR;
    }

    std::string tag;
    std::vector<jst_tensor> devIn;
    std::vector<const Tensor*> hostIn;
    std::vector<size_t> hostBytes;
    jst_tensor devOut{};
    jst_module module{};
    jst_runtime runtime{};
};

}  // namespace Jetstream::Modules::Mi355x
