// integration/device_hip/runtime_native_hip_impl.cc -- REFERENCE-SIDE code: would live at src/runtime/native/hip/impl.cc
// (INTEGRATION.md section 3).  The runtime of one flowgraph segment on the HIP device, with the CPU runtime's contract
// (src/runtime/native/cpu/impl.cc:98-148: order, SKIP propagation, YIELD / TIMEOUT end the cycle quietly, anything else
// fails it) and the CUDA runtime's shape (src/runtime/native/cuda/impl.cc:185-272: ONE stream per segment, modules only
// enqueue, one synchronise per cycle).  Runtime::Runtime (src/runtime/runtime.cc:17-61) reaches the factory through
// core_hip_device.patch.  A segment made only of library modules can hand the whole cycle to jst_runtime_* instead
// (graph capture, fusion, cycle batching): that is what cyberether_amd/jetstream.py's Runtime does stand-alone.
#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE

#include <chrono>

#include <hip/hip_runtime_api.h>

#include <jetstream/detail/runtime_impl.hh>
#include <jetstream/runtime_context_native_hip.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/module.hh>

namespace Jetstream {

struct NativeHipRuntime : public Runtime::Impl {
 public:
    virtual ~NativeHipRuntime() { (void)destroy(); }

    Result create(const Runtime::Modules& modules) override {
        modulesMap.clear();
        moduleNames.clear();
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
            JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Failed to create the stream of runtime '{}'.", name);
            return Result::ERROR;
        }
        for (const auto& [moduleName, module] : modules) {
            if (module->device() != DeviceType::HIP || module->runtime() != RuntimeType::NATIVE) {
                JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Module '{}' is incompatible (DeviceType::{}, RuntimeType::{}).",
                          moduleName, module->device(), module->runtime());
                (void)destroy();
                return Result::ERROR;
            }
            const auto result = context(module)->computeInitialize();
            if (result != Result::SUCCESS && result != Result::RELOAD) {
                (void)context(module)->computeDeinitialize();
                (void)destroy();
                return result;
            }
            Module::Timing timing;
            timing.runtime = name;
            timing.device = GetDevicePrettyName(device);
            timing.backend = GetRuntimePrettyName(backend);
            module->timing(timing);
            modulesMap[moduleName] = module;
            moduleNames.push_back(moduleName);
        }
        return Result::SUCCESS;
    }

    Result destroy() override {
        Result result = Result::SUCCESS;
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto it = moduleNames.rbegin(); it != moduleNames.rend(); ++it) {
            const auto r = context(modulesMap.at(*it))->computeDeinitialize();
            if (result == Result::SUCCESS && r != Result::SUCCESS && r != Result::RELOAD) result = r;
        }
        modulesMap.clear();
        moduleNames.clear();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        return result;
    }

    Result compute(const std::vector<std::string>& modules, std::unordered_set<std::string>& skippedModules,
                   std::unordered_set<std::string>& failedModules) override {
        const auto& targets = modules.empty() ? moduleNames : modules;
        const auto start = std::chrono::steady_clock::now();
        std::vector<std::shared_ptr<Module>> submitted;
        for (const auto& moduleName : targets) {
            if (!modulesMap.contains(moduleName)) {
                failedModules.insert(moduleName);
                JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Context for module '{}' not found.", moduleName);
                return Result::ERROR;
            }
            const auto& module = modulesMap.at(moduleName);
            if (skippedModules.contains(moduleName) || hasSkippedInputs(module, skippedModules)) {
                skippedModules.insert(moduleName);
                continue;
            }
            const auto result = context(module)->computeSubmit(stream);
            if (result == Result::YIELD || result == Result::TIMEOUT) {
                (void)hipStreamSynchronize(stream);
                return result;
            }
            if (result != Result::SUCCESS && result != Result::RELOAD && result != Result::SKIP) {
                (void)hipStreamSynchronize(stream);
                failedModules.insert(moduleName);
                return result;
            }
            if (result == Result::SKIP) skippedModules.insert(moduleName);
            else submitted.push_back(module);
        }
        // one synchronise per cycle: the outputs are complete when compute() returns (what the scheduler and the present
        // thread assume of every runtime)
        if (hipStreamSynchronize(stream) != hipSuccess) {
            JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] The stream of runtime '{}' failed.", name);
            return Result::ERROR;
        }
        const F32 elapsedMs = std::chrono::duration<F32, std::milli>(std::chrono::steady_clock::now() - start).count();
        for (const auto& module : submitted) {  // the cycle's time, shared evenly: per-module device times need events
            auto timing = module->timing();
            timing.cycles += 1;
            timing.computeTime += elapsedMs / static_cast<F32>(submitted.size());
            module->timing(timing);
        }
        return Result::SUCCESS;
    }

 private:
    static std::shared_ptr<NativeHipRuntimeContext> context(const std::shared_ptr<Module>& module) {
        return std::dynamic_pointer_cast<NativeHipRuntimeContext>(module->context()->runtime());
    }

    hipStream_t stream = nullptr;
    Runtime::Modules modulesMap;
    std::vector<std::string> moduleNames;
};

std::shared_ptr<Runtime::Impl> NativeHipRuntimeFactory() { return std::make_shared<NativeHipRuntime>(); }

}  // namespace Jetstream

#endif  // JETSTREAM_BACKEND_HIP_AVAILABLE
