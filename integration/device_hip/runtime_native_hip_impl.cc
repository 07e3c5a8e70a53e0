// integration/device_hip/runtime_native_hip_impl.cc -- REFERENCE-SIDE code: would live at src/runtime/native/hip/impl.cc
// (INTEGRATION.md section 3).  The runtime of one flowgraph segment on the HIP device, with the CPU runtime's contract
// (src/runtime/native/cpu/impl.cc:98-148: order, SKIP propagation, YIELD / TIMEOUT end the cycle quietly, anything else
// fails it) and the CUDA runtime's shape (src/runtime/native/cuda/impl.cc:185-272: ONE stream per segment, modules only
// enqueue, one synchronise per cycle).  Runtime::Runtime (src/runtime/runtime.cc:17-61) reaches the factory through
// core_hip_device.patch.
//
// SEGMENT HAND-OFF.  When every module of the segment stands on a library module (NativeHipRuntimeContext::libraryModule),
// create() builds ONE jst_runtime over their handles (JST_RUNTIME_GRAPH | JST_RUNTIME_FUSE) and compute() forwards the
// cycle to it: the library orders the same modules by the same data-flow edges (hip_library_module.hh: consumers hold
// their producers' library tensors), settles the static ones as src/scheduler_synchronous.cc:534-546 does, captures the
// steady-state cycle into a hipGraph and submits fused units -- spectrum_fused(multiply+fft+amplitude+range)+indices is one
// kernel.  The scheduler still decides WHEN a cycle runs and which modules it names; a mixed segment (a module with kernels
// of its own) keeps the module-by-module loop.  What is visible when compute() returns is bit-identical either way, except
// that intermediates a fused unit never materialises (the product, the spectrum, the amplitude) are not written.
//
// DEFERRED CYCLES (opt-in, jetstream_hip_runtime_configure(.., deferCycles >= 1)): for a RESIDENT source (a ring whose next
// cycles' inputs are already in HBM: file replay, benchmarks) compute() only counts the cycle; every `deferCycles` cycles
// the counted cycles are ENQUEUED as ONE jst_runtime_compute(n) -- with JST_RUNTIME_BATCH: one launch per unit for the whole
// span -- and compute() returns without waiting, so the scheduler's bookkeeping of the next span overlaps the device's work
// on this one.  jetstream_hip_runtime_flush() (and destroy()) is the synchronisation point: it enqueues what is still
// counted, copies the outputs the library turned into rings (latest slot, device to device, on the runtime's stream) into
// the reference's tensors and waits -- a reader behind flush() sees what n synchronous cycles would have left.
// deferCycles = 1: every cycle is enqueued at once, still without a wait.  Default (0): every cycle is synchronous.
#ifdef JETSTREAM_BACKEND_HIP_AVAILABLE

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <jetstream/detail/runtime_impl.hh>
#include <jetstream/runtime_context_native_hip.hh>
#include <jetstream/scheduler_context.hh>
#include <jetstream/module_context.hh>
#include <jetstream/module.hh>
#include <jetstream_hip.h>  // this repo's include/

namespace Jetstream {

struct NativeHipRuntime;

namespace {
// 0: module by module; 1: hand library segments over -- cycle-batched SPANS (deferCycles > 1) replayed from a hipGraph, single cycles
// launched directly: on this ROCm a hipGraphLaunch of the cycle's two or three kernel nodes costs more host time than their direct
// launches (per 1024 x 4096 cycle, tools/reference_driven_bench.py: synchronous 41.5 us replayed against 34.1 direct, enqueued
// without a wait 27.9 against 22.4; spans: equal); 2: direct launches always; 3: hipGraph always
std::atomic<int> gHandOff{1};
std::atomic<uint64_t> gDeferCycles{0};
std::mutex gLiveMutex;
std::set<NativeHipRuntime*> gLive;
}  // namespace

struct NativeHipRuntime : public Runtime::Impl {
 public:
    virtual ~NativeHipRuntime() { (void)destroy(); }

    Result create(const Runtime::Modules& modules) override {
        modulesMap.clear();
        moduleNames.clear();
        std::vector<jst_module> handles;
        bool everyModuleIsLibrary = gHandOff.load() != 0 && !modules.empty();
        for (const auto& [moduleName, module] : modules) {
            if (module->device() != DeviceType::HIP || module->runtime() != RuntimeType::NATIVE || !context(module)) {
                JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Module '{}' is incompatible (DeviceType::{}, RuntimeType::{}).",
                          moduleName, module->device(), module->runtime());
                return Result::ERROR;
            }
            void* handle = context(module)->libraryModule();
            if (handle) handles.push_back(static_cast<jst_module>(handle));
            else everyModuleIsLibrary = false;
        }
        if (everyModuleIsLibrary) {
            deferCycles = gDeferCycles.load();
            const int mode = gHandOff.load();
            const bool graph = mode == 3 || (mode == 1 && deferCycles > 1);
            const uint32_t flags = (graph ? (uint32_t)JST_RUNTIME_GRAPH : 0u) | JST_RUNTIME_FUSE | (deferCycles > 1 ? (uint32_t)JST_RUNTIME_BATCH : 0u);
            if (jst_runtime_create(handles.data(), (uint32_t)handles.size(), flags, &library) != JST_SUCCESS) {
                JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Runtime '{}': the library refused the segment: {}", name, jst_last_error());
                library = {};
                return Result::ERROR;
            }
            batchedSpans = deferCycles > 1 && jst_runtime_batched(library);   // else: spans of per-cycle launches
        } else if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) {
            JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Failed to create the stream of runtime '{}'.", name);
            return Result::ERROR;
        }
        for (const auto& [moduleName, module] : modules) {
            if (!library) {
                const auto result = context(module)->computeInitialize();
                if (result != Result::SUCCESS && result != Result::RELOAD) {
                    (void)context(module)->computeDeinitialize();
                    (void)destroy();
                    return result;
                }
            }
            Module::Timing timing;
            timing.runtime = name;
            timing.device = GetDevicePrettyName(device);
            timing.backend = GetRuntimePrettyName(backend);
            module->timing(timing);
            modulesMap[moduleName] = module;
            moduleNames.push_back(moduleName);
        }
        std::lock_guard<std::mutex> lock(gLiveMutex);
        gLive.insert(this);
        return Result::SUCCESS;
    }

    Result destroy() override {
        {
            std::lock_guard<std::mutex> lock(gLiveMutex);
            gLive.erase(this);
        }
        Result result = Result::SUCCESS;
        if (library) {
            result = flush();
            (void)jst_runtime_destroy(library);  // deinitialises its modules in reverse order
            library = {};
        } else {
            if (stream) (void)hipStreamSynchronize(stream);
            for (auto it = moduleNames.rbegin(); it != moduleNames.rend(); ++it) {
                const auto r = context(modulesMap.at(*it))->computeDeinitialize();
                if (result == Result::SUCCESS && r != Result::SUCCESS && r != Result::RELOAD) result = r;
            }
        }
        modulesMap.clear();
        moduleNames.clear();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        pending = 0;
        return result;
    }

    Result compute(const std::vector<std::string>& modules, std::unordered_set<std::string>& skippedModules,
                   std::unordered_set<std::string>& failedModules) override {
        const auto& targets = modules.empty() ? moduleNames : modules;
        for (const auto& moduleName : targets) {
            if (!modulesMap.contains(moduleName)) {
                failedModules.insert(moduleName);
                JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Context for module '{}' not found.", moduleName);
                return Result::ERROR;
            }
        }
        return library ? computeHandedOff(targets, skippedModules, failedModules) : computeModules(targets, skippedModules, failedModules);
    }

    // everything counted so far runs now; on return the reference's tensors hold the latest cycle
    Result flush() {
        if (!library) return Result::SUCCESS;
        jst_result r = launchPending();
        if (r == JST_SUCCESS && unpublished) {
            r = publishLatest();
            unpublished = false;
        }
        if (r == JST_SUCCESS) r = jst_runtime_synchronize(library);
        if (r != JST_SUCCESS) {
            JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Runtime '{}': deferred cycles failed: {}", name, jst_last_error());
            return static_cast<Result>(r);
        }
        return Result::SUCCESS;
    }

    std::string units() const {
        if (!library) return {};
        std::string text(4096, '\0');
        (void)jst_runtime_units(library, text.data(), text.size());
        text.resize(std::strlen(text.c_str()));
        return text;
    }
    bool batched() const { return library && batchedSpans; }

 private:
    static std::shared_ptr<NativeHipRuntimeContext> context(const std::shared_ptr<Module>& module) {
        return std::dynamic_pointer_cast<NativeHipRuntimeContext>(module->context()->runtime());
    }

    // the counted cycles go to the device as ONE span -- not waited for: the scheduler counts the next span meanwhile
    jst_result launchPending() {
        if (pending == 0) return JST_SUCCESS;
        const uint64_t cycles = pending;
        pending = 0;
        unpublished = true;
        return jst_runtime_compute(library, cycles, 0);
    }

    jst_result publishLatest() {
        void* libraryStream = jst_runtime_stream(library);
        for (const auto& moduleName : moduleNames) {
            const Result r = context(modulesMap.at(moduleName))->publishOutputs(libraryStream);
            if (r != Result::SUCCESS) return JST_ERROR_;
        }
        return JST_SUCCESS;
    }

    Result computeHandedOff(const std::vector<std::string>& targets, std::unordered_set<std::string>& skippedModules,
                            std::unordered_set<std::string>& failedModules) {
        // The library runs the segment as a whole.  A cycle in which the scheduler (throttling) or an upstream segment
        // (SKIP) holds back one of its modules is skipped as a whole: every dynamic module of a fused chain depends on
        // the segment's sources, and the static ones settle in the next cycle that runs.
        for (const auto& moduleName : targets) {
            if (skippedModules.contains(moduleName) || hasSkippedInputs(modulesMap.at(moduleName), skippedModules)) {
                for (const auto& n : targets) skippedModules.insert(n);
                return Result::SUCCESS;
            }
        }
        const auto start = std::chrono::steady_clock::now();
        if (deferCycles >= 1 && settled) {
            if (++pending >= deferCycles && launchPending() != JST_SUCCESS) {
                JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Runtime '{}': {}", name, jst_last_error());
                for (const auto& n : targets) failedModules.insert(n);
                return Result::ERROR;
            }
        } else {
            const jst_result r = jst_runtime_compute(library, 1, 1);
            if (r == JST_YIELD || r == JST_TIMEOUT || r == JST_SKIP) {
                if (r == JST_SKIP) for (const auto& n : targets) skippedModules.insert(n);
                return r == JST_SKIP ? Result::SUCCESS : static_cast<Result>(r);
            }
            if (r != JST_SUCCESS && r != JST_RELOAD) {
                JST_ERROR("[RUNTIME_IMPL_NATIVE_HIP] Runtime '{}': {}", name, jst_last_error());
                for (const auto& n : targets) failedModules.insert(n);
                return static_cast<Result>(r);
            }
            settled = true;  // the first cycle (eager: static modules run, tables are uploaded) is never deferred
        }
        const F32 elapsedMs = std::chrono::duration<F32, std::milli>(std::chrono::steady_clock::now() - start).count();
        for (const auto& moduleName : targets) {
            const auto& module = modulesMap.at(moduleName);
            auto timing = module->timing();
            timing.cycles += 1;
            timing.computeTime += elapsedMs / static_cast<F32>(targets.size());
            module->timing(timing);
        }
        return Result::SUCCESS;
    }

    Result computeModules(const std::vector<std::string>& targets, std::unordered_set<std::string>& skippedModules,
                          std::unordered_set<std::string>& failedModules) {
        const auto start = std::chrono::steady_clock::now();
        std::vector<std::shared_ptr<Module>> submitted;
        for (const auto& moduleName : targets) {
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

    hipStream_t stream = nullptr;
    jst_runtime library{};
    uint64_t deferCycles = 0, pending = 0;
    bool settled = false, unpublished = false, batchedSpans = false;
    Runtime::Modules modulesMap;
    std::vector<std::string> moduleNames;
};

std::shared_ptr<Runtime::Impl> NativeHipRuntimeFactory() { return std::make_shared<NativeHipRuntime>(); }

}  // namespace Jetstream

// Process-wide knobs of the HIP runtimes created from here on (a host application's settings page, a benchmark): handOff = 0
// keeps every segment module by module, 1 (the default) hands library segments over (hipGraph replay for cycle-batched spans, direct
// launches of the fused units for single cycles), 2 / 3 force direct launches / the hipGraph; deferCycles > 1 turns on deferred cycles (see the top of this file).
extern "C" void jetstream_hip_runtime_configure(int handOff, uint64_t deferCycles) {
    Jetstream::gHandOff.store(handOff);
    Jetstream::gDeferCycles.store(deferCycles);
}
// Runs what the live HIP runtimes still hold back and waits for it; returns the number of runtimes that failed.
extern "C" int jetstream_hip_runtime_flush(void) {
    std::lock_guard<std::mutex> lock(Jetstream::gLiveMutex);
    int failed = 0;
    for (auto* runtime : Jetstream::gLive)
        if (runtime->flush() != Jetstream::Result::SUCCESS) ++failed;
    return failed;
}
// "unit\nunit\n.." of every live handed-off runtime (what the library fused), '|' between runtimes; returns the length needed
extern "C" size_t jetstream_hip_runtime_units(char* buffer, size_t capacity) {
    std::lock_guard<std::mutex> lock(Jetstream::gLiveMutex);
    std::string all;
    for (auto* runtime : Jetstream::gLive) {
        if (!all.empty()) all += "|";
        all += runtime->units();
        if (runtime->batched()) all += "[batched]";
    }
    if (buffer && capacity) {
        const size_t n = all.size() < capacity - 1 ? all.size() : capacity - 1;
        std::memcpy(buffer, all.data(), n);
        buffer[n] = '\0';
    }
    return all.size();
}

#endif  // JETSTREAM_BACKEND_HIP_AVAILABLE
