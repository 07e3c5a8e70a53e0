// module.cc -- Module framework entry, Registry and the NativeHip Runtime (see module.hh).
#include "module.hh"

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "../modules/modules.hh"
#include "switches.hh"

namespace jst {

// ---- config helpers ----------------------------------------------------------------------------
bool ConfigBool(const Config& c, const std::string& key, bool fallback, bool* ok) {
    if (ok) *ok = true;
    auto it = c.find(key);
    if (it == c.end()) return fallback;
    std::string v = it->second;
    std::transform(v.begin(), v.end(), v.begin(), [](unsigned char ch) { return std::tolower(ch); });
    if (v == "true" || v == "1" || v == "yes" || v == "on") return true;
    if (v == "false" || v == "0" || v == "no" || v == "off") return false;
    if (ok) *ok = false;
    return fallback;
}
U64 ConfigU64(const Config& c, const std::string& key, U64 fallback, bool* ok) {
    if (ok) *ok = true;
    auto it = c.find(key);
    if (it == c.end()) return fallback;
    char* end = nullptr;
    const std::string& v = it->second;
    if (v.empty() || v[0] == '-') {
        if (ok) *ok = false;
        return fallback;
    }
    const unsigned long long r = std::strtoull(v.c_str(), &end, 10);
    if (end == v.c_str() || *end != '\0') {
        if (ok) *ok = false;
        return fallback;
    }
    return r;
}
F64 ConfigF64(const Config& c, const std::string& key, F64 fallback, bool* ok) {
    if (ok) *ok = true;
    auto it = c.find(key);
    if (it == c.end()) return fallback;
    char* end = nullptr;
    const double r = std::strtod(it->second.c_str(), &end);
    if (end == it->second.c_str() || *end != '\0') {
        if (ok) *ok = false;
        return fallback;
    }
    return r;
}
std::string ConfigStr(const Config& c, const std::string& key, const std::string& fallback) {
    auto it = c.find(key);
    return it == c.end() ? fallback : it->second;
}

// ---- Module ------------------------------------------------------------------------------------
Result Module::construct(const std::string& name, const Config& config,
                         const std::map<std::string, Tensor>& inputs) {
    name_ = name;
    config_ = config;
    inputs_ = inputs;
    outputs_.clear();
    input_ports_.clear();
    output_ports_.clear();
    created_ = false;

    JST_CHECK(validate());
    JST_CHECK(define());

    const bool discontiguous = (taint_ & DISCONTIGUOUS) != 0;
    const bool cross_device = (taint_ & CROSS_DEVICE) != 0;

    for (const auto& port : input_ports_) {  // src/module.cc:124-129
        if (!inputs_.count(port)) {
            JST_ERROR("[MODULE] Module '%s' requested missing input '%s'.", name_.c_str(),
                      port.c_str());
            return Result::ERROR;
        }
    }
    for (const auto& [port, tensor] : inputs_) {  // src/module.cc:133-155
        if (tensor.device() != device() && !cross_device) {
            JST_ERROR("[MODULE] Input tensor device ('%s', DeviceType::%s) doesn't match the module "
                      "device ('%s', DeviceType::%s).",
                      port.c_str(), DeviceName(tensor.device()), name_.c_str(), DeviceName(device()));
            return Result::ERROR;
        }
        if (!tensor.validShape()) {
            JST_ERROR("[MODULE] Input tensor ('%s') is invalid.", port.c_str());
            return Result::ERROR;
        }
        if (tensor.size() == 0) {
            JST_ERROR("[MODULE] Module ('%s') input tensor ('%s') size is zero.", name_.c_str(),
                      port.c_str());
            return Result::ERROR;
        }
        if (!tensor.contiguous() && !discontiguous) {
            JST_ERROR("[MODULE] Contiguous tensor expected for module ('%s') input tensor ('%s').",
                      name_.c_str(), port.c_str());
            return Result::ERROR;
        }
    }

    JST_CHECK(create());

    for (const auto& port : output_ports_) {
        if (!outputs_.count(port)) {
            JST_ERROR("[MODULE] Module '%s' did not produce output '%s'.", name_.c_str(),
                      port.c_str());
            return Result::ERROR;
        }
    }
    created_ = true;
    return Result::SUCCESS;
}

Result Module::teardown() {
    if (!created_) return Result::SUCCESS;
    created_ = false;
    return destroy();
}

Result Module::reconfigure(const Config& config, bool validateOnly) {
    if (!created_) return Result::RECREATE;  // DESTROYED / ERRORED (src/module.cc:234-236)
    Config candidate = config_;
    for (const auto& kv : config) candidate[kv.first] = kv.second;
    if (candidate == config_) return Result::SUCCESS;
    const Config staged = config_;
    auto restore = [&]() {  // validate() parses config_ into the members: put both back
        config_ = staged;
        (void)validate();
    };
    config_ = candidate;
    const Result v = validate();
    if (v != Result::SUCCESS && v != Result::RELOAD) {
        restore();
        return v;
    }
    if (validateOnly) {
        restore();
        return Result::SUCCESS;
    }
    const Result r = reconfigureImpl(staged);
    if (r != Result::SUCCESS && r != Result::RELOAD) {
        restore();
        return r;
    }
    ++config_generation_;
    return Result::SUCCESS;
}

// ---- Registry ----------------------------------------------------------------------------------
namespace {
std::string registry_key(const std::string& type, DeviceType d, RuntimeType r,
                         const std::string& provider) {
    return type + "|" + DeviceName(d) + "|" + (r == RuntimeType::NATIVE ? "native" : "?") + "|" +
           provider;
}
}  // namespace

Registry& Registry::instance() {
    static Registry r;
    return r;
}
void Registry::add(const std::string& type, DeviceType device, RuntimeType runtime,
                   const std::string& provider, ModuleFactory factory) {
    factories_[registry_key(type, device, runtime, provider)] = std::move(factory);
}
std::unique_ptr<Module> Registry::build(const std::string& type, DeviceType device,
                                        RuntimeType runtime, const std::string& provider) const {
    auto it = factories_.find(registry_key(type, device, runtime, provider));
    if (it == factories_.end()) {
        JST_ERROR("[REGISTRY] No module '%s' registered for (device=%s, runtime=native, "
                  "provider=%s).",
                  type.c_str(), DeviceName(device), provider.c_str());
        return nullptr;
    }
    auto m = it->second();
    m->setProvider(provider);
    return m;
}
std::vector<std::string> Registry::listAvailableModules(const std::string& type) const {
    std::vector<std::string> out;
    for (const auto& kv : factories_) {
        if (type.empty() || kv.first.compare(0, type.size() + 1, type + "|") == 0)
            out.push_back(kv.first);
    }
    return out;
}

// ---- Runtime -----------------------------------------------------------------------------------
Runtime::Runtime() = default;
Runtime::~Runtime() { (void)destroy(); }

Result Runtime::planOrder(const std::vector<Module*>& modules) {
    // Kahn's algorithm over "module B reads storage that module A produced"
    // (src/scheduler_synchronous.cc:574-696).  Ties keep insertion order; with FUSE a ready
    // consumer of the module placed last goes first, so that producer/consumer chains end up
    // adjacent and the fusion hooks (which look at neighbours) can see them.
    // The edges do not depend on the order the caller lists the modules in (a host framework may hand them over from a
    // hash map: integration/device_hip/runtime_native_hip_impl.cc).  An input handle remembers the module that published
    // it (ProducerAttribute travels with every clone / view of the handle): that is the exact edge, also through view
    // modules (reshape, a bypassing cast) whose output shares its input's storage.  A tensor without that attribute falls
    // back to storage identity: the module that WRITES the storage.
    const size_t n = modules.size();
    std::map<const Module*, size_t> index_of;
    for (size_t i = 0; i < n; ++i) index_of.emplace(modules[i], i);
    auto republishes = [&](size_t i, const void* storage) {
        for (const auto& kv : modules[i]->inputs())
            if (kv.second.storageId() == storage) return true;
        return false;
    };
    std::map<const void*, size_t> producer;  // storage -> the module that WRITES it (a re-publisher of a view is reached through its tag only:
                                             // as a storage-level producer it could sit behind its own reader and close a cycle)
    for (size_t i = 0; i < n; ++i)
        for (const auto& kv : modules[i]->outputs()) {
            const void* storage = kv.second.storageId();
            if (storage && !republishes(i, storage)) producer.emplace(storage, i);
        }
    std::vector<std::set<size_t>> deps(n);
    std::vector<std::vector<size_t>> users(n);
    std::vector<size_t> outside_inputs(n, 0);
    for (size_t i = 0; i < n; ++i)
        for (const auto& kv : modules[i]->inputs()) {
            size_t from = n;
            const AttrValue* tag = kv.second.attribute(ProducerAttribute);
            if (const U64* address = tag ? std::get_if<U64>(tag) : nullptr) {
                const auto it = index_of.find(reinterpret_cast<const Module*>(static_cast<uintptr_t>(*address)));
                if (it != index_of.end()) from = it->second;
            }
            if (from == n) {
                const auto it = producer.find(kv.second.storageId());
                if (it != producer.end()) from = it->second;
            }
            if (from == n) ++outside_inputs[i];  // a tensor nobody in this runtime produces: new data every cycle, as far as we know
            if (from != n && from != i && deps[i].insert(from).second) users[from].push_back(i);
        }
    std::vector<size_t> indeg(n);
    for (size_t i = 0; i < n; ++i) indeg[i] = deps[i].size();
    std::vector<bool> done(n, false), placed_static(n, false);
    ordered_.clear();
    size_t last = n;
    for (size_t placed = 0; placed < n; ++placed) {
        size_t pick = n;
        if ((flags_ & FUSE) && last != n)
            for (size_t u : users[last])
                if (!done[u] && indeg[u] == 0) {
                    pick = u;
                    break;
                }
        // then a ready module of a STATIC chain (a table: window -> invert -> reshape, filter_taps -> pad -> fft -> reshape):
        // tables first, whatever order the caller listed them in, so that they never sit between the members of a dynamic
        // chain the fusion hooks want adjacent (pad -> fft | taps ... | multiply -> fold)
        for (size_t i = 0; pick == n && i < n; ++i) {
            if (done[i] || indeg[i] != 0 || !(modules[i]->taint() & (STATIC_OUTPUT | STATELESS))) continue;
            bool settles = modules[i]->inputs().empty() ? (modules[i]->taint() & STATIC_OUTPUT) != 0 : outside_inputs[i] == 0;
            for (size_t d : deps[i]) settles = settles && placed_static[d];
            if (settles) pick = i, placed_static[i] = true;
        }
        for (size_t i = 0; pick == n && i < n; ++i)
            if (!done[i] && indeg[i] == 0) pick = i;
        if (pick == n) {
            JST_ERROR("[SCHEDULER] Module graph contains a cycle.");
            return Result::ERROR;
        }
        done[pick] = true;
        last = pick;
        ordered_.push_back(modules[pick]);
        for (size_t u : users[pick]) --indeg[u];
    }
    order_names_.clear();
    for (Module* m : ordered_) order_names_.push_back(m->name());
    return Result::SUCCESS;
}

void Runtime::computePeriod() {
    period_ = 1;
    for (Module* m : ordered_) {  // least common multiple of the modules' host-state periods
        U64 a = period_, b = m->cyclePeriod() ? m->cyclePeriod() : 1;
        while (b) { const U64 t = a % b; a = b; b = t; }
        period_ = period_ / a * (m->cyclePeriod() ? m->cyclePeriod() : 1);
    }
}

Result Runtime::flushUnits() {
    for (auto& u : units_)
        if (u.flush) JST_CHECK(u.flush(stream_));
    return Result::SUCCESS;
}

Result Runtime::planUnits() {
    units_.clear();
    for (Module* m : ordered_) m->resetPlan();  // a decision of an earlier runtime does not outlive it
    for (Module* m : ordered_)
        if (auto* spec = dynamic_cast<modules::Spectrogram*>(m)) spec->combined = spec->indexFed = false;
    for (Module* m : ordered_)
        if (auto* cast = dynamic_cast<modules::Cast*>(m)) cast->fusedIntoSpectrum = false;
    // Static settlement: a STATIC_OUTPUT module with no inputs, or a STATELESS/STATIC module
    // whose inputs all come from settled producers, runs once
    // (src/scheduler_synchronous.cc:534-546,670-693).
    std::set<const void*>& static_storage = static_storage_;
    static_storage.clear();
    std::vector<bool> is_static(ordered_.size(), false);
    for (size_t i = 0; i < ordered_.size(); ++i) {
        Module* m = ordered_[i];
        const bool eligible = (m->taint() & (STATIC_OUTPUT | STATELESS)) != 0;
        bool all_inputs_static = true;
        for (const auto& kv : m->inputs())
            if (!static_storage.count(kv.second.storageId())) all_inputs_static = false;
        const bool source_static = m->inputs().empty() && (m->taint() & STATIC_OUTPUT);
        if (eligible && (source_static || (!m->inputs().empty() && all_inputs_static))) {
            is_static[i] = true;
            for (const auto& kv : m->outputs()) static_storage.insert(kv.second.storageId());
        }
    }
    for (size_t i = 0; i < ordered_.size();) {
        Unit u;
        size_t consumed = 0;
        if ((flags_ & FUSE) && !is_static[i] && tryFuseSpectrum(i, u, consumed)) {
            if (u.name.size() > 8 && u.name.compare(u.name.size() - 8, 8, "(elided)") == 0) u.has_kernels = false;  // nothing reaches the stream
            units_.push_back(std::move(u));
            i += consumed;
            continue;
        }
        Module* m = ordered_[i];
        u.name = m->name();
        u.modules = {m};
        u.submit = [m](hipStream_t s) { return m->computeSubmit(s); };
        u.is_static = is_static[i];
        u.has_kernels = m->launchesKernels();
        units_.push_back(std::move(u));
        ++i;
    }
    computePeriod();  // a Spectrogram taken into the spectrum unit alternates between two output slots
    // Event pairs go around the units that launch kernels, plus ONE kernel-less dynamic unit
    // whose (empty) pair measures what the pair itself costs on this stream.
    calibration_unit_.clear();
    for (auto& u : units_) {
        u.timed = u.has_kernels;
        if (!u.has_kernels && !u.is_static && calibration_unit_.empty()) {
            calibration_unit_ = u.name;
            u.timed = true;
        }
    }
    unit_names_.clear();
    for (auto& u : units_) {
        unit_names_.push_back(u.name);
        u.span.name = u.name;
        if ((flags_ & TIMING) && u.timed) {
            u.span.begin.assign(period_, nullptr);
            u.span.end.assign(period_, nullptr);
            u.span.recorded.assign(period_, false);
            u.span.sampleCycles.assign(period_, 1);
            for (U64 s = 0; s < period_; ++s) {
                JST_HIP_CHECK(hipEventCreate(&u.span.begin[s]), "hipEventCreate");
                JST_HIP_CHECK(hipEventCreate(&u.span.end[s]), "hipEventCreate");
            }
        }
    }
    return planBranches();
}

// Which units may run side by side inside a captured cycle.  Unit i runs behind unit j < i when one of them writes storage
// the other touches (the members' declared inputs and outputs; views share their storage's id; a unit that launches nothing
// -- a view, a bypass cast, a host-side cursor -- and a settled static unit write nothing during a cycle).  Branch
// assignment: a unit continues the branch of its latest dependency that is still that branch's tail, else it opens a
// new branch (the first unit of all takes branch 0 = the runtime's stream); at most kMaxBranches, then it queues behind
// its latest dependency.
Result Runtime::planBranches() {
    // MEASURED (r05 experiments s): multi-fm.yml 154-177 us per cycle with 2-5 branches against 136 us as one chain -- this
    // ROCm's graph executor pays more for a cross-branch edge than the overlap of launch-floor kernels returns.  The plan
    // stays (JST_RUNTIME_MAX_BRANCHES=n opts in); the default is one chain.
    int kMaxBranches = 1;
    if (const int n = switch_value(SW_RUNTIME_MAX_BRANCHES)) kMaxBranches = n;
    for (auto& u : units_) {
        u.deps.clear();
        u.branch = 0;
    }
    if (flags_ & PIPELINE) return Result::SUCCESS;
    std::vector<std::set<const void*>> reads(units_.size()), writes(units_.size());
    for (size_t i = 0; i < units_.size(); ++i) {
        const Unit& u = units_[i];
        if (u.is_static || !u.has_kernels) continue;
        for (Module* m : u.modules) {
            for (const auto& kv : m->inputs()) reads[i].insert(kv.second.storageId());
            for (const auto& kv : m->outputs()) writes[i].insert(kv.second.storageId());
            m->planStorage(reads[i], writes[i]);
        }
    }
    auto meets = [](const std::set<const void*>& a, const std::set<const void*>& b) {
        for (const void* x : a)
            if (b.count(x)) return true;
        return false;
    };
    std::vector<long> tail;  // tail[b] = last unit on branch b
    for (size_t i = 0; i < units_.size(); ++i) {
        Unit& u = units_[i];
        if (u.is_static || !u.has_kernels) continue;
        for (size_t j = 0; j < i; ++j)
            if (meets(writes[j], reads[i]) || meets(writes[j], writes[i]) || meets(reads[j], writes[i])) u.deps.push_back(j);
        int pick = -1;
        for (size_t k = u.deps.size(); k-- > 0 && pick < 0;) {
            const int b = units_[u.deps[k]].branch;
            if (tail[(size_t)b] == (long)u.deps[k]) pick = b;
        }
        if (pick < 0) {
            if ((int)tail.size() < kMaxBranches) {
                pick = (int)tail.size();
                tail.push_back(-1);
            } else {
                pick = u.deps.empty() ? 0 : units_[u.deps.back()].branch;
            }
        }
        u.branch = pick;
        tail[(size_t)pick] = (long)i;
    }
    if (tail.size() < 2) {
        for (auto& u : units_) u.branch = 0;
        return Result::SUCCESS;
    }
    while (branch_streams_.size() < tail.size()) {
        hipStream_t s = stream_;
        if (!branch_streams_.empty()) JST_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
        branch_streams_.push_back(s);
        hipEvent_t e = nullptr;
        JST_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
        branch_join_.push_back(e);
    }
    if (!branch_fork_) JST_HIP_CHECK(hipEventCreateWithFlags(&branch_fork_, hipEventDisableTiming), "hipEventCreate");
    for (auto& u : units_)
        if (!u.done) JST_HIP_CHECK(hipEventCreateWithFlags(&u.done, hipEventDisableTiming), "hipEventCreate");
    return Result::SUCCESS;
}

// Recognise multiply(a = signal, b = static broadcast window) -> fft(forward, CF32) -> amplitude
// [-> range] laid out consecutively in the order, with no other consumer of the intermediates
// (the block wiring of src/domains/dsp/spectrum_engine/block_impl.cc:120-217).
bool Runtime::tryFuseSpectrum(size_t at, Unit& unit, size_t& consumed) {
    return modules::TryFuseSpectrum(ordered_, at, unit.name, unit.modules, unit.submit, consumed,
                                    (flags_ & COMBINE) != 0 && (flags_ & PIPELINE) == 0, &unit.flush,
                                    (flags_ & PIPELINE) == 0,
                                    (flags_ & BATCH) && (flags_ & GRAPH) && !(flags_ & (PIPELINE | COMBINE)) ? &unit.batch : nullptr,
                                    &static_storage_) ||
           modules::TryFuseFilter(ordered_, at, unit.name, unit.modules, unit.submit, consumed) ||
           modules::TryFuseMultiplyFft(ordered_, at, unit.name, unit.modules, unit.submit, consumed) ||
           modules::TryFuseAgcChain(ordered_, at, unit.name, unit.modules, unit.submit, consumed) ||
           modules::TryElideDuplicate(ordered_, at, unit.name, unit.modules, unit.submit, consumed) ||
           modules::TryFuseAmplitudeRange(ordered_, at, unit.name, unit.modules, unit.submit, consumed);
}

Result Runtime::create(const std::vector<Module*>& modules, U32 flags) {
    if (created_) JST_CHECK(destroy());
    flags_ = flags;
    span_graphs_.clear();  // destroy() released them; a runtime object never carries graphs across create()
    span_clock_ = 0;
    timed_once_ = false;
    last_timed_cycle_ = 0;
    for (Module* m : modules) {
        if (!m || !m->created()) {
            JST_ERROR("[RUNTIME] Every module must be created before Runtime::create.");
            return Result::ERROR;
        }
    }
    JST_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    // From here on a failure tears down what was built (stream, initialised modules, events): destroy() does
    // nothing for a runtime that never reached created_.
    size_t initialised = 0;
    auto fail = [&](Result r) {
        for (size_t i = initialised; i-- > 0;) (void)ordered_[i]->computeDeinitialize();
        created_ = true;  // let destroy() release units, graphs, streams
        std::vector<Module*> none;
        ordered_.swap(none);
        (void)destroy();
        return r;
    };
    {
        const Result r = planOrder(modules);
        if (r != Result::SUCCESS) return fail(r);
    }
    for (Module* m : ordered_) {
        const Result r = m->computeInitialize();
        if (r != Result::SUCCESS) {
            JST_ERROR("[RUNTIME] computeInitialize failed for module '%s': %s", m->name().c_str(),
                      last_error());
            return fail(r);
        }
        ++initialised;
    }
    computePeriod();
    {
        Result r = planUnits();
        if (r == Result::SUCCESS && (flags_ & PIPELINE) && (flags_ & GRAPH)) r = planPipeline();
        if (r == Result::SUCCESS) r = planBatch();
        if (r != Result::SUCCESS) return fail(r);
    }
    cycles_ = 0;
    created_ = true;
    for (Module* m : ordered_) {  // from here on the planned storage stays where it is (Tensor::rebind refuses)
        for (const auto& kv : m->inputs()) kv.second.runtimeBound(+1);
        for (const auto& kv : m->outputs()) kv.second.runtimeBound(+1);
    }
    return Result::SUCCESS;
}

// BATCH planning.  With the inputs of a whole ring period resident in HBM (a ring source that is not live), the cycles
// of a captured period need not be one launch per unit and cycle: the persistent fused spectrum kernel pays its ramp,
// cold start and tail once per LAUNCH (~4 of the ~17 us of a 1024-transform launch), so the unit runs the
// transforms of all `n` slots of a span as one launch into output rings of as many slots, and the index-fed Spectrogram
// walks the n index tensors in one launch with its state tile in registers.  What is visible after compute() is what
// the per-cycle submissions leave: every cycle's output sits in its ring slot (the handles show the latest), the
// Spectrogram's state went through the same n decays and hit updates.  Batched only when EVERY dynamic unit can do it:
// the one fused spectrum unit with span support, modules that are spanCapable(), and kernel-less modules (their host
// state moves through advanceHostState), and sinks of the batched unit (see below); anything else (a live source, two
// spectrum units, a dynamic module in front of the unit) leaves the runtime per cycle.  JST_RUNTIME_NO_BATCH=1 is the
// A/B switch.
Result Runtime::planBatch() {
    batched_ = false;
    if (!(flags_ & BATCH) || !(flags_ & GRAPH) || !(flags_ & FUSE) || (flags_ & (PIPELINE | COMBINE)) || period_ < 2 ||
        switch_value(SW_RUNTIME_NO_BATCH) != 0)
        return Result::SUCCESS;
    size_t fused = units_.size();
    for (size_t i = 0; i < units_.size(); ++i) {
        Unit& u = units_[i];
        if (u.is_static) continue;
        if (u.batch) {
            if (fused != units_.size()) return Result::SUCCESS;  // two spectrum units: per cycle
            fused = i;
            continue;
        }
        for (Module* m : u.modules) {
            if (!m->capturable()) return Result::SUCCESS;
            if (!m->launchesKernels() && m->cyclePeriod() != 1 && m->cyclePeriod() != period_) return Result::SUCCESS;
        }
    }
    if (fused == units_.size()) return Result::SUCCESS;
    SpanSupport& b = units_[fused].batch;
    if (!b.phase.valid() || b.phase.ringSlots() != period_) return Result::SUCCESS;
    // Kernel modules without a span form may still sit in a batched runtime when they are SINKS of the batched unit: a
    // surface (waterfall, lineplot, a second Spectrogram ...) with no outputs whose inputs are all views of the batched
    // unit's output or of the source ring.  Every cycle's data is in its ring slot when the span's big launches are
    // through, so such a module simply runs its n per-cycle submissions behind them, slot after slot (submitBatched).
    for (size_t i = 0; i < units_.size(); ++i) {
        Unit& u = units_[i];
        u.per_cycle_in_span = false;
        if (u.is_static || i == fused) continue;
        for (Module* m : u.modules) {
            if (!m->launchesKernels()) continue;
            // every kernel module of a batched runtime -- span form or sink -- reads what the spectrum unit (or the source
            // ring) leaves in the rings: it runs BEHIND that unit and every input of it shares storage with them.  Anything
            // else (a Lineplot of the static window ordered in front of the unit, say) keeps the runtime per cycle.
            if (i < fused) return Result::SUCCESS;
            if (m->spanCapable()) {
                for (const auto& kv : m->inputs()) {
                    bool from_ring = kv.second.storageId() == b.phase.storageId();
                    for (Module* prod : units_[fused].modules)
                        for (const auto& out : prod->outputs()) from_ring |= out.second.storageId() == kv.second.storageId();
                    if (!from_ring) return Result::SUCCESS;
                }
                continue;
            }
            bool sink = (m->taint() & SURFACE) != 0 && m->outputs().empty() && m->cyclePeriod() == 1 && u.modules.size() == 1;
            for (const auto& kv : m->inputs()) {
                bool from_ring = kv.second.storageId() == b.phase.storageId();
                for (Module* prod : units_[fused].modules)
                    for (const auto& out : prod->outputs()) from_ring |= out.second.storageId() == kv.second.storageId();
                sink &= from_ring;
            }
            if (!sink) return Result::SUCCESS;
            u.per_cycle_in_span = true;
        }
    }
    JST_CHECK(b.prepare(period_));
    batch_unit_ = fused;
    batched_ = true;
    return Result::SUCCESS;
}

// `n` consecutive cycles from the current phase, one submission per unit.  Kernel-less dynamic modules (the source)
// expose the span's FIRST cycle before the units that read them run (advanceHostState(1)) and move on to its last
// cycle afterwards, so the host side ends where n per-cycle submissions would have left it.
Result Runtime::submitBatched(U64 n, bool record_events) {
    U64 first = ~0ull;
    Result r = Result::SUCCESS;
    for (auto& u : units_) {
        if (u.is_static && u.settled) continue;
        const bool rec = record_events && !u.span.begin.empty();
        if (rec) JST_HIP_CHECK(hipEventRecord(u.span.begin[0], stream_), "hipEventRecord");
        if (u.batch) {
            first = u.batch.phase.ringSlot();
            r = u.batch.submit_span(stream_, first, n);
        } else if (u.per_cycle_in_span) {
            // a sink of the batched unit: its n per-cycle submissions, each looking at its cycle's slot of the rings
            SpanSupport& b = units_[batch_unit_].batch;
            if (first == ~0ull) {
                JST_ERROR("[RUNTIME] Batched span: '%s' runs before the spectrum unit that feeds it.", u.name.c_str());
                r = Result::ERROR;
            }
            const U64 ring = b.phase.ringSlots();
            for (U64 c = 0; c < n && r == Result::SUCCESS; ++c) {
                const U64 slot = (first + c) % ring;
                JST_CHECK(b.phase.ringSelect(slot));
                for (Tensor& t : b.rings) JST_CHECK(t.ringSelect(slot % t.ringSlots()));
                r = u.submit(stream_);
                if (r == Result::RELOAD) r = Result::SUCCESS;
            }
        } else {
            for (Module* m : u.modules) {
                if (!m->launchesKernels()) {
                    m->advanceHostState(1);
                } else if (first == ~0ull) {
                    JST_ERROR("[RUNTIME] Batched span: '%s' runs before the spectrum unit that feeds it.", u.name.c_str());
                    r = Result::ERROR;
                } else {
                    r = m->computeSubmitSpan(stream_, first, n);
                }
                if (r != Result::SUCCESS) break;
            }
        }
        if (r != Result::SUCCESS) {
            JST_ERROR("[RUNTIME] Batched submission failed in '%s' (%s): %s", u.name.c_str(), ResultName(r), last_error());
            return r;
        }
        if (rec) {
            JST_HIP_CHECK(hipEventRecord(u.span.end[0], stream_), "hipEventRecord");
            u.span.recorded[0] = true;
            u.span.sampleCycles[0] = n;
        }
    }
    for (auto& u : units_) {
        if ((u.is_static && u.settled) || u.batch) continue;
        for (Module* m : u.modules)
            if (!m->launchesKernels() && n > 1) m->advanceHostState(n - 1);
    }
    return Result::SUCCESS;
}

// BATCH: a reader of the promoted output rings sees the most recent cycle's slot (graph replays do not run the host
// side that selects it).
Result Runtime::showLatestSlot() {
    if (!batched_) return Result::SUCCESS;
    SpanSupport& b = units_[batch_unit_].batch;
    for (Tensor& t : b.rings) JST_CHECK(t.ringSelect(b.phase.ringSlot() % t.ringSlots()));
    return Result::SUCCESS;
}

// PIPELINE planning: SURFACE units (spectrogram, waterfall, lineplot: pure consumers with their
// own state) move to a second stream.  hipGraph branches inside ONE graph are executed one after
// the other on this ROCm, but two graphs launched on two streams run on two hardware queues and do
// overlap, so each lane gets its own graph of period_ cycles: the surface graph of period k runs
// beside the producer graph of period k+1.  Every tensor the surfaces read from a dynamic producer
// becomes a ring of 2*period_ slots (period k uses half k%2); the producers of period k+2 wait for
// the surfaces of period k before they overwrite that half.
Result Runtime::planPipeline() {
    std::vector<size_t> surfaces;
    for (size_t i = 0; i < units_.size(); ++i) {
        bool all_surface = !units_[i].is_static;
        for (Module* mod : units_[i].modules)
            all_surface &= (mod->taint() & SURFACE) != 0 && mod->outputs().empty();
        if (all_surface) surfaces.push_back(i);
    }
    if (surfaces.empty()) return Result::SUCCESS;
    std::vector<Tensor> ring;
    std::set<const void*> seen;
    for (size_t si : surfaces)
        for (Module* mod : units_[si].modules)
            for (const auto& kv : mod->inputs()) {
                for (size_t ui = 0; ui < units_.size(); ++ui) {
                    if (units_[ui].is_static || ui == si) continue;
                    for (Module* prod : units_[ui].modules)
                        for (const auto& out : prod->outputs())
                            if (out.second.storageId() == kv.second.storageId() &&
                                prod->launchesKernels() && seen.insert(kv.second.storageId()).second)
                                ring.push_back(kv.second);
                }
            }
    if (ring.empty()) return Result::SUCCESS;
    for (Tensor& t : ring)
        if (t.ringSlots() != 1) return Result::SUCCESS;  // already a ring (a source): leave as is
    for (Tensor& t : ring) JST_CHECK(t.promoteToRing(2 * period_));
    pipelined_ = ring;
    for (size_t si : surfaces) units_[si].lane = 1;
    JST_HIP_CHECK(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking), "hipStreamCreate");
    for (auto& lane : lane_done_)
        for (auto& e : lane) JST_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    return Result::SUCCESS;
}

// One lane's units for period_ consecutive cycles on ring half `half`, as a graph of its own.
Result Runtime::captureLane(int lane, int half, bool timing) {
    hipStream_t s = lane == 0 ? stream_ : side_stream_;
    JST_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
    Result r = Result::SUCCESS;
    for (U64 c = 0; c < period_ && r == Result::SUCCESS; ++c) {
        for (Tensor& t : pipelined_) JST_CHECK(t.ringSelect((U64)half * period_ + c));
        for (auto& u : units_) {
            if (u.lane != lane || (u.is_static && u.settled)) continue;
            // No event records: one issued under capture is a dependency edge, nothing is recorded when the graph
            // replays.  A PIPELINE runtime's unit timers are fed by its eager cycles only (settling, non-period tails).
            const bool rec = false;
            (void)timing;
            if (rec) JST_HIP_CHECK(hipEventRecord(u.span.begin[c], s), "hipEventRecord");
            r = u.submit(s);
            if (r != Result::SUCCESS && r != Result::RELOAD) {
                JST_ERROR("[RUNTIME] computeSubmit failed in '%s' (%s): %s", u.name.c_str(), ResultName(r),
                          last_error());
                break;
            }
            r = Result::SUCCESS;
            if (rec) JST_HIP_CHECK(hipEventRecord(u.span.end[c], s), "hipEventRecord");
        }
    }
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (r != Result::SUCCESS) {
        if (g) (void)hipGraphDestroy(g);
        return r;
    }
    JST_HIP_CHECK(e, "hipStreamEndCapture");
    lane_graph_[lane][half] = g;
    JST_HIP_CHECK(hipGraphInstantiate(&lane_exec_[lane][half], g, nullptr, nullptr, 0), "hipGraphInstantiate");
    return Result::SUCCESS;
}

// Both lanes idle, readers see the most recent cycle's slot.
Result Runtime::joinLanes() {
    JST_HIP_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    if (side_stream_) JST_HIP_CHECK(hipStreamSynchronize(side_stream_), "hipStreamSynchronize");
    side_pending_ = false;
    for (Tensor& t : pipelined_) JST_CHECK(t.ringSelect(last_slot_));
    return Result::SUCCESS;
}

Result Runtime::destroy() {
    if (!created_) return Result::SUCCESS;
    created_ = false;
    // Every captured graph -- period, lane and the span cache -- goes while the streams still exist (dropGraphs joins
    // both lanes first): a span graph that outlived destroy() held the previous modules' kernel arguments and a
    // create() on the same object would have replayed it for a matching (phase, length).
    if (stream_) (void)dropGraphs();
    (void)hipStreamSynchronize(stream_);
    if (side_stream_) {
        (void)hipStreamSynchronize(side_stream_);
        (void)hipStreamDestroy(side_stream_);
        side_stream_ = nullptr;
    }
    for (int lane = 0; lane < 2; ++lane)
        for (int half = 0; half < 2; ++half) {
            if (lane_exec_[lane][half]) (void)hipGraphExecDestroy(lane_exec_[lane][half]);
            if (lane_graph_[lane][half]) (void)hipGraphDestroy(lane_graph_[lane][half]);
            if (lane_done_[lane][half]) (void)hipEventDestroy(lane_done_[lane][half]);
            lane_exec_[lane][half] = nullptr;
            lane_graph_[lane][half] = nullptr;
            lane_done_[lane][half] = nullptr;
        }
    pipelined_.clear();
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    graph_exec_ = nullptr;
    graph_ = nullptr;
    for (auto& u : units_) {
        for (hipEvent_t e : u.span.begin) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : u.span.end) if (e) (void)hipEventDestroy(e);
        if (u.done) (void)hipEventDestroy(u.done);
    }
    units_.clear();
    for (size_t b = 1; b < branch_streams_.size(); ++b) {
        (void)hipStreamSynchronize(branch_streams_[b]);
        (void)hipStreamDestroy(branch_streams_[b]);
    }
    branch_streams_.clear();
    for (hipEvent_t e : branch_join_) if (e) (void)hipEventDestroy(e);
    branch_join_.clear();
    if (branch_fork_) (void)hipEventDestroy(branch_fork_);
    branch_fork_ = nullptr;
    for (size_t i = ordered_.size(); i-- > 0;) {  // reverse order (native/cuda/impl.cc:126-137)
        (void)ordered_[i]->computeDeinitialize();
        ordered_[i]->resetPlan();
        for (const auto& kv : ordered_[i]->inputs()) kv.second.runtimeBound(-1);
        for (const auto& kv : ordered_[i]->outputs()) kv.second.runtimeBound(-1);
    }
    ordered_.clear();
    if (stream_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
    return Result::SUCCESS;
}

// The result convention of src/runtime/native/cpu/impl.cc:98-148: SUCCESS | RELOAD continue; SKIP marks the
// unit's outputs skipped for this cycle and every unit reading a skipped tensor is skipped in turn;
// YIELD | TIMEOUT end the cycle quietly (a source without data); anything else fails the cycle.
Result Runtime::submitAll(bool record_events, U64 slot, bool count_cycles, bool fork) {
    std::set<U64> skipped;  // producers (ProducerAttribute) whose outputs do not exist this cycle
    auto producer_of = [](const Tensor& t) -> U64 {
        const AttrValue* a = t.attribute(ProducerAttribute);
        return a && std::holds_alternative<U64>(*a) ? std::get<U64>(*a) : 0;
    };
    fork = fork && branch_streams_.size() > 1 && !record_events;
    std::vector<char> started(branch_streams_.size(), 0);
    if (fork) JST_HIP_CHECK(hipEventRecord(branch_fork_, stream_), "hipEventRecord");
    for (auto& u : units_) {
        if (u.is_static && u.settled) continue;
        bool starved = false;
        for (Module* m : u.modules)
            for (const auto& kv : m->inputs()) starved |= skipped.count(producer_of(kv.second)) != 0;
        if (starved) {
            for (Module* m : u.modules) skipped.insert((U64)reinterpret_cast<uintptr_t>(m));
            continue;
        }
        hipStream_t on = stream_;
        if (fork && u.has_kernels) {
            on = branch_streams_[(size_t)u.branch];
            if (u.branch != 0 && !started[(size_t)u.branch]) {  // the branch joins the capture behind the cycle's start
                JST_HIP_CHECK(hipStreamWaitEvent(on, branch_fork_, 0), "hipStreamWaitEvent");
                started[(size_t)u.branch] = 1;
            }
            for (size_t d : u.deps) {
                const Unit& p = units_[d];
                if (p.branch != u.branch && p.done_epoch == capture_epoch_)  // (not recorded: it did not run in this capture)
                    JST_HIP_CHECK(hipStreamWaitEvent(on, p.done, 0), "hipStreamWaitEvent");
            }
        }
        const bool rec = record_events && slot < u.span.begin.size();
        if (rec) JST_HIP_CHECK(hipEventRecord(u.span.begin[slot], stream_), "hipEventRecord");
        const Result r = u.submit(on);
        if (r == Result::YIELD || r == Result::TIMEOUT || (r != Result::SUCCESS && r != Result::RELOAD && r != Result::SKIP)) {
            // The cycle ends here; whatever it did enqueue is behind stream_.  Modules that ran still hear that their cycle
            // is closed (a live ring source records the slot's free event there: without it the producer's next push to
            // that slot sat out its 200 ms timeout).  Under a forked capture the branches go back into stream_ first.
            if (fork)
                for (size_t b = 1; b < branch_streams_.size(); ++b) {
                    if (!started[b]) continue;
                    (void)hipEventRecord(branch_join_[b], branch_streams_[b]);
                    (void)hipStreamWaitEvent(stream_, branch_join_[b], 0);
                }
            for (auto& w : units_)
                for (Module* m : w.modules) m->cycleSubmitted(stream_);
            if (r != Result::YIELD && r != Result::TIMEOUT)
                JST_ERROR("[RUNTIME] computeSubmit failed in '%s' (%s): %s", u.name.c_str(), ResultName(r), last_error());
            return r;
        }
        if (r == Result::SKIP)
            for (Module* m : u.modules) skipped.insert((U64)reinterpret_cast<uintptr_t>(m));
        if (fork && u.has_kernels) {
            JST_HIP_CHECK(hipEventRecord(u.done, on), "hipEventRecord");
            u.done_epoch = capture_epoch_;
        }
        if (rec) {
            JST_HIP_CHECK(hipEventRecord(u.span.end[slot], stream_), "hipEventRecord");
            u.span.recorded[slot] = true;
            u.span.sampleCycles[slot] = 1;
        }
        if (count_cycles)
            for (Module* m : u.modules) m->timing.cycles++;
    }
    if (fork)  // every branch back into the runtime's stream: the next cycle (and the end of the capture) sits behind all of them
        for (size_t b = 1; b < branch_streams_.size(); ++b) {
            if (!started[b]) continue;
            JST_HIP_CHECK(hipEventRecord(branch_join_[b], branch_streams_[b]), "hipEventRecord");
            JST_HIP_CHECK(hipStreamWaitEvent(stream_, branch_join_[b], 0), "hipStreamWaitEvent");
        }
    for (auto& u : units_)
        for (Module* m : u.modules) m->cycleSubmitted(stream_);
    return Result::SUCCESS;
}

Result Runtime::harvestTiming() {
    if (!timing_pending_) return Result::SUCCESS;
    timing_pending_ = false;
    for (auto& u : units_) {
        for (size_t s = 0; s < u.span.begin.size(); ++s) {
            if (!u.span.recorded[s]) continue;
            u.span.recorded[s] = false;
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, u.span.begin[s], u.span.end[s]) == hipSuccess) {
                u.span.totalMs += ms;
                u.span.count += 1;
                u.span.cycles += s < u.span.sampleCycles.size() ? u.span.sampleCycles[s] : 1;
                for (Module* m : u.modules) m->timing.computeTimeMs += ms / (F64)u.modules.size();
            }
        }
    }
    return Result::SUCCESS;
}

Result Runtime::eagerCycle(bool& needs_sync, bool overwrite_samples) {
    const bool timing = (flags_ & TIMING) != 0;
    bool any_unsettled_static = false;
    for (auto& u : units_) any_unsettled_static |= (u.is_static && !u.settled);
    // Event pairs rotate over the period_ slots; the host only waits when a slot that still holds
    // an unread sample is about to be re-recorded (once per period_ cycles, not per cycle).
    const U64 slot = cycles_ % period_;
    if (timing) {
        bool busy = false;
        for (auto& u : units_) busy |= (slot < u.span.recorded.size() && u.span.recorded[slot]);
        if (busy && !overwrite_samples) {
            JST_HIP_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
            timing_pending_ = true;
            JST_CHECK(harvestTiming());
        }
    }
    if (pipelined()) {  // everything on the segment stream, after whatever the side lane still runs
        if (side_pending_) JST_CHECK(joinLanes());
        last_slot_ = cycles_ % (2 * period_);
        for (Tensor& t : pipelined_) JST_CHECK(t.ringSelect(last_slot_));
    }
    {
        const Result r = submitAll(timing, slot, true);
        if (r == Result::YIELD || r == Result::TIMEOUT) return r;  // no data: the cycle did not happen
        if (r != Result::SUCCESS) return r;
    }
    timing_pending_ = timing_pending_ || timing;
    ++cycles_;
    // STATIC modules settle after one successful cycle (scheduler_synchronous.cc:534-546).
    for (auto& u : units_)
        if (u.is_static) u.settled = true;
    needs_sync |= any_unsettled_static;
    return Result::SUCCESS;
}

U64 Runtime::configGenerations() const {
    U64 g = 0;
    for (const Module* m : ordered_) g += m->configGeneration();
    return g;
}

// A reconfigured module has new kernel arguments: the captured graphs hold the old ones.
Result Runtime::dropGraphs() {
    JST_CHECK(joinLanes());
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_) (void)hipGraphDestroy(graph_);
    graph_exec_ = nullptr;
    graph_ = nullptr;
    for (int lane = 0; lane < 2; ++lane)
        for (int half = 0; half < 2; ++half) {
            if (lane_exec_[lane][half]) (void)hipGraphExecDestroy(lane_exec_[lane][half]);
            if (lane_graph_[lane][half]) (void)hipGraphDestroy(lane_graph_[lane][half]);
            lane_exec_[lane][half] = nullptr;
            lane_graph_[lane][half] = nullptr;
        }
    lane_launches_ = 0;
    for (auto& kv : span_graphs_) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    span_graphs_.clear();
    return Result::SUCCESS;
}

// A failure between hipStreamBeginCapture and hipStreamEndCapture must not leave the stream capturing (every later
// call on it would fail) nor half-built graphs behind (graphActive() would launch null executables).
Result Runtime::abortCapture(Result r) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(stream_, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    (void)dropGraphs();
    return r;
}

Result Runtime::launchSpan(U64 n, bool timing) {
    const U64 phase = cycles_ % period_;
    // A cycle-batched span is one launch per unit: two or three kernels.  JST_RUNTIME_EAGER_SPANS=1 (A/B switch) submits
    // them directly instead of replaying a graph of them.
    // (read per call: bench.py measures both forms in one process -- `alt_eager_spans`)
    const bool eager_spans = switch_value(SW_RUNTIME_EAGER_SPANS) != 0;
    if (eager_spans && batched_) {
        JST_CHECK(submitBatched(n, false));
        for (auto& u : units_) {
            if (u.is_static && u.settled) continue;
            for (Module* m : u.modules) m->timing.cycles += n;
        }
        cycles_ += n;
        return Result::SUCCESS;
    }
    auto key = std::make_pair(phase, n);
    auto it = span_graphs_.find(key);
    if (it == span_graphs_.end()) {
        if (span_graphs_.size() >= 64) {
            // Bounded cache: the least recently used entry goes, and only behind a stream fence -- after an
            // unsynchronised compute() its executable may still be in flight.
            auto victim = span_graphs_.begin();
            for (auto j = span_graphs_.begin(); j != span_graphs_.end(); ++j)
                if (j->second.last_use < victim->second.last_use) victim = j;
            JST_HIP_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
            if (victim->second.exec) (void)hipGraphExecDestroy(victim->second.exec);
            if (victim->second.graph) (void)hipGraphDestroy(victim->second.graph);
            span_graphs_.erase(victim);
        }
        JST_HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
        Result r = Result::SUCCESS;
        // No event-record nodes in span graphs: the unit timers live in the period graph (every timingStride()-th
        // cycle) and in eager cycles; a span is a head or a tail of at most period - 1 cycles.
        if (batched_) r = submitBatched(n, false);
        ++capture_epoch_;
        for (U64 c = 0; !batched_ && c < n && r == Result::SUCCESS; ++c)
            r = submitAll(false, (phase + c) % period_, false, true);  // advances the host cursors
        if (r != Result::SUCCESS) return abortCapture(r);
        SpanGraph sg;
        if (hipStreamEndCapture(stream_, &sg.graph) != hipSuccess || !sg.graph) return abortCapture(Result::ERROR);
        if (hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0) != hipSuccess) {
            (void)hipGraphDestroy(sg.graph);
            JST_ERROR("[RUNTIME] hipGraphInstantiate failed for a %llu-cycle span.", (unsigned long long)n);
            return Result::ERROR;
        }
        captured_generation_ = configGenerations();
        it = span_graphs_.emplace(key, sg).first;
    } else {
        for (Module* m : ordered_) m->advanceHostState(n);  // a replay does not run the host side
    }
    it->second.last_use = ++span_clock_;
    JST_HIP_CHECK(hipGraphLaunch(it->second.exec, stream_), "hipGraphLaunch");
    for (auto& u : units_) {
        if (u.is_static && u.settled) continue;
        for (Module* m : u.modules) m->timing.cycles += n;
    }
    (void)timing;
    cycles_ += n;
    return Result::SUCCESS;
}

Result Runtime::compute(U64 cycles, bool sync) {
    if (!created_) {
        JST_ERROR("[RUNTIME] compute() before create().");
        return Result::ERROR;
    }
    if (graphActive() && configGenerations() != captured_generation_) JST_CHECK(dropGraphs());
    const bool timing = (flags_ & TIMING) != 0;
    bool needs_sync = sync;
    while (cycles > 0) {
        bool any_unsettled_static = false;
        for (auto& u : units_) any_unsettled_static |= (u.is_static && !u.settled);
        // TIMING: a hipEventRecord issued under stream capture is only a dependency edge of the capture -- nothing is
        // recorded when the graph replays (and hipEventRecordExternal is rejected by this ROCm) -- so a timed runtime
        // does not capture whole periods: the first cycle of every period runs EAGERLY between real event records
        // (same kernels, same stream, back to back with the graphs) and the other period - 1 cycles replay as a
        // span graph.  (Round 1's "in-graph" unit timers were in fact fed by the eager cycles between replays.)
        // Every kTimedPeriodStride-th period is timed that way (the eager cycle costs ~1.5 us of launch gaps and a
        // pair of event packets per unit: at every fourth period that was 1.5 % of the step); the
        // periods in between replay as one graph, captured from phase 0 without event records.
        const U64 kTimedPeriodStride = period_ > 1 ? 16 : 64;
        const bool timed_periods = timing && !pipelined();
        const bool aligned0 = (cycles_ % period_) == 0;
        // (counted in cycles, not in whole-period replays: a caller that submits fewer cycles than a period per call
        // still gets its unit timers sampled every kTimedPeriodStride periods)
        const bool timed_now = timed_periods && aligned0 &&
                               (!timed_once_ || cycles_ - last_timed_cycle_ >= kTimedPeriodStride * period_);
        bool use_graph = (flags_ & GRAPH) && !any_unsettled_static && cycles >= period_ &&
                         (!timed_periods || (aligned0 && !timed_now));
        if (use_graph && !periodGraphActive()) {
            bool capturable = true;
            for (auto& u : units_)
                for (Module* m : u.modules) capturable &= (u.is_static || m->capturable());
            if (!capturable) {
                flags_ &= ~GRAPH;  // a module needs per-cycle host arguments: stay eager
                use_graph = false;
            } else if (pipelined()) {
                // Four graphs: {producer lane, surface lane} x {ring half 0, 1}.  Host-side cursors
                // advance by a whole period per capture, i.e. come back to the phase they had.
                capture_phase_ = cycles_ % period_;
                JST_CHECK(joinLanes());
                for (int half = 0; half < 2; ++half)
                    for (int lane = 0; lane < 2; ++lane) JST_CHECK(captureLane(lane, half, timing));
                captured_generation_ = configGenerations();
                for (auto& u : units_) std::fill(u.span.recorded.begin(), u.span.recorded.end(), false);
            } else {
                // Capture period_ consecutive cycles.  Host-side cursors (ring sources) advance
                // during capture exactly as they would while running, so after the capture they
                // are back at the phase they started from and the replay starts there too.
                capture_phase_ = cycles_ % period_;
                JST_HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal),
                              "hipStreamBeginCapture");
                Result r = Result::SUCCESS;
                if (batched_) r = submitBatched(period_, false);  // one launch per unit for the whole period
                ++capture_epoch_;
                for (U64 c = 0; !batched_ && c < period_ && r == Result::SUCCESS; ++c)
                    r = submitAll(false, c, false, true);  // event records under capture record nothing on replay (see above)
                if (r != Result::SUCCESS) return abortCapture(r);
                hipGraph_t g = nullptr;
                if (hipStreamEndCapture(stream_, &g) != hipSuccess || !g) return abortCapture(Result::ERROR);
                graph_ = g;
                if (hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0) != hipSuccess) {
                    graph_exec_ = nullptr;
                    (void)dropGraphs();
                    JST_ERROR("[RUNTIME] hipGraphInstantiate failed.");
                    return Result::ERROR;
                }
                captured_generation_ = configGenerations();
            }
        }
        if (use_graph && !pipelined() && period_ > 1 && cycles > period_ && cycles < 2 * period_ &&
            (cycles_ % period_) == capture_phase_ && switch_value(SW_RUNTIME_NO_SPANS) == 0) {
            // One period and a tail: ONE graph of exactly these cycles (cached per (phase, cycles) like every span)
            // instead of the period graph followed by a span graph -- a launch less per call, which is what a short
            // timed region (bench.py --steps 20 with 16 ring slots) is made of.
            const U64 n = cycles;
            JST_CHECK(launchSpan(n, timing));
            cycles -= n;
            continue;
        }
        if (use_graph && (cycles_ % period_) == capture_phase_) {
            // No harvest between replays: the in-graph event nodes are simply re-recorded, and the
            // final synchronise reads the last replay's period_ samples per unit.  Replays
            // therefore queue back to back with no host round trip.
            if (pipelined()) {
                const int h = (int)(lane_launches_ % 2);
                if (lane_launches_ >= 2)  // this half is free once the surfaces of two periods ago are done
                    JST_HIP_CHECK(hipStreamWaitEvent(stream_, lane_done_[1][h], 0), "hipStreamWaitEvent");
                JST_HIP_CHECK(hipGraphLaunch(lane_exec_[0][h], stream_), "hipGraphLaunch");
                JST_HIP_CHECK(hipEventRecord(lane_done_[0][h], stream_), "hipEventRecord");
                JST_HIP_CHECK(hipStreamWaitEvent(side_stream_, lane_done_[0][h], 0), "hipStreamWaitEvent");
                JST_HIP_CHECK(hipGraphLaunch(lane_exec_[1][h], side_stream_), "hipGraphLaunch");
                JST_HIP_CHECK(hipEventRecord(lane_done_[1][h], side_stream_), "hipEventRecord");
                last_slot_ = (U64)h * period_ + (period_ - 1);
                side_pending_ = true;
                ++lane_launches_;
            } else {
                JST_HIP_CHECK(hipGraphLaunch(graph_exec_, stream_), "hipGraphLaunch");
            }
            // a replay does not run the host side: a whole period brings every cursor back to where it was, but a
            // module may still need to know that cycles went by (Spectrogram riding on the spectrum launches)
            for (Module* m : ordered_) m->advanceHostState(period_);
            for (auto& u : units_) {
                if (u.is_static && u.settled) continue;
                for (Module* m : u.modules) m->timing.cycles += period_;
            }
            timing_pending_ = timing_pending_ || timing;
            cycles_ += period_;
            cycles -= period_;
            continue;
        }
        static const bool no_spans = getenv("JST_RUNTIME_NO_SPANS") != nullptr;  // A/B switch: eager heads and tails
        if (timed_now && (!no_spans || period_ == 1) && (flags_ & GRAPH) && !any_unsettled_static) {
            bool capturable = true;
            for (auto& u : units_)
                for (Module* m : u.modules) capturable &= (u.is_static || m->capturable());
            if (capturable && batched_) {
                // A cycle-batched runtime times what it replays: a whole period submitted EAGERLY, one launch per unit,
                // between real event records (the sample then covers period_ cycles: unitMeanCycles).  A call too short
                // to hold a period leaves the timers alone and replays as a span.
                if (cycles >= period_) {
                    bool busy = false;
                    for (auto& u : units_) busy |= (!u.span.recorded.empty() && u.span.recorded[0]);
                    if (busy) {
                        JST_HIP_CHECK(hipStreamSynchronize(stream_), "hipStreamSynchronize");
                        timing_pending_ = true;
                        JST_CHECK(harvestTiming());
                    }
                    last_timed_cycle_ = cycles_;
                    JST_CHECK(submitBatched(period_, true));
                    timed_once_ = true;
                    timing_pending_ = true;
                    for (auto& u : units_) {
                        if (u.is_static && u.settled) continue;
                        for (Module* m : u.modules) m->timing.cycles += period_;
                    }
                    cycles_ += period_;
                    cycles -= period_;
                    continue;
                }
            } else if (capturable) {
                last_timed_cycle_ = cycles_;
                const Result r = eagerCycle(needs_sync, true);  // the timed cycle: real event records
                if (r != Result::SUCCESS) return r;
                timed_once_ = true;
                // the rest of the period (or of the call, when it is shorter) replays as a span graph
                const U64 rest = std::min<U64>(cycles - 1, period_ - 1);
                if (rest) JST_CHECK(launchSpan(rest, timing));
                cycles -= 1 + rest;
                continue;
            }
        }
        if (!no_spans && (flags_ & GRAPH) && !any_unsettled_static && !pipelined() && period_ > 1) {
            // Not a whole period from here (the head of a call that starts off the captured phase, or its tail):
            // replay a graph of exactly those cycles instead of running them eagerly.
            bool capturable = true;
            for (auto& u : units_)
                for (Module* m : u.modules) capturable &= (u.is_static || m->capturable());
            if (capturable) {
                U64 n = cycles < period_ ? cycles : period_ - 1;
                if (periodGraphActive() || timed_periods) {
                    const U64 base = timed_periods ? 0 : capture_phase_;
                    const U64 off = (cycles_ + period_ - base) % period_;
                    if (off) n = std::min<U64>(n, period_ - off);  // up to the next period boundary
                }
                JST_CHECK(launchSpan(n, timing));
                cycles -= n;
                continue;
            }
            flags_ &= ~GRAPH;
        }
        {
            const Result r = eagerCycle(needs_sync, false);
            if (r == Result::YIELD || r == Result::TIMEOUT) {  // quiet end: what was queued still completes
                JST_CHECK(flushUnits());
                if (sync) {
                    JST_CHECK(joinLanes());
                    JST_CHECK(harvestTiming());
                }
                return r;
            }
            if (r != Result::SUCCESS) return r;
        }
        --cycles;
    }
    JST_CHECK(flushUnits());  // work a unit deferred past its cycle (a spectrogram riding on the next launch)
    JST_CHECK(showLatestSlot());
    if (needs_sync) {
        JST_CHECK(joinLanes());
        JST_CHECK(harvestTiming());
    }
    return Result::SUCCESS;
}

Result Runtime::synchronize() {
    if (!stream_) return Result::SUCCESS;
    JST_CHECK(joinLanes());
    return harvestTiming();
}

F64 Runtime::unitMeanMs(const std::string& name) {
    for (auto& u : units_)
        if (u.name == name && u.span.count) return u.span.totalMs / (F64)u.span.count;
    return -1.0;
}

F64 Runtime::eventPairOverheadMs() {
    return calibration_unit_.empty() ? -1.0 : unitMeanMs(calibration_unit_);
}

void Runtime::resetTiming() {
    for (auto& u : units_) {
        u.span.totalMs = 0.0;
        u.span.count = 0;
        u.span.cycles = 0;
    }
}

F64 Runtime::unitMeanCycles(const std::string& name) {
    for (auto& u : units_)
        if (u.name == name && u.span.count) return (F64)u.span.cycles / (F64)u.span.count;
    return -1.0;
}

}  // namespace jst
