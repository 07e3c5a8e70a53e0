// module.hh -- Module contract, 4-key Registry, per-segment NativeHip Runtime and the synchronous
// Scheduler of the MI355X backend.
//
// The contract is the reference's:
//   lifecycle  validate -> define -> create -> destroy            src/module.cc:47-212,
//                                                                 include/jetstream/detail/module_impl.hh:43-47
//   compute    computeInitialize / computeSubmit(stream) / computeDeinitialize
//                                                                 include/jetstream/runtime_context_native_cuda.hh:32-34
//   registry   exact (type, device, runtime, provider) lookup     src/registry.cc:583-622
//   runtime    one stream per segment, per-module events, ONE host sync per cycle
//                                                                 src/runtime/native/cuda/impl.cc:185-272
//   scheduler  Kahn order, STATIC_OUTPUT settlement               src/scheduler_synchronous.cc:534-546,574-696
// What is new (MI355X-first): the runtime captures the steady-state cycle into a hipGraph and
// replays it, and the scheduler recognises the spectrum chain Multiply(window) -> FFT ->
// Amplitude [-> Range] and submits it as ONE kernel (8 B in + 4 B out per sample).
#pragma once

#include <functional>
#include <map>
#include <set>

#include "tensor.hh"

namespace jst {

using Config = std::map<std::string, std::string>;
constexpr const char* ProducerAttribute = "producer";

bool ConfigBool(const Config& c, const std::string& key, bool fallback, bool* ok = nullptr);
U64 ConfigU64(const Config& c, const std::string& key, U64 fallback, bool* ok = nullptr);
F64 ConfigF64(const Config& c, const std::string& key, F64 fallback, bool* ok = nullptr);
std::string ConfigStr(const Config& c, const std::string& key, const std::string& fallback);

struct Timing {  // include/jetstream/module.hh:25-31
    U64 cycles = 0;
    F64 computeTimeMs = 0.0;
};

class Module {
 public:
    virtual ~Module() = default;

    // Framework entry (src/module.cc:47-212): config -> validate -> define -> input checks ->
    // create.  On failure the module is left uncreated and the error is in last_error().
    Result construct(const std::string& name, const Config& config,
                     const std::map<std::string, Tensor>& inputs);
    Result teardown();
    // Module::reconfigure (src/module.cc:233-290): overlay `config` on the staged configuration; an
    // unchanged configuration is SUCCESS, an invalid one is rejected with the staged one intact, a
    // change the module cannot absorb in place is RECREATE (the caller tears down and rebuilds), anything
    // else is applied by reconfigureImpl().  validateOnly stops after validation.  The runtime re-captures
    // its hipGraph when a module's configuration generation moved.
    Result reconfigure(const Config& config, bool validateOnly);
    U64 configGeneration() const { return config_generation_; }

    // ---- Module::Impl hooks --------------------------------------------------------------------
    virtual const char* type() const = 0;
    virtual Result validate() { return Result::SUCCESS; }
    virtual Result define() = 0;
    virtual Result create() = 0;
    virtual Result destroy() { return Result::SUCCESS; }
    // Module::Impl::reconfigure (detail/module_impl.hh:47; default src/module_impl.cc:61-63): called with
    // the members already parsed from the NEW configuration by validate(); `previous` is the staged one.
    virtual Result reconfigureImpl(const Config& previous) { (void)previous; return Result::RECREATE; }

    // ---- NativeHipRuntimeContext ---------------------------------------------------------------
    virtual Result computeInitialize() { return Result::SUCCESS; }
    virtual Result computeSubmit(hipStream_t stream) = 0;
    virtual Result computeDeinitialize() { return Result::SUCCESS; }
    // Called on the compute thread once EVERY unit of the cycle this module's computeSubmit opened has been enqueued on
    // `stream` (Runtime::submitAll): from here on an event recorded on the stream is behind all of the cycle's readers.
    virtual void cycleSubmitted(hipStream_t /*stream*/) {}
    // True when computeSubmit touches no device state through the host (graph-capturable).
    virtual bool capturable() const { return true; }
    // Number of consecutive cycles after which this module's host-side state repeats (a ring
    // source with R slots: R).  The runtime captures that many cycles into one hipGraph.
    virtual U64 cyclePeriod() const { return 1; }
    // Moves that host-side state forward by 'cycles' cycles without submitting anything: a cached hipGraph of
    // those cycles is about to be replayed (a capture advances the state itself, a replay does not).
    virtual void advanceHostState(U64 /*cycles*/) {}
    // Cycle batching (MI355X-first; no counterpart in the reference): a module that can take `n` CONSECUTIVE compute
    // cycles as one piece of work -- the cycles whose inputs sit in ring slots first_slot, first_slot + 1, ... (mod the
    // ring) of the tensors the runtime's planner promoted for it -- with the result every one of the `n` per-cycle
    // submissions would have left (Runtime::planBatch).  spanCapable() is asked at planning time.
    virtual bool spanCapable() const { return false; }
    virtual Result computeSubmitSpan(hipStream_t /*stream*/, U64 /*first_slot*/, U64 /*n*/) { return Result::ERROR; }
    // A runtime's planner may rewire a module for a fused / elided unit (a duplicate whose readers are pointed at its source).
    // Called at the top of every plan and when the runtime is destroyed: no planning decision outlives the runtime that took it.
    virtual void resetPlan() {}
    // Storage this module touches OUTSIDE its ports, as planned: state tensors, side outputs a fused unit writes for it, an
    // operand the planner pointed elsewhere.  Runtime::planBranches derives the edges between parallel branches from the
    // ports plus these (ports alone miss the Waterfall's ring, the Spectrogram's row indices, an elided duplicate's readers).
    virtual void planStorage(std::set<const void*>& /*reads*/, std::set<const void*>& /*writes*/) const {}
    // False for view/bookkeeping modules whose computeSubmit enqueues nothing on the stream.
    virtual bool launchesKernels() const { return true; }
    // Named internal state tensors (spectrogram/waterfall "frequencyBins"), for read-back.
    virtual const Tensor* state(const std::string&) const { return nullptr; }

    const std::string& name() const { return name_; }
    const std::string& provider() const { return provider_; }
    void setProvider(const std::string& p) { provider_ = p; }
    DeviceType device() const { return DeviceType::HIP; }
    const Config& config() const { return config_; }
    const std::map<std::string, Tensor>& inputs() const { return inputs_; }
    const std::map<std::string, Tensor>& outputs() const { return outputs_; }
    U64 taint() const { return taint_; }
    bool created() const { return created_; }
    Timing timing;

 protected:
    Result defineTaint(U64 taint) { taint_ = taint; return Result::SUCCESS; }
    Result defineInterfaceInput(const std::string& port) { input_ports_.push_back(port); return Result::SUCCESS; }
    Result defineInterfaceOutput(const std::string& port) { output_ports_.push_back(port); return Result::SUCCESS; }
    // outputs()[port].produced(name(), port, tensor) (include/jetstream/tensor_link.hh:22-34): the link
    // remembers its producer -- here as an attribute that travels with every copy of the handle, so
    // that the runtime can tell a skipped producer's tensor from another view of the same storage
    void produced(const std::string& port, const Tensor& t) {
        Tensor link = t;
        (void)link.setAttribute(ProducerAttribute, AttrValue{(U64)reinterpret_cast<uintptr_t>(this)});
        outputs_[port] = link;
    }

    std::string name_;
    std::string provider_ = "generic";
    Config config_;
    std::map<std::string, Tensor> inputs_;
    std::map<std::string, Tensor> outputs_;
    std::vector<std::string> input_ports_, output_ports_;
    U64 taint_ = CLEAN;
    bool created_ = false;
    U64 config_generation_ = 0;
};

// ---- Registry ----------------------------------------------------------------------------------
using ModuleFactory = std::function<std::unique_ptr<Module>()>;

class Registry {
 public:
    static Registry& instance();
    void add(const std::string& type, DeviceType device, RuntimeType runtime,
             const std::string& provider, ModuleFactory factory);
    // Exact 4-key match, no fallback (src/registry.cc:605-618).
    std::unique_ptr<Module> build(const std::string& type, DeviceType device, RuntimeType runtime,
                                  const std::string& provider) const;
    std::vector<std::string> listAvailableModules(const std::string& type = "") const;

 private:
    std::map<std::string, ModuleFactory> factories_;
};

struct ModuleRegistrar {
    ModuleRegistrar(const char* type, DeviceType device, RuntimeType runtime, const char* provider,
                    ModuleFactory f) {
        Registry::instance().add(type, device, runtime, provider, std::move(f));
    }
};

#define JST_REGISTER_MODULE(Impl, type_string, device, runtime, provider)                  \
    static ::jst::ModuleRegistrar jst_registrar_##Impl(type_string, device, runtime, provider, \
                                                       [] { return std::unique_ptr<::jst::Module>(new Impl()); })

// ---- Runtime -----------------------------------------------------------------------------------
struct KernelSpan {  // hipEvent pairs around one execution unit, for live per-kernel timing
    std::string name;
    std::vector<hipEvent_t> begin, end;  // one pair per cycle of the capture period
    std::vector<bool> recorded;
    std::vector<U64> sampleCycles;       // compute cycles the pair of that slot brackets (1, or a whole batched period)
    F64 totalMs = 0.0;
    U64 count = 0;
    U64 cycles = 0;                      // compute cycles covered by the `count` samples
};

// What a fused spectrum unit offers to a cycle-batched runtime (modules::TryFuseSpectrum fills it when the unit reads a
// resident ring and feeds one index-fed Spectrogram): `prepare` turns the unit's outputs into rings of as many slots as
// the source has, `submit_span` runs the transforms of n consecutive ring slots as ONE launch.
struct SpanSupport {
    std::function<Result(U64 slots)> prepare;
    std::function<Result(hipStream_t, U64 first_slot, U64 n)> submit_span;
    Tensor phase;                        // the ring whose selected slot is the current cycle's slot (the source's output)
    std::vector<Tensor> rings;           // the promoted outputs (the runtime shows a reader the latest cycle's slot)
    explicit operator bool() const { return static_cast<bool>(submit_span); }
};

class Runtime {
 public:
    // BATCH: cycle batching -- the cycles of a captured ring period (or of a span of it) run as ONE launch per unit
    // (needs GRAPH and FUSE and a chain the planner can batch: resident ring source -> fused spectrum unit -> index-fed
    // Spectrogram; anything else silently stays per cycle).
    enum Flags : U32 { NONE = 0, GRAPH = 1 << 0, FUSE = 1 << 1, TIMING = 1 << 2, PIPELINE = 1 << 3, COMBINE = 1 << 4, BATCH = 1 << 5 };

    Runtime();
    ~Runtime();

    // Takes the modules of one device segment (not owned).  Orders them (Kahn over tensor
    // storage identity), runs computeInitialize, plans static settlement and fusion.
    Result create(const std::vector<Module*>& modules, U32 flags);
    Result destroy();

    // 'cycles' compute cycles: submit every unsettled unit on the segment stream, or replay the
    // captured hipGraph (which holds period() cycles).  sync=true ends with the reference's
    // stream synchronise (src/runtime/native/cuda/impl.cc:244); sync=false leaves the work
    // queued so consecutive calls run back to back.
    Result compute(U64 cycles, bool sync);
    Result synchronize();
    U64 period() const { return period_; }

    hipStream_t stream() const { return stream_; }
    const std::vector<std::string>& order() const { return order_names_; }
    const std::vector<std::string>& units() const { return unit_names_; }
    bool graphActive() const { return periodGraphActive() || !span_graphs_.empty(); }
    bool periodGraphActive() const { return graph_exec_ != nullptr || lane_exec_[0][0] != nullptr; }
    // Mean device time (ms) of the named unit over the cycles run with TIMING; <0 if unknown.
    F64 unitMeanMs(const std::string& name);
    // Mean duration of an EMPTY event pair recorded in the same graph/stream (one kernel-less
    // unit keeps its pair for this purpose): the cost of the measurement itself.
    F64 eventPairOverheadMs();
    // Mean number of compute cycles one timing sample of the named unit covers: 1, or the ring period for a
    // cycle-batched runtime (whose timed launches carry a whole period each); <0 if unknown.
    F64 unitMeanCycles(const std::string& name);
    bool batched() const { return batched_; }
    size_t branches() const { return branch_streams_.size() > 1 ? branch_streams_.size() : 1; }
    void resetTiming();

 private:
    struct Unit {
        std::string name;
        std::vector<Module*> modules;      // 1 module, or the fused chain
        std::function<Result(hipStream_t)> submit;
        std::function<Result(hipStream_t)> flush;  // optional: work the unit defers to the end of a compute call
        bool is_static = false;            // STATIC_OUTPUT with settled inputs: runs once
        bool has_kernels = true;           // false: nothing reaches the stream
        bool timed = true;                 // carries an event pair
        int lane = 0;                      // PIPELINE: 1 = surface lane (runs beside the next cycle)
        bool settled = false;
        SpanSupport batch;                 // BATCH: the fused spectrum unit's multi-cycle launch
        bool per_cycle_in_span = false;    // BATCH: a sink that reads the batched unit's rings: n per-cycle submissions inside a span
        KernelSpan span;
        // BRANCHES (planBranches): the units this one must run behind (a tensor one writes and the other reads or writes)
        // and the capture stream it is submitted on
        std::vector<size_t> deps;
        int branch = 0;
        hipEvent_t done = nullptr;         // recorded behind the unit under capture (an edge of the graph, nothing at run time)
        U64 done_epoch = 0;                // the capture that recorded it
    };
    Result planOrder(const std::vector<Module*>& modules);
    Result planUnits();
    void computePeriod();
    Result flushUnits();
    bool tryFuseSpectrum(size_t at, Unit& unit, size_t& consumed);
    std::set<const void*> static_storage_;  // storage of the tensors statically settled units produce (planUnits)
    // In a captured period only every timingStride()-th cycle carries event-record nodes: a pair
    // costs ~2 us of queue time, sampling keeps Module::Timing live at a quarter of that cost.
    U64 timingStride() const { return period_ >= 8 ? 4 : 1; }
    Result submitAll(bool record_events, U64 event_slot, bool count_cycles, bool fork = false);
    // BRANCHES: a flowgraph with independent chains behind one source (the reference's multi-fm.yml: three spectrum chains,
    // a Filter and a demodulator, every kernel a handful of workgroups) is captured as a graph with PARALLEL branches --
    // the units of a chain on one capture stream, forks and joins as event edges -- instead of one serial chain of
    // launch-floor kernels.  Only under capture, only when the plan has more than one branch.
    Result planBranches();
    std::vector<hipStream_t> branch_streams_;  // [0] = stream_
    std::vector<hipEvent_t> branch_join_;
    hipEvent_t branch_fork_ = nullptr;
    U64 capture_epoch_ = 0;
    Result harvestTiming();
    Result eagerCycle(bool& needs_sync, bool overwrite_samples);
    // 'n' < period() cycles starting at the current phase as a hipGraph of their own (captured on first use, cached
    // per (phase, n)): the head and tail of a compute() call that is not a whole number of periods replay like
    // the periods in between instead of running eagerly.
    Result launchSpan(U64 n, bool timing);
    Result abortCapture(Result r);
    // BATCH: `n` consecutive cycles from the current phase, one submission per unit (under capture, or eagerly between
    // real event records when record_events).
    Result planBatch();
    Result submitBatched(U64 n, bool record_events);
    Result showLatestSlot();
    bool batched_ = false;
    size_t batch_unit_ = 0;

    hipStream_t stream_ = nullptr;
    U32 flags_ = 0;
    std::vector<Module*> ordered_;
    std::vector<std::string> order_names_, unit_names_;
    std::vector<Unit> units_;
    hipGraph_t graph_ = nullptr;
    hipGraphExec_t graph_exec_ = nullptr;
    U64 cycles_ = 0;
    U64 period_ = 1;
    U64 captured_generation_ = 0;  // sum of the modules' configuration generations baked into the graphs
    U64 configGenerations() const;
    Result dropGraphs();
    U64 capture_phase_ = 0;
    struct SpanGraph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        U64 last_use = 0;  // LRU stamp
    };
    std::map<std::pair<U64, U64>, SpanGraph> span_graphs_;  // (phase, cycles) -> graph
    std::string calibration_unit_;
    // PIPELINE: SURFACE units run as their own graphs on a second stream (a second hardware queue),
    // one period behind the producers; the tensors in between are rings of two periods.
    hipStream_t side_stream_ = nullptr;
    std::vector<Tensor> pipelined_;        // producer->surface tensors, promoted to 2*period-slot rings
    hipGraph_t lane_graph_[2][2] = {};     // [lane][half of the ring]
    hipGraphExec_t lane_exec_[2][2] = {};
    hipEvent_t lane_done_[2][2] = {};      // lane L finished its latest launch on half h
    U64 lane_launches_ = 0;
    U64 last_slot_ = 0;                    // ring slot of the most recent cycle (what a reader sees)
    bool side_pending_ = false;
    bool pipelined() const { return !pipelined_.empty(); }
    Result planPipeline();
    Result captureLane(int lane, int half, bool timing);
    Result joinLanes();
    bool timing_pending_ = false;
    bool timed_once_ = false;       // an eager timed cycle has run since create()
    U64 last_timed_cycle_ = 0;      // cycle index of the last eager timed cycle (phase 0 of its period)
    U64 span_clock_ = 0;            // LRU stamp source of the span-graph cache
    bool created_ = false;
};

}  // namespace jst
