// modules.hh -- the Jetstream DSP/visualization modules of the hot path, registered for
// (DeviceType::HIP, RuntimeType::NATIVE, "generic").  One class per reference module; each header
// comment names the reference files it stands in for.  Port names, config keys, validation rules
// and error texts follow the reference so a flowgraph node only has to switch `device:`.
#pragma once

#include <condition_variable>
#include <mutex>

#include "../jst/module.hh"
#include "../kernels/kernels.hh"

namespace jst::modules {

// A failed launch becomes the module's diagnostic (JST_ERROR) and Result::ERROR.
inline Result hip_result(hipError_t e, const char* what) {
    if (e == hipSuccess) return Result::SUCCESS;
    JST_ERROR("[HIP] %s failed: %s", what, hipGetErrorString(e));
    return Result::ERROR;
}

// Device-resident pocketfft twiddle table W[k] = exp(+j 2 pi k / n), k in [0,n), cached per n.
Result GetTwiddles(U64 n, const float2** table);
Result GetPassTwiddles(U64 n, const float2** table);  // per-pass layout for the tiled kernels
// Host generator (exposed for tests through the C ABI): pocketfft sincos_2pibyn<float> scheme.
void ComputeTwiddles(U64 n, float* interleaved);

bool MakeEwLayout(const Tensor& out, const Tensor* a, const Tensor* b, dev::EwLayout& L);

// Scheduler hook: recognise multiply -> fft -> amplitude [-> range] starting at ordered[at].
// allow_combine: a Spectrogram that is the only consumer of the fused output may ride on the next cycle's launch
// (then `flush` is set: the runtime calls it at the end of every compute call to run the waiting spectrogram).
// allow_side: the one Spectrogram that quantises the fused output (height <= 256) may be fed with one-byte row indices
// written by the fused kernel beside its values (Spectrogram::indexFed) instead of re-reading the values.
// batch: filled when the unit can run the transforms of several consecutive ring slots as one launch (a cycle-batched
// runtime, Runtime::planBatch): the signal is a dense view of a resident ring and the unit feeds one index-fed Spectrogram.
bool TryFuseSpectrum(const std::vector<Module*>& ordered, size_t at, std::string& name,
                     std::vector<Module*>& members, std::function<Result(hipStream_t)>& submit,
                     size_t& consumed, bool allow_combine = false,
                     std::function<Result(hipStream_t)>* flush = nullptr, bool allow_side = true,
                     SpanSupport* batch = nullptr, const std::set<const void*>* static_storage = nullptr);

// Filter-chain fusions (filter_modules.cc): pad -> fft (zeros synthesised in the FFT's first load)
// and multiply -> fold (the broadcast product is never materialised).  Same contract.
bool TryFuseFilter(const std::vector<Module*>& ordered, size_t at, std::string& name,
                   std::vector<Module*>& members, std::function<Result(hipStream_t)>& submit,
                   size_t& consumed);

// Small-chain fusions (chain_fusions.cc): multiply(window) -> fft when no Amplitude follows (an AGC in between), and
// amplitude -> range as one elementwise pass.  Same contract.
bool TryFuseMultiplyFft(const std::vector<Module*>& ordered, size_t at, std::string& name, std::vector<Module*>& members,
                        std::function<Result(hipStream_t)>& submit, size_t& consumed);
bool TryFuseAmplitudeRange(const std::vector<Module*>& ordered, size_t at, std::string& name, std::vector<Module*>& members,
                           std::function<Result(hipStream_t)>& submit, size_t& consumed);
// duplicate whose readers all walk strides themselves (fft_windowed's Multiply, Fm): the readers take the view, no copy
// (filter_modules.cc).
bool TryElideDuplicate(const std::vector<Module*>& ordered, size_t at, std::string& name, std::vector<Module*>& members,
                       std::function<Result(hipStream_t)>& submit, size_t& consumed);
// agc (one tile per lane) -> amplitude -> range [-> waterfall] in the AGC's launch (ingest_modules.cc).  Same contract.
bool TryFuseAgcChain(const std::vector<Module*>& ordered, size_t at, std::string& name, std::vector<Module*>& members,
                     std::function<Result(hipStream_t)>& submit, size_t& consumed);

// The Filter block's plan (src/domains/dsp/filter/block_impl.cc:40-168, CalculateCandidatePlan): how long the
// convolution is, whether the block resamples by spectral folding and, if so, each head's fold offset.  Host logic of
// the block, kept next to the modules it wires so that a C / C++ consumer of the C ABI does not have to re-derive it.
struct FilterPlan {
    U64 padSize = 0, convolutionSize = 0, resamplerSize = 0;
    bool resample = false;
    F32 resampledSampleRate = 0.0f;
    std::vector<U64> resamplerOffsets;  // one per head when resample
};
Result CalculateFilterPlan(F32 sampleRate, F32 bandwidth, const std::vector<F32>& center, U64 taps, U64 heads,
                           U64 signalSize, FilterPlan& plan);

// src/domains/dsp/window/{module_impl.cc, module_impl_native_cpu.cc, module_impl_native_cuda.cc}
class Window : public Module {
 public:
    const char* type() const override { return "window"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeSubmit(hipStream_t stream) override;
    bool capturable() const override { return false; }  // host evaluation + upload (runs once: STATIC_OUTPUT)
    Tensor output;
    U64 size = 1024;
    std::vector<float> hostTaps;
};

// src/domains/dsp/invert/{module_impl.cc, module_impl_native_cpu.cc:79-103}
class Invert : public Module {
 public:
    const char* type() const override { return "invert"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeSubmit(hipStream_t stream) override;
    Tensor input, output;
    Index resolvedAxis = 0;
    U64 axisInnerSize = 1, axisLength = 1;
};

// src/domains/core/reshape/{module_impl.cc, module_impl_native_cpu.cc:17-19} -- view only
class Reshape : public Module {
 public:
    const char* type() const override { return "reshape"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeSubmit(hipStream_t) override { return Result::SUCCESS; }
    bool launchesKernels() const override { return false; }
    Shape target;
};

// src/domains/core/cast/{module_impl.cc:8-113, module_impl_native_cpu.cc:42-283}: integer sample
// formats -> F32 / CF32 (scalers 128, 32768, 2^31), F32 -> CF32, same-dtype bypass (alias).
class Cast : public Module {
 public:
    const char* type() const override { return "cast"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeSubmit(hipStream_t stream) override;
    bool launchesKernels() const override { return !bypass && !fusedIntoSpectrum; }
    Tensor input, output;
    DataType outputDtype = DataType::None;
    F32 scaler = 1.0f;
    bool bypass = false;
    // Set by the fusion planner (TryFuseSpectrum) when the spectrum unit reads this module's INPUT itself -- raw CI16 /
    // CI8 / CU8 samples converted in the transform's first load -- and nobody else reads the output: the module then
    // launches nothing.  A decision of one runtime; Runtime::planUnits clears it.
    bool fusedIntoSpectrum = false;
};

// src/domains/core/multiply/{module_impl.cc:10-132, module_impl_native_cpu.cc:86-100}
class Multiply : public Module {
 public:
    const char* type() const override { return "multiply"; }
    // Add (core/add) shares the broadcast planning; only names, taint and the op differ.
    virtual const char* tag() const { return "MULTIPLY"; }
    virtual const char* outputPort() const { return "product"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeSubmit(hipStream_t stream) override;
    Tensor a, b, c;  // a, b are the validated broadcast views
    void planStorage(std::set<const void*>& reads, std::set<const void*>&) const override {  // `a` may have been pointed at a duplicate's source
        reads.insert(a.storageId());
        reads.insert(b.storageId());
    }
    Shape outputShape;
};

// src/domains/core/multiply_constant/{module_impl.cc, module_impl_native_cpu.cc:92-100}
class MultiplyConstant : public Module {
 public:
    const char* type() const override { return "multiply_constant"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result reconfigureImpl(const Config& previous) override;
    Result computeSubmit(hipStream_t stream) override;
    Tensor input, output;
    F32 constant = 1.0f;
};

// src/domains/dsp/fft/{module_impl.cc:8-86, module_impl_native_cpu.cc:90-167,
// module_impl_native_cuda.cc:307-519}
class Fft : public Module {
 public:
    const char* type() const override { return "fft"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeInitialize() override;
    Result computeDeinitialize() override;
    Result computeSubmit(hipStream_t stream) override;
    Result layout(dev::FftLayout& L) const;
    // one dense batch of `length`-point transforms, in place in `data` (the Bluestein inner FFTs)
    Result innerTransform(float2* data, U64 length, U64 transforms, bool fwd, hipStream_t stream);
    // the complex transform of length n described by L (cfftp passes or Bluestein)
    Result submitComplex(const dev::FftLayout& L, const float2* in, float2* out, bool fwd,
                         hipStream_t stream);
    // F32 input (fft/module_impl_native_cpu.cc:142-167): r2r_fftpack (halfcomplex) or r2c
    bool realInput = false, complexOut = false;
    Tensor realA, realB, realTw, realLine;  // dense F32 work rows, rfftp twiddles, Bluestein line
    Tensor input, output, scratchA, scratchB, scratchH;
    // which kernels run the passes at the working length (n, or the Bluestein length):
    // register/LDS kernels (2^k <= 16384), the LDS-tiled mixed-radix path, or one launch per pass
    bool useGlobalPasses = false, useTiled = false;
    bool forward = true, complexOutput = false;
    Index resolvedAxis = 0;
    const float2* twiddles = nullptr;
    // Bluestein plan (pocketfft fftblue, pocketfft.hh:2362-2432): chosen exactly when pocketfft_c
    // would (kernels::fft_bluestein_size); bluesteinSize = n2, 0 otherwise.
    U64 bluesteinSize = 0;
    Tensor akf, bk, bkf;
};

// src/domains/dsp/amplitude/{module_impl.cc:8-60, module_impl_native_cpu.cc:73-99}
class Amplitude : public Module {
 public:
    const char* type() const override { return "amplitude"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeSubmit(hipStream_t stream) override;
    Tensor input, output;
    U64 normalizationSize = 1;
    F32 scalingCoeff = 0.0f;
};

// src/domains/core/range/{module_impl.cc:16-62, module_impl_native_cpu.cc:67-82}
class Range : public Module {
 public:
    const char* type() const override { return "range"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result reconfigureImpl(const Config& previous) override;
    Result computeSubmit(hipStream_t stream) override;
    void updateCoefficients();
    Tensor input, output;
    F32 min = -1.0f, max = 1.0f, scalingCoeff = 0.0f, offsetCoeff = 0.5f;
};

// src/domains/visualization/spectrogram/{module_impl.cc:16-114, module_impl_native_cpu.cc:61-87}
// (the reference has no GPU implementation of this module)
class Spectrogram : public Module {
 public:
    const char* type() const override { return "spectrogram"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeSubmit(hipStream_t stream) override;
    const Tensor* state(const std::string& key) const override {
        return key == "frequencyBins" ? &frequencyBins : nullptr;
    }
    Tensor input, frequencyBins;  // state: F32 {width, height}, laid out [height][width]
    void planStorage(std::set<const void*>& reads, std::set<const void*>& writes) const override {
        for (const Tensor* t : {&frequencyBins, &hitCounts, &schedWords, &combineCtrl})
            if (t->valid()) writes.insert(t->storageId());
        if (rowIndices.valid()) reads.insert(rowIndices.storageId());
    }
    // config merge = "counts" (not in the reference, whose display is per process): the module publishes this cycle's
    // integer hit counts as output "counts" (U32 {width, height}, laid out like the state) and leaves the state
    // alone -- the counts of all ranks are summed (RCCL all-reduce) and applied by a spectrogram_merge module.
    bool countsOnly = false;
    Tensor hitCounts;
    U64 height = 256, numberOfElements = 0, numberOfBatches = 0;
    U64 inputElementStride = 0, inputBatchStride = 0;
    F32 decayFactor = 1.0f;
    // Carried by the NEXT cycle's fused spectrum launch (TryFuseSpectrum, kernels::launch_spectrum_spectrogram_fused):
    // the spectrum output is then a ring of two slots, cycle k writes slot k & 1, and this module keeps the count of
    // cycles submitted or replayed that way (host side of the ring) and whether one is still waiting for its
    // spectrogram (the flush at the end of a compute call runs it).
    // Fed with ROW INDICES by the fused spectrum unit that produces its input (TryFuseSpectrum, allow_side): the unit
    // writes, beside every F32 value, the one-byte index this module would derive from it into `rowIndices`
    // (U8, batches * width bytes, TILE-MAJOR [width / 128][batches][128], 0 = no hit), and computeSubmit reads those instead of the values
    // (kernels::launch_spectrogram_index).  Same state, bit for bit; a decision of the runtime's planner, reset by it.
    bool indexFed = false;
    Tensor rowIndices;
    // Device words of the feeding unit's 4096-point kernel (kernels::spectrum_sched_words(): the counters its long launches
    // hand their last rounds out from); owned here because this module is the unit's partner for the lifetime of the plan.
    Tensor schedWords;
    // Cycle batching (Runtime::planBatch): rowIndices is then a ring of as many slots as the source has, and a span of n
    // cycles is ONE launch over n consecutive index tensors with the state tile in registers in between
    // (kernels::launch_spectrogram_index_span; cycle c of the span reads slot (first + c) mod R: one launch whatever the span).
    bool spanCapable() const override { return indexFed && !countsOnly && !combined; }
    Result computeSubmitSpan(hipStream_t stream, U64 first_slot, U64 n) override;
    bool combined = false, combinedPending = false;
    U64 combinedCycle = 0;
    Tensor combineCtrl;  // two zeroed device words {pending, ticket}
    U64 cyclePeriod() const override { return combined ? 2 : 1; }
    void advanceHostState(U64 cycles) override {
        if (!combined || cycles == 0) return;
        combinedCycle += cycles;
        combinedPending = true;
    }
};

// The second half of the exact multi-GPU spectrogram: input "counts" = U32 {width, height} hit counts already
// summed over the ranks; state frequencyBins decays by 0.999^batches (config: the batches of ALL ranks, the `decayFactor`
// of spectrogram/module_impl.cc:104 for the merged batch) and takes min(v + 0.02f, 1.0f) count-times per bin
// (module_impl_native_cpu.cc:61-87 applies it once per hit): bit-identical to one Spectrogram over the union of the batches.
class SpectrogramMerge : public Module {
 public:
    const char* type() const override { return "spectrogram_merge"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    Result computeSubmit(hipStream_t stream) override;
    const Tensor* state(const std::string& key) const override {
        return key == "frequencyBins" ? &frequencyBins : nullptr;
    }
    Tensor counts, frequencyBins;
    U64 totalBatches = 0;
    F32 decayFactor = 1.0f;
};

// src/domains/visualization/waterfall/{module_impl.cc, ring_state.hh:16-56,
// module_impl_native_cpu.cc:53-78, module_impl_native_cuda.cc:18-149}
class Waterfall : public Module {
 public:
    const char* type() const override { return "waterfall"; }
    Result validate() override;
    Result define() override;
    Result create() override;
    // waterfall/module_impl.cc:121-130: `interpolate` (a render option) moves in place, the height does not
    Result reconfigureImpl(const Config& previous) override {
        return ConfigU64(previous, "height", 512) != height ? Result::RECREATE : Result::SUCCESS;
    }
    Result computeSubmit(hipStream_t stream) override;
    const Tensor* state(const std::string& key) const override {
        if (key == "frequencyBins") return &frequencyBins;
        if (key == "ringState") return &ringState;
        return nullptr;
    }
    Tensor input, frequencyBins, ringState;  // ringState: device U64[4]
    void planStorage(std::set<const void*>&, std::set<const void*>& writes) const override {
        if (frequencyBins.valid()) writes.insert(frequencyBins.storageId());
        if (ringState.valid()) writes.insert(ringState.storageId());
    }
    U64 height = 512, numberOfElements = 0, numberOfBatches = 0;
    U64 inputElementStride = 0, inputBatchStride = 0;
};

// Synthetic stand-in for the Soapy source's OUTPUT CONTRACT (src/domains/io/soapy/
// module_impl.cc:197-201): CF32 [batches, samples], batchAxis 0, sampleAxis 1, attributes
// sampleRate / frequency -- backed by an HBM-resident ring of `slots` batches.  Each compute
// cycle exposes the next slot (no copy: downstream kernels read the slot in place).
class RingSource : public Module {
 public:
    const char* type() const override { return "ring_source"; }
    ~RingSource() override;
    Result validate() override;
    Result define() override;
    Result create() override;
    Result destroy() override;
    Result computeSubmit(hipStream_t stream) override;
    void cycleSubmitted(hipStream_t stream) override;
    Result computeDeinitialize() override;
    U64 cyclePeriod() const override { return live ? 1 : slots; }
    void advanceHostState(U64 cycles) override;
    bool launchesKernels() const override { return false; }
    bool capturable() const override { return !live; }  // live: every cycle asks the host-side counters
    Result reconfigureImpl(const Config& previous) override;
    Tensor output;
    DataType sampleType = DataType::CF32;  // config dtype: CF32 | CI16 | CI8 | CU8 (raw SDR sample formats, cast downstream)
    U64 batches = 8, samples = 2048, slots = 1, cursor = 0;
    // live: the host fills slot (published % slots), then raises `published` by reconfigure(); a cycle
    // consumes one published slot, or YIELDs when there is none (io/soapy/module_impl_native_cpu.cc:47-60)
    bool live = false;
    // batches made available so far: raised from the host through reconfigure("published") by a caller that fills the
    // slots itself (round-1 interface), and by the producer interface below (ringPush / ringCommit)
    U64 publishedConfig = 0, pushed = 0, consumed = 0;
    U64 published() const { return publishedConfig + pushed; }
    bool first = true;

    // ---- producer side of a live source: the HBM replacement of the Soapy thread's CircularBuffer --------------------
    // (include/jetstream/tools/circular_buffer.hh:31-48, src/tools/circular_buffer.cc, the 8192-sample pushes of
    // soapy/module_impl.cc:375-399 and the waitForSize of module_impl_native_cpu.cc:39-45.)  Samples arrive in chunks of
    // ANY size; they are assembled into batches in PINNED staging memory (acquire / commit hands the producer the
    // staging memory itself, so a driver can read straight into it; push() is acquire + memcpy + commit); every
    // completed batch goes to ring slot (published % slots) with an asynchronous H2D copy on the source's own upload
    // stream.  The library -- not the caller -- keeps an upload from overwriting a slot whose consuming cycle has not
    // finished: a cycle's completion is an event on the compute stream, recorded BY THE COMPUTE THREAD once every unit of
    // the cycle is enqueued (cycleSubmitted; a cycle that failed half way: when the next one is submitted), and the upload
    // stream waits for it.  A producer that reaches the slot of a cycle still being enqueued waits for that record.  A full ring (every slot published and unconsumed) follows the
    // overflow policy of the reference's buffer: "overwrite" (default, OverwriteOldest: the oldest unconsumed batch is
    // dropped) or "reject" (the push returns INCOMPLETE and nothing of it is taken); both count an overflow.
    Result ringAcquire(void** ptr, U64* max_elements);
    Result ringCommit(U64 elements);
    Result ringPush(const void* samples, U64 elements);
    Result ringWait(U64 elements, U32 timeout_ms);
    Result ringClear();
    U64 ringSize();      // elements published and not yet consumed, plus the partial batch in staging
    U64 ringCapacity() const { return slots * batches * samples; }
    U64 ringOverflows();
    size_t ringElementBytes() const { return elementBytes; }

 private:
    Result ensureProducer();
    Result publishStagedBatch(std::unique_lock<std::mutex>& lock);  // mu held (released while waiting for a cycle to close)
    static constexpr U64 kStaging = 4;
    std::mutex mu;
    std::condition_variable dataAvailable;
    std::condition_variable cycleClosed;  // the compute thread recorded the pending slot's free event
    bool rejectOnOverflow = false;
    hipStream_t uploadStream = nullptr;
    void* staging[kStaging] = {};
    hipEvent_t stagingFree[kStaging] = {};   // the H2D copy out of staging buffer i has finished
    bool stagingBusy[kStaging] = {};
    U64 stagingIndex = 0, stagingFill = 0;   // current staging buffer, elements already in it
    std::vector<hipEvent_t> slotUploaded;    // per ring slot: its latest H2D has finished (compute waits for it)
    std::vector<hipEvent_t> slotFree;        // per ring slot: the cycle that consumed it has finished (uploads wait)
    std::vector<uint8_t> slotUploadValid, slotFreeValid;
    I64 pendingFreeSlot = -1;                // consumed by the latest cycle; its completion event is still to be recorded
    hipStream_t lastComputeStream = nullptr;
    U64 overflowCount = 0;
    U64 clearEpoch = 0;                      // ringClear() calls so far (publishStagedBatch re-validates across its unlocked wait)
    size_t elementBytes = 8;
};

}  // namespace jst::modules
