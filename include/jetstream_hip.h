/*
 * jetstream_hip.h -- C ABI of libjetstream_hip.so, the MI355X (gfx950) compute backend for
 * CyberEther's Jetstream DSP module graph.
 *
 * The reference's module interface is a C++ virtual ABI inside libjetstream and is not stable
 * across compilers; what crosses a DSO boundary there is ONE C symbol, jetstream_plugin_abi
 * (include/jetstream/plugin.hh:48-87, validated at src/plugin.cc:1190-1224).  This header is the
 * flat equivalent of the calls the framework makes on a module of the hot path, so that a
 * binding (C++, ctypes, cgo...) can drive the HIP modules with plain pointers and sizes:
 *
 *   Registry::BuildModule(type, device, runtime, provider)   include/jetstream/registry.hh:119-125
 *   Module::create(name, config, inputs)                      src/module.cc:47-212
 *   outputs()["port"]                                         include/jetstream/tensor_link.hh:22-34
 *   Runtime::create(modules) / Runtime::compute()             include/jetstream/runtime.hh:22-40
 *   NativeCudaRuntimeContext::compute{Initialize,Submit,Deinitialize}
 *                                                             include/jetstream/runtime_context_native_cuda.hh:32-34
 *   Tensor(device, dtype, shape) / views / attributes / copy  include/jetstream/memory/tensor.hh:24-135
 *
 * Conventions: every call returns a jst_result (the reference's Result enum,
 * include/jetstream/types.hh:19-30); the text of the last error on the calling thread is
 * jst_last_error() (what JST_ERROR logged).  Handles are opaque and owned by the caller until
 * the matching *_destroy.  Tensors handed to a module are shared (ref-counted storage): a
 * module never writes its inputs.  Shapes, strides and offsets are in ELEMENTS.
 * Nothing here needs torch; device pointers may be borrowed from any allocator
 * (jst_tensor_wrap).
 */
#ifndef JETSTREAM_HIP_H
#define JETSTREAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t jst_result; /* include/jetstream/types.hh:19-30 */
enum {
    JST_SUCCESS = 0,
    JST_ERROR_ = 1,
    JST_WARNING = 2,
    JST_FATAL = 3,
    JST_SKIP = 4,
    JST_YIELD = 5,
    JST_RELOAD = 6,
    JST_RECREATE = 7,
    JST_TIMEOUT = 8,
    JST_INCOMPLETE = 9
};

/* include/jetstream/memory/types.hh:22-29; HIP uses the free bit 1<<6. */
enum { JST_DEVICE_NONE = 1 << 0, JST_DEVICE_CPU = 1 << 1, JST_DEVICE_HIP = 1 << 6 };

/* include/jetstream/memory/types.hh (DataType); the integer sample formats feed `cast`. */
enum {
    JST_DTYPE_F32 = 1, JST_DTYPE_CF32 = 2, JST_DTYPE_F64 = 3, JST_DTYPE_U64 = 4,
    JST_DTYPE_I8 = 5, JST_DTYPE_CI8 = 6, JST_DTYPE_I16 = 7, JST_DTYPE_CI16 = 8, JST_DTYPE_U8 = 9,
    JST_DTYPE_CU8 = 10, JST_DTYPE_U16 = 11, JST_DTYPE_CU16 = 12, JST_DTYPE_I32 = 13,
    JST_DTYPE_CI32 = 14, JST_DTYPE_U32 = 15, JST_DTYPE_CU32 = 16, JST_DTYPE_CF64 = 17
};

/* Runtime flags. */
enum {
    JST_RUNTIME_GRAPH = 1 << 0,  /* capture the steady-state cycle(s) into a hipGraph */
    JST_RUNTIME_FUSE = 1 << 1,   /* submit Multiply->FFT->Amplitude[->Range] as one kernel */
    JST_RUNTIME_TIMING = 1 << 2, /* hipEvent pair around every execution unit */
    JST_RUNTIME_PIPELINE = 1 << 3, /* with GRAPH: SURFACE units (spectrogram, waterfall, lineplot) run on
                                     a second captured stream beside the NEXT cycle's producers; the
                                     tensors they read are double-buffered */
    JST_RUNTIME_COMBINE = 1 << 4, /* with FUSE: a Spectrogram that is the only reader of the fused spectrum unit's
                                     output rides on the NEXT cycle's spectrum launch (one kernel per cycle, output a
                                     ring of two slots); the one still waiting when a compute call ends is run then,
                                     so results and their visibility after jst_runtime_compute are unchanged */
    JST_RUNTIME_BATCH = 1 << 5    /* with GRAPH and FUSE: CYCLE BATCHING.  When the chain is a resident ring_source
                                     (R slots) -> fused spectrum unit -> ONE index-fed Spectrogram, the cycles of a
                                     captured ring period -- and of every span of it, wrapping or lapping the ring
                                     included -- run as ONE launch per unit: the persistent spectrum kernel takes the
                                     transforms of all the span's slots (its ramp, cold start and tail are paid once
                                     per launch, not once per cycle), the range output and the row indices become
                                     rings of R slots (cycle c writes slot c mod R; the tensor handles show the latest
                                     cycle), and the Spectrogram walks the span's index tensors in one launch with its
                                     state tile in registers.  The LDS-tiled spectrum unit (beyond 16384 points) and the
                                     Lineplot have span forms too; a waterfall on the same output runs its per-cycle
                                     submissions behind the span's launches.  What is visible after
                                     jst_runtime_compute is bit-identical to the per-cycle submissions; any other
                                     chain silently stays per cycle (jst_runtime_batched tells). */
};

typedef struct jst_tensor_s* jst_tensor;
typedef struct jst_module_s* jst_module;
typedef struct jst_runtime_s* jst_runtime;
typedef struct jst_comm_s* jst_comm;

/* Tensors of up to 8 axes.  A NARROWING of the reference's tensor contract, on purpose: its fast iterators take rank <= 16
 * (include/jetstream/tools/automatic_iterator.hh:182) and the generic path any rank; nothing on the named path -- Soapy
 * [B, N], the Filter chain's [B, heads, N], FM's [B, lanes.., S, 2], the reference tests' rank-4 non-contiguous cases --
 * goes beyond rank 4, the descriptor below is passed by value across the ABI, and the device-side layouts (EwLayout,
 * FftLayout: kernel arguments) are sized by it.  jst_tensor_create / _wrap / _reshape / _expand_dims refuse more axes
 * with JST_ERROR and a message in jst_last_error(). */
#define JST_MAX_RANK 8

/* The fields a module reads from a Tensor (tensor.hh:56-79, axis.hh:19-23). axis = -1: unset. */
typedef struct jst_tensor_desc {
    void* data;       /* buffer base of the selected ring slot (device tensors: NOT offset) */
    uint64_t offset;  /* elements */
    uint8_t dtype;
    uint8_t device;
    uint32_t rank;
    uint64_t shape[JST_MAX_RANK];
    uint64_t stride[JST_MAX_RANK];
    int64_t sample_axis, batch_axis, channel_axis;
} jst_tensor_desc;

/* ---- library ---------------------------------------------------------------------------- */
/* The plugin handshake symbol the reference's loader looks up (plugin.hh:48-87).  HANDSHAKE ONLY: Plugin::load validates
 * this record and then drains the plugin's JST_REGISTER_MODULE queues (src/plugin.cc:1131-1137) -- C++ objects this C
 * library cannot fill, so a reference build that dlopens the file as a plugin finds a valid plugin with ZERO
 * registrations.  Modules reach the reference through the bindings of INTEGRATION.md sections 2-3 (translation units
 * compiled INTO the reference, or into a plugin of its own, which call the jst_* entry points below). */
typedef struct JetstreamPluginAbi {
    uint32_t magic;   /* 0x4a535450 "JSTP" */
    uint32_t size;    /* sizeof(JetstreamPluginAbi) */
    uint32_t abi_version; /* 1 */
} JetstreamPluginAbi;
extern const JetstreamPluginAbi jetstream_plugin_abi;

const char* jst_version(void);
const char* jst_last_error(void);
const char* jst_result_name(jst_result r);
/* Number of visible HIP devices (0 without a GPU); jst_device_set selects one per process. */
int jst_device_count(void);
jst_result jst_device_set(int ordinal);
/* Registry::ListAvailableModules: writes "type|device|runtime|provider" lines. Returns count. */
size_t jst_registry_list(char* buffer, size_t capacity);

/* ---- tensors (src/memory/tensor.cc) ----------------------------------------------------- */
jst_result jst_tensor_create(uint8_t device, uint8_t dtype, uint32_t rank, const uint64_t* shape,
                             jst_tensor* out);
jst_result jst_tensor_create_ring(uint8_t device, uint8_t dtype, uint32_t rank,
                                  const uint64_t* shape, uint64_t slots, jst_tensor* out);
/* Borrow external memory (e.g. a torch tensor's data_ptr).  stride may be NULL (dense). */
jst_result jst_tensor_wrap(void* ptr, size_t bytes, uint8_t device, uint8_t dtype, uint32_t rank,
                           const uint64_t* shape, const uint64_t* stride, uint64_t offset,
                           jst_tensor* out);
/* Move the STORAGE of t -- and every view of it, e.g. the output tensor a module handed out -- onto caller-owned memory
 * of at least the same size (no ownership, contents dropped).  For a host framework that has already allocated the
 * buffer a module is to write into: the reference's Impl::create() allocates `output` before a device binding gets a
 * say (Tensor::create(device(), ..), e.g. fft/module_impl.cc:80-83; integration/device_hip/).  Single-slot storage,
 * before the first compute. */
jst_result jst_tensor_rebind(jst_tensor t, void* ptr, size_t bytes);
jst_result jst_tensor_clone(jst_tensor t, jst_tensor* out); /* new view, shared storage */
/* A new handle on `base`'s STORAGE with a geometry of the caller's (elements; stride NULL = dense): a reference Tensor copy
 * after slice / permute / broadcastTo (src/memory/tensor.cc:196-306), e.g. the input a consumer module sees.  Unlike
 * jst_tensor_wrap the storage identity -- what the runtime derives its data-flow edges and fusions from -- is kept.
 * Attributes are copied from `base`.  Bounds are checked against one ring slot. */
jst_result jst_tensor_view(jst_tensor base, uint32_t rank, const uint64_t* shape, const uint64_t* stride, uint64_t offset,
                           jst_tensor* out);
jst_result jst_tensor_destroy(jst_tensor t);
jst_result jst_tensor_describe(jst_tensor t, jst_tensor_desc* out);
jst_result jst_tensor_ring_select(jst_tensor t, uint64_t slot);
/* views, mutating the handle (tensor.cc:196-306) */
jst_result jst_tensor_reshape(jst_tensor t, uint32_t rank, const uint64_t* shape);
jst_result jst_tensor_expand_dims(jst_tensor t, uint64_t axis);
jst_result jst_tensor_squeeze_dims(jst_tensor t, uint64_t axis);
jst_result jst_tensor_slice(jst_tensor t, uint64_t axis, uint64_t begin, uint64_t end,
                            uint64_t step);
jst_result jst_tensor_permute(jst_tensor t, uint32_t rank, const uint64_t* axes);
jst_result jst_tensor_broadcast_to(jst_tensor t, uint32_t rank, const uint64_t* shape);
/* attributes: "sampleAxis" | "batchAxis" | "channelAxis" (Index), "sampleRate", ... (F64) */
jst_result jst_tensor_set_attribute_u64(jst_tensor t, const char* key, uint64_t value);
jst_result jst_tensor_set_attribute_f64(jst_tensor t, const char* key, double value);
/* vector attributes: "channelOffsets" (vector<U64>), "channelPhaseIncrements" / "center" (vector<F64>) */
jst_result jst_tensor_set_attribute_u64v(jst_tensor t, const char* key, const uint64_t* values,
                                         uint64_t count);
jst_result jst_tensor_set_attribute_f64v(jst_tensor t, const char* key, const double* values,
                                         uint64_t count);
jst_result jst_tensor_remove_attribute(jst_tensor t, const char* key);
/* scalar read-back (Index attributes widen to double); values[0..count) for the vector kinds with *count
 * set to the stored length (pass capacity in *count).  JST_ERROR when the key is absent: the metadata a
 * module publishes ("frequency", "sampleRate", "center", ...) is part of its output contract. */
jst_result jst_tensor_get_attribute_f64v(jst_tensor t, const char* key, double* values, uint64_t* count);
/* dense copies, synchronous on return (tensor.cc:882-963) */
jst_result jst_tensor_copy_from_host(jst_tensor t, const void* src, size_t bytes);
jst_result jst_tensor_copy_to_host(jst_tensor t, void* dst, size_t bytes);
/* Tensor::copyFrom(source, context) (include/jetstream/memory/tensor.hh:52): dense device-to-device copy of `src` into `dst`
 * (same byte size), enqueued on `hip_stream` (NULL: the null stream) -- e.g. a trace taken out of a module's state before a
 * collective overwrites it in place. */
jst_result jst_tensor_copy(jst_tensor dst, jst_tensor src, void* hip_stream);
/* asynchronous H2D on the library's side stream (pinned source recommended); the returned
 * work is ordered before the next jst_runtime_compute of any runtime via an event. */
jst_result jst_tensor_copy_from_host_async(jst_tensor t, const void* src, size_t bytes);

/* ---- modules (src/module.cc, include/jetstream/registry.hh) ------------------------------ */
/* config: "key=value" strings (the reference's Parser::Map of the module's JST_MODULE_PARAMS).
 * inputs: parallel arrays of port names and tensors. */
jst_result jst_module_create(const char* type, uint8_t device, const char* provider,
                             const char* name, const char* const* config, uint32_t n_config,
                             const char* const* input_ports, const jst_tensor* input_tensors,
                             uint32_t n_inputs, jst_module* out);
jst_result jst_module_destroy(jst_module m);
/* Module::reconfigure (src/module.cc:233-290, Module::Impl::reconfigure detail/module_impl.hh:47): overlay
 * "key=value" entries on the module's configuration.  JST_SUCCESS: applied in place (or unchanged);
 * JST_RECREATE (7): valid, but this module cannot absorb the change without being rebuilt -- nothing was
 * changed; JST_ERROR: rejected by validate(), nothing was changed.  validate_only != 0 stops after the
 * validation.  A runtime holding the module re-captures its hipGraph on the next compute(). */
jst_result jst_module_reconfigure(jst_module m, const char* const* config, uint32_t n_config, int validate_only);
jst_result jst_module_output(jst_module m, const char* port, jst_tensor* out);
/* internal state tensors: spectrogram/waterfall "frequencyBins", waterfall "ringState" */
jst_result jst_module_state(jst_module m, const char* key, jst_tensor* out);
uint64_t jst_module_taint(jst_module m);
/* Module::Timing (include/jetstream/module.hh:25-31) */
jst_result jst_module_timing(jst_module m, uint64_t* cycles, double* compute_time_ms);
/* direct runtime-context hooks, for harnesses that own the stream */
jst_result jst_module_compute_initialize(jst_module m);
jst_result jst_module_compute_submit(jst_module m, void* hip_stream);
jst_result jst_module_compute_deinitialize(jst_module m);

/* ---- live ring source: the producer side ------------------------------------------------------
 * What the Soapy thread does with its host CircularBuffer in the reference -- push chunks of at most 8192 samples
 * (src/domains/io/soapy/module_impl.cc:375-399) into tools/circular_buffer.hh:31-48, which the compute thread pops one
 * batch at a time after waitForSize (soapy/module_impl_native_cpu.cc:39-60) -- against a `ring_source{live=true}`
 * module whose ring lives in HBM.  `count` / `size` are in ELEMENTS of the source's sample format (config dtype:
 * CF32 | CI16 | CI8 | CU8; one element = one complex sample).  Chunks of any size are assembled into batches in pinned
 * staging memory; every completed batch is uploaded asynchronously (a stream of the source's own) into ring slot
 * published % slots and becomes the input of one compute cycle (the source YIELDs when none is ready).  The LIBRARY
 * keeps an upload from landing in a slot whose consuming cycle has not finished (per-slot events; the caller needs no
 * synchronise of its own).  A full ring follows config overflow = "overwrite" (default, CircularBuffer's
 * OverwriteOldest: the oldest unconsumed batch is dropped) or "reject" (JST_INCOMPLETE, nothing of the push taken);
 * either way jst_ring_overflows counts it.  Thread-safe against the compute thread.
 *   jst_ring_push     circular_buffer.hh push(): copies `count` elements out of caller memory (any memory)
 *   jst_ring_acquire / jst_ring_commit   zero-copy form: the producer (a driver's readStream) writes up to *max_count
 *                     elements straight into the pinned staging memory at *ptr, then commits how many it wrote
 *   jst_ring_wait     waitForSize(): blocks until `size` elements are buffered (JST_SUCCESS) or JST_TIMEOUT
 *   jst_ring_size / _capacity / _overflows / _clear   size() (published, unconsumed batches + the partial batch),
 *                     capacity() (slots x batch), overflows(), clear() */
jst_result jst_ring_push(jst_module source, const void* samples, uint64_t count);
jst_result jst_ring_acquire(jst_module source, void** ptr, uint64_t* max_count);
jst_result jst_ring_commit(jst_module source, uint64_t count);
jst_result jst_ring_wait(jst_module source, uint64_t size, uint32_t timeout_ms);
jst_result jst_ring_clear(jst_module source);
uint64_t jst_ring_size(jst_module source);
uint64_t jst_ring_capacity(jst_module source);
uint64_t jst_ring_overflows(jst_module source);

/* ---- block plans ----------------------------------------------------------------------------- */
/* The Filter block's plan (src/domains/dsp/filter/block_impl.cc:40-168, CalculateCandidatePlan): convolution size,
 * whether the block resamples by spectral folding, the pad size after resampling and one fold offset per head.  A
 * consumer wiring the block's modules itself (see the runtime notes below) configures pad / fold / unpad from it.
 * sample_rate, bandwidth and center are F32 like the block's config (filter/block.hh); offsets: `heads` entries
 * (all 0 when the plan does not resample). */
typedef struct jst_filter_plan_desc {
    uint64_t pad_size, convolution_size, resampler_size;
    int32_t resample;
    float resampled_sample_rate;
} jst_filter_plan_desc;
jst_result jst_filter_plan(float sample_rate, float bandwidth, const float* center, uint64_t centers,
                           uint64_t taps, uint64_t heads, uint64_t signal_size, jst_filter_plan_desc* plan,
                           uint64_t* offsets);

/* ---- runtime (src/runtime/native/cuda/impl.cc + src/scheduler_synchronous.cc) ------------ */
/* Blocks are wiring: a C/C++ consumer of this header creates the MODULES a reference block expands to and hands
 * them to one runtime (any order: the runtime orders them by data flow).  With JST_RUNTIME_FUSE the runtime
 * replaces these module patterns by one kernel each -- only when the intermediates have no other consumer, the
 * modules share a provider and follow each other in the data-flow order; otherwise the modules run one by one,
 * with the same results:
 *   spectrum_engine (spectrum_engine/block_impl.cc:120-217):
 *       window{size=N} -> invert -> reshape{shape=[1,..,N]}            (STATIC: run once, then settled)
 *       multiply{a = signal CF32[.., N], b = that window}  ->  fft{forward=true}  ->  amplitude  [-> range]
 *     fused unit "spectrum_fused(multiply+fft+amplitude[+range])": needs multiply.b STATIC and broadcast along every
 *     axis but the sample axis, fft forward on CF32, N a power of two in [256, 16384] or any length whose pocketfft
 *     plan uses radices <= 11 (tiled kernels); a Spectrogram consuming the range output passes its height to the
 *     "fast" provider's bin guard.  A cast{CI16 | CI8 | CU8 -> CF32} whose output feeds only that multiply (the block's
 *     own leading cast_input) joins the unit -- "spectrum_fused(cast+multiply+..)": the kernel reads the raw samples and
 *     converts them in the transform's first load (N <= 16384); the cast module then launches nothing.
 *     When exactly ONE spectrogram{height <= 256} quantises the whole range output (dense {batches, n} rows,
 *     n = 1024 .. 8192), the unit -- then named "spectrum_fused(..)+indices" -- also writes, beside every F32 value, the
 *     one-byte row index that module derives from it ((u32)(value * height), 0 = no hit), and the spectrogram reads
 *     those instead of re-reading the values: same state bit for bit, every other reader of the output unaffected
 *     (spectrogram/module_impl_native_cpu.cc:61-87; JST_NO_SPECTROGRAM_SIDE=1 keeps the value path).
 *   spectrogram{merge=counts} -> [all-reduce of the U32 counts over the ranks] -> spectrogram_merge{batches=total}: the
 *     exact multi-GPU display (not in the reference, whose display is per process): bit-identical to one spectrogram
 *     over the union of the batches.
 *   filter (filter/block_impl.cc:350-582), per the plan of CalculateCandidatePlan (:40-168):
 *       pad -> fft                      fused: the zeros are synthesised in the transform's first load
 *       multiply{fft output, taps spectrum} -> fold     fused: the broadcast product is never materialised
 *       pad -> fft -> multiply -> fold  ONE unit "fft_padded_fold(..)" when the transform runs on the LDS-tiled kernels
 *                                       (mixed-radix length), the taps spectrum is broadcast over the transforms
 *                                       (any number of heads: one operand row and one fold offset per head) and
 *                                       the fold's aliases fit one workgroup: neither the spectrum nor the
 *                                       product is written
 *       fft{forward=false} -> multiply_constant [-> phase_correction] -> unpad -> overlap_add     ONE unit
 *                                       "ifft_unpad_overlap(..)" / "ifft_phase_unpad_overlap(..)" (tiled transform): scale,
 *                                       phase correction and body / tail split on the transform's last store, one small
 *                                       kernel for the overlap region + state (+ the next cycle's correction table)
 *       multiply_constant [-> phase_correction] -> unpad -> overlap_add as modules otherwise
 *   around an AGC (spectrum_engine with enableAgc): multiply -> fft "fft_windowed(..)", and
 *       agc (one tile per lane) -> amplitude -> range [-> waterfall]  ONE unit "agc_amplitude_range[_waterfall](..)";
 *       amplitude -> range on their own: "amplitude_range(..)"
 *   duplicate (the dense copy behind a `slice` block's view) whose every reader walks strides itself (the Multiply of an
 *       "fft_windowed(..)", an fm): "<name>(elided)" -- the readers take the view, the copy's output tensor is not written
 *     provider "fast" on a single head centred on 0 Hz replaces the whole chain by fir_taps + fir_decimate.
 * jst_runtime_units reports what was fused. */
jst_result jst_runtime_create(const jst_module* modules, uint32_t n, uint32_t flags,
                              jst_runtime* out);
jst_result jst_runtime_destroy(jst_runtime r);
/* run `cycles` compute cycles; sync != 0 ends with a stream synchronise */
jst_result jst_runtime_compute(jst_runtime r, uint64_t cycles, int sync);
jst_result jst_runtime_synchronize(jst_runtime r);
void* jst_runtime_stream(jst_runtime r);
uint64_t jst_runtime_period(jst_runtime r);
int jst_runtime_graph_active(jst_runtime r);
/* newline-separated names: execution order of modules / execution units after fusion */
size_t jst_runtime_order(jst_runtime r, char* buffer, size_t capacity);
size_t jst_runtime_units(jst_runtime r, char* buffer, size_t capacity);
/* mean device milliseconds of a unit (prefix match on its name), JST_RUNTIME_TIMING only */
double jst_runtime_unit_mean_ms(jst_runtime r, const char* unit_prefix);
/* mean duration of an EMPTY hipEvent pair recorded in the same graph (cost of the measurement
 * itself; < 0 when the runtime has no kernel-less dynamic unit to carry it) */
double jst_runtime_event_overhead_ms(jst_runtime r);
/* mean number of compute cycles ONE timing sample of that unit covers: 1, or the ring period for a cycle-batched
 * runtime (its timed launches carry a whole period each); < 0 without samples */
double jst_runtime_unit_mean_cycles(jst_runtime r, const char* unit_prefix);
/* 1 when JST_RUNTIME_BATCH took effect (the planner could batch every dynamic unit), else 0 */
int jst_runtime_batched(jst_runtime r);
/* number of parallel branches a captured cycle is laid out on (1 = one serial chain, the default).  With
 * JST_RUNTIME_MAX_BRANCHES=n independent chains behind one source -- the reference's multi-fm.yml -- are captured as a hipGraph
 * with forks and joins instead of one chain (the reference's scheduler, src/scheduler_synchronous.cc:574-696, runs them one
 * after the other); measured slower on ROCm 7.2 (cross-branch edges cost more than the overlap returns), hence opt-in. */
int jst_runtime_branches(jst_runtime r);
jst_result jst_runtime_reset_timing(jst_runtime r);

/* ---- collectives: one communicator per process (one process per GPU), RCCL over xGMI ----------------
 * The reference has no multi-device code; a C++ host that runs one device ordinal per process
 * (include/jetstream/backend/config.hh:39-41) reaches the path's two cross-rank exchanges here, without torch:
 *   - the exact multi-GPU Spectrogram: `spectrogram{merge=counts}` writes this rank's U32[H, N] hit counts,
 *     jst_comm_allreduce(counts, JST_COMM_SUM) sums them in place, `spectrogram_merge` applies them (bit-exact);
 *   - BASELINE config 5's averaged spectrum: jst_comm_allreduce(lineplot average F32[N], JST_COMM_SUM, average = 1)
 *     once per reporting interval (256 KiB at N = 65536: latency bound -- one collective per interval, never per cycle).
 * RCCL is dlopen'ed on first use; world = 1 needs no RCCL and reduces nothing.  The 128-byte id is ncclUniqueId: rank 0
 * creates it (jst_comm_unique_id) and ships it to the other ranks over whatever the host already has (MPI, a file, a
 * TCP store ...); every rank then calls jst_comm_init with the same bytes.  The all-reduce runs in place on the
 * tensor's own HBM on `hip_stream` (e.g. jst_runtime_stream(rt)): no host round trip. */
enum { JST_COMM_SUM = 0, JST_COMM_MAX = 1 };
#define JST_COMM_ID_BYTES 128
int jst_comm_available(void); /* 1 when librccl.so can be loaded in this process */
jst_result jst_comm_unique_id(uint8_t* id128);
jst_result jst_comm_init(uint32_t rank, uint32_t world, const uint8_t* id128 /* may be NULL when world == 1 */,
                         jst_comm* out);
jst_result jst_comm_destroy(jst_comm c);
uint32_t jst_comm_rank(jst_comm c);
uint32_t jst_comm_world(jst_comm c);
uint64_t jst_comm_calls(jst_comm c);  /* all-reduces issued so far */
int jst_comm_uses_rccl(jst_comm c);   /* 1 when the communicator holds an RCCL communicator (world > 1) */
/* dense F32 or U32 HIP tensor, in place; average: F32 sums only, divided by the world size behind the reduce */
jst_result jst_comm_allreduce(jst_comm c, jst_tensor t, int op, int average, void* hip_stream);

/* ---- test/bench probes -------------------------------------------------------------------- */
/* The A/B switches the differential tests and the benches flip inside one process: JST_FFT_KERNEL (slot | pipe | wave | quad:
 * kernel family of the fused spectrum unit), JST_QUAD_STATIC, JST_FM_SERIAL, JST_RUNTIME_MAX_BRANCHES (n), JST_RUNTIME_NO_BATCH,
 * JST_RUNTIME_EAGER_SPANS, JST_RUNTIME_NO_SPANS.  The environment variable of the same name is read ONCE (the first time the
 * library asks); from then on only this call changes a switch -- value NULL or "" unsets it.  JST_ERROR for an unknown name.
 * Nothing on a launch path calls getenv. */
jst_result jst_debug_set(const char* name, const char* value);
/* Host twiddle generator used for the FFT tables: W[k] = exp(+j 2 pi k/n), interleaved. */
jst_result jst_fft_twiddles(uint64_t n, float* interleaved_out);
/* Which kernel family a complex transform of length n (the pass length: Bluestein sizes resolved by the caller) runs on:
 * JST_FFT_PATH_REGISTER (fft_lds.hh, one workgroup per transform in registers + LDS), _TILE (fft_tiled.hip, one kernel),
 * _TILE_PAIR (fft_tiled.hip, columns + blocks kernels), _PASSES (fft_global.hip, one launch per pass through HBM). */
#define JST_FFT_PATH_REGISTER 0
#define JST_FFT_PATH_TILE 1
#define JST_FFT_PATH_TILE_PAIR 2
#define JST_FFT_PATH_PASSES 3
int jst_probe_fft_path(uint64_t n);
/* The Soapy-shaped producer loop in native code: `count` elements of the source's sample format pushed as consecutive
 * jst_ring_push calls of at most `chunk` elements (soapy/module_impl.cc:375-399).  bench.py: host_fed.push_8192. */
jst_result jst_probe_ring_push_chunks(jst_module source, const void* samples, uint64_t count, uint64_t chunk);
/* out[i] = device restatement of libm tanhf(in[i]); both DEVICE pointers. */
jst_result jst_probe_tanhf(const float* in_device, float* out_device, uint64_t count);
/* The fused Amplitude -> Range epilogues on arbitrary complex inputs (interleaved re,im; DEVICE
 * pointers): out_exact = the generic provider's arithmetic, out_fast = the fast provider's with the
 * Spectrogram bin guard for heights guard_h0 / guard_h1 (0 = none).  Test hook for the guarantee that
 * trunc(out_fast * height) == trunc(out_exact * height). */
jst_result jst_probe_amplitude_range(const float* in_device, float* out_exact_device, float* out_fast_device,
                                     uint64_t count, float amplitude_coeff, float range_scale,
                                     float range_offset, float guard_h0, float guard_h1);

/* Exhaustive device sweep over all 2^32 float bit patterns of the one-float functions behind the exact
 * Amplitude -> Range epilogue (everything from the power re^2+im^2 on): which = 0 the main path's sqrt vs the
 * correctly rounded one, 1 its tanhf vs the class-ladder restatement of glibc's, 2 amplitude->range (main path
 * + bail-out) vs the general form, 3 amplitude alone, 4 provider "fast" WITH the Spectrogram bin guard: bin at
 * `height` equal to the exact provider's on every power, 5 the same WITHOUT the guard (bins move: shows the
 * sweep can tell).  Returns the number of mismatching arguments, how many the main path answered itself, and
 * the first mismatching bit pattern.  Synchronous; ~0.1 s per sweep on MI355X. */
jst_result jst_probe_exact_sweep(int which, float amplitude_coeff, float range_scale, float range_offset,
                                 float height, uint64_t* mismatches, uint64_t* visited, uint32_t* first_bad);

#ifdef __cplusplus
}
#endif
#endif /* JETSTREAM_HIP_H */
