// kernels.hh -- host-visible launch interface of the gfx950 kernels.  Plain structs and
// functions; the module layer (../modules) is the only caller.
#pragma once

#include "../jst/switches.hh"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace jst::dev {

constexpr int kMaxOuterRank = 7;
constexpr int kMaxRank = 8;

// Addressing of the batch ("outer") axes around the transform axis.  Strides in ELEMENTS of the
// respective tensor (src/memory/tensor.cc:94-109); offsets already include Tensor::offset()
// because a device Tensor::data() does not (tensor.cc:1090-1095).
struct FftLayout {
    uint64_t transforms;  // product of outer shape
    int32_t outer_rank;   // 0..kMaxOuterRank
    uint64_t outer_shape[kMaxOuterRank];
    int64_t in_outer_stride[kMaxOuterRank];
    int64_t out_outer_stride[kMaxOuterRank];
    int64_t in_axis_stride;
    int64_t out_axis_stride;
    uint64_t in_offset;
    uint64_t out_offset;
    // A span of a RING (cycle batching, one batch axis only): transform t of the launch is row (ring_first + t) mod
    // ring_transforms of tensors that hold ring_transforms rows (all the slots of the ring one behind the other), so a span
    // that wraps the ring -- or laps it -- is still ONE launch.  ring_transforms = 0: no ring, transform t is row t.
    uint64_t ring_first;
    uint64_t ring_transforms;
};

// N-ary strided elementwise traversal (the device counterpart of
// include/jetstream/tools/automatic_iterator.hh:108-343): a common shape, per-operand element
// strides (0 on broadcast axes) and element offsets.  Operand 0 is the output.
struct EwLayout {
    uint64_t size;
    int32_t rank;
    int32_t contiguous;  // every operand dense row-major with the common shape
    uint64_t shape[kMaxRank];
    int64_t stride[3][kMaxRank];
    uint64_t offset[3];
};

// FM demodulator coefficients (dsp/fm/module_impl.cc:108-172) and lane addressing.
struct FmCoeffs {
    float ref, pilot_inc, pilot_alpha, deemph_alpha;
    float notch[5];  // b0 b1 b2 a1 a2
    float lp[3][5];
    int wide, deemph_enabled;
};
struct FmLayout {
    uint64_t lanes, batches, samples;
    int32_t lane_rank;
    uint64_t lane_shape[kMaxRank];
    int64_t in_lane_stride[kMaxRank], out_lane_stride[kMaxRank];
    int64_t in_batch_stride, in_sample_stride, out_batch_stride, out_sample_stride,
        out_channel_stride;
    uint64_t in_offset, out_offset;
};

}  // namespace jst::dev

namespace jst::kernels {

using jst::dev::FmCoeffs;
using jst::dev::FmLayout;

using jst::dev::EwLayout;
using jst::dev::FftLayout;

// ---- FFT (fft_kernels.hip) ---------------------------------------------------------------------
bool fft_lds_supported(uint64_t n);
bool fft_fused_supported(uint64_t n);
hipError_t launch_fft_c2c(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                          const float2* in, float2* out, hipStream_t stream);
// General lengths (pocketfft's cfftp for any n >= 2: radix 2/3/4/5/7/8/11 passes plus the generic
// odd radix): one Stockham pass per launch through HBM (fft_global.hip).  scratch_a/b: dense
// CF32[transforms * n] each (b may be null when the plan has at most two passes); scratch_h: one
// more of the same size, needed only when fft_plan_has_generic_radix(n).
int fft_plan_factors(uint64_t n, uint32_t* factors /*[64]*/);
bool fft_global_supported(uint64_t n);
bool fft_plan_has_generic_radix(uint64_t n);
hipError_t launch_fft_c2c_global(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                                 const float2* in, float2* out, float2* scratch_a,
                                 float2* scratch_b, float2* scratch_h, hipStream_t stream);
// LDS-tiled mixed-radix path (fft_tiled.hip): plans with radices <= 11; one kernel when a
// transform fits an LDS tile (n <= 8192), two kernels (columns, then blocks) up to 2^26 points.
// scratch: dense CF32[transforms * n], needed when fft_tiled_may_use_scratch(n).
// NOTE: the tiled launchers take the PER-PASS twiddle table (fft_pass_twiddle_*), not W.
uint64_t fft_pass_twiddle_count(uint64_t n);
void fft_pass_twiddle_fill(uint64_t n, const float* w_interleaved, float* out_interleaved);
bool fft_tiled_supported(uint64_t n);
bool fft_tiled_needs_scratch(uint64_t n);
// true also for lengths that fit one tile but run as columns + blocks kernels when a launch has only a few transforms
bool fft_tiled_may_use_scratch(uint64_t n);
hipError_t launch_fft_c2c_tiled(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                                const float2* in, float2* out, float2* scratch, hipStream_t stream);
// Pad (zeros appended along the transform axis) fused into the first load: `in` is the UNPADDED
// tensor (L.in_* describe it), `valid` its extent along the axis, n the padded transform length.
hipError_t launch_fft_c2c_tiled_windowed(uint64_t n, const FftLayout& L, const float2* W, const float2* in,
                                         const float2* window, int64_t window_stride, float2* out, float2* scratch,
                                         hipStream_t stream);
hipError_t launch_fft_c2c_tiled_padded(uint64_t n, uint64_t valid, bool forward, const FftLayout& L,
                                       const float2* W, const float2* in, float2* out,
                                       float2* scratch, hipStream_t stream);
// The same transform with Multiply -> Fold behind its last pass (filter/block_impl.cc:444-497): bin m of the output
// is the F64 mean of the `n / fold` products spectrum[idx] * h[idx], idx = (m - offset + g * fold) mod n, g ascending
// (dsp/fold/module_impl_native_cpu.cc:103-172); neither the spectrum nor the product is written.  `out` is dense
// [transforms, fold]; `h` is addressed along the transform axis only (an operand broadcast over the transforms).
// chan_offsets (device, optional): per-channel offsets, channel of transform t = (t / chan_div) % chan_count.
struct FoldProductArgs {
    float2* out;
    const float2* h;
    int64_t h_stride;
    uint64_t fold, offset;
    const uint64_t* chan_offsets;
    uint64_t chan_count, chan_div;
    bool spectrum_first;
    // heads > 1: `heads` operand rows h + hd * h_head_stride (one per head, fold offset chan_offsets[hd]) meet ONE
    // spectrum per transform; out is dense [transforms, heads, fold] (the Filter block's multi-head form)
    uint64_t heads = 1;
    int64_t h_head_stride = 0;
};
bool fft_tiled_fold_supported(uint64_t n, uint64_t transforms, uint64_t fold);
hipError_t launch_fft_c2c_tiled_padded_fold(uint64_t n, uint64_t valid, bool forward, const FftLayout& L,
                                            const float2* W, const float2* in, float2* scratch,
                                            const FoldProductArgs& f, hipStream_t s);
// The transform with multiply_constant -> unpad behind its last pass (filter/block_impl.cc:499-560): element `pos` of
// transform t, times `constant` (complex x real), goes to body[t * body_len + pos] when pos < body_len and to
// tail[t * (n - body_len) + pos - body_len] otherwise; t counts the outer axes of L row-major.  L's output side is ignored.
hipError_t launch_fft_c2c_tiled_scaled_unpad(uint64_t n, bool forward, const FftLayout& L, const float2* W,
                                             const float2* in, float2* scratch, float2* body, float2* tail,
                                             float constant, uint64_t body_len, hipStream_t s);
// ... with phase_correction between multiply_constant and unpad (inverse transforms only): `corr` is the module's
// [channels, batches] table; transform t belongs to batch (t / batch_div) % batches, channel (t / chan_div) % channels
hipError_t launch_fft_c2c_tiled_scaled_phase_unpad(uint64_t n, const FftLayout& L, const float2* W, const float2* in,
                                                   float2* scratch, float2* body, float2* tail, float constant,
                                                   uint64_t body_len, const float2* corr, uint64_t batches,
                                                   uint64_t batch_div, uint64_t channels, uint64_t chan_div, hipStream_t s);


// The 4096-point fused spectrum chain with the side output as ONE WAVEFRONT PER TRANSFORM (fft_wave.hip / fft_wave.hh): dense
// CF32 rows in, F32 rows + one-byte row indices out (the arguments of launch_spectrum_fused_side).
bool spectrum_wave_selected();
hipError_t launch_spectrum_wave_side(const FftLayout& L, const float2* W, const float2* in, const float2* window, float* out,
                                     float amp_coeff, float range_scale, float range_offset, bool fast, float guard_h0,
                                     float guard_h1, uint8_t* side, float side_height, uint32_t side_batches,
                                     uint32_t side_pitch, bool real_window, hipStream_t stream);
hipError_t launch_spectrum_fused_tiled(uint64_t n, const FftLayout& L, const float2* W,
                                       const float2* in, const float2* window,
                                       int64_t window_stride, float* out, float amp_coeff,
                                       bool with_range, float range_scale, float range_offset,
                                       bool fast, float guard_h0, float guard_h1, float2* scratch,
                                       hipStream_t stream);
// pocketfft_c's plan choice (pocketfft.hh:2472-2489): 0 = cfftp of n, else the Bluestein
// convolution length n2 = good_size_cmplx(2n-1).  The three elementwise steps of fftblue::fft
// (:2370-2399) around the two n2-point transforms; akf: dense CF32[transforms * n2].
uint64_t fft_bluestein_size(uint64_t n);
uint64_t fft_bluestein_size_scaled(uint64_t n, double direct_cost_factor);

// ---- real-input transforms (rfft.hip): pocketfft rfftp, radices 2/3/4/5 ------------------------
// Dense F32[transforms][n] work rows; tw = rfft_twiddle_fill layout (rfftp::comp_twiddle).
int rfft_plan_factors(uint64_t n, uint32_t* factors /*[64]*/);
bool rfft_supported(uint64_t n);        // plan uses radices <= 5 only (no radfg / radbg)
uint64_t rfft_bluestein_size(uint64_t n);  // pocketfft_r's choice: 0 = rfftp, else n2
uint64_t rfft_twiddle_count(uint64_t n);
void rfft_twiddle_fill(uint64_t n, const float* w_interleaved, float* out);
hipError_t launch_rfft_gather(const FftLayout& L, float* dense, const float* in, uint64_t n, hipStream_t s);
hipError_t launch_rfft_scatter(const FftLayout& L, float* out, const float* dense, uint64_t n,
                               bool complex_out, hipStream_t s);
hipError_t launch_rfft_passes(uint64_t n, bool r2hc, uint64_t transforms, float* a, float* b,
                              const float* tw, float** result, hipStream_t s);
hipError_t launch_rfft_blue_in(float2* c, const float* dense, uint64_t transforms, uint64_t n, bool r2hc,
                               hipStream_t s);
hipError_t launch_rfft_blue_out(float* dense, const float2* c, uint64_t transforms, uint64_t n, bool r2hc,
                                hipStream_t s);
hipError_t launch_bluestein_pre(bool forward, const FftLayout& L, float2* akf, const float2* in,
                                const float2* bk, uint64_t n, uint64_t n2, hipStream_t stream);
hipError_t launch_bluestein_mul(bool forward, float2* akf, const float2* bkf, uint64_t transforms,
                                uint64_t n2, hipStream_t stream);
hipError_t launch_bluestein_post(bool forward, const FftLayout& L, float2* out, const float2* akf,
                                 const float2* bk, uint64_t n, uint64_t n2, hipStream_t stream);
// Multiply(window) -> FFT(forward) -> Amplitude [-> Range] in one pass over HBM.
hipError_t launch_spectrum_fused(uint64_t n, const FftLayout& L, const float2* W,
                                 const float2* in, const float2* window, int64_t window_stride,
                                 float* out, float amp_coeff, bool with_range, float range_scale,
                                 float range_offset, bool fast, float guard_h0, float guard_h1,
                                 hipStream_t stream);
// Cast (CI16 / CI8 / CU8 -> CF32) -> Multiply -> FFT -> Amplitude [-> Range]: raw SDR samples straight into the
// transform's first load.  in_format: 1 = CI16, 2 = CI8, 3 = CU8; scaler: the Cast module's divisor.
hipError_t launch_spectrum_fused_cast(uint64_t n, const FftLayout& L, const float2* W, const void* in, int in_format,
                                      float scaler, const float2* window, int64_t window_stride, float* out,
                                      float amp_coeff, bool with_range, float range_scale, float range_offset, bool fast,
                                      float guard_h0, float guard_h1, hipStream_t stream);
// The same chain with the Spectrogram consumer's row index as a one-byte SIDE OUTPUT (fft_side.hip): the index of
// out[t][x] is (u32)(out[t][x] * height) when that hits (1 <= value * height < height), else 0.  `side` holds
// transforms / side_batches index tensors one behind the other, each TILE-MAJOR U8[n / 128][side_pitch][128] (the 128
// columns of a group for all rows of one compute cycle contiguous: what the column-tiled consumer walks; rows
// [side_batches, side_pitch) are padding -- with side_pitch = side_batches = 1024 the groups lie 128 KiB apart and
// every store of a row lands on the same few HBM channels: spectrum_side_pitch() picks the pad).
uint64_t spectrum_side_pitch(uint64_t batches);
// in_format: 0 = CF32, 1 = CI16, 2 = CI8, 3 = CU8 (scaler as above).  Range is always on; window dense.
// real_window: the caller KNOWS every imaginary part of the window to be +-0; with provider "fast" the Multiply then runs
// as two products per sample (fft_lds.hh: RealOperand), bit-identical for finite input.
bool spectrum_side_supported(uint64_t n, const FftLayout& L, int64_t window_stride, uint64_t height);
hipError_t launch_spectrum_fused_side(uint64_t n, const FftLayout& L, const float2* W, const void* in, int in_format,
                                      float scaler, const float2* window, float* out, float amp_coeff,
                                      float range_scale, float range_offset, bool fast, float guard_h0, float guard_h1,
                                      uint8_t* side, uint64_t height, uint64_t side_batches, uint64_t side_pitch, bool real_window,
                                      hipStream_t stream, uint32_t* sched = nullptr);
// sched: spectrum_sched_words() zeroed U32 words of device memory that belong to the calling unit (launches that share them
// must be ordered on one stream), or null.  With them the 4096-point kernel (fft_quad.hh) hands the last rounds of a long
// launch out dynamically; it leaves the words zeroed behind every launch.
uint64_t spectrum_sched_words();
// guard_h0/h1 (fast + range only): heights of the Spectrogram modules that will quantise the output;
// elements whose value * height lies within the fast path's error of a bin edge are computed with the
// exact arithmetic instead, so the bins equal the exact provider's (dev::BinGuard, device_math.hh).
// Test probe: both epilogues on arbitrary complex inputs (DEVICE pointers).
hipError_t launch_amplitude_range_probe(float* out_exact, float* out_fast, const float2* in, uint64_t count,
                                        float amp_coeff, float range_scale, float range_offset,
                                        float guard_h0, float guard_h1, hipStream_t stream);

// ---- elementwise modules (elementwise.hip) -----------------------------------------------------
hipError_t launch_multiply_cf32(const EwLayout& L, float2* c, const float2* a, const float2* b,
                                hipStream_t stream);
hipError_t launch_multiply_f32(const EwLayout& L, float* c, const float* a, const float* b,
                               hipStream_t stream);
hipError_t launch_amplitude_cf32(const EwLayout& L, float* out, const float2* in, float coeff,
                                 bool fast, hipStream_t stream);
hipError_t launch_amplitude_f32(const EwLayout& L, float* out, const float* in, float coeff,
                                hipStream_t stream);
hipError_t launch_range_f32(const EwLayout& L, float* out, const float* in, float scale,
                            float offset, bool fast, hipStream_t stream);
hipError_t launch_amplitude_range(const EwLayout& L, float* out, const void* in, bool in_is_complex, float coeff, float scale,
                                  float offset, bool fast, hipStream_t stream);
hipError_t launch_multiply_constant_cf32(const EwLayout& L, float2* out, const float2* in,
                                         float constant, hipStream_t stream);
hipError_t launch_multiply_constant_f32(const EwLayout& L, float* out, const float* in,
                                        float constant, hipStream_t stream);
// Invert: flat (row-major) index -> coordinate on the sample axis = (index / inner) % length
// (invert/module_impl_native_cpu.cc:79-103).  Output always CF32.
hipError_t launch_invert(const EwLayout& L, float2* out, const void* in, bool in_is_complex,
                         uint64_t inner, uint64_t length, hipStream_t stream);
// Add (core/add/module_impl_native_cpu.cc:83-98), broadcast like multiply.
hipError_t launch_add_cf32(const EwLayout& L, float2* c, const float2* a, const float2* b,
                           hipStream_t stream);
hipError_t launch_add_f32(const EwLayout& L, float* c, const float* a, const float* b,
                          hipStream_t stream);
// Cast (core/cast/module_impl_native_cpu.cc:137-283): integer sample formats -> F32 / CF32 with
// the reference's scalers (128, 32768, 2^31), and F32 -> CF32 (imag = 0).
enum class CastKind { I8, U8, I16, U16, I32, U32, CI8, CU8, CI16, CU16, CI32, CU32, F32_TO_CF32 };
hipError_t launch_cast(const EwLayout& L, void* out, const void* in, CastKind kind, float scaler,
                       hipStream_t stream);
// AGC (dsp/agc/module_impl_native_cpu.cc:20-160): tiled RMS gain with linear interpolation
// between tiles, all gain arithmetic in F64.  gains: F64 [lanes][tiles][2] scratch (start, end).
struct AgcParams {
    uint64_t lanes, samples, tile, tiles;
    int32_t lane_rank;
    uint64_t lane_shape[jst::dev::kMaxRank];
    int64_t in_lane_stride[jst::dev::kMaxRank], out_lane_stride[jst::dev::kMaxRank];
    int64_t in_sample_stride, out_sample_stride;
    uint64_t in_offset, out_offset;
    double reference, epsilon, min_gain, max_gain, max_gain_change;
};
hipError_t launch_agc(void* out, const void* in, bool complex, double* gains, const AgcParams& p,
                      hipStream_t stream);
// What follows a one-tile-per-lane CF32 AGC in the spectrum_engine block (block_impl.cc:183-217: agc -> amplitude -> range,
// then a Waterfall reads the block's output) in the AGC's own launch: the workgroup that found a lane's gain forms the
// level of every sample it scales (amplitude/module_impl_native_cpu.cc:73-86, range/module_impl_native_cpu.cc:67-82, the
// functions of device_math.hh the standalone kernels use), stores it dense [lanes, samples], and -- with `ring` -- copies
// the row into the Waterfall's ring as waterfall_kernel would (device cursor, last workgroup advances it).
struct AgcTail {
    float* level = nullptr;  // dense [lanes, samples]
    float coeff = 0, scale = 0, offset = 0;
    int fast = 0;
    float* ring = nullptr;       // Waterfall ring [height, samples] (nullptr: no Waterfall rides along)
    uint64_t* ring_state = nullptr;
    uint64_t height = 0;
};
// `out` may be nullptr when nobody but the tail reads the scaled signal.  p.tiles must be 1.
hipError_t launch_agc_tail(void* out, const void* in, double* gains, const AgcParams& p, const AgcTail& tail, hipStream_t stream);
// Window: Blackman taps evaluated in F64 (window/module_impl_native_cpu.cc:20-37).
hipError_t launch_window(float2* out, uint64_t n, hipStream_t stream);
// libm-faithful tanhf sweep helper for the parity tests (out[i] = libm_tanhf(in[i])).
// Squelch: peak[0] = max |x| (NaNs ignored like std::max, complex magnitude = libm hypotf), device scalar
hipError_t launch_peak_abs(float* peak, const void* in, uint64_t count, bool complex, hipStream_t stream);
// ones_tensor: count elements of 1 (elem_bytes 4 = F32, 8 with pair = CF32 (1,0), 8 = F64, 16 = CF64 (1,0))
hipError_t launch_fill_ones(void* out, uint64_t count, int elem_bytes, bool pair, hipStream_t stream);
hipError_t launch_divide_f32(float* x, uint64_t count, float divisor, hipStream_t stream);
hipError_t launch_tanhf_probe(float* out, const float* in, uint64_t count, hipStream_t stream);
// Exhaustive sweeps over all 2^32 float bit patterns (exact_sweep.hip): which = 0 sqrt of the main path, 1 tanhf
// main path, 2 amplitude->range from the power (main + bail-out) vs the general form, 3 amplitude alone, 4 fast
// provider with the bin guard vs exact Spectrogram bins at `height`, 5 the same without the guard.  Synchronous.
hipError_t launch_exact_sweep(int which, float coeff, float scale, float offset, float height,
                              uint64_t* mismatches, uint64_t* visited, uint32_t* first_bad);

// ---- Spectrogram (spectrogram.hip) -------------------------------------------------------------
// bins: F32 [height][width] state, updated in place:
//   bins *= decay; for every (batch, x): f = in*height; if 1 <= f < height: bins[x + (u64)f*width]
//   = min(bins + 0.02, 1)   (spectrogram/module_impl_native_cpu.cc:61-87)
// The fused spectrum kernel of cycle k and the Spectrogram of cycle k - 1 in one launch (fft_kernels.hip): `out` is the
// half of the two-slot output ring this cycle writes, `spec_in` the other half (read only while ctrl[0] != 0), `bins` the
// spectrogram state, ctrl two zero-initialised device words {pending, ticket}.  4096-point dense batches only.
bool spectrum_spectrogram_supported(uint64_t n, const FftLayout& L, int64_t window_stride, uint64_t height);
hipError_t launch_spectrum_spectrogram_fused(const FftLayout& L, const float2* W, const float2* in, const float2* window,
                                             float* out, float amp_coeff, bool with_range, float range_scale,
                                             float range_offset, bool fast, float guard_h0, float guard_h1, float* bins,
                                             const float* spec_in, uint64_t height, float decay, uint32_t* ctrl,
                                             hipStream_t stream);
size_t spectrogram_lds_bytes(uint64_t height);
hipError_t launch_spectrogram(float* bins, const float* in, uint64_t in_offset, uint64_t batches,
                              uint64_t width, uint64_t height, int64_t batch_stride,
                              int64_t elem_stride, float decay, hipStream_t stream);
// The Spectrogram fed with the fused spectrum kernel's one-byte row indices (tile-major U8[width / 128][batches][128],
// 0 = no hit) instead of the values: same state update, a quarter of the bytes, a quarter of the wavefronts.
bool spectrogram_index_supported(uint64_t batches, uint64_t width, uint64_t height);
hipError_t launch_spectrogram_index(float* bins, const uint8_t* idx, uint64_t batches, uint64_t pitch, uint64_t width,
                                    uint64_t height, float decay, hipStream_t stream);
// The same over `cycles` consecutive compute cycles in ONE launch: `cycles` index tensors one behind the other (the
// side output of a fused spectrum launch that carried that many ring slots), the state tile in registers in between.
bool spectrogram_index_span_supported(uint64_t batches, uint64_t width, uint64_t height, uint64_t cycles);
// idx: the FIRST slot of a ring of ring_slots index tensors; the span's cycle c reads slot (first_slot + c) mod ring_slots
// (ring_slots = 0: `cycles` tensors one behind the other from idx on, no wrap).
hipError_t launch_spectrogram_index_span(float* bins, const uint8_t* idx, uint64_t batches, uint64_t pitch, uint64_t width,
                                         uint64_t height, float decay, uint64_t cycles, uint64_t first_slot,
                                         uint64_t ring_slots, hipStream_t stream);
// The exact multi-GPU merge of spectrograms: this cycle's hit COUNTS as a U32[height][width] tensor
// (no state touched) -- all-reduce(sum) them over the ranks -- then one shared decay and the count-times update.
hipError_t launch_spectrogram_counts(uint32_t* counts, const float* in, uint64_t in_offset, uint64_t batches,
                                     uint64_t width, uint64_t height, int64_t batch_stride, int64_t elem_stride,
                                     hipStream_t stream);
hipError_t launch_spectrogram_apply_counts(float* bins, const uint32_t* counts, uint64_t cells, float decay,
                                           hipStream_t stream);

// ---- Waterfall (waterfall.hip) -----------------------------------------------------------------
// ring: F32 [height][width]; state: device u64[4] = {writeIndex, dirtyRows, ticket, pad}.
// waterfall/ring_state.hh:16-56 + module_impl_native_cpu.cc:53-78, cursor kept on the device so
// that the launch is hipGraph-replayable.
hipError_t launch_waterfall(float* ring, uint64_t* state, const float* in, uint64_t in_offset,
                            uint64_t batches, uint64_t width, uint64_t height,
                            int64_t batch_stride, int64_t elem_stride, hipStream_t stream);

hipError_t launch_lineplot(float* points, float* average, const float* in, uint64_t in_offset,
                           uint64_t batches, uint64_t elements, int64_t batch_stride,
                           int64_t elem_stride, uint64_t decimation, float normalization,
                           float averaging, hipStream_t stream);
// `cycles` consecutive compute cycles of a cycle-batched span in one launch: cycle c reads slot (first_slot + c) mod
// ring_slots of the input ring (in_ring = slot 0, slots slot_stride elements apart); the average stays in a register.
hipError_t launch_lineplot_span(float* points, float* average, const float* in_ring, uint64_t in_offset, uint64_t slot_stride,
                                uint64_t first_slot, uint64_t ring_slots, uint64_t cycles, uint64_t batches, uint64_t elements,
                                int64_t batch_stride, int64_t elem_stride, uint64_t decimation, float normalization,
                                float averaging, hipStream_t stream);

// ---- Filter / FM side chains (filter_kernels.hip) ----------------------------------------------
hipError_t launch_pad(void* out, const void* in, bool complex, uint64_t outer, uint64_t in_axis,
                      uint64_t out_axis, uint64_t inner, hipStream_t s);
hipError_t launch_unpad(void* body, void* tail, const void* in, bool complex, uint64_t outer,
                        uint64_t in_axis, uint64_t body_axis, uint64_t inner, hipStream_t s);
hipError_t launch_fold(float* out, const float* in, bool complex, uint64_t outer, uint64_t axis_size,
                       uint64_t fold_size, uint64_t inner, uint64_t scalar_offset,
                       const uint64_t* chan_offsets, uint64_t chan_count, uint64_t chan_inner,
                       hipStream_t s);
// Multiply (broadcast) fused into Fold: out[o, k] = mean_g( a[..] * b[..] ) with the product formed
// exactly like the Multiply module (cmul_full, F32) and accumulated like Fold (F64).  P: the
// multiply's operand layout (operand 0 = the product tensor that is never materialised).
hipError_t launch_fold_product_cf32(float2* out, const float2* a, const float2* b, const EwLayout& P,
                                    uint64_t axis_size, uint64_t fold_size, uint64_t scalar_offset,
                                    const uint64_t* chan_offsets, uint64_t chan_count,
                                    uint64_t chan_inner, hipStream_t s);
// Direct-form polyphase FIR + decimation of a continuous CF32 stream with real taps (fir.hip): the
// time-domain equivalent of the Filter block's FFT overlap-add chain, provider "fast".
bool fir_decimate_supported(uint64_t row_samples, uint64_t taps, uint64_t decimation);
size_t fir_table_floats(uint64_t taps, uint64_t decimation, uint64_t heads);
// table: the taps re-laid out per (polyphase branch, chunk) for scalar loads; depends only on the taps
hipError_t launch_fir_table(float* table, const float2* taps, uint64_t ntaps, uint64_t decimation,
                            uint64_t heads, hipStream_t s);
hipError_t launch_fir_decimate(float2* out, const float2* in, const float* table, float2* history,
                               uint64_t rows, uint64_t row_samples, uint64_t ntaps, uint64_t decimation,
                               uint64_t heads, hipStream_t s);
hipError_t launch_overlap_add(void* out, const void* buf, const void* ovl, void* prev, bool complex,
                              uint32_t rank, int32_t batch_axis, const uint64_t* buf_shape,
                              const uint64_t* ovl_shape, hipStream_t s);
// overlap_add when `out` already holds the buffer (written there by the producer): only the overlap region is touched
// -- out += previous overlap (first batch) / the previous batch's overlap -- and the state takes the last batch's
// overlap, by the thread that read it.
hipError_t launch_overlap_heads(void* out, const void* ovl, void* prev, bool complex, uint32_t rank,
                                int32_t batch_axis, const uint64_t* buf_shape, const uint64_t* ovl_shape,
                                hipStream_t s);
// the same launch also advances a phase_correction state and writes the NEXT cycle's correction table (corr != nullptr)
hipError_t launch_overlap_heads_phase(void* out, const void* ovl, void* prev, bool complex, uint32_t rank,
                                      int32_t batch_axis, const uint64_t* buf_shape, const uint64_t* ovl_shape,
                                      float2* corr, double* phases, const double* increments, uint64_t channels,
                                      uint64_t batches, hipStream_t s);
hipError_t launch_phase_table_prime(float2* corr, double* phases, const double* increments, uint64_t channels,
                                    uint64_t batches, hipStream_t s);

hipError_t launch_phase_correction(const EwLayout& L, float2* out, const float2* in, float2* corr,
                                   double* phases, const double* increments, uint64_t batches,
                                   uint64_t batch_inner, uint64_t channels, uint64_t channel_inner,
                                   hipStream_t s);
hipError_t launch_filter_taps(float2* out, double sample_rate, double bandwidth, const double* center,
                              uint64_t heads, uint64_t taps, hipStream_t s);
// op: 0 add, 1 sub, 2 mul, 3 div; L: operand 0 = output, operand 1 = input indexed over the OUTPUT
// shape (reduced axis extent 1); (r, r_stride) walk the reduced axis
hipError_t launch_arithmetic(const EwLayout& L, void* out, const void* in, bool complex, int op,
                             uint64_t r, int64_t r_stride, hipStream_t s);
// cosine oscillator: phases = F64 scratch [count], state = F64[1] carried phase
// Signal generator (dsp/signal_generator/module_impl_native_cpu.cc:159-375).  state: F64[4] =
// {oscillatorPhase, chirpTime, noise counter (u64 bits), unused}; phases: F64[count] scratch.
enum class SignalShape { Sine, Cosine, Square, Triangle, Sawtooth, Noise, Dc, Chirp };
struct SignalParams {
    SignalShape shape;
    double amplitude, frequency, sample_rate, dc_offset, noise_variance, chirp_start, chirp_end,
        chirp_duration;
};
hipError_t launch_signal_generator(float* out, double* phases, double* state, uint64_t count,
                                   bool complex_out, const SignalParams& p, hipStream_t s);
// AM: envelope + DC blocker, states = lanes x {previous envelope, previous output} (zero = fresh)
size_t am_state_bytes();
hipError_t launch_am(float* out, const float2* in, void* states, float alpha, const FmLayout& L, hipStream_t s);
size_t fm_state_bytes();
hipError_t launch_fm(float* out, const float2* in, void* states, const FmCoeffs& k, const FmLayout& L,
                     hipStream_t s);

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: remember it per (function,
// device), not per process (a process that drives several GPUs raises it once on each).
hipError_t raise_dynamic_lds(const void* kernel, int bytes);

}  // namespace jst::kernels
