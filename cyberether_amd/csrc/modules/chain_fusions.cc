// chain_fusions.cc -- planner patterns for the small chains that sit AROUND the headline fusion (round 5).
//
// A flowgraph such as the reference's multi-fm.yml (examples/flowgraphs/multi-fm.yml) is launch bound: 8 x 8000 samples per
// cycle and ~30 kernels, most of them at the 4-5 us launch floor.  Its spectrum chains carry an AGC between the transform and
// the Amplitude (spectrum_engine/block_impl.cc:183-197), so the `multiply -> fft -> amplitude -> range` unit of
// modules.cc does not form; what still fuses without changing a bit:
//   multiply(window) -> fft(forward)   the product formed in the transform's first load (LDS-tiled kernels), the spectrum
//                                       stored as CF32 for whoever reads it next                       (TryFuseMultiplyFft)
//   amplitude -> range                  one elementwise pass, the level in a register in between     (TryFuseAmplitudeRange)
// (the AGC itself is one launch when a lane is one tile: kernels/agc.hip).  Same contract as TryFuseSpectrum: the
// intermediates have no other reader and are not written.
#include <cstdlib>
#include <cstring>

#include "modules.hh"
#include "../kernels/kernels.hh"

namespace jst::modules {

using dev::EwLayout;
using dev::FftLayout;

namespace {

bool sole_reader(const std::vector<Module*>& ordered, const Tensor& t, const Module* consumer) {
    for (const Module* m : ordered) {
        if (m == consumer) continue;
        if (const auto* c = dynamic_cast<const Cast*>(m); c && c->bypass) continue;  // a pure alias reads nothing
        for (const auto& kv : m->inputs())
            if (kv.second.storageId() == t.storageId()) return false;
    }
    return true;
}

template <class T>
T* dptr(const Tensor& t) {
    return static_cast<T*>(t.data());
}

}  // namespace

bool TryFuseAmplitudeRange(const std::vector<Module*>& ordered, size_t at, std::string& name, std::vector<Module*>& members,
                           std::function<Result(hipStream_t)>& submit, size_t& consumed) {
    static const bool off = std::getenv("JST_NO_CHAIN_FUSION") != nullptr;
    if (off || at + 1 >= ordered.size()) return false;
    auto* amp = dynamic_cast<Amplitude*>(ordered[at]);
    auto* rng = dynamic_cast<Range*>(ordered[at + 1]);
    if (!amp || !rng || rng->input.storageId() != amp->output.storageId()) return false;
    if (!sole_reader(ordered, amp->output, rng)) return false;
    // the Range reads exactly what the Amplitude writes: same dense layout, no offset games in between
    if (!amp->output.contiguous() || !rng->input.contiguous() || amp->output.offset() != rng->input.offset() ||
        amp->output.shape() != rng->input.shape() || rng->output.shape() != amp->input.shape())
        return false;
    const bool fast = amp->provider() == "fast";
    if ((rng->provider() == "fast") != fast) return false;  // one arithmetic flavour per kernel
    if (amp->input.dtype() != DataType::CF32 && (amp->input.dtype() != DataType::F32 || fast)) return false;
    members = {amp, rng};
    consumed = 2;
    name = "amplitude_range(" + amp->name() + "+" + rng->name() + ")";
    submit = [amp, rng, fast](hipStream_t stream) -> Result {
        EwLayout L;
        if (!MakeEwLayout(rng->output, &amp->input, nullptr, L)) return Result::ERROR;
        return hip_result(kernels::launch_amplitude_range(L, dptr<float>(rng->output), amp->input.data(),
                                                          amp->input.dtype() == DataType::CF32, amp->scalingCoeff,
                                                          rng->scalingCoeff, rng->offsetCoeff, fast, stream),
                          "amplitude + range kernel");
    };
    return true;
}

bool TryFuseMultiplyFft(const std::vector<Module*>& ordered, size_t at, std::string& name, std::vector<Module*>& members,
                        std::function<Result(hipStream_t)>& submit, size_t& consumed) {
    static const bool off = std::getenv("JST_NO_CHAIN_FUSION") != nullptr;
    if (off || at + 1 >= ordered.size()) return false;
    auto* mul = dynamic_cast<Multiply*>(ordered[at]);
    auto* fft = dynamic_cast<Fft*>(ordered[at + 1]);
    if (!mul || !fft || std::string(mul->type()) != "multiply") return false;
    if (fft->input.storageId() != mul->c.storageId() || !sole_reader(ordered, mul->c, fft)) return false;
    if (mul->c.dtype() != DataType::CF32 || !fft->forward || fft->output.dtype() != DataType::CF32) return false;
    // geometry as TryFuseSpectrum: transform along the LAST axis of dense tensors, the window broadcast over every other axis
    const Tensor& sig = mul->a;
    const Tensor& win = mul->b;
    const Index axis = fft->resolvedAxis;
    if (axis + 1 != sig.rank() || sig.shape() != mul->c.shape() || win.rank() != sig.rank()) return false;
    const U64 n = sig.shape(axis);
    // the LDS-tiled kernels only (mixed-radix lengths: 8000 = 2^6 5^3 in multi-fm.yml); what the register kernels cover keeps
    // its two launches
    if (!fft->useTiled || fft->bluesteinSize != 0 || !kernels::fft_tiled_supported(n)) return false;
    if (!fft->input.contiguous() || !fft->output.contiguous() || fft->input.offset() != mul->c.offset()) return false;
    if (win.shape(axis) != n) return false;
    for (Index ax = 0; ax < win.rank(); ++ax)
        if (ax != axis && win.shape(ax) != 1 && win.stride(ax) != 0) return false;
    if (sig.rank() - 1 > (Index)dev::kMaxOuterRank) return false;
    members = {mul, fft};
    consumed = 2;
    name = "fft_windowed(" + mul->name() + "+" + fft->name() + ")";
    submit = [mul, fft, n, axis](hipStream_t stream) -> Result {
        const Tensor& sig = mul->a;
        const Tensor& win = mul->b;
        const Tensor& out = fft->output;
        FftLayout L;
        std::memset(&L, 0, sizeof(L));
        L.transforms = 1;
        int r = 0;
        for (Index ax = 0; ax < sig.rank(); ++ax) {
            if (ax == axis) continue;
            L.outer_shape[r] = sig.shape(ax);
            L.in_outer_stride[r] = (int64_t)sig.stride(ax);
            L.out_outer_stride[r] = (int64_t)out.stride(ax);
            L.transforms *= sig.shape(ax);
            ++r;
        }
        L.outer_rank = r;
        L.in_axis_stride = (int64_t)sig.stride(axis);
        L.out_axis_stride = (int64_t)out.stride(axis);
        L.in_offset = sig.offset();
        L.out_offset = out.offset();
        return hip_result(kernels::launch_fft_c2c_tiled_windowed(n, L, fft->twiddles, static_cast<const float2*>(sig.data()),
                                                                 static_cast<const float2*>(win.data()) + win.offset(),
                                                                 (int64_t)win.stride(axis), dptr<float2>(out),
                                                                 static_cast<float2*>(fft->scratchA.data()), stream),
                          "windowed transform (tiled) kernel");
    };
    return true;
}

}  // namespace jst::modules
