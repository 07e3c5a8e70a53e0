// integration/device_hip/modules/side_chains.cc -- REFERENCE-SIDE code: in the reference's tree one module_impl_native_hip.cc per module
// directory (src/domains/{core,dsp,visualization}/<module>/; the CUDA peers are laid out that way), here one translation unit:
// the modules of the Filter block (src/domains/dsp/filter/block_impl.cc:350-582: filter_taps, expand_dims, pad, fft, reshape,
// multiply, fold, multiply_constant, phase_correction, unpad, overlap_add), of the decimator (reshape, arithmetic), of the
// `slice` block (slice, duplicate), the demodulators (fm, am), the generators and converters either side (signal_generator, agc,
// squelch, add, squeeze_dims, flatten, permutation, ones_tensor) and the sinks with state (waterfall, lineplot) on
// (DeviceType::HIP, RuntimeType::NATIVE) -- each the reference's own Impl on device tensors with its compute hooks on the
// library module of the same type (native_hip_module.hh: LibraryBackedModule).  With them the reference's `filter` block and
// the chains of examples/flowgraphs/multi-fm.yml run device-resident (tests/test_gpu_reference_device_hip.py).
#include "domains/core/add/module_impl.hh"
#include "domains/core/arithmetic/module_impl.hh"
#include "domains/core/duplicate/module_impl.hh"
#include "domains/core/expand_dims/module_impl.hh"
#include "domains/core/flatten/module_impl.hh"
#include "domains/core/multiply_constant/module_impl.hh"
#include "domains/core/ones_tensor/module_impl.hh"
#include "domains/core/pad/module_impl.hh"
#include "domains/core/permutation/module_impl.hh"
#include "domains/core/slice/module_impl.hh"
#include "domains/core/squeeze_dims/module_impl.hh"
#include "domains/core/unpad/module_impl.hh"
#include "domains/dsp/agc/module_impl.hh"
#include "domains/dsp/am/module_impl.hh"
#include "domains/dsp/filter_taps/module_impl.hh"
#include "domains/dsp/fm/module_impl.hh"
#include "domains/dsp/fold/module_impl.hh"
#include "domains/dsp/overlap_add/module_impl.hh"
#include "domains/dsp/phase_correction/module_impl.hh"
#include "domains/dsp/signal_generator/module_impl.hh"
#include "domains/dsp/squelch/module_impl.hh"
#include "domains/visualization/lineplot/module_impl.hh"
#include "domains/visualization/waterfall/module_impl.hh"

#include "native_hip_module.hh"

namespace Jetstream::Modules {

using AddImplNativeHip = LibraryBackedModule<AddImpl>;
using ArithmeticImplNativeHip = LibraryBackedModule<ArithmeticImpl>;
using DuplicateImplNativeHip = LibraryBackedModule<DuplicateImpl>;   // HIP -> HIP: the dense copy behind a `slice` block's view
using ExpandDimsImplNativeHip = LibraryBackedModule<ExpandDimsImpl>;
using FlattenImplNativeHip = LibraryBackedModule<FlattenImpl>;
using MultiplyConstantImplNativeHip = LibraryBackedModule<MultiplyConstantImpl>;
using OnesTensorImplNativeHip = LibraryBackedModule<OnesTensorImpl>;
using PadImplNativeHip = LibraryBackedModule<PadImpl>;
using PermutationImplNativeHip = LibraryBackedModule<PermutationImpl>;
using SliceImplNativeHip = LibraryBackedModule<SliceImpl>;
using SqueezeDimsImplNativeHip = LibraryBackedModule<SqueezeDimsImpl>;
using UnpadImplNativeHip = LibraryBackedModule<UnpadImpl>;
using AgcImplNativeHip = LibraryBackedModule<AgcImpl>;
using AmImplNativeHip = LibraryBackedModule<AmImpl>;
using FilterTapsImplNativeHip = LibraryBackedModule<FilterTapsImpl>;
using FmImplNativeHip = LibraryBackedModule<FmImpl>;
using FoldImplNativeHip = LibraryBackedModule<FoldImpl>;
using OverlapAddImplNativeHip = LibraryBackedModule<OverlapAddImpl>;
using PhaseCorrectionImplNativeHip = LibraryBackedModule<PhaseCorrectionImpl>;
using SignalGeneratorImplNativeHip = LibraryBackedModule<SignalGeneratorImpl>;
using SquelchImplNativeHip = LibraryBackedModule<SquelchImpl>;

// sinks: no output port, their state lives in the reference module's own tensor (the present half reads it there)
struct WaterfallImplNativeHip : public LibraryBackedModule<WaterfallImpl> {
    Result presentInitialize() override { return createPresent(); }
    Result presentSubmit() override { return present(); }

 protected:
    Result bindStates() override { return library.bindState("frequencyBins", frequencyBins); }
};
struct LineplotImplNativeHip : public LibraryBackedModule<LineplotImpl> {
    Result presentInitialize() override { return createPresent(); }
    Result presentSubmit() override { return present(); }

 protected:
    Result bindStates() override { return library.bindState("signalPoints", signalPoints); }
};

JST_REGISTER_HIP_LIBRARY_MODULE(AddImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(ArithmeticImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(DuplicateImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(ExpandDimsImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(FlattenImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(MultiplyConstantImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(OnesTensorImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(PadImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(PermutationImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(SliceImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(SqueezeDimsImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(UnpadImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(AgcImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(AmImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(FilterTapsImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(FmImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(FoldImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(OverlapAddImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(PhaseCorrectionImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(SignalGeneratorImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(SquelchImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(WaterfallImplNativeHip);
JST_REGISTER_HIP_LIBRARY_MODULE(LineplotImplNativeHip);

}  // namespace Jetstream::Modules
