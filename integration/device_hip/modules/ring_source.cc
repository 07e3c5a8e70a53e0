// integration/device_hip/modules/ring_source.cc -- REFERENCE-SIDE code: would live at src/domains/io/ring_source/ (module.hh, block.hh,
// module_impl_native_hip.cc, block_impl.cc; INTEGRATION.md sections 3 and 5).  The HBM-resident stand-in for the Soapy source's
// output contract (src/domains/io/soapy/module_impl.cc:197-201: CF32 [batches, samples], batchAxis 0, sampleAxis 1,
// "sampleRate" / "frequency" attributes): `slots` batches live in ONE device allocation and a compute cycle SELECTS the next
// slot instead of copying it (soapy/module_impl_native_cpu.cc:39-60 pops the host CircularBuffer into the output tensor).
// Everything is the library's `ring_source` module; this unit gives it a reference-side Config, a module of
// (DeviceType::HIP, RuntimeType::NATIVE) and a block.  slots = 1 is a plain device-resident source.  The reference tensor
// `buffer` borrows slot 0; consumers that are library modules follow the ring through the producer's library tensor
// (hip_library_module.hh: TensorDirectory), which is also what lets the HIP runtime batch the cycles of a resident ring.
// A producer thread feeds a live ring through jst_ring_push / _acquire / _commit on `libraryModule()`.
#include <jetstream/block.hh>
#include <jetstream/detail/block_impl.hh>
#include <jetstream/detail/module_impl.hh>
#include <jetstream/memory/axis.hh>
#include <jetstream/module.hh>

#include "native_hip_module.hh"

namespace Jetstream::Modules {

struct RingSource : public Module::Config {
    U64 batches = 8;
    U64 samples = 2048;
    U64 slots = 1;
    std::string dataType = "CF32";
    F32 sampleRate = 2.0e6f;
    F32 frequency = 96.9e6f;

    JST_MODULE_TYPE(ring_source);
    JST_MODULE_PARAMS(batches, samples, slots, dataType, sampleRate, frequency);
};

struct RingSourceImpl : public Module::Impl, public DynamicConfig<RingSource> {
    Result validate() override {
        const auto& config = *candidate();
        if (config.batches == 0 || config.samples == 0 || config.slots == 0) {
            JST_ERROR("[MODULE_RING_SOURCE] batches, samples and slots must be positive.");
            return Result::ERROR;
        }
        const DataType dtype = NameToDataType(config.dataType);
        if (dtype != DataType::CF32 && dtype != DataType::CI16 && dtype != DataType::CI8 && dtype != DataType::CU8) {
            JST_ERROR("[MODULE_RING_SOURCE] Unsupported sample format '{}' (CF32, CI16, CI8, CU8).", config.dataType);
            return Result::ERROR;
        }
        return Result::SUCCESS;
    }
    Result define() override { return defineInterfaceOutput("buffer"); }

 protected:
    Tensor output;
};

struct RingSourceImplNativeHip : public NativeHipModule<RingSourceImpl> {
    Result create() override {
        // the library allocates the ring; the reference's tensor borrows slot 0 of it
        JST_CHECK(library.create("MODULE_RING_SOURCE_NATIVE_HIP", "ring_source", "generic", name(),
                                 {"batches=" + std::to_string(batches), "samples=" + std::to_string(samples), "slots=" + std::to_string(slots),
                                  "dtype=" + dataType, "sampleRate=" + Hip::Number(sampleRate), "frequency=" + Hip::Number(frequency)},
                                 {}, {}));
        jst_tensor ring{};
        jst_tensor_desc d{};
        if (jst_module_output(library.handle(), "buffer", &ring) != JST_SUCCESS || jst_tensor_describe(ring, &d) != JST_SUCCESS) {
            JST_ERROR("[MODULE_RING_SOURCE_NATIVE_HIP] {}", jst_last_error());
            if (ring) (void)jst_tensor_destroy(ring);
            return Result::ERROR;
        }
        const Result r = output.create(d.data, DeviceType::HIP, NameToDataType(dataType), {batches, samples});
        if (r == Result::SUCCESS) Hip::TensorDirectory::Get().publish(name(), "buffer", ring);
        (void)jst_tensor_destroy(ring);
        JST_CHECK(r);
        JST_CHECK(SetSignalAxes(output, {.sample = Index{1}, .batch = Index{0}}));
        JST_CHECK(output.setAttribute("sampleRate", F32{sampleRate}));
        JST_CHECK(output.setAttribute("frequency", F32{frequency}));
        outputs()["buffer"].produced(name(), "buffer", output);
        return Result::SUCCESS;
    }
    Result destroy() override { return library.destroy(); }
};

JST_REGISTER_MODULE(RingSourceImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "generic");
JST_REGISTER_MODULE(RingSourceImplNativeHip, DeviceType::HIP, RuntimeType::NATIVE, "fast");

}  // namespace Jetstream::Modules

namespace Jetstream::Blocks {

struct RingSource : public Block::Config {
    U64 batches = 8;
    U64 samples = 2048;
    U64 slots = 1;
    std::string dataType = "CF32";
    F32 sampleRate = 2.0e6f;
    F32 frequency = 96.9e6f;

    JST_BLOCK_TYPE(ring_source);
    JST_BLOCK_DOMAIN("IO");
    JST_BLOCK_PARAMS(batches, samples, slots, dataType, sampleRate, frequency);
    JST_BLOCK_DESCRIPTION("Ring Source", "Device-resident sample ring.",
                          "Batches of IQ samples resident in HBM; a compute cycle selects the next slot.");
};

struct RingSourceBlockImpl : public Block::Impl, public DynamicConfig<Blocks::RingSource> {
    Result configure() override {
        moduleConfig->batches = batches;
        moduleConfig->samples = samples;
        moduleConfig->slots = slots;
        moduleConfig->dataType = dataType;
        moduleConfig->sampleRate = sampleRate;
        moduleConfig->frequency = frequency;
        return Result::SUCCESS;
    }
    Result define() override { return defineInterfaceOutput("buffer", "Output", "The selected batch of the ring."); }
    Result create() override {
        JST_CHECK(moduleCreate("ring_source", moduleConfig, {}));
        return moduleExposeOutput("buffer", {"ring_source", "buffer"});
    }
    std::shared_ptr<Modules::RingSource> moduleConfig = std::make_shared<Modules::RingSource>();
};

JST_REGISTER_BLOCK(RingSourceBlockImpl, {"ring_source"});

}  // namespace Jetstream::Blocks
