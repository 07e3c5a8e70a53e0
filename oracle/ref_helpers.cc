// ref_helpers.cc -- extern "C" shims around two header-inline pieces of the REFERENCE, compiled IN PLACE from
// /root/reference (never copied): Backend::ApproxLog10 (include/jetstream/backend/devices/cpu/helpers.hh:59-74, what
// the Amplitude module calls) and the waterfall ring arithmetic (src/domains/visualization/waterfall/ring_state.hh:16-56).
// TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into oracle/_ref/libref_helpers.so when the reference tree is
// present; tests/test_oracle_ref_helpers.py sweeps oracle/jst_oracle.c's restatements against it.  The include path
// puts oracle/ref_shim first (a stub config.hh and jst::fmt forwarded to the fmt headers torch ships): logger.hh only
// needs its declarations to parse, nothing here logs.
#include <cstdint>

#include "jetstream/backend/devices/cpu/helpers.hh"
#include "ring_state.hh"

extern "C" {

float ref_approx_log10(float x) { return Jetstream::Backend::ApproxLog10(x); }

// bulk form for the sweep: out[i] = ApproxLog10(float with bit pattern first + i)
void ref_approx_log10_bits(uint32_t first, uint64_t count, float* out) {
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t b = first + (uint32_t)i;
        float x;
        __builtin_memcpy(&x, &b, 4);
        out[i] = Jetstream::Backend::ApproxLog10(x);
    }
}

void ref_waterfall_plan(uint64_t write_index, uint64_t batches, uint64_t height, uint64_t* out3) {
    const auto p = Jetstream::Modules::PlanWaterfallWrite(write_index, batches, height);
    out3[0] = p.sourceRow;
    out3[1] = p.destinationRow;
    out3[2] = p.rowCount;
}

// state = {writeIndex, dirtyRows}; advances it like the module does after a write and reports the dirty plan
void ref_waterfall_advance(uint64_t* state2, uint64_t batches, uint64_t height, uint64_t* dirty3) {
    Jetstream::Modules::WaterfallRingState s{state2[0], state2[1]};
    s.advance(batches, height);
    const auto d = s.dirtyPlan(height);
    state2[0] = s.writeIndex;
    state2[1] = s.dirtyRows;
    dirty3[0] = d.startRow;
    dirty3[1] = d.firstRowCount;
    dirty3[2] = d.secondRowCount;
}

}  // extern "C"
