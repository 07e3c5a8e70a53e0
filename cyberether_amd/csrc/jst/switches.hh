// switches.hh -- the A/B switches the differential tests and the benches flip INSIDE one process (kernel family of the fused
// spectrum unit, static hand-out of the quad kernel, the serial FM walk, the runtime's launch forms).  Each is read from the
// environment ONCE -- the first time anyone asks -- and from then on changes only through jst_debug_set (include/jetstream_hip.h):
// no getenv on a launch path.  Closed experiments (JST_SPAN_KERNEL, JST_SPAN_THREADS, JST_TILED_*) are gone.
#pragma once

namespace jst {

enum Switch : int {
    SW_FFT_KERNEL = 0,       // JST_FFT_KERNEL=slot|pipe|wave|quad: value = first character, 0 = the library's own choice
    SW_QUAD_STATIC,          // JST_QUAD_STATIC: the quad kernel's static round robin throughout
    SW_FM_SERIAL,            // JST_FM_SERIAL: one-thread-per-lane walk of the FM recurrences
    SW_RUNTIME_MAX_BRANCHES, // JST_RUNTIME_MAX_BRANCHES=n: parallel hipGraph branches (value = n)
    SW_RUNTIME_NO_BATCH,     // JST_RUNTIME_NO_BATCH
    SW_RUNTIME_EAGER_SPANS,  // JST_RUNTIME_EAGER_SPANS
    SW_RUNTIME_NO_SPANS,     // JST_RUNTIME_NO_SPANS
    SW_FIR_DIRECT,           // JST_FIR_DIRECT: provider fast's FIR on the vector FMAs (the round-1 direct form) instead of the MFMA form
    SW_COUNT
};

int switch_value(Switch which);                           // 0 = unset
bool switch_set(const char* name, const char* value);     // value null or "" = unset; false: no such switch

}  // namespace jst
