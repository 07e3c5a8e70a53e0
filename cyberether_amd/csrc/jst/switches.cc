// switches.cc -- see switches.hh.
#include "switches.hh"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace jst {

namespace {
struct Entry {
    const char* name;
    bool first_char;  // the value is the first character of the text (else: its integer value, 1 when not a number)
};
constexpr Entry kEntries[SW_COUNT] = {
    {"JST_FFT_KERNEL", true},       {"JST_QUAD_STATIC", false},        {"JST_FM_SERIAL", false},       {"JST_RUNTIME_MAX_BRANCHES", false},
    {"JST_RUNTIME_NO_BATCH", false}, {"JST_RUNTIME_EAGER_SPANS", false}, {"JST_RUNTIME_NO_SPANS", false},
    {"JST_FIR_DIRECT", false},
};
std::atomic<int> g_value[SW_COUNT];
std::once_flag g_once;

int parse(const Entry& e, const char* text) {
    if (!text || !text[0]) return 0;
    if (e.first_char) return (int)(unsigned char)text[0];
    const int v = std::atoi(text);
    return v > 0 ? v : 1;
}
void init() {
    std::call_once(g_once, [] {
        for (int i = 0; i < SW_COUNT; ++i) g_value[i].store(parse(kEntries[i], std::getenv(kEntries[i].name)), std::memory_order_relaxed);
    });
}
}  // namespace

int switch_value(Switch which) {
    init();
    return g_value[which].load(std::memory_order_relaxed);
}

bool switch_set(const char* name, const char* value) {
    init();
    for (int i = 0; i < SW_COUNT; ++i)
        if (name && std::strcmp(name, kEntries[i].name) == 0) {
            g_value[i].store(parse(kEntries[i], value), std::memory_order_relaxed);
            return true;
        }
    return false;
}

}  // namespace jst
