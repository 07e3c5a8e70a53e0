// types.hh -- vocabulary types of the MI355X Jetstream backend.  Values mirror the reference so
// that codes cross the C ABI unchanged: Result = include/jetstream/types.hh:19-30, DeviceType =
// include/jetstream/memory/types.hh:22-29 (HIP takes the free bit 1<<6 noted in SURVEY 8b),
// Taint = include/jetstream/module.hh:53-63.
#pragma once

#include <stdint.h>

#include <cstdarg>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <variant>
#include <vector>

namespace jst {

using U8 = uint8_t;
using U32 = uint32_t;
using U64 = uint64_t;
using I64 = int64_t;
using F32 = float;
using F64 = double;
using Index = U64;
using Shape = std::vector<U64>;

enum class Result : uint16_t {
    SUCCESS = 0,
    ERROR = 1,
    WARNING = 2,
    FATAL = 3,
    SKIP = 4,
    YIELD = 5,
    RELOAD = 6,
    RECREATE = 7,
    TIMEOUT = 8,
    INCOMPLETE = 9,
};

enum class DeviceType : uint8_t {
    None = 1 << 0,
    CPU = 1 << 1,
    HIP = 1 << 6,
};

enum class RuntimeType : uint8_t { NATIVE = 0 };

enum class DataType : uint8_t {
    None = 0, F32 = 1, CF32 = 2, F64 = 3, U64 = 4, I8 = 5, CI8 = 6, I16 = 7, CI16 = 8, U8 = 9,
    CU8 = 10, U16 = 11, CU16 = 12, I32 = 13, CI32 = 14, U32 = 15, CU32 = 16, CF64 = 17
};

inline size_t DataTypeSize(DataType t) {
    switch (t) {
        case DataType::F32: return 4;
        case DataType::CF32: return 8;
        case DataType::F64: return 8;
        case DataType::U64: return 8;
        case DataType::I8: return 1;
        case DataType::U8: return 1;
        case DataType::CI8: return 2;
        case DataType::I16: return 2;
        case DataType::CI16: return 4;
        case DataType::CU8: return 2;
        case DataType::U16: return 2;
        case DataType::CU16: return 4;
        case DataType::I32: return 4;
        case DataType::CI32: return 8;
        case DataType::U32: return 4;
        case DataType::CU32: return 8;
        case DataType::CF64: return 16;
        default: return 0;
    }
}
const char* DataTypeName(DataType t);
DataType NameToDataType(const std::string& name);  // "CF32" -> CF32, unknown -> None
const char* DeviceName(DeviceType d);
DeviceType StringToDevice(const std::string& s);  // "hip" | "cpu" | ... case-insensitive
const char* ResultName(Result r);

enum Taint : U64 {
    CLEAN = 0,
    IN_PLACE = 1 << 0,
    DISCONTIGUOUS = 1 << 1,
    SURFACE = 1 << 2,
    CROSS_DEVICE = 1 << 4,
    STATIC_OUTPUT = 1 << 6,
    STATELESS = 1 << 7,
};

// ---- logging / last error (JST_ERROR in the reference logs and becomes the block diagnostic) ---
void log_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void log_debug(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_error();

#define JST_ERROR(...) ::jst::log_error(__VA_ARGS__)
#define JST_DEBUG(...) ::jst::log_debug(__VA_ARGS__)

// include/jetstream/macros.hh:54-61
#define JST_CHECK(x)                                                                    \
    do {                                                                                \
        const ::jst::Result jst_check_result_ = (x);                                    \
        if (jst_check_result_ != ::jst::Result::SUCCESS &&                              \
            jst_check_result_ != ::jst::Result::RELOAD)                                 \
            return jst_check_result_;                                                   \
    } while (0)

#define JST_HIP_CHECK(call, what)                                                       \
    do {                                                                                \
        const hipError_t jst_hip_err_ = (call);                                         \
        if (jst_hip_err_ != hipSuccess) {                                               \
            JST_ERROR("[HIP] %s failed: %s", what, hipGetErrorString(jst_hip_err_));    \
            return ::jst::Result::ERROR;                                                \
        }                                                                               \
    } while (0)

}  // namespace jst
