// comm.cc -- the collective of the path behind the C ABI: ONE communicator per process (one process per GPU), RCCL over
// xGMI loaded at run time (dlopen: the library has no link-time dependency on RCCL and a single-GPU user never loads
// it).  The reference has no multi-device code; what crosses ranks on this path is
//   * the integer hit counts of the exact multi-GPU Spectrogram (U32[H, N]; spectrogram{merge=counts} ->
//     all-reduce(sum) -> spectrogram_merge: bit-exact, the update commutes with summing counts), and
//   * the averaged spectrum of BASELINE config 5 (F32[N] per reporting interval: sum, then / world),
// both IN PLACE on the module's own HBM tensor, on the caller's stream (no host round trip, no torch).
#include "comm.hh"

#include <dlfcn.h>

#include <mutex>

#include "../kernels/kernels.hh"

namespace jst {

namespace {

// the slice of rccl.h this file needs (values: /opt/rocm/include/rccl/rccl.h:40-43,448-466)
struct NcclUniqueId { char internal[128]; };
using NcclComm = void*;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclMax = 2, kNcclUint32 = 3, kNcclFloat32 = 7 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* why = dlerror();  // ONE call: dlerror() clears the message it returns, a second call answers NULL
            r.error = std::string("RCCL is not loadable (") + (why ? why : "librccl.so not found") + ")";
            return;
        }
        const auto sym = [&](const char* s) { return dlsym(r.handle, s); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
            r.error = "RCCL lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r;
}

Result rccl_check(int code, const char* what) {
    if (code == kNcclSuccess) return Result::SUCCESS;
    Rccl& r = rccl();
    JST_ERROR("[COMM] %s failed: %s", what, r.GetErrorString ? r.GetErrorString(code) : "RCCL error");
    return Result::ERROR;
}

}  // namespace

bool Comm::available(std::string* why) {
    Rccl& r = rccl();
    if (!r.handle && why) *why = r.error;
    return r.handle != nullptr;
}

Result Comm::uniqueId(uint8_t id[kIdBytes]) {
    Rccl& r = rccl();
    if (!r.handle) {
        JST_ERROR("[COMM] %s", r.error.c_str());
        return Result::ERROR;
    }
    NcclUniqueId u{};
    JST_CHECK(rccl_check(r.GetUniqueId(&u), "ncclGetUniqueId"));
    std::memcpy(id, u.internal, kIdBytes);
    return Result::SUCCESS;
}

Result Comm::create(uint32_t rank, uint32_t world, const uint8_t* id) {
    if (world == 0 || rank >= world) {
        JST_ERROR("[COMM] invalid rank %u of %u.", rank, world);
        return Result::ERROR;
    }
    rank_ = rank;
    world_ = world;
    // a one-rank communicator reduces nothing and needs no RCCL -- unless the caller hands over an id: then a REAL one-rank
    // RCCL communicator is opened (ncclCommInitRank(nranks = 1) is legal), and every all-reduce goes through the library,
    // the dtype / op enum slice above and the caller's stream exactly as it does at world > 1
    if (world == 1 && !id) return Result::SUCCESS;
    if (!id) {
        JST_ERROR("[COMM] a communicator of %u ranks needs rank 0's unique id.", world);
        return Result::ERROR;
    }
    Rccl& r = rccl();
    if (!r.handle) {
        JST_ERROR("[COMM] world size %u needs RCCL: %s", world, r.error.c_str());
        return Result::ERROR;
    }
    NcclUniqueId u{};
    std::memcpy(u.internal, id, kIdBytes);
    return rccl_check(r.CommInitRank(&comm_, (int)world, u, (int)rank), "ncclCommInitRank");
}

Comm::~Comm() {
    if (comm_) (void)rccl().CommDestroy(comm_);
}

// In place on a dense F32 or U32 tensor; `average`: F32 only, sum then divide by the world size.
Result Comm::allReduce(Tensor& t, Op op, bool average, hipStream_t stream) {
    if (t.device() != DeviceType::HIP || !t.contiguous() || (t.dtype() != DataType::F32 && t.dtype() != DataType::U32)) {
        JST_ERROR("[COMM] all-reduce takes a dense F32 or U32 HIP tensor.");
        return Result::ERROR;
    }
    if (average && (t.dtype() != DataType::F32 || op != Op::SUM)) {
        JST_ERROR("[COMM] the average is defined for F32 sums.");
        return Result::ERROR;
    }
    ++calls_;
    if (!comm_) return Result::SUCCESS;  // world 1 without RCCL
    char* base = static_cast<char*>(t.data()) + t.offset() * 4;
    JST_CHECK(rccl_check(rccl().AllReduce(base, base, (size_t)t.size(), t.dtype() == DataType::F32 ? kNcclFloat32 : kNcclUint32,
                                           op == Op::SUM ? kNcclSum : kNcclMax, comm_, stream),
                         "ncclAllReduce"));
    if (average)
        JST_HIP_CHECK(kernels::launch_divide_f32(reinterpret_cast<float*>(base), t.size(), (float)world_, stream),
                      "divide kernel");
    return Result::SUCCESS;
}

}  // namespace jst
