// comm.hh -- communicator of the path's two cross-rank exchanges (jst/comm.cc): RCCL over xGMI, loaded at run time.
#pragma once

#include <cstring>
#include <string>

#include "tensor.hh"

namespace jst {

class Comm {
 public:
    static constexpr size_t kIdBytes = 128;  // ncclUniqueId
    enum class Op { SUM = 0, MAX = 1 };
    static bool available(std::string* why = nullptr);  // can RCCL be loaded in this process?
    static Result uniqueId(uint8_t id[kIdBytes]);       // rank 0 creates it, every rank passes it to create()
    Comm() = default;
    ~Comm();
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    Result create(uint32_t rank, uint32_t world, const uint8_t* id);
    Result allReduce(Tensor& t, Op op, bool average, hipStream_t stream);
    uint32_t rank() const { return rank_; }
    uint32_t world() const { return world_; }
    uint64_t calls() const { return calls_; }
    bool usesRccl() const { return comm_ != nullptr; }

 private:
    void* comm_ = nullptr;
    uint32_t rank_ = 0, world_ = 1;
    uint64_t calls_ = 0;
};

}  // namespace jst
