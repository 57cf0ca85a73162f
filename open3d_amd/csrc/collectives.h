// Internal view of o3dmi_comm_t (include/o3d_mi355x_host.h): the collectives
// the sharded hot path needs -- a sum all-reduce of a few float64 (ICP), an
// all-gather of small fixed-size records (block counts, block IDs) and an
// all-to-all of variable-size byte ranges (voxel rows to their owners) -- over
// one of two transports: RCCL (librccl.so, resolved at run time; xGMI between
// the GPUs of a node) or a caller-provided table of functions (tests; other
// runtimes). Everything is enqueued on the caller's stream.
#pragma once

#include <cstdint>

#include "common.h"
#include "o3d_mi355x_host.h"

struct o3dmi_comm {
    int rank = 0;
    int world = 1;
    void* nccl = nullptr;    // ncclComm_t when the transport is RCCL
    bool owns_nccl = false;  // created here (o3dmi_comm_create_rccl)
    bool custom = false;
    o3dmi_transport_t table = {};
    void* user = nullptr;

    int AllreduceSumF64(double* dev, int64_t n, hipStream_t s);
    int Allgather(const void* send_dev, void* recv_dev, int64_t bytes_per_rank,
                  hipStream_t s);
    // byte counts / offsets per peer (host arrays of `world` entries)
    int Alltoallv(const void* send_dev, const int64_t* send_bytes,
                  const int64_t* send_offsets, void* recv_dev,
                  const int64_t* recv_bytes, const int64_t* recv_offsets,
                  hipStream_t s);
};

namespace o3dmi {
// The communicator of the calling host thread (o3dmi_set_comm), or NULL.
o3dmi_comm* ThreadComm();
}  // namespace o3dmi
