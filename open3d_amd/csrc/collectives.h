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

    // AgreeStatus's words: one per rank on the device, mirrored in pinned
    // host memory (allocated at the first use, released with the communicator)
    int* status_dev = nullptr;
    int* status_host = nullptr;
    ~o3dmi_comm();

    int AllreduceSumF64(double* dev, int64_t n, hipStream_t s);
    int Allgather(const void* send_dev, void* recv_dev, int64_t bytes_per_rank,
                  hipStream_t s);
    // byte counts / offsets per peer (host arrays of `world` entries)
    int Alltoallv(const void* send_dev, const int64_t* send_bytes,
                  const int64_t* send_offsets, void* recv_dev,
                  const int64_t* recv_bytes, const int64_t* recv_offsets,
                  hipStream_t s);
    // For host calls that are collectives with a rank-local stage in front
    // (render a band, build a chunk ...): every rank passes what its stage
    // returned and every rank gets the same verdict -- O3DMI_OK if all stages
    // succeeded, its own status if its own stage failed, O3DMI_ERR_PEER if
    // only a peer's did -- so that no rank enters the data collectives without
    // the others. One 4-byte all-gather and one host wait on `s`.
    int AgreeStatus(int local_status, hipStream_t s);
};

namespace o3dmi {
// The communicator of the calling host thread (o3dmi_set_comm), or NULL.
o3dmi_comm* ThreadComm();
}  // namespace o3dmi
