// Internal interface of the SLICED block touch: block-ownership sharding
// (scheme A) without replicated front work.
//
// Scheme A as round 1-3 built it lets every rank run the whole front role of
// every frame (block touch of all rays + the per-pixel prepare pass) and keep
// only the blocks it owns; at 8 ranks that replicated work is what a rank
// mostly does (profiles/r3a_emulate_rank0_of_8.json: 2.9x at 8). SURVEY 8(e)
// specifies the split this file implements:
//
//   every GPU runs DepthTouch (VoxelBlockGridCPU.cpp:117-201) on a 1/G slice of
//   the rays, the candidate keys are all-gathered, each GPU keeps the keys with
//   OwnerOf(key) == rank, activates and integrates only those.
//
// Frames are handled in CHUNKS of kChunkGroups frame groups (a group = the
// frames one integrate launch applies to register-resident blocks):
//
//   side stream   TouchSliceKernel   rank r's band of 16x16-ray tiles of every
//                                    frame of the chunk: candidates -> LDS
//                                    de-duplication per tile -> per-group
//                                    table {key -> frame bits} (GroupTables)
//                 PackSliceKernel    tables -> wire segment {counts, records},
//                                    tables back to the empty state
//                 all-gather         o3dmi_comm (RCCL over xGMI): one call per
//                                    chunk, fixed-size segments
//                 ApplySliceKernel   records of all ranks -> keys this rank
//                                    owns -> insert into the block hash + the
//                                    receiver's per-group tables (frame bits
//                                    OR-ed over the ranks that saw the block)
//                 BuildReadyKernel   tables -> the groups' READY lists (the
//                                    integrate role's work lists, stream_path.h)
//   main stream   FrameStepKernel    integrate role only, one launch per group,
//                                    reading the RAW depth / colour images (no
//                                    prepare pass: a rank would prepare every
//                                    pixel to use an eighth of them)
//
// The side-stream work of chunk c+1 overlaps the integrate launches of chunk
// c. Results are what the replicated scheme produces: each rank's grid holds
// exactly the blocks it owns, bit-identical to the single-GPU grid.
#pragma once

#include "common.h"
#include "stream_path.h"
#include "touch_device.h"

namespace o3dmi {

constexpr int kChunkGroups = 16;

// One candidate of the wire format: a block key and the frames of its group
// (bit f = frame f of the group) whose rays touch the block on this rank's
// slice.
struct alignas(16) SliceRecord {
    unsigned long long key;
    unsigned bits;
    unsigned pad;
};

// Wire segment of one rank and one chunk: header + kChunkGroups x capacity
// records. count[g] > capacity (or kSliceFlagTable) = the slice of group g did
// not fit: every rank sees it (the header is all-gathered) and takes the same
// fallback.
struct SliceHeader {
    int count[kChunkGroups];
    int flags;
    int capacity;
    int pad[14];
};
static_assert(sizeof(SliceHeader) == 128, "wire format");
constexpr int kSliceFlagTable = 1;   // a per-group table overflowed
constexpr int kSliceFlagKeyRange = 2;

inline int64_t SliceSegmentBytes(int capacity) {
    return (int64_t)sizeof(SliceHeader) +
           (int64_t)kChunkGroups * capacity * (int64_t)sizeof(SliceRecord);
}

// Per-group open-addressing tables {key -> bits}, self-cleaning (their
// consumer returns every slot it reads to the empty state).
struct GroupTables {
    unsigned long long* keys;  // [kChunkGroups][slots], kEmptyKey when free
    unsigned* bits;            // [kChunkGroups][slots], 0 when free
    unsigned* hslot;           // [kChunkGroups][slots] (receiver only): slot of
                               // the key in the block hash
    unsigned* list;            // [kChunkGroups][list_cap] claimed slots
    int* count;                // [kChunkGroups]
    int* flags;                // [1]
    unsigned mask;             // slots - 1
    int list_cap;              // slots / 2
};

// Per-frame inputs of the side-stream kernels (device array, one per frame of
// the call).
struct SliceFrame {
    float pose[3][4];  // inverse extrinsic, as TouchParams::cam.e
    const uint16_t* depth;
};

int AllocGroupTables(GroupTables* t, int slots, bool receiver, hipStream_t s);
void FreeGroupTables(GroupTables* t);

// Rank `slice_rank`'s band of ray tiles of frames [f0, f0 + n) (n <= 16 x
// frames_per_group) -> sender tables.
int LaunchTouchSlice(const TouchParams& shared, const SliceFrame* frames_dev,
                     int f0, int n, int frames_per_group, int slice_rank,
                     int slice_world, const GroupTables& tables,
                     hipStream_t s);
// sender tables -> wire segment (device), tables cleaned.
int LaunchPackSlice(const GroupTables& tables, void* segment_dev, int capacity,
                    hipStream_t s);
// `world` wire segments -> keys this rank owns -> block hash + receiver tables.
int LaunchApplySlice(o3dmi_hash* block_hash, const void* gathered_dev,
                     int world, int capacity, const GroupTables& tables,
                     int overflow_stamp, hipStream_t s);
// receiver tables -> ready lists [kChunkGroups][ready_cap] + counts, tables
// cleaned; publishes {map size, overflow stamp, blocks of the chunk, stamp} in
// the host-mapped status (may be null).
int LaunchBuildReady(o3dmi_hash* block_hash, const GroupTables& tables,
                     ReadyEntry* ready, int ready_cap, int* ready_count,
                     int* status_host, int stamp, hipStream_t s);

}  // namespace o3dmi
