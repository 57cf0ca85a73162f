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
// Frames are handled in CHUNKS of up to kChunkFrames frames:
//
//   side stream   TouchSliceKernel   rank r's band of 16x16-ray tiles of every
//                                    frame of the chunk: candidates -> LDS
//                                    de-duplication per tile -> chunk table
//                                    {key -> one bit per frame of the chunk}
//                 PackSliceKernel    table -> wire segment {count, records},
//                                    table back to the empty state
//                 all-gather         o3dmi_comm (RCCL over xGMI): one call per
//                                    chunk, fixed-size segments
//                 ApplySliceKernel   records of all ranks -> keys this rank
//                                    owns -> insert into the block hash + the
//                                    receiver's chunk table (frame bits OR-ed
//                                    over the ranks that saw the block)
//                 BuildChunkKernel   table -> the chunk's work list {key,
//                                    buffer index, frame bits}
//   main stream   ChunkIntegrateKernel  ONE launch per chunk: a work item is
//                                    (block, 512-voxel part) and applies ALL
//                                    frames of the chunk that touch the block,
//                                    in frame order, to voxel state that stays
//                                    in registers -- reading the RAW depth /
//                                    colour images (no prepare pass: a rank
//                                    would prepare every pixel to use an
//                                    eighth of them).
//
// Why one launch per chunk: a rank's share of a 12-frame group is ~100 blocks,
// less than one round of the chip, so a launch per group lasts as long as one
// work item's chain of dependent memory round trips (~16 us + 1.6 us per frame,
// profiles/r4e) whatever the rank count -- 8 ranks were no faster than 3. With
// the whole chunk in one launch that chain is paid once per 192 frames and the
// launch is bound by the rank's share of the gather / arithmetic work.
//
// The side-stream work of chunk c+1 overlaps the integrate launch of chunk c.
// Results are what the replicated scheme produces: each rank's grid holds
// exactly the blocks it owns, bit-identical to the single-GPU grid.
#pragma once

#include "common.h"
#include "stream_path.h"
#include "touch_device.h"

namespace o3dmi {

constexpr int kChunkGroups = 16;   // a chunk = 16 x frames_per_launch frames
constexpr int kChunkFrames = kChunkGroups * kMaxGroup;  // <= 256
constexpr int kChunkWords = kChunkFrames / 32;

// One candidate of the wire format: a block key and the frames of the chunk
// (bit f of word f / 32 = frame f) whose rays touch the block on the sending
// rank's slice.
struct alignas(16) SliceRecord {
    unsigned long long key;
    unsigned bits[kChunkWords];
    unsigned pad[2];
};
static_assert(sizeof(SliceRecord) == 48, "wire format");

// Wire segment of one rank and one chunk: header + `capacity` records.
// count > capacity (or kSliceFlagTable) = the slice did not fit: every rank
// sees it (the header is all-gathered) and takes the same fallback.
struct SliceHeader {
    int count;
    int flags;
    int capacity;
    int pad[29];
};
static_assert(sizeof(SliceHeader) == 128, "wire format");
constexpr int kSliceFlagTable = 1;   // a chunk table overflowed
constexpr int kSliceFlagKeyRange = 2;
constexpr int kSliceFlagSender = 4;  // (receiver side) a SENDER's slice did not
                                     // fit: seen by every rank alike
constexpr int kSliceFlagAbort = 8;   // (wire) the sending rank is leaving the
                                     // call with an error: its segment is
                                     // empty and every rank leaves at this
                                     // chunk (o3dmi_vbg_integrate_frames is
                                     // COLLECTIVE on the sliced path)
constexpr int kSliceFlagPeerAbort = 16;  // (receiver side) a segment said so

inline int64_t SliceSegmentBytes(int capacity) {
    return (int64_t)sizeof(SliceHeader) +
           (int64_t)capacity * (int64_t)sizeof(SliceRecord);
}

// Open-addressing table {key -> frame bits of the chunk}, self-cleaning (its
// consumer returns every slot it reads to the empty state).
struct ChunkTable {
    unsigned long long* keys;  // [slots], kEmptyKey when free
    unsigned* bits;            // [slots][kChunkWords], 0 when free
    unsigned* hslot;           // [slots] (receiver only): slot of the key in
                               // the block hash
    unsigned* list;            // [list_cap] claimed slots
    int* count;                // [1]
    int* flags;                // [1]
    unsigned mask;             // slots - 1
    int list_cap;              // slots / 2
};

// One entry of a chunk's work list: everything the integrate launch needs to
// start on a block.
struct alignas(16) ChunkEntry {
    unsigned long long key;  // PackKey(x, y, z)
    int block_idx;
    unsigned pad;
    unsigned bits[kChunkWords];
};
static_assert(sizeof(ChunkEntry) == 48, "work list entry");

// Per-frame inputs of the chunk's integrate launch (device array): what the
// per-group launches carry as kernel arguments does not fit there for 256
// frames.
struct IntegFrame {
    float ext[3][4];  // extrinsic, as Camera::Make keeps it
    const uint16_t* depth;
    const uint8_t* color;
    const PixelRec* recs;  // the frame's prepared records (records form)
    const void* pad;
};

// Per-frame inputs of the side-stream kernels (device array, one per frame of
// the call).
struct SliceFrame {
    float pose[3][4];  // inverse extrinsic, as TouchParams::cam.e
    const uint16_t* depth;
};

int AllocChunkTable(ChunkTable* t, int slots, bool receiver, hipStream_t s);
void FreeChunkTable(ChunkTable* t);

// Rank `slice_rank`'s band of ray tiles of frames [f0, f0 + n) (n <=
// kChunkFrames) -> sender table.
int LaunchTouchSlice(const TouchParams& shared, const SliceFrame* frames_dev,
                     int f0, int n, int slice_rank, int slice_world,
                     const ChunkTable& table, hipStream_t s);
// sender table -> wire segment (device), table cleaned.
int LaunchPackSlice(const ChunkTable& table, void* segment_dev, int capacity,
                    hipStream_t s);
// `world` wire segments -> keys this rank owns -> block hash + receiver table.
int LaunchApplySlice(o3dmi_hash* block_hash, const void* gathered_dev,
                     int world, int capacity, const ChunkTable& table,
                     int overflow_stamp, hipStream_t s);
// receiver table -> the chunk's work list + its length, table cleaned;
// publishes {map size, overflow, blocks of the chunk, stamp} in the host-mapped
// status (may be null). overflow: > 0 = stamp of the chunk that ran out of
// buffer indices; -1 = a sender's segment / table was too small (every rank
// reads that in the gathered headers: a collective redo); -2 = THIS rank's
// receiver table was too small (a local redo, no collective); -3 = a rank is
// leaving the call with an error (kSliceFlagAbort): every rank leaves.
int LaunchBuildChunk(o3dmi_hash* block_hash, const ChunkTable& table,
                     ChunkEntry* entries, int entries_cap, int* entries_count,
                     int* status_host, int stamp, hipStream_t s);

// Integrate launch of one chunk (vbg_stream.hip).
struct ChunkIntegrateArgs {
    int n_frames;                 // <= kChunkFrames
    const IntegFrame* frames;     // device, n_frames entries
    const ChunkEntry* entries;    // device work list
    const int* count;             // device: its length
    int64_t entries_cap;
    int grid_hint;
    int rows, cols;
    bool with_color;
    float* tsdf;
    void* weight;
    void* color;
    int grid_dtype;
    const double* depth_intrinsic;
    int resolution;
    float voxel_size, sdf_trunc, depth_max, depth_scale;
    bool depth_div_short;
    bool raw;                     // gather from the raw images (IntegFrame::
                                  // depth / color) instead of the prepared
                                  // records (IntegFrame::recs)
    int* size_host;               // host-mapped status of the integrate roles
    int status_stamp;
    int* prof_count;
    int* prof_frame_blocks;
    int* prof_map_size;
};
int LaunchChunkIntegrate(o3dmi_hash* block_hash, const ChunkIntegrateArgs& a,
                         hipStream_t s);

}  // namespace o3dmi
