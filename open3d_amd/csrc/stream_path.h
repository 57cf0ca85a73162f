// Internal interface of the frame-stream fast path (not part of the C ABI):
// the two kernels one RGB-D frame costs when it is integrated with
// o3dmi_vbg_integrate_frame(s).
//
//   front role      block touch + activation in the main hash (first
//                   workgroups) fused with a per-pixel "prepare" pass
//   integrate role  per-voxel TSDF / weight / colour update over the frame's
//                   block list
// (vbg_stream.hip; one launch can carry the front role of frame k+1 next to
// the integrate role of frame k)
//
// Both are pure re-cuts of the reference's arithmetic (DepthTouchCPU,
// VoxelBlockGridCPU.cpp:117-201; IntegrateCPU, VoxelBlockGridImpl.h:151-308):
// every per-pixel sub-expression of the integrate lambda that does not depend
// on the voxel -- float(depth)/depth_scale and the colour pixel selected by
// Unproject(depth K) -> Project(colour K) -> round -- is evaluated once per
// pixel by the prepare pass with the reference's exact operation order and
// stored as an 8-byte record, so the per-voxel kernel gathers one record
// instead of a depth sample plus three colour bytes and skips four of its
// seven correctly-rounded divisions. Results are bit-identical.
#pragma once

#include "common.h"

namespace o3dmi {

// Frames are processed in groups of up to kMaxGroup consecutive frames: the
// integrate role applies the frames of a group one after the other to a block
// whose voxel state stays in registers, so that state is read and written
// once per group instead of once per frame. Frame order inside the group is
// preserved and a block only receives the frames that touched it (per-slot
// frame bits, TouchSlot), so the result is identical to frame-by-frame
// integration.
constexpr int kMaxGroup = 16;  // = kTouchBits, the frame bits of a touch word
static_assert(kMaxGroup == kTouchBits, "one touch bit per frame of a group");
// Group size when the caller does not choose: a group may add up to (frames x
// per-frame frustum bound) blocks, which the run-ahead capacity policy holds
// against the hash capacity -- 8 frames fit a 262 144-block map, 16 want twice
// that.
constexpr int kDefaultGroup = 8;
// The integrate role's wide form keeps the records of kGroupChunk frames in
// registers at a time (a group of 16 = four chunks on the same register-
// resident voxel state).
constexpr int kGroupChunk = 4;

// One entry of a group's block list: hash slot + block key.
struct alignas(16) FrameBlock {
    int slot, x, y, z;
};

// One entry of a group's READY list: everything the integrate role needs to
// start on a block -- buffer index, packed block key, frame bits -- in one
// 16-byte load. Written by the front roles' last touch workgroup once every
// touch workgroup of the group has arrived (the frame bits of a block are
// final only then), in the launch BEFORE the one whose integrate role reads
// it. Without it a work item's header is two dependent round trips (list
// entry -> buffer index + touch word) in front of its voxel state and record
// gathers.
struct alignas(16) ReadyEntry {
    unsigned long long key;  // PackKey(x, y, z)
    int block_idx;           // buffer index of the block
    unsigned bits;           // frames of the group that touched it
};

// Per-pixel prepared record (u16 depth / u8 colour inputs).
struct alignas(8) PixelRec {
    float d;        // float(depth) / depth_scale
    unsigned rgba;  // r | g<<8 | b<<16 | (colour pixel in bounds)<<24
};

// Front role of ONE frame.
struct FrameFrontArgs {
    const uint16_t* depth;  // {rows, cols}
    const uint8_t* color;   // {color_rows, color_cols, 3} or null
    int rows, cols, color_rows, color_cols;
    const double* depth_intrinsic;  // host 3x3
    const double* color_intrinsic;  // host 3x3
    const double* extrinsic;        // host 4x4
    int resolution;
    float voxel_size, sdf_trunc, depth_scale, depth_max;
    int stride;
    unsigned long long group_stamp;  // > 0, increases per group
    int group_bit;                   // index of this frame in its group
    int touch_plane;                 // plane of the per-slot touch words
                                     // (group sequence & 1)
    // Optional per-column / per-row tables of the prepare pass (device, cols /
    // rows ints): the colour pixel Unproject -> Project -> round selects is
    // separable, column of the colour image from the depth column alone and
    // row from the row alone (-1 = outside the colour image). Built on the
    // host with the same float32 operations (PrepTables); null = evaluate per
    // pixel. depth_div_short: float(depth) / depth_scale may take the short
    // constant-divisor form (host-verified for all 65536 depth values).
    const int* col_lut;
    const int* row_lut;
    bool depth_div_short;
    bool prep_identity;   // col_lut[u] == u and row_lut[v] == v everywhere
    PixelRec* recs;       // {rows, cols} out, + 1 sentinel record {0, 0}
    FrameBlock* list;     // the group's list (shared by its frames)
    int64_t list_capacity;
    int* count;           // the group's count; 0 before the group's first frame
    ReadyEntry* ready;    // the group's ready list (may be null: not built)
    int* tickets;         // 9 arrival counters of the group's touch
                          // workgroups, zero between launches
    int* touch_status;    // host-mapped {map size, overflow stamp, group block
                          // count, group stamp}: published by the group's last
                          // touch workgroup (may be null)
    bool prepare_only;    // no block touch: only the per-pixel prepare pass
                          // (the sliced path touches per rank band and
                          // prepares per chunk, sliced_path.h)
};

// Integrate role of ONE group.
struct IntegrateStreamArgs {
    int n_frames;                          // 1..kMaxGroup
    unsigned long long group_stamp;        // the stamp the group's front roles
    int touch_plane;                       // used, and their touch plane
    const PixelRec* recs[kMaxGroup];
    const double* extrinsic[kMaxGroup];    // host 4x4 each
    int rows, cols;
    bool with_color;
    const FrameBlock* list;
    const ReadyEntry* ready;  // built by the group's front roles, or null
    const int* count;     // device: live length of `list`
    int64_t list_capacity;
    int grid_hint;        // expected number of blocks (sizes the grid)
    float* tsdf;
    void* weight;
    void* color;          // may be null
    int grid_dtype;       // O3DMI_U16 | O3DMI_F32
    const double* depth_intrinsic;
    int resolution;
    float voxel_size, sdf_trunc, depth_max;
    // bookkeeping done by workgroup 0 (any may be null):
    int* zero_counter;    // device int reset to 0 (a future group's count)
    int* size_host;       // host-mapped {heap_top, error flags, count, stamp}
    int status_stamp;     // value published in size_host[3]
    int* prof_count;      // device int receiving the live count
    int* prof_frame_blocks;  // device int receiving sum over blocks of
                             // popcount(frame bits) = block-frames integrated
    int* prof_map_size;      // device int receiving the map size (heap top) the
                             // role sees when it starts
};

// One launch running the front roles of up to kMaxGroup frames and / or the
// integrate role of one group. The workgroups of the front roles are
// dispatched first and overlap the integrate role inside the same kernel: the
// two touch disjoint scratch (double-buffered lists / records / counters /
// per-slot touch words, all selected by the group's sequence parity) and the
// hash map tolerates concurrent insertion of new keys next to lookups of
// existing ones. The integrate role verifies the stamp of every touch word it
// reads and raises kErrTouchStamp on a foreign one.
int LaunchFrameStep(o3dmi_hash* block_hash, const FrameFrontArgs* fronts,
                    int n_fronts, const IntegrateStreamArgs* integ,
                    hipStream_t s);

// Starts (without waiting for it) the on-device proof that the integrate
// role's short division forms are exact for this truncation distance; launches
// use them once it has finished (vbg_stream.hip VerifyFastDivision).
// Returns the forms usable now: 0 = IEEE only (proof running, failed or
// disabled), 1 = sdf / trunc and 1 / (w + 1), 2 / 3 = also 1 / z with one /
// two Newton steps. `wait`: block until the proof has finished.
int PrefetchFastDivision(float sdf_trunc, bool wait);

// After a frame-stream group ran out of buffer indices (HashView::counters[3],
// InsertKey): with the stream drained, turns the slots that got no index into
// tombstones, brings heap_top back to the capacity and clears the overflow
// stamp, so that the map is a consistent FULL map again (ready for Reserve).
// `wanted` receives the number of indices the dropped groups asked for in all
// (>= capacity).
int RecoverOverflow(o3dmi_hash* block_hash, hipStream_t s, int64_t* wanted);

// Host evaluation of the prepare pass's per-column / per-row sub-expressions
// (IntegrateCPU's colour-pixel selection, VoxelBlockGridImpl.h:277-289, with
// TransformIndexer's float32 arithmetic): col[u] / row[v] = rounded colour
// pixel coordinate or -1. Returns whether the short constant-divisor form of
// float(depth) / depth_scale equals the IEEE division for all 65536 depths.
bool PrepTables(const double* depth_intrinsic, const double* color_intrinsic,
                int rows, int cols, int color_rows, int color_cols,
                float depth_scale, int* col, int* row);

// Strict upper bound on the number of distinct blocks one depth frame can
// touch: every touched block lies inside the viewing pyramid (z-depth <=
// depth_max) dilated by one block diagonal, so their count is at most that
// volume / block volume (and at most 4 per ray).
int64_t FrustumBlockBound(const double* intrinsic, int rows, int cols,
                          float depth_max, float block_size, int stride);

}  // namespace o3dmi
