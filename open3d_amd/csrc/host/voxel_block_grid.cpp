// Host-side VoxelBlockGrid for the MI355X backend: the orchestration of
// t::geometry::VoxelBlockGrid (cpp/open3d/t/geometry/VoxelBlockGrid.cpp:65-117
// ctor, :212-245 GetUniqueBlockCoordinates, :292-326 Integrate, :328-402
// RayCast) and of core::HashMap::Activate's capacity policy
// (cpp/open3d/core/hashmap/HashMap.cpp:166-181) on top of the kernel C ABI.
//
// What differs from the reference is where counts live: the reference reads
// Size() and the candidate count back to the host every frame; here the
// host keeps an upper bound of the map size and only synchronises when that
// bound says a Reserve() might be needed, so the steady-state frame stream
// (o3dmi_vbg_integrate_frame) issues kernels back to back.

#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../common.h"
#include "../npz.h"
#include "../stream_path.h"
#include "../sliced_path.h"
#include "../touch_device.h"
#include "../collectives.h"
#include "o3d_mi355x_host.h"

using namespace o3dmi;

struct o3dmi_vbg {
    float voxel_size = 0;
    int64_t block_resolution = 0;
    std::vector<std::string> attr_names;
    std::vector<int> attr_dtypes;
    std::vector<int> attr_channels;
    o3dmi_hash_t* block_hashmap = nullptr;
    o3dmi_hash_t* frustum_hashmap = nullptr;  // lazily created (cpp:224-235)
    int64_t frustum_capacity = 0;
    int owner_rank = 0, owner_world = 1;      // block-ownership sharding
    // Host-side upper bound of block_hashmap's Size().
    int64_t size_bound = 0;
    // Device scratch for the frame-stream path.
    int32_t* frame_indices = nullptr;
    int64_t frame_indices_capacity = 0;
    int32_t* frame_count = nullptr;
    int32_t* scratch_buf_indices = nullptr;
    int64_t scratch_capacity = 0;
    int32_t frame_stamp = 0;
    // Pinned read-back of {heap_top, error flags} behind each frame's touch.
    int* size_host = nullptr;
    hipEvent_t size_event = nullptr;
    bool size_event_pending = false;
    // Frame-stream fast path (stream_path.h): double-buffered prepared pixel
    // records and block lists, a ring of 4 device counters and a host-mapped
    // status word written by the integrate role.
    PixelRec* recs[2][kMaxGroup] = {};
    int64_t recs_pixels = 0;
    FrameBlock* lists[2] = {nullptr, nullptr};
    ReadyEntry* ready[2] = {nullptr, nullptr};  // same capacity as the lists
    int* front_tickets = nullptr;               // device int[2][16]
    int64_t lists_capacity = 0;
    int* ring_counters = nullptr;        // device int[4]
    // o3dmi_vbg_ray_cast_dev without a caller's range map: the grid's own,
    // left clean ({lo, hi} in every cell) by the ray cast that consumed it
    float* own_range = nullptr;
    int64_t own_range_cells = 0;
    bool own_range_clean = false;
    float own_range_lo = 0, own_range_hi = 0;
    // ... and, for images whose ray cast runs in rounds, the last cast's
    // per-tile durations and this cast's tile order (longest first)
    unsigned long long* rc_cost = nullptr;
    int* rc_order = nullptr;
    int64_t rc_tiles = 0;
    unsigned rc_seq = 0;
    volatile int* stream_status = nullptr;  // host-mapped int[8]: [0..3]
                                         // published by the integrate roles,
                                         // [4..7] by the groups' last touch
                                         // workgroup (stream_path.h)
    int touch_seen_stamp = 0;            // newest touch status taken in
    int touch_seen_size = 0;
    int stream_overflow = 0;             // stamp of the first group that ran
                                         // out of buffer indices (sticky until
                                         // StreamIntegrate has recovered)
    // What recent groups really added to the map (for the run-ahead policy).
    int recent_new[4] = {0, 0, 0, 0};
    int recent_n = 0;
    int64_t stream_seq = 0;              // groups issued on the fast path
    int known_size = 0;                  // map size after frame `known_stamp`
    int known_stamp = 0;
    bool known_valid = false;            // false after any non-stream activation
    // Prepare-pass tables (stream_path.h PrepTables) and what they were
    // built for.
    int* prep_col = nullptr;
    int* prep_row = nullptr;
    std::vector<int> prep_host;
    double prep_key[24] = {0};
    bool prep_valid = false, prep_div_short = false, prep_identity = false;
    int last_count = 1024;
    // Which path integrated the most recent frame (for
    // o3dmi_vbg_export_last_frame_blocks): 0 none, 1 frame-stream, 2 generic.
    int last_path = 0;
    int64_t last_seq = 0;                // frame-stream group sequence number
    // bench.py measurement hook (o3dmi_vbg_profile_begin/end).
    bool profiling = false;
    std::vector<hipEvent_t> prof_events;  // 2 per frame, around the launch carrying the integrate work
    int prof_frames = 0, prof_max = 0, prof_stride = 1, prof_seen = 0;
    int64_t prof_launch_frames = 0;  // frames carried by the bracketed launches
    int32_t* prof_counts = nullptr;  // device: [prof_max] block-frames, then
                                     // [prof_max] distinct blocks, then
                                     // [prof_max] map size, per launch
    int64_t prof_distinct_blocks = 0;  // of the last profile_end
    // per bracketed launch of the last profile_end (o3dmi_vbg_profile_launches)
    std::vector<float> prof_launch_ms;
    std::vector<int32_t> prof_launch_counts;  // 3 x launches, as prof_counts

    // Sliced block touch (sliced_path.h): block-ownership sharding with the
    // touch split over the ranks. Buffers are per grid; the side stream runs
    // chunk c + 1's touch / exchange / apply while the caller's stream runs
    // chunk c's integrate launches.
    struct Sliced {
        hipStream_t side = nullptr;
        hipEvent_t ev_side[2] = {nullptr, nullptr};  // chunk set ready
        hipEvent_t ev_main[2] = {nullptr, nullptr};  // chunk set consumed
        hipEvent_t ev_enter = nullptr;
        ChunkTable send_table = {}, recv_table = {};
        int table_slots = 0;  // sender table (doubles on every rank alike)
        int recv_slots = 0;   // receiver table (may grow on one rank alone)
        int capacity = 0;  // records of a wire segment
        int world = 0;
        // the ranks' agreed verdict on the proven short divisions for one
        // truncation distance (0 = not asked yet, 1 = all have them, -1 = no)
        float agreed_trunc = 0.0f;
        int agreed_fast_div = 0;
        const void* agreed_comm = nullptr;
        void* send_seg[2] = {nullptr, nullptr};
        void* gathered[2] = {nullptr, nullptr};
        ChunkEntry* entries[2] = {nullptr, nullptr};  // [entries_cap]
        int* entries_count[2] = {nullptr, nullptr};   // device int
        int entries_cap = 0;
        SliceFrame* frames_dev = nullptr;   // touch: pose (inverse extrinsic)
        IntegFrame* iframes_dev = nullptr;  // integrate: extrinsic + images
        int64_t frames_cap = 0;
        std::vector<SliceFrame> frames_host;
        std::vector<IntegFrame> iframes_host;
        // records form: the prepared records of a chunk's frames, two sets of
        // kChunkFrames images of (pixels + 1) records
        PixelRec* chunk_recs[2] = {nullptr, nullptr};
        int64_t chunk_recs_pixels = 0;
        int chunk_recs_frames = 0;  // frames each set holds
        int64_t chunks_done = 0;  // statistics (o3dmi_vbg_sliced_stats)
        int64_t reapplied = 0;
    } sliced;
    int sliced_slots_wanted = 8192;
    // o3dmi_vbg_allgather_owned_blocks has replicated the other ranks' blocks
    // here: a further owner-partitioned merge would send them back to their
    // owners and count their weights again
    bool replicated = false;

    int AttrIndex(const char* name) const {
        for (size_t i = 0; i < attr_names.size(); ++i)
            if (attr_names[i] == name) return (int)i;
        return -1;
    }
};

namespace {

int DtypeSize(int dt) {
    switch (dt) {
        case O3DMI_F32: return 4;
        case O3DMI_F64: return 8;
        case O3DMI_U16: return 2;
        case O3DMI_U8: return 1;
        case O3DMI_I32: return 4;
        case O3DMI_I64: return 8;
        default: return 0;
    }
}

// HashMap::Activate's pre-amble (HashMap.cpp:166-176): if Size()+length would
// exceed the capacity, Reserve(max(new_size, 2*capacity)). `size_bound` avoids
// the Size() read-back while it proves no growth is needed.
int EnsureCapacity(o3dmi_vbg* g, int64_t length, o3dmi_stream_t stream) {
    if (g->size_event_pending) {
        O3DMI_HIP_CHECK(hipEventSynchronize(g->size_event));
        g->size_event_pending = false;
        g->size_bound = g->size_host[0];
    }
    int64_t capacity = o3dmi_hash_capacity(g->block_hashmap);
    if (g->size_bound + length <= capacity) {
        g->size_bound += length;
        return O3DMI_OK;
    }
    int64_t size = 0;
    int st = o3dmi_hash_size(g->block_hashmap, stream, &size);
    if (st) return st;
    int64_t new_size = size + length;
    if (new_size > capacity) {
        int64_t target = new_size > capacity * 2 ? new_size : capacity * 2;
        st = o3dmi_hash_reserve(g->block_hashmap, target, stream);
        if (st) return st;
    }
    g->size_bound = new_size;
    return O3DMI_OK;
}

int EnsureScratch(o3dmi_vbg* g, int64_t m) {
    if (m <= g->scratch_capacity) return O3DMI_OK;
    (void)hipFree(g->scratch_buf_indices);
    g->scratch_buf_indices = nullptr;
    O3DMI_HIP_CHECK(hipMalloc((void**)&g->scratch_buf_indices,
                              sizeof(int32_t) * (size_t)m));
    g->scratch_capacity = m;
    return O3DMI_OK;
}

int GridDtype(o3dmi_vbg* g, int* out) {
    int wi = g->AttrIndex("weight");
    int ci = g->AttrIndex("color");
    int wdt = wi >= 0 ? g->attr_dtypes[(size_t)wi] : O3DMI_F32;
    int cdt = ci >= 0 ? g->attr_dtypes[(size_t)ci] : wdt;
    if (wdt == O3DMI_F32 && cdt == O3DMI_F32) *out = O3DMI_F32;
    else if (wdt == O3DMI_U16 && cdt == O3DMI_U16) *out = O3DMI_U16;
    else {
        SetLastError(
                "Unsupported value data type combination. Expected (float, "
                "float) or (uint16, uint16)");
        return O3DMI_ERR_INVALID_ARG;
    }
    return O3DMI_OK;
}

int RunIntegrate(o3dmi_vbg* g, const int32_t* indices, int64_t n,
                 const int32_t* n_dev, const void* depth, int drows, int dcols,
                 const void* color, int crows, int ccols, int input_dtype,
                 const double* Kd, const double* Kc, const double* T,
                 float depth_scale, float depth_max, float trunc_mult,
                 o3dmi_stream_t stream) {
    int ti = g->AttrIndex("tsdf"), wi = g->AttrIndex("weight"),
        ci = g->AttrIndex("color");
    if (ti < 0 || wi < 0) {
        SetLastError(
                "TSDF and/or weight not allocated in blocks, please implement "
                "customized integration.");
        return O3DMI_ERR_INVALID_ARG;
    }
    O3DMI_REQUIRE(g->attr_dtypes[(size_t)ti] == O3DMI_F32,
                  "tsdf must be Float32");
    int grid_dtype;
    int st = GridDtype(g, &grid_dtype);
    if (st) return st;
    bool integrate_color = color != nullptr && (int64_t)crows * ccols > 0;
    void* cbuf = (ci >= 0 && integrate_color)
                         ? o3dmi_hash_value_buffer(g->block_hashmap, ci)
                         : nullptr;
    return o3dmi_vbg_integrate(
            depth, drows, dcols, integrate_color ? color : nullptr, crows,
            ccols, input_dtype, indices, n, n_dev,
            o3dmi_hash_key_buffer(g->block_hashmap),
            (float*)o3dmi_hash_value_buffer(g->block_hashmap, ti),
            o3dmi_hash_value_buffer(g->block_hashmap, wi), cbuf, grid_dtype, Kd,
            Kc ? Kc : Kd, T, (int)g->block_resolution, g->voxel_size,
            g->voxel_size * trunc_mult, depth_scale, depth_max, stream);
}

}  // namespace

namespace {

// Block keys of a frame-stream group list -> {n,3} int32, count copied.
__global__ void ExportListKeysKernel(const FrameBlock* __restrict__ list,
                                     const int* __restrict__ count,
                                     int64_t capacity,
                                     int32_t* __restrict__ out_keys,
                                     int32_t* __restrict__ out_count) {
    int64_t n = *count;
    if (n > capacity) n = capacity;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const FrameBlock b = list[i];
        out_keys[3 * i + 0] = b.x;
        out_keys[3 * i + 1] = b.y;
        out_keys[3 * i + 2] = b.z;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = (int32_t)n;
}

// Same from a list of buffer indices (generic path).
__global__ void ExportIndexKeysKernel(const int32_t* __restrict__ indices,
                                      const int* __restrict__ count,
                                      int64_t capacity,
                                      const int32_t* __restrict__ key_buffer,
                                      int32_t* __restrict__ out_keys,
                                      int32_t* __restrict__ out_count) {
    int64_t n = *count;
    if (n > capacity) n = capacity;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t* k = key_buffer + 3 * (int64_t)indices[i];
        out_keys[3 * i + 0] = k[0];
        out_keys[3 * i + 1] = k[1];
        out_keys[3 * i + 2] = k[2];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = (int32_t)n;
}


// Folds foreign blocks into the grid (o3dmi_vbg_merge_blocks): one thread per
// voxel, block b of the foreign set lands in buffer row indices[b]. Streaming:
// reads both sides once, writes the grid side once.
template <typename W>
__global__ void MergeBlocksKernel(const int32_t* __restrict__ indices,
                                  int64_t n_voxels_total, int voxels_per_block,
                                  float* __restrict__ tsdf,
                                  W* __restrict__ weight, W* __restrict__ color,
                                  const float* __restrict__ src_tsdf,
                                  const W* __restrict__ src_weight,
                                  const W* __restrict__ src_color) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
         i < n_voxels_total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / voxels_per_block;
        const int64_t v = i - b * voxels_per_block;
        const int64_t d = (int64_t)indices[b] * voxels_per_block + v;
        const float w2 = (float)src_weight[i];
        if (w2 == 0) continue;
        const float w1 = (float)weight[d];
        if (w1 == 0) {
            tsdf[d] = src_tsdf[i];
            weight[d] = src_weight[i];
            if (color)
                for (int c = 0; c < 3; ++c) color[3 * d + c] = src_color[3 * i + c];
            continue;
        }
        const float inv = 1.0f / (w1 + w2);
        tsdf[d] = (w1 * tsdf[d] + w2 * src_tsdf[i]) * inv;
        if (color)
            for (int c = 0; c < 3; ++c)
                color[3 * d + c] = (W)((w1 * (float)color[3 * d + c] +
                                        w2 * (float)src_color[3 * i + c]) *
                                       inv);
        // a uint16 weight saturates (long per-rank streams can pass 65535;
        // float -> uint16 of a larger value is undefined)
        const float wsum = w1 + w2;
        if constexpr (sizeof(W) == 2)
            weight[d] = (W)(wsum < 65535.0f ? wsum : 65535.0f);
        else
            weight[d] = (W)wsum;
    }
}

// ---- owner-partitioned exchange of a frame-sharded grid -----------------------
constexpr int kMaxWorld = 64;
struct OwnerOffsets {
    int v[kMaxWorld];
};

// owner of every active block + blocks per owner
__global__ void OwnerCountKernel(const int32_t* __restrict__ active, int64_t n,
                                 const int* __restrict__ key_buffer, int world,
                                 int32_t* __restrict__ owner,
                                 int* __restrict__ counts) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int* k = key_buffer + 3 * (int64_t)active[i];
        const int o = OwnerOf(PackKey(k[0], k[1], k[2]), world);
        owner[i] = o;
        atomicAdd(&counts[o], 1);
    }
}

// buffer indices grouped by owner (order inside a group: as the atomics fall;
// nothing downstream depends on it -- rows are matched by key)
__global__ void OwnerGroupKernel(const int32_t* __restrict__ active, int64_t n,
                                 const int32_t* __restrict__ owner,
                                 OwnerOffsets offsets, int* __restrict__ cursor,
                                 int32_t* __restrict__ grouped) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int o = owner[i];
        grouped[offsets.v[o] + atomicAdd(&cursor[o], 1)] = active[i];
    }
}

// value rows of erased blocks back to the all-zero state a fresh buffer has
// (Activate never touches values: HashBackendBuffer.cpp:16-78)
__global__ void ZeroRowsKernel(uint8_t* __restrict__ rows,
                               const int32_t* __restrict__ indices, int64_t n,
                               int64_t row_units) {
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
        uint4* d = (uint4*)rows + (int64_t)indices[r] * row_units;
        for (int64_t i = threadIdx.x; i < row_units; i += blockDim.x)
            d[i] = make_uint4(0, 0, 0, 0);
    }
}

}  // namespace

static void FreeSliced(o3dmi_vbg* g);

extern "C" {

int o3dmi_vbg_create(int n_attrs, const char* const* attr_names,
                     const int* attr_dtypes, const int* attr_channels,
                     float voxel_size, int64_t block_resolution,
                     int64_t block_count, o3dmi_stream_t stream,
                     o3dmi_vbg_t** out) {
    O3DMI_REQUIRE(out != nullptr, "out is null");
    O3DMI_REQUIRE(voxel_size > 0, "voxel size must be positive");
    O3DMI_REQUIRE(block_resolution > 0, "block resolution must be positive");
    O3DMI_REQUIRE(n_attrs > 0 && n_attrs <= 8 && attr_names && attr_dtypes &&
                          attr_channels,
                  "Number of attribute dtypes/channels mismatch with names.");
    auto* g = new o3dmi_vbg();
    g->voxel_size = voxel_size;
    g->block_resolution = block_resolution;
    int64_t dsizes[8];
    int64_t res3 = block_resolution * block_resolution * block_resolution;
    for (int i = 0; i < n_attrs; ++i) {
        int sz = DtypeSize(attr_dtypes[i]);
        if (sz == 0 || attr_channels[i] <= 0) {
            delete g;
            SetLastError("bad attribute dtype / channels");
            return O3DMI_ERR_INVALID_ARG;
        }
        g->attr_names.emplace_back(attr_names[i]);
        g->attr_dtypes.push_back(attr_dtypes[i]);
        g->attr_channels.push_back(attr_channels[i]);
        dsizes[i] = res3 * attr_channels[i] * sz;
    }
    int st = o3dmi_hash_create(block_count, n_attrs, dsizes, stream,
                               &g->block_hashmap);
    if (st == O3DMI_OK) {
        hipError_t e = hipMalloc((void**)&g->frame_count, sizeof(int32_t) * 4);
        if (e == hipSuccess)
            e = hipHostMalloc((void**)&g->size_host, sizeof(int) * 4);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&g->size_event, hipEventDisableTiming);
        if (e != hipSuccess) st = O3DMI_ERR_HIP;
    }
    if (st != O3DMI_OK) {
        o3dmi_vbg_destroy(g);
        return st;
    }
    // The frame stream's short division forms are proven per truncation
    // distance on the device, asynchronously (vbg_stream.hip): start the proof
    // for the API's default multiplier now, so that it is over before frames
    // arrive.
    if (block_resolution % 4 == 0)
        (void)PrefetchFastDivision(voxel_size * 8.0f, false);
    *out = g;
    return O3DMI_OK;
}

int o3dmi_vbg_division_forms(float voxel_size, float trunc_voxel_multiplier,
                             int wait) {
    return PrefetchFastDivision(voxel_size * trunc_voxel_multiplier, wait != 0);
}

// VoxelBlockGrid::To(device, copy) (VoxelBlockGrid.cpp, via
// HashMap::To, core/hashmap/HashMap.cpp:230-255): the same grid on another
// (or the same) device -- attribute layout and voxel size carried over, the
// block hash map cloned with its value rows, scratch state fresh.
int o3dmi_vbg_to_device(o3dmi_vbg_t* g, int device, o3dmi_vbg_t** out) {
    O3DMI_REQUIRE(g != nullptr && out != nullptr, "null argument");
    int cur = 0;
    O3DMI_HIP_CHECK(hipGetDevice(&cur));
    o3dmi_hash_t* hm = nullptr;
    int st = o3dmi_hash_to_device(g->block_hashmap, device, &hm);
    if (st) return st;
    auto* n = new o3dmi_vbg();
    n->voxel_size = g->voxel_size;
    n->block_resolution = g->block_resolution;
    n->attr_names = g->attr_names;
    n->attr_dtypes = g->attr_dtypes;
    n->attr_channels = g->attr_channels;
    n->owner_rank = g->owner_rank;
    n->owner_world = g->owner_world;
    n->block_hashmap = hm;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess)
        e = hipMalloc((void**)&n->frame_count, sizeof(int32_t) * 4);
    if (e == hipSuccess)
        e = hipHostMalloc((void**)&n->size_host, sizeof(int) * 4);
    if (e == hipSuccess)
        e = hipEventCreateWithFlags(&n->size_event, hipEventDisableTiming);
    if (e != hipSuccess) {
        o3dmi_vbg_destroy(n);
        (void)hipSetDevice(cur);
        SetLastError(std::string("o3dmi_vbg_to_device: ") +
                     hipGetErrorString(e));
        return O3DMI_ERR_HIP;
    }
    (void)hipSetDevice(cur);
    *out = n;
    return O3DMI_OK;
}

int o3dmi_vbg_destroy(o3dmi_vbg_t* g) {
    if (!g) return O3DMI_OK;
    o3dmi_hash_destroy(g->block_hashmap);
    o3dmi_hash_destroy(g->frustum_hashmap);
    (void)hipFree(g->frame_indices);
    (void)hipFree(g->frame_count);
    (void)hipFree(g->scratch_buf_indices);
    if (g->size_host) (void)hipHostFree(g->size_host);
    if (g->size_event) (void)hipEventDestroy(g->size_event);
    for (int i = 0; i < 2; ++i) {
        for (int f = 0; f < kMaxGroup; ++f) (void)hipFree(g->recs[i][f]);
        (void)hipFree(g->lists[i]);
        (void)hipFree(g->ready[i]);
        if (i == 0) (void)hipFree(g->prep_col);
    }
    (void)hipFree(g->front_tickets);
    (void)hipFree(g->ring_counters);
    (void)hipFree(g->own_range);
    (void)hipFree(g->rc_cost);
    (void)hipFree(g->rc_order);
    if (g->stream_status) (void)hipHostFree((void*)g->stream_status);
    for (hipEvent_t e : g->prof_events) (void)hipEventDestroy(e);
    (void)hipFree(g->prof_counts);
    FreeSliced(g);
    delete g;
    return O3DMI_OK;
}

int o3dmi_vbg_set_block_ownership(o3dmi_vbg_t* g, int rank, int world) {
    O3DMI_REQUIRE(g != nullptr, "grid is null");
    int st = o3dmi_hash_set_ownership(g->block_hashmap, rank, world);
    if (st) return st;
    if (g->frustum_hashmap &&
        (st = o3dmi_hash_set_ownership(g->frustum_hashmap, rank, world)))
        return st;
    g->owner_rank = rank;
    g->owner_world = world;
    return O3DMI_OK;
}

o3dmi_hash_t* o3dmi_vbg_hashmap(o3dmi_vbg_t* g) {
    return g ? g->block_hashmap : nullptr;
}

void* o3dmi_vbg_attribute(o3dmi_vbg_t* g, const char* name, int* dtype,
                          int* channels) {
    if (!g || !name) return nullptr;
    int i = g->AttrIndex(name);
    if (i < 0) return nullptr;  // "Attribute {} not found, return empty tensor."
    if (dtype) *dtype = g->attr_dtypes[(size_t)i];
    if (channels) *channels = g->attr_channels[(size_t)i];
    return o3dmi_hash_value_buffer(g->block_hashmap, i);
}

int o3dmi_vbg_get_unique_block_coordinates(
        o3dmi_vbg_t* g, const void* depth_dev, int depth_dtype, int rows,
        int cols, const double* intrinsic, const double* extrinsic,
        float depth_scale, float depth_max, float trunc_voxel_multiplier,
        int32_t* out_coords_dev, int64_t* m_out, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && depth_dev && out_coords_dev && m_out, "null argument");
    const int64_t down_factor = 4;
    const int64_t est_sample_multiplier = 4;
    int64_t capacity = (cols / down_factor) * (rows / down_factor) *
                       est_sample_multiplier;
    O3DMI_REQUIRE(capacity > 0, "depth image too small");
    if (g->frustum_hashmap == nullptr || g->frustum_capacity < capacity) {
        o3dmi_hash_destroy(g->frustum_hashmap);
        g->frustum_hashmap = nullptr;
        int st = o3dmi_hash_create(capacity, 0, nullptr, stream,
                                   &g->frustum_hashmap);
        if (st) return st;
        g->frustum_capacity = capacity;
        if ((st = o3dmi_hash_set_ownership(g->frustum_hashmap, g->owner_rank,
                                           g->owner_world)))
            return st;
    }
    int st = o3dmi_vbg_depth_touch(
            g->frustum_hashmap, depth_dev, depth_dtype, rows, cols, intrinsic,
            extrinsic, out_coords_dev, capacity, g->frame_count,
            (int)g->block_resolution, g->voxel_size,
            g->voxel_size * trunc_voxel_multiplier, depth_scale, depth_max,
            (int)down_factor, stream);
    if (st) return st;
    // The count rides to pinned memory in front of the size query's own copy
    // and wait: one round trip to the host per call (as upstream: the size of
    // the returned tensor), not one per word.
    g->size_host[2] = 0;
    O3DMI_HIP_CHECK(hipMemcpyAsync(&g->size_host[2], g->frame_count,
                                   sizeof(int32_t), hipMemcpyDeviceToHost,
                                   (hipStream_t)stream));
    int64_t dummy = 0;
    st = o3dmi_hash_size(g->frustum_hashmap, stream, &dummy);  // syncs + errors
    if (st) return st;
    const int32_t count = g->size_host[2];
    *m_out = count;
    if (count == 0) {
        SetLastError(o3dmi_status_string(O3DMI_ERR_NO_BLOCKS));
        return O3DMI_ERR_NO_BLOCKS;
    }
    return O3DMI_OK;
}

int o3dmi_vbg_get_unique_block_coordinates_pcd(
        o3dmi_vbg_t* g, const float* points_dev, int64_t n,
        float trunc_voxel_multiplier, int32_t* out_coords_dev,
        int64_t out_capacity, int64_t* m_out, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && m_out && n >= 0 && out_capacity >= 0, "null argument");
    *m_out = 0;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(points_dev && out_coords_dev, "null argument");
    const int64_t est_neighbor_multiplier = 8;  // VoxelBlockGrid.cpp:251
    const int64_t capacity = n * est_neighbor_multiplier;
    if (g->frustum_hashmap == nullptr || g->frustum_capacity < capacity) {
        o3dmi_hash_destroy(g->frustum_hashmap);
        g->frustum_hashmap = nullptr;
        g->frustum_capacity = 0;
        int st = o3dmi_hash_create(capacity, 0, nullptr, stream,
                                   &g->frustum_hashmap);
        if (st) return st;
        g->frustum_capacity = capacity;
        if ((st = o3dmi_hash_set_ownership(g->frustum_hashmap, g->owner_rank,
                                           g->owner_world)))
            return st;
    }
    int st = o3dmi_vbg_pointcloud_touch(
            g->frustum_hashmap, points_dev, n, out_coords_dev, out_capacity,
            g->frame_count, (int)g->block_resolution, g->voxel_size,
            g->voxel_size * trunc_voxel_multiplier, stream);
    if (st) return st;
    g->size_host[2] = 0;
    O3DMI_HIP_CHECK(hipMemcpyAsync(&g->size_host[2], g->frame_count,
                                   sizeof(int32_t), hipMemcpyDeviceToHost,
                                   (hipStream_t)stream));
    int64_t dummy = 0;
    st = o3dmi_hash_size(g->frustum_hashmap, stream, &dummy);  // syncs + errors
    if (st) return st;
    const int64_t count = g->size_host[2];
    if (count > out_capacity) {
        *m_out = out_capacity;
        SetLastError("GetUniqueBlockCoordinates: more blocks than "
                     "out_capacity rows");
        return O3DMI_ERR_CAPACITY;
    }
    *m_out = count;
    return O3DMI_OK;
}

int o3dmi_vbg_get_voxel_indices(o3dmi_vbg_t* g, const int32_t* buf_indices_dev,
                                int64_t n_blocks, int64_t* voxel_indices_dev,
                                o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g, "null argument");
    return o3dmi_vbg_voxel_indices(buf_indices_dev, n_blocks,
                                   (int)g->block_resolution, voxel_indices_dev,
                                   stream);
}

int o3dmi_vbg_get_voxel_coordinates(o3dmi_vbg_t* g,
                                    const int64_t* voxel_indices_dev,
                                    int64_t n_voxels, int64_t* voxel_coords_dev,
                                    o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && n_voxels >= 0, "null argument");
    if (n_voxels == 0) return O3DMI_OK;
    // the word the kernel flags a bad buffer index in: the frame counter's
    // neighbour of the grid's device scratch, read back with the result
    int32_t* err = g->frame_count + 1;
    O3DMI_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int32_t),
                                   (hipStream_t)stream));
    int st = o3dmi_vbg_voxel_coordinates(
            voxel_indices_dev, n_voxels,
            o3dmi_hash_key_buffer(g->block_hashmap),
            o3dmi_hash_capacity(g->block_hashmap), (int)g->block_resolution,
            voxel_coords_dev, err, stream);
    if (st) return st;
    g->size_host[2] = 0;
    O3DMI_HIP_CHECK(hipMemcpyAsync(&g->size_host[2], err, sizeof(int32_t),
                                   hipMemcpyDeviceToHost,
                                   (hipStream_t)stream));
    O3DMI_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    O3DMI_REQUIRE(g->size_host[2] == 0,
                  "GetVoxelCoordinates: buffer index out of range");
    return O3DMI_OK;
}

int o3dmi_vbg_get_voxel_coordinates_and_flattened_indices(
        o3dmi_vbg_t* g, const int32_t* buf_indices_dev, int64_t n_blocks,
        float* voxel_coords_dev, int64_t* flattened_indices_dev,
        o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g, "null argument");
    return o3dmi_vbg_voxel_coordinates_and_flattened_indices(
            buf_indices_dev, n_blocks, o3dmi_hash_key_buffer(g->block_hashmap),
            (int)g->block_resolution, g->voxel_size, voxel_coords_dev,
            flattened_indices_dev, stream);
}

int o3dmi_vbg_integrate_blocks(o3dmi_vbg_t* g, const int32_t* block_coords_dev,
                               int64_t m, const void* depth_dev,
                               int depth_rows, int depth_cols,
                               const void* color_dev, int color_rows,
                               int color_cols, int input_dtype,
                               const double* depth_intrinsic,
                               const double* color_intrinsic,
                               const double* extrinsic, float depth_scale,
                               float depth_max, float trunc_voxel_multiplier,
                               o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && block_coords_dev && depth_dev && depth_intrinsic &&
                          extrinsic,
                  "null argument");
    O3DMI_REQUIRE(m >= 0, "m < 0");
    if (m == 0) return O3DMI_OK;
    g->known_valid = false;
    int st = EnsureCapacity(g, m, stream);
    if (st) return st;
    if ((st = EnsureScratch(g, m))) return st;
    // block_hashmap_->Activate(block_coords, ...); ->Find(block_coords, ...)
    st = o3dmi_hash_activate(g->block_hashmap, block_coords_dev, m, nullptr,
                             nullptr, nullptr, stream);
    if (st) return st;
    st = o3dmi_hash_find(g->block_hashmap, block_coords_dev, m, nullptr,
                         g->scratch_buf_indices, nullptr, stream);
    if (st) return st;
    return RunIntegrate(g, g->scratch_buf_indices, m, nullptr, depth_dev,
                        depth_rows, depth_cols, color_dev, color_rows,
                        color_cols, input_dtype, depth_intrinsic,
                        color_intrinsic, extrinsic, depth_scale, depth_max,
                        trunc_voxel_multiplier, stream);
}

static int IntegrateFrameGeneric(o3dmi_vbg_t* g, const void* depth_dev,
                              int depth_rows, int depth_cols,
                              const void* color_dev, int color_rows,
                              int color_cols, int input_dtype,
                              const double* depth_intrinsic,
                              const double* color_intrinsic,
                              const double* extrinsic, float depth_scale,
                              float depth_max, float trunc_voxel_multiplier,
                              o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && depth_dev && depth_intrinsic && extrinsic,
                  "null argument");
    const int stride = 4;
    int64_t max_new = (int64_t)(depth_cols / stride) * (depth_rows / stride) * 4;
    O3DMI_REQUIRE(max_new > 0, "depth image too small");
    if (g->frame_indices_capacity < max_new) {
        (void)hipFree(g->frame_indices);
        g->frame_indices = nullptr;
        O3DMI_HIP_CHECK(hipMalloc((void**)&g->frame_indices,
                                  sizeof(int32_t) * (size_t)max_new));
        g->frame_indices_capacity = max_new;
    }
    // Capacity policy (HashMap::Activate, HashMap.cpp:166-176, with `length`
    // = the most blocks one frame can create). The exact map size after the
    // previous frame's activation was copied to pinned memory right behind
    // that frame's touch kernel; waiting for it does not drain the GPU (the
    // previous frame's integrate kernel is still running) and keeps the
    // Reserve decision exact instead of heuristic.
    hipStream_t s = (hipStream_t)stream;
    if (g->size_event_pending) {
        O3DMI_HIP_CHECK(hipEventSynchronize(g->size_event));
        g->size_event_pending = false;
        if (g->size_host[1] & kErrKeyRange) {
            SetLastError("block coordinate outside +-2^20");
            return O3DMI_ERR_KEY_RANGE;
        }
        if (g->size_host[1] & kErrCapacity) {
            SetLastError("hash map capacity exceeded");
            return O3DMI_ERR_CAPACITY;
        }
        g->size_bound = g->size_host[0];
    }
    int64_t capacity = o3dmi_hash_capacity(g->block_hashmap);
    if (g->size_bound + max_new > capacity) {
        int64_t size = 0;
        int st = o3dmi_hash_size(g->block_hashmap, stream, &size);
        if (st) return st;
        g->size_bound = size;
        if (size + max_new > capacity) {
            int64_t need = size + max_new;
            int64_t target = need > capacity * 2 ? need : capacity * 2;
            st = o3dmi_hash_reserve(g->block_hashmap, target, stream);
            if (st) return st;
        }
    }

    int dt = input_dtype == O3DMI_F32 ? O3DMI_F32 : O3DMI_U16;
    g->known_valid = false;
    g->frame_stamp += 1;
    const bool prof = g->profiling && g->prof_frames < g->prof_max &&
                      g->prof_stride > 0 &&
                      (g->prof_seen++ % g->prof_stride) == 0;
    int st = o3dmi_vbg_touch_activate(
            g->block_hashmap, depth_dev, dt, depth_rows, depth_cols,
            depth_intrinsic, extrinsic, g->frame_indices, max_new,
            g->frame_count, (int)g->block_resolution, g->voxel_size,
            g->voxel_size * trunc_voxel_multiplier, depth_scale, depth_max,
            stride, g->frame_stamp, stream);
    if (st) return st;
    O3DMI_HIP_CHECK(hipMemcpyAsync(g->size_host, g->block_hashmap->view.counters,
                                   sizeof(int) * 2, hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipEventRecord(g->size_event, s));
    g->size_event_pending = true;
    g->size_bound += max_new;  // until the read-back lands
    if (prof) {
        O3DMI_HIP_CHECK(hipMemcpyAsync(g->prof_counts + g->prof_frames,
                                       g->frame_count, sizeof(int32_t),
                                       hipMemcpyDeviceToDevice, s));
        O3DMI_HIP_CHECK(hipEventRecord(
                g->prof_events[(size_t)g->prof_frames * 2 + 0], s));
    }
    g->last_path = 2;
    st = RunIntegrate(g, g->frame_indices, max_new, g->frame_count, depth_dev,
                      depth_rows, depth_cols, color_dev, color_rows,
                      color_cols, input_dtype, depth_intrinsic,
                      color_intrinsic, extrinsic, depth_scale, depth_max,
                      trunc_voxel_multiplier, stream);
    if (prof) {
        O3DMI_HIP_CHECK(hipEventRecord(
                g->prof_events[(size_t)g->prof_frames * 2 + 1], s));
        g->prof_launch_frames += 1;
        g->prof_frames += 1;
    }
    return st;
}


// ---- frame-stream fast path ------------------------------------------------

static int EnsureStreamBuffers(o3dmi_vbg* g, int rows, int cols,
                               int64_t list_cap) {
    const int64_t px = (int64_t)rows * cols;
    if (g->recs_pixels < px) {
        for (int i = 0; i < 2; ++i)
            for (int f = 0; f < kMaxGroup; ++f) {
                (void)hipFree(g->recs[i][f]);
                g->recs[i][f] = nullptr;
                // + 1: the sentinel record behind the image
                O3DMI_HIP_CHECK(hipMalloc((void**)&g->recs[i][f],
                                          sizeof(PixelRec) * (size_t)(px + 1)));
            }
        g->recs_pixels = px;
    }
    if (g->lists_capacity < list_cap) {
        for (int i = 0; i < 2; ++i) {
            (void)hipFree(g->lists[i]);
            g->lists[i] = nullptr;
            O3DMI_HIP_CHECK(hipMalloc((void**)&g->lists[i],
                                      sizeof(FrameBlock) * (size_t)list_cap));
            (void)hipFree(g->ready[i]);
            g->ready[i] = nullptr;
            O3DMI_HIP_CHECK(hipMalloc((void**)&g->ready[i],
                                      sizeof(ReadyEntry) * (size_t)list_cap));
        }
        g->lists_capacity = list_cap;
    }
    if (!g->front_tickets) {
        O3DMI_HIP_CHECK(hipMalloc((void**)&g->front_tickets, sizeof(int) * 32));
        O3DMI_HIP_CHECK(hipMemset(g->front_tickets, 0, sizeof(int) * 32));
    }
    if (!g->ring_counters) {
        O3DMI_HIP_CHECK(hipMalloc((void**)&g->ring_counters, sizeof(int) * 4));
        O3DMI_HIP_CHECK(hipMemset(g->ring_counters, 0, sizeof(int) * 4));
        int* st = nullptr;
        O3DMI_HIP_CHECK(hipHostMalloc((void**)&st, sizeof(int) * 8,
                                      hipHostMallocMapped |
                                              hipHostMallocCoherent));
        for (int i = 0; i < 8; ++i) st[i] = 0;
        g->stream_status = st;
    }
    return O3DMI_OK;
}

// Builds / re-uses the prepare-pass tables for this image geometry.
static int EnsurePrepTables(o3dmi_vbg* g, const double* dk, const double* ck,
                            int rows, int cols, int crows, int ccols,
                            float depth_scale, hipStream_t s) {
    double key[24] = {0};
    for (int i = 0; i < 9; ++i) key[i] = dk[i];
    for (int i = 0; i < 9; ++i) key[9 + i] = (ck ? ck : dk)[i];
    key[18] = rows; key[19] = cols; key[20] = crows; key[21] = ccols;
    key[22] = depth_scale;
    if (g->prep_valid && std::memcmp(key, g->prep_key, sizeof(key)) == 0)
        return O3DMI_OK;
    // the previous tables may still be read by launches in flight
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(g->prep_col);
    g->prep_col = g->prep_row = nullptr;
    g->prep_valid = false;
    O3DMI_HIP_CHECK(hipMalloc((void**)&g->prep_col,
                              sizeof(int) * (size_t)(rows + cols)));
    g->prep_row = g->prep_col + cols;
    g->prep_host.assign((size_t)(rows + cols), -1);
    g->prep_div_short =
            PrepTables(dk, ck, rows, cols, crows, ccols, depth_scale,
                       g->prep_host.data(), g->prep_host.data() + cols);
    g->prep_identity = crows == rows && ccols == cols;
    for (int u = 0; u < cols && g->prep_identity; ++u)
        g->prep_identity = g->prep_host[(size_t)u] == u;
    for (int v = 0; v < rows && g->prep_identity; ++v)
        g->prep_identity = g->prep_host[(size_t)cols + (size_t)v] == v;
    O3DMI_HIP_CHECK(hipMemcpyAsync(g->prep_col, g->prep_host.data(),
                                   sizeof(int) * (size_t)(rows + cols),
                                   hipMemcpyHostToDevice, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    std::memcpy(g->prep_key, key, sizeof(key));
    g->prep_valid = true;
    return O3DMI_OK;
}

// Reads the status words the device publishes -- by every integrate role as
// its first action ({map size, error flags, group block count, stamp}) and by
// the last touch workgroup of every group ({map size after the group's touch,
// overflow stamp, group block count, stamp}); never blocks. `overflow` (may be
// null) receives the stamp of the first group that ran out of buffer indices,
// 0 when none did.
static int PollStreamStatus(o3dmi_vbg* g, int* overflow = nullptr) {
    const volatile int* ts = g->stream_status + 4;
    const int tstamp = __atomic_load_n((const int*)&ts[3], __ATOMIC_ACQUIRE);
    if (tstamp - g->touch_seen_stamp > 0) {
        const int size = ts[0], ovf = ts[1];
        const int tstamp2 =
                __atomic_load_n((const int*)&ts[3], __ATOMIC_ACQUIRE);
        if (tstamp2 == tstamp) {
            // what the groups since the last word seen added, per group
            const int groups = tstamp - g->touch_seen_stamp;
            if (size >= g->touch_seen_size && g->known_valid) {
                const int per = (size - g->touch_seen_size + groups - 1) / groups;
                g->recent_new[g->recent_n++ & 3] = per;
            }
            g->touch_seen_stamp = tstamp;
            g->touch_seen_size = size;
            if (tstamp - g->known_stamp > 0) {
                g->known_size = size;
                g->known_stamp = tstamp;
            }
            if (ovf != 0 && g->stream_overflow == 0) g->stream_overflow = ovf;
        }
    }
    if (overflow) *overflow = g->stream_overflow;
    const int stamp = __atomic_load_n((const int*)&g->stream_status[3],
                                      __ATOMIC_ACQUIRE);
    if (stamp != g->known_stamp && stamp != 0) {
        const int size = g->stream_status[0];
        const int err = g->stream_status[1];
        const int count = g->stream_status[2];
        // Re-check the stamp: a newer group may have overwritten the words.
        const int stamp2 = __atomic_load_n((const int*)&g->stream_status[3],
                                           __ATOMIC_ACQUIRE);
        if (stamp2 == stamp) {
            if (stamp - g->known_stamp > 0) {
                g->known_size = size < o3dmi_hash_capacity(g->block_hashmap)
                                        ? size
                                        : (int)o3dmi_hash_capacity(
                                                  g->block_hashmap);
                g->known_stamp = stamp;
            }
            g->last_count = count;
        }
        if (err & kErrKeyRange) {
            SetLastError("block coordinate outside +-2^20");
            return O3DMI_ERR_KEY_RANGE;
        }
        if (err & kErrCapacity) {
            SetLastError("hash map capacity exceeded");
            return O3DMI_ERR_CAPACITY;
        }
        if (err & (kErrTouchStamp | kErrProbe)) {
            SetLastError(err & kErrProbe
                                 ? "hash map probe sequence wrapped"
                                 : "frame-stream touch word of another group");
            return O3DMI_ERR_INTERNAL;
        }
    }
    return O3DMI_OK;
}

// ---- run-ahead capacity policy ------------------------------------------------
// HashMap::Activate reserves when Size() + M would pass the capacity
// (HashMap.cpp:166-176), M being the frame's block count, which the reference
// has on the host because it synchronises every frame. The frame stream
// issues groups ahead of the GPU, so neither Size() nor M is known when a
// group is issued. Two bounds:
//   strict   known size + (groups not yet reported + 1) x the frustum bound of
//            a group (FrustumBlockBound per frame): cannot overflow, needs no
//            confirmation. A map with a few hundred thousand blocks of
//            head-room is issued this way, as in rounds 1-3.
//   estimate the same with what recent groups REALLY added (twice the largest
//            of the last four + a margin) in place of the frustum bound. A
//            group issued on the estimate may run out of buffer indices: the
//            device then drops that group and every later one as a whole
//            (InsertKey / the last touch workgroup, vbg_stream.hip), reports
//            the stamp, and StreamIntegrate reserves and replays from the
//            dropped group's first frame. Such groups are CONFIRMED before the
//            call that issued them returns (their frames are only known to be
//            alive until then).
// A Reserve therefore happens when the map really is too small (or, drained,
// when even the estimate does not fit), not because of the frustum bound.
static int64_t EstimatedGroupNew(const o3dmi_vbg* g, int64_t strict) {
    if (g->recent_n == 0) {
        // nothing observed yet (a cold start): an eighth of the free map per
        // group in flight, so that the first groups of a stream pipeline too
        const int64_t room = o3dmi_hash_capacity(g->block_hashmap) -
                             (int64_t)g->known_size;
        const int64_t guess = room / 8 > 1024 ? room / 8 : 1024;
        return guess < strict ? guess : strict;
    }
    int m = 0;
    for (int i = 0; i < 4 && i < g->recent_n; ++i)
        if (g->recent_new[i] > m) m = g->recent_new[i];
    const int64_t est = 2 * (int64_t)m + 64;
    return est < strict ? est : strict;
}

enum class Issue { kStrict, kEstimate, kNo };

// The map size is exact again (stream drained, size read from the map).
static void SetExactSize(o3dmi_vbg* g, int64_t size) {
    g->known_size = (int)size;
    g->known_stamp = g->frame_stamp;
    g->touch_seen_stamp = g->frame_stamp;
    g->touch_seen_size = (int)size;
    g->known_valid = true;
}

// Non-blocking: may one more group be issued behind those in flight?
static Issue StreamMayIssue(o3dmi_vbg* g, int64_t strict_new, bool allow_est,
                            bool may_spin = true) {
    if (!g->known_valid) return Issue::kNo;
    const int64_t capacity = o3dmi_hash_capacity(g->block_hashmap);
    // The host runs ahead of the GPU; when a bound fails only because too
    // many issued groups have not reported their map size yet, give the
    // status words a moment to catch up instead of draining the pipeline.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        if (PollStreamStatus(g) != O3DMI_OK) return Issue::kNo;  // surfaced later
        if (g->stream_overflow != 0) return Issue::kNo;  // recovery first
        const int64_t unknown = (int64_t)g->frame_stamp - g->known_stamp;
        if ((int64_t)g->known_size + (unknown + 1) * strict_new <= capacity)
            return Issue::kStrict;
        const int64_t est = EstimatedGroupNew(g, strict_new);
        if (allow_est && est < strict_new &&
            (int64_t)g->known_size + (unknown + 1) * est <= capacity)
            return Issue::kEstimate;
        // Even a fully reported pipeline would not fit: the blocking path
        // decides (drained go-ahead or Reserve).
        if ((int64_t)g->known_size + 2 * (allow_est ? est : strict_new) >
            capacity)
            return Issue::kNo;
        if (unknown <= 1 || !may_spin) return Issue::kNo;
        if (std::chrono::steady_clock::now() - t0 >
            std::chrono::microseconds(500))
            return Issue::kNo;
    }
}

// Blocking form, for a group issued with nothing overlapping it: waits until
// every issued group has reported (they complete without the host), then
// applies the policy to the exact size. With `allow_est` a drained map that is
// not full issues the group whatever the bounds say -- the device reports an
// overflow and the caller recovers (Reserve to max(wanted, 2 x capacity), the
// reference's growth rule, then replay).
static int StreamEnsureCapacity(o3dmi_vbg* g, int64_t strict_new,
                                bool allow_est, hipStream_t s, Issue* how,
                                int* overflow) {
    *how = Issue::kStrict;
    int st = PollStreamStatus(g, overflow);
    if (st || *overflow) return st;
    if (!g->known_valid) {
        // Something else activated blocks since the last fast-path group (or
        // this is the first one): take the exact size from the map itself.
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        int64_t size = 0;
        st = o3dmi_hash_size(g->block_hashmap, (o3dmi_stream_t)s, &size);
        if (st) return st;
        SetExactSize(g, size);
        g->recent_n = 0;
    }
    const Issue quick = StreamMayIssue(g, strict_new, allow_est, false);
    if (quick != Issue::kNo) {
        *how = quick;
        return O3DMI_OK;
    }
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    // a group issued on the estimate may have overflowed the map meanwhile:
    // the caller recovers (the map must not be sized or reserved before)
    if ((st = PollStreamStatus(g, overflow)) || *overflow) return st;
    int64_t size = 0;
    st = o3dmi_hash_size(g->block_hashmap, (o3dmi_stream_t)s, &size);
    if (st) return st;
    SetExactSize(g, size);
    const int64_t capacity = o3dmi_hash_capacity(g->block_hashmap);
    // Drained, on the estimate: go unless the map is outright full -- an
    // overflow is recoverable (drop + replay), a Reserve the stream did not
    // need is not. The estimate only decides how far AHEAD groups are issued.
    const int64_t need_new = allow_est ? 1 : strict_new;
    if (size + need_new > capacity) {
        const int64_t need = size + need_new;
        const int64_t target = need > capacity * 2 ? need : capacity * 2;
        st = o3dmi_hash_reserve(g->block_hashmap, target, (o3dmi_stream_t)s);
        if (st) return st;
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    }
    *how = size + strict_new <= o3dmi_hash_capacity(g->block_hashmap)
                   ? Issue::kStrict
                   : Issue::kEstimate;
    return O3DMI_OK;
}

// Per-frame inputs of the fast path.
struct StreamFrame {
    const void* depth;
    const void* color;
    const double* extrinsic;
};
struct StreamCommon {
    int depth_rows, depth_cols, color_rows, color_cols;
    const double* depth_intrinsic;
    const double* color_intrinsic;
    float depth_scale, depth_max, trunc;
    int64_t frame_new;  // strict bound on blocks one frame can touch
    int grid_dtype;
    int ti, wi, ci;
    bool with_color;
};

// A group of up to kMaxGroup consecutive frames whose front roles have been
// (or are being) issued.
struct StreamGroup {
    int n = 0;
    int64_t seq = 0;  // group sequence number (selects the scratch buffers)
    int stamp = 0;    // touch / status stamp
    const StreamFrame* frames = nullptr;
};

// Front-role arguments of the frames of a new group; advances the stamp.
static StreamGroup MakeGroup(o3dmi_vbg* g, const StreamCommon& c,
                             const StreamFrame* frames, int n,
                             FrameFrontArgs* fa) {
    StreamGroup grp;
    grp.n = n;
    grp.seq = g->stream_seq;
    g->frame_stamp += 1;
    grp.stamp = g->frame_stamp;
    grp.frames = frames;
    const int par = (int)(grp.seq & 1);
    for (int f = 0; f < n; ++f) {
        FrameFrontArgs& a = fa[f];
        a.depth = (const uint16_t*)frames[f].depth;
        a.color = c.with_color ? (const uint8_t*)frames[f].color : nullptr;
        a.rows = c.depth_rows;
        a.cols = c.depth_cols;
        a.color_rows = c.color_rows;
        a.color_cols = c.color_cols;
        a.depth_intrinsic = c.depth_intrinsic;
        a.color_intrinsic = c.color_intrinsic ? c.color_intrinsic
                                              : c.depth_intrinsic;
        a.extrinsic = frames[f].extrinsic;
        a.resolution = (int)g->block_resolution;
        a.voxel_size = g->voxel_size;
        a.sdf_trunc = g->voxel_size * c.trunc;
        a.depth_scale = c.depth_scale;
        a.depth_max = c.depth_max;
        a.stride = 4;
        a.group_stamp = (unsigned long long)grp.stamp;
        a.group_bit = f;
        a.touch_plane = par;
        a.col_lut = g->prep_valid ? g->prep_col : nullptr;
        a.row_lut = g->prep_valid ? g->prep_row : nullptr;
        a.depth_div_short = g->prep_valid && g->prep_div_short;
        a.prep_identity = g->prep_valid && g->prep_identity;
        a.recs = g->recs[par][f];
        a.list = g->lists[par];
        a.list_capacity = g->lists_capacity;
        a.count = g->ring_counters + (grp.seq & 3);
        a.ready = g->ready[par];
        a.tickets = g->front_tickets + 16 * par;
        a.touch_status = (int*)g->stream_status + 4;
        a.prepare_only = false;
    }
    g->stream_seq += 1;
    g->size_bound = o3dmi_hash_capacity(g->block_hashmap);  // generic path: re-read
    return grp;
}

static void MakeIntegArgs(o3dmi_vbg* g, const StreamCommon& c,
                          const StreamGroup& grp, bool prof,
                          IntegrateStreamArgs* ia) {
    const int par = (int)(grp.seq & 1);
    ia->n_frames = grp.n;
    ia->group_stamp = (unsigned long long)grp.stamp;
    ia->touch_plane = par;
    for (int f = 0; f < grp.n; ++f) {
        ia->recs[f] = g->recs[par][f];
        ia->extrinsic[f] = grp.frames[f].extrinsic;
    }
    ia->rows = c.depth_rows;
    ia->cols = c.depth_cols;
    ia->with_color = c.with_color;
    ia->list = g->lists[par];
    ia->ready = g->ready[par];
    ia->count = g->ring_counters + (grp.seq & 3);
    ia->list_capacity = g->lists_capacity;
    ia->grid_hint = g->last_count;
    ia->tsdf = (float*)o3dmi_hash_value_buffer(g->block_hashmap, c.ti);
    ia->weight = o3dmi_hash_value_buffer(g->block_hashmap, c.wi);
    ia->color = c.with_color ? o3dmi_hash_value_buffer(g->block_hashmap, c.ci)
                             : nullptr;
    ia->grid_dtype = c.grid_dtype;
    ia->depth_intrinsic = c.depth_intrinsic;
    ia->resolution = (int)g->block_resolution;
    ia->voxel_size = g->voxel_size;
    ia->sdf_trunc = g->voxel_size * c.trunc;
    ia->depth_max = c.depth_max;
    ia->zero_counter = g->ring_counters + ((grp.seq + 2) & 3);
    ia->size_host = (int*)g->stream_status;
    ia->status_stamp = grp.stamp;
    ia->prof_count = prof ? g->prof_counts + g->prof_max + g->prof_frames
                          : nullptr;
    ia->prof_frame_blocks = prof ? g->prof_counts + g->prof_frames : nullptr;
    ia->prof_map_size =
            prof ? g->prof_counts + 2 * g->prof_max + g->prof_frames : nullptr;
}

static bool StreamPathApplies(const o3dmi_vbg* g, int input_dtype) {
    int ti = g->AttrIndex("tsdf"), wi = g->AttrIndex("weight");
    return input_dtype == O3DMI_U16 && (g->block_resolution % 4) == 0 &&
           ti >= 0 && wi >= 0 && g->attr_dtypes[(size_t)ti] == O3DMI_F32;
}

// Waits (spinning on the host-mapped words; no runtime call) until the touch
// of group `stamp` has reported or `overflow` is set.
static int WaitTouchReported(o3dmi_vbg* g, int stamp, hipStream_t s,
                             int* overflow) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        int st = PollStreamStatus(g, overflow);
        if (st) return st;
        if (*overflow != 0 || g->touch_seen_stamp - stamp >= 0) return O3DMI_OK;
        if (std::chrono::steady_clock::now() - t0 >
            std::chrono::milliseconds(20)) {
            // not a spin any more: let the runtime wait, then read once more
            O3DMI_HIP_CHECK(hipStreamSynchronize(s));
            if ((st = PollStreamStatus(g, overflow))) return st;
            return O3DMI_OK;
        }
    }
}

// Integrates frames[0..n) strictly in order on stream `s`, `group` frames per
// integrate launch. The front roles of group k+1 share the launch of group
// k's integrate role whenever the capacity policy allows it without waiting.
static int StreamIntegrate(o3dmi_vbg* g, const StreamCommon& c0,
                           const StreamFrame* frames, int n, int group,
                           hipStream_t s) {
    StreamCommon c = c0;
    if (group < 1) group = 1;
    if (group > kMaxGroup) group = kMaxGroup;
    c.frame_new = FrustumBlockBound(
            c.depth_intrinsic, c.depth_rows, c.depth_cols, c.depth_max,
            g->voxel_size * (float)g->block_resolution, 4);
    O3DMI_REQUIRE(c.frame_new > 0, "depth image too small");
    const int64_t group_new = c.frame_new * group;
    c.ti = g->AttrIndex("tsdf");
    c.wi = g->AttrIndex("weight");
    c.ci = g->AttrIndex("color");
    c.with_color = c.ci >= 0 && (int64_t)c.color_rows * c.color_cols > 0 &&
                   n > 0 && frames[0].color != nullptr;
    int st = GridDtype(g, &c.grid_dtype);
    if (st) return st;
    if ((st = EnsureStreamBuffers(g, c.depth_rows, c.depth_cols,
                                  c.frame_new * kMaxGroup)))
        return st;

    if ((st = EnsurePrepTables(g, c.depth_intrinsic, c.color_intrinsic,
                                      c.depth_rows, c.depth_cols,
                                      c.color_rows, c.color_cols,
                                      c.depth_scale, s))) {
        return st;
    }
    // The short division forms are proven asynchronously (vbg_stream.hip); a
    // batch call never waits for the proof (its launches take the IEEE forms
    // until it is over). A ONE-frame call is the interactive API -- a loop of
    // them is latency-bound and would run beside the proof's kernels for its
    // first ~10 ms: there the (one-time) wait is taken up front, as rounds 1-3
    // did for every caller.
    (void)PrefetchFastDivision(g->voxel_size * c.trunc, n == 1);
    // O3DMI_STRICT_CAPACITY=1 (A / B): rounds 1-3's policy, the frustum bound
    // only. Groups on the estimate need more than one frame per call to pay
    // (the confirmation is a wait).
    static const bool strict_only =
            std::getenv("O3DMI_STRICT_CAPACITY") != nullptr;
    const bool allow_est = !strict_only && n > 1;

    // Groups issued on the estimate and not yet confirmed: {stamp, first
    // frame}. All of them belong to this call.
    struct Pending {
        int stamp, f0;
    };
    std::vector<Pending> pending;
    bool issued = false;  // front roles of `cur` already in flight
    StreamGroup cur;
    FrameFrontArgs fa[kMaxGroup];
    int f = 0;
    for (;;) {
        int overflow = 0;
        while (f < n && overflow == 0) {
            if (!issued) {
                Issue how;
                if ((st = StreamEnsureCapacity(g, group_new, allow_est, s,
                                               &how, &overflow)))
                    return st;
                if (overflow != 0) break;
                const int m = n - f < group ? n - f : group;
                cur = MakeGroup(g, c, frames + f, m, fa);
                if (how == Issue::kEstimate)
                    pending.push_back({cur.stamp, f});
                if ((st = LaunchFrameStep(g->block_hashmap, fa, m, nullptr, s)))
                    return st;
            }
            g->last_path = 1;
            g->last_seq = cur.seq;
            const int next_f = f + cur.n;
            const bool prof = g->profiling && g->prof_frames < g->prof_max &&
                              g->prof_stride > 0 &&
                              (g->prof_seen++ % g->prof_stride) == 0;
            hipEvent_t* pe = prof ? &g->prof_events[(size_t)g->prof_frames * 2]
                                  : nullptr;
            IntegrateStreamArgs ia;
            MakeIntegArgs(g, c, cur, prof, &ia);
            StreamGroup nxt;
            // O3DMI_NO_FUSE=1 (diagnostics): front roles in their own launches
            // so that a kernel trace shows the two roles separately.
            static const bool no_fuse = std::getenv("O3DMI_NO_FUSE") != nullptr;
            Issue how = Issue::kNo;
            if (!no_fuse && next_f < n)
                how = StreamMayIssue(g, group_new, allow_est);
            const bool fuse = how != Issue::kNo;
            int m = 0;
            if (fuse) {
                m = n - next_f < group ? n - next_f : group;
                nxt = MakeGroup(g, c, frames + next_f, m, fa);
                if (how == Issue::kEstimate)
                    pending.push_back({nxt.stamp, next_f});
            }
            if (pe) O3DMI_HIP_CHECK(hipEventRecord(pe[0], s));
            if ((st = LaunchFrameStep(g->block_hashmap, fuse ? fa : nullptr, m,
                                      &ia, s)))
                return st;
            if (pe) {
                O3DMI_HIP_CHECK(hipEventRecord(pe[1], s));
                g->prof_launch_frames += cur.n;
                g->prof_frames += 1;
            }
            issued = fuse;
            if (fuse) cur = nxt;
            f = next_f;
            // confirmations that have arrived (never waits)
            if (!pending.empty()) {
                if ((st = PollStreamStatus(g, &overflow))) return st;
                while (!pending.empty() && overflow == 0 &&
                       g->touch_seen_stamp - pending.front().stamp >= 0)
                    pending.erase(pending.begin());
            }
        }
        // Every group issued on the estimate is confirmed before the call
        // returns: its touch reports from inside the launch BEFORE the one
        // that integrates it, so this wait ends while that launch is still
        // queued or running -- the GPU does not go idle over it.
        if (overflow == 0 && !pending.empty())
            if ((st = WaitTouchReported(g, pending.back().stamp, s, &overflow)))
                return st;
        if (overflow == 0) break;
        // A group ran out of buffer indices: it and every later group were
        // dropped on the device. Drain, make the map consistent, reserve,
        // replay from the dropped group's first frame.
        int replay_from = -1;
        for (const Pending& pg : pending)
            if (pg.stamp == overflow) replay_from = pg.f0;
        if (replay_from < 0) {
            // An overflow stamp of a group this call issued on the STRICT
            // bound (or a one-frame call): a probe wrap recorded as an
            // overflow (touch_device.h). Nothing to replay from -- but the map
            // must not stay in the "overflow not recovered" state, in which
            // every later size / export / reserve call is refused: recover
            // the slots, rebuild the crowded table, then report.
            (void)hipStreamSynchronize(s);
            int64_t w = 0;
            (void)RecoverOverflow(g->block_hashmap, s, &w);
            (void)hipMemsetAsync(g->ring_counters, 0, sizeof(int) * 4, s);
            (void)hipMemsetAsync(g->front_tickets, 0, sizeof(int) * 32, s);
            const int64_t cap = o3dmi_hash_capacity(g->block_hashmap);
            (void)o3dmi_hash_reserve(g->block_hashmap, w > cap ? w : cap,
                                     (o3dmi_stream_t)s);
            (void)hipStreamSynchronize(s);
            g->stream_overflow = 0;
            g->known_valid = false;
            SetLastError("frame stream: overflow reported for a group this "
                         "call did not issue on an estimate (hash table probe "
                         "sequence wrapped); the map was recovered, the "
                         "call's remaining frames were not integrated");
            return O3DMI_ERR_INTERNAL;
        }
        int64_t wanted = 0;
        if ((st = RecoverOverflow(g->block_hashmap, s, &wanted))) return st;
        O3DMI_HIP_CHECK(hipMemsetAsync(g->ring_counters, 0, sizeof(int) * 4, s));
        O3DMI_HIP_CHECK(hipMemsetAsync(g->front_tickets, 0, sizeof(int) * 32, s));
        const int64_t capacity = o3dmi_hash_capacity(g->block_hashmap);
        const int64_t target = wanted > capacity * 2 ? wanted : capacity * 2;
        if ((st = o3dmi_hash_reserve(g->block_hashmap, target,
                                     (o3dmi_stream_t)s)))
            return st;
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        int64_t size = 0;
        if ((st = o3dmi_hash_size(g->block_hashmap, (o3dmi_stream_t)s, &size)))
            return st;
        SetExactSize(g, size);
        g->stream_overflow = 0;
        pending.clear();
        issued = false;
        f = replay_from;
    }
    return O3DMI_OK;
}

// ---- sliced block touch (block-ownership sharding, sliced_path.h) -----------

static void FreeSliced(o3dmi_vbg* g) {
    o3dmi_vbg::Sliced& z = g->sliced;
    if (z.side) (void)hipStreamDestroy(z.side);
    for (int i = 0; i < 2; ++i) {
        if (z.ev_side[i]) (void)hipEventDestroy(z.ev_side[i]);
        if (z.ev_main[i]) (void)hipEventDestroy(z.ev_main[i]);
        (void)hipFree(z.send_seg[i]);
        (void)hipFree(z.gathered[i]);
        (void)hipFree(z.entries[i]);
        (void)hipFree(z.entries_count[i]);
    }
    if (z.ev_enter) (void)hipEventDestroy(z.ev_enter);
    if (z.table_slots) {
        FreeChunkTable(&z.send_table);
        FreeChunkTable(&z.recv_table);
    }
    (void)hipFree(z.frames_dev);
    (void)hipFree(z.iframes_dev);
    (void)hipFree(z.chunk_recs[0]);
    (void)hipFree(z.chunk_recs[1]);
    // (what the ranks agreed on is not a buffer: it outlives a re-allocation
    // that one rank may do on its own)
    const float agreed_trunc = z.agreed_trunc;
    const int agreed_fast_div = z.agreed_fast_div;
    const void* agreed_comm = z.agreed_comm;
    z = o3dmi_vbg::Sliced();
    z.agreed_trunc = agreed_trunc;
    z.agreed_fast_div = agreed_fast_div;
    z.agreed_comm = agreed_comm;
}

// Buffers for `world` ranks, `capacity` records per wire segment and chunk
// tables of `slots` slots. Growing frees and re-creates everything (the
// streams are drained first).
static int EnsureSliced(o3dmi_vbg* g, int world, int capacity, int slots,
                        hipStream_t s) {
    o3dmi_vbg::Sliced& z = g->sliced;
    if (z.side && z.world >= world && z.capacity >= capacity &&
        z.table_slots >= slots)
        return O3DMI_OK;
    if (z.side) O3DMI_HIP_CHECK(hipDeviceSynchronize());
    if (world < z.world) world = z.world;
    if (capacity < z.capacity) capacity = z.capacity;
    if (slots < z.table_slots) slots = z.table_slots;
    SliceFrame* keep_frames = z.frames_dev;
    IntegFrame* keep_iframes = z.iframes_dev;
    const int64_t keep_cap = z.frames_cap;
    const int64_t keep_chunks = z.chunks_done, keep_re = z.reapplied;
    const int keep_recv_slots = z.recv_slots;
    PixelRec* keep_recs[2] = {z.chunk_recs[0], z.chunk_recs[1]};
    const int64_t keep_px = z.chunk_recs_pixels;
    const int keep_recs_frames = z.chunk_recs_frames;
    z.frames_dev = nullptr;
    z.iframes_dev = nullptr;
    z.chunk_recs[0] = z.chunk_recs[1] = nullptr;
    FreeSliced(g);
    z.frames_dev = keep_frames;
    z.iframes_dev = keep_iframes;
    z.frames_cap = keep_cap;
    z.chunk_recs[0] = keep_recs[0];
    z.chunk_recs[1] = keep_recs[1];
    z.chunk_recs_pixels = keep_px;
    z.chunk_recs_frames = keep_recs_frames;
    z.chunks_done = keep_chunks;
    z.reapplied = keep_re;
    O3DMI_HIP_CHECK(hipStreamCreateWithFlags(&z.side, hipStreamNonBlocking));
    const int64_t seg = SliceSegmentBytes(capacity);
    for (int i = 0; i < 2; ++i) {
        O3DMI_HIP_CHECK(hipEventCreateWithFlags(&z.ev_side[i],
                                                hipEventDisableTiming));
        O3DMI_HIP_CHECK(hipEventCreateWithFlags(&z.ev_main[i],
                                                hipEventDisableTiming));
        O3DMI_HIP_CHECK(hipMalloc(&z.send_seg[i], (size_t)seg));
        O3DMI_HIP_CHECK(hipMalloc(&z.gathered[i], (size_t)seg * world));
        O3DMI_HIP_CHECK(hipMalloc(
                (void**)&z.entries[i],
                sizeof(ChunkEntry) *
                        (size_t)((keep_recv_slots > slots ? keep_recv_slots
                                                          : slots) / 2)));
        O3DMI_HIP_CHECK(hipMalloc((void**)&z.entries_count[i], sizeof(int)));
        O3DMI_HIP_CHECK(hipMemsetAsync(z.entries_count[i], 0, sizeof(int),
                                       z.side));
    }
    O3DMI_HIP_CHECK(hipEventCreateWithFlags(&z.ev_enter, hipEventDisableTiming));
    int st;
    const int recv_slots = keep_recv_slots > slots ? keep_recv_slots : slots;
    if ((st = AllocChunkTable(&z.send_table, slots, false, z.side))) return st;
    if ((st = AllocChunkTable(&z.recv_table, recv_slots, true, z.side)))
        return st;
    z.table_slots = slots;
    z.recv_slots = recv_slots;
    z.entries_cap = recv_slots / 2;
    z.capacity = capacity;
    z.world = world;
    O3DMI_HIP_CHECK(hipStreamSynchronize(z.side));
    return O3DMI_OK;
}

// Per-frame tables of the side-stream touch (pose as TouchParams keeps it) and
// of the chunk integrate launch (extrinsic as Camera::Make keeps it, images).
static int UploadSliceFrames(o3dmi_vbg* g, const StreamCommon& c,
                             const StreamFrame* frames, int n,
                             TouchParams* shared, int chunk_frames = 0) {
    o3dmi_vbg::Sliced& z = g->sliced;
    if (z.frames_cap < n) {
        if (z.frames_dev) O3DMI_HIP_CHECK(hipDeviceSynchronize());
        (void)hipFree(z.frames_dev);
        (void)hipFree(z.iframes_dev);
        z.frames_dev = nullptr;
        z.iframes_dev = nullptr;
        int64_t cap = 256;
        while (cap < n) cap <<= 1;
        O3DMI_HIP_CHECK(hipMalloc((void**)&z.frames_dev,
                                  sizeof(SliceFrame) * (size_t)cap));
        O3DMI_HIP_CHECK(hipMalloc((void**)&z.iframes_dev,
                                  sizeof(IntegFrame) * (size_t)cap));
        z.frames_cap = cap;
    }
    z.frames_host.resize((size_t)n);
    z.iframes_host.resize((size_t)n);
    for (int f = 0; f < n; ++f) {
        const TouchParams tp = MakeTouchParams(
                c.depth_intrinsic, frames[f].extrinsic, c.depth_rows,
                c.depth_cols, 4, (int)g->block_resolution, g->voxel_size,
                g->voxel_size * c.trunc, c.depth_scale, c.depth_max);
        std::memcpy(z.frames_host[(size_t)f].pose, tp.cam.e,
                    sizeof(z.frames_host[(size_t)f].pose));
        z.frames_host[(size_t)f].depth = (const uint16_t*)frames[f].depth;
        if (f == 0) *shared = tp;
        const Camera cf = Camera::Make(c.depth_intrinsic, frames[f].extrinsic,
                                       g->voxel_size);
        std::memcpy(z.iframes_host[(size_t)f].ext, cf.e,
                    sizeof(z.iframes_host[(size_t)f].ext));
        z.iframes_host[(size_t)f].depth = (const uint16_t*)frames[f].depth;
        z.iframes_host[(size_t)f].color =
                c.with_color ? (const uint8_t*)frames[f].color : nullptr;
        // records form: frame f = slot f % chunk of chunk set (f / chunk) & 1
        z.iframes_host[(size_t)f].recs =
                (chunk_frames > 0 && z.chunk_recs[0])
                        ? z.chunk_recs[(f / chunk_frames) & 1] +
                                  (size_t)(f % chunk_frames) *
                                          (size_t)(z.chunk_recs_pixels + 1)
                        : nullptr;
        z.iframes_host[(size_t)f].pad = nullptr;
    }
    // Synchronous copies: the tables are free (every launch of the previous
    // call that reads them has completed: the touch launches were waited for
    // through their chunk status, the integrate launches by the wait below)
    // and complete before the first launch of this call is issued.
    O3DMI_HIP_CHECK(hipMemcpy(z.frames_dev, z.frames_host.data(),
                              sizeof(SliceFrame) * (size_t)n,
                              hipMemcpyHostToDevice));
    O3DMI_HIP_CHECK(hipMemcpy(z.iframes_dev, z.iframes_host.data(),
                              sizeof(IntegFrame) * (size_t)n,
                              hipMemcpyHostToDevice));
    return O3DMI_OK;
}

// What a chunk's apply reported (host-mapped words 4..7 of stream_status).
struct ChunkStatus {
    int map_size, overflow, blocks;
};
static int WaitChunkStatus(o3dmi_vbg* g, int stamp, hipStream_t side,
                           ChunkStatus* out) {
    const volatile int* ts = g->stream_status + 4;
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    for (;;) {
        if (__atomic_load_n((const int*)&ts[3], __ATOMIC_ACQUIRE) == stamp) {
            out->map_size = ts[0];
            out->overflow = ts[1];
            out->blocks = ts[2];
            return O3DMI_OK;
        }
        if (synced) {
            SetLastError("sliced touch: the chunk's apply never reported");
            return O3DMI_ERR_INTERNAL;
        }
        if (std::chrono::steady_clock::now() - t0 >
            std::chrono::milliseconds(50)) {
            O3DMI_HIP_CHECK(hipStreamSynchronize(side));
            synced = true;
        }
    }
}

// How far a call of the sliced path got with its collectives: what an error
// exit needs to leave the other ranks in step (StreamIntegrateSliced).
struct SlicedProgress {
    int n_chunks = 0;
    int first_gathers = 0;  // chunks whose FIRST all-gather was issued (the
                            // redo of a chunk is collective by construction:
                            // every rank reads the same gathered headers)
    bool left_in_step = false;  // the error exit was agreed on by all ranks
                                // (o3dmi_comm::AgreeStatus): nobody waits in
                                // an all-gather, no abort segment to send
};

// The frames of one call through the sliced path. `gathered_in`: per chunk the
// `world` wire segments of all ranks (emulation / tests: the stand-in for the
// all-gather; the own segment is replaced by the one computed here), or null:
// all-gather over `comm`.
static int StreamIntegrateSlicedBody(o3dmi_vbg* g, const StreamCommon& c0,
                                     const StreamFrame* frames, int n,
                                     int group, hipStream_t s, o3dmi_comm* comm,
                                     const void* const* gathered_in,
                                     SlicedProgress* progress) {
    StreamCommon c = c0;
    if (group < 1) group = 1;
    if (group > kMaxGroup) group = kMaxGroup;
    const int world = g->owner_world, rank = g->owner_rank;
    O3DMI_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad ownership");
    O3DMI_REQUIRE(!comm || (comm->world == world && comm->rank == rank),
                  "sliced touch: the communicator's rank / world differ from "
                  "the grid's block ownership");
    c.ti = g->AttrIndex("tsdf");
    c.wi = g->AttrIndex("weight");
    c.ci = g->AttrIndex("color");
    c.with_color = c.ci >= 0 && (int64_t)c.color_rows * c.color_cols > 0 &&
                   n > 0 && frames[0].color != nullptr;
    // Everything rank-local in front of the first all-gather -- argument
    // checks that depend on this rank's grid, buffers, tables, the frame-table
    // upload -- runs as ONE stage whose status the ranks agree on (ADVICE r5:
    // a rank whose hipMalloc failed here used to return with no all-gather
    // issued, and its peers waited in chunk 0's for ever).
    int st;
    o3dmi_vbg::Sliced& z = g->sliced;
    bool raw_form = false;
    int chunk_frames = 0;
    TouchParams shared;
    auto setup = [&]() -> int {
    int st = GridDtype(g, &c.grid_dtype);
    if (st) return st;
    O3DMI_REQUIRE((c.depth_cols % 4) == 0 &&
                          (!c.with_color || (c.color_rows == c.depth_rows &&
                                             c.color_cols == c.depth_cols)),
                  "sliced touch needs depth and colour images of one size "
                  "(width % 4 == 0)");
    // status words, depth-division verdict (PrepTables) for this depth scale
    if ((st = EnsureStreamBuffers(g, c.depth_rows, c.depth_cols, 1024)))
        return st;
    if ((st = EnsurePrepTables(g, c.depth_intrinsic, c.color_intrinsic,
                               c.depth_rows, c.depth_cols, c.color_rows,
                               c.color_cols, c.depth_scale, s)))
        return st;
    O3DMI_REQUIRE(!c.with_color || g->prep_identity ||
                          !(std::getenv("O3DMI_SLICED_RAW") &&
                            std::getenv("O3DMI_SLICED_RAW")[0] == '1'),
                  "sliced touch, raw form, needs the same intrinsics for depth "
                  "and colour");
    // Sizes: a rank's band sees about 1 / world of a chunk's blocks plus the
    // band's rim; start from a generous guess and double on overflow (every
    // rank reads the same headers, so every rank doubles together).
    if ((st = EnsureSliced(g, world, z.capacity ? z.capacity : 4096,
                           z.table_slots ? z.table_slots
                                         : g->sliced_slots_wanted,
                           s)))
        return st;

    // Two forms of the chunk's integrate launch (same results):
    //   records  the per-pixel prepare pass of the chunk's frames runs on the
    //           side stream (every rank prepares every pixel: 4.3 MB of
    //           traffic per frame, replicated); the launch gathers one 8-byte
    //           record per voxel and frame, as the single-GPU stream does
    //   raw     no prepare pass and no record buffers (2 x 256 images of
    //           8 B / pixel); the launch gathers depth and colour separately
    //           and divides per voxel: 1 088 instead of 877 VALU instructions
    //           per 4 frames
    // Emulated shares (profiles/r4zb_emu_*, r4zg_emu_*; work list longest
    // first, compact lane map): records is ahead at 2 ranks (207 k frames/s
    // against 175 k), raw from 4 on, where the replicated prepare pass weighs
    // as much as a rank's share of the voxel work (328 k against 296 k at 4,
    // 527 k against 413 k at 8). O3DMI_SLICED_RAW=0 / 1 picks one (read per
    // call: tests switch it).
    const char* raw_env = std::getenv("O3DMI_SLICED_RAW");
    raw_form = raw_env ? raw_env[0] == '1'
                       : (world >= 4 && (!c.with_color || g->prep_identity));
    // (A software-pipelined form of the chunk launch -- next round's gathers
    // in flight during this round's arithmetic -- made no difference at 4 and
    // 8 emulated ranks, profiles/r4p, r4zc: dropped.)
    chunk_frames = kChunkGroups * group;
    const int64_t px = (int64_t)c.depth_rows * c.depth_cols;
    // The records form keeps two sets of prepared records, sized by what the
    // call needs (a chunk of this call's frames -- not the 256-frame maximum:
    // 0.6 GB per set at VGA); a call that takes the raw form gives them back.
    const int recs_frames = n < chunk_frames ? n : chunk_frames;
    if (raw_form && z.chunk_recs[0]) {
        O3DMI_HIP_CHECK(hipDeviceSynchronize());
        for (int i = 0; i < 2; ++i) {
            (void)hipFree(z.chunk_recs[i]);
            z.chunk_recs[i] = nullptr;
        }
        z.chunk_recs_pixels = 0;
        z.chunk_recs_frames = 0;
    }
    if (!raw_form &&
        (z.chunk_recs_pixels < px || z.chunk_recs_frames < recs_frames)) {
        O3DMI_HIP_CHECK(hipDeviceSynchronize());
        const int64_t new_px = z.chunk_recs_pixels > px ? z.chunk_recs_pixels
                                                        : px;
        const int new_fr = z.chunk_recs_frames > recs_frames
                                   ? z.chunk_recs_frames
                                   : recs_frames;
        for (int i = 0; i < 2; ++i) {
            (void)hipFree(z.chunk_recs[i]);
            z.chunk_recs[i] = nullptr;
            O3DMI_HIP_CHECK(hipMalloc((void**)&z.chunk_recs[i],
                                      sizeof(PixelRec) * (size_t)(new_px + 1) *
                                              (size_t)new_fr));
        }
        z.chunk_recs_pixels = new_px;
        z.chunk_recs_frames = new_fr;
    }
    // every launch of the previous call that reads the frame tables is over
    // once its last integrate launch is (the side stream waited for it)
    O3DMI_HIP_CHECK(hipStreamSynchronize(z.side));
    if ((st = UploadSliceFrames(g, c, frames, n, &shared,
                                raw_form ? 0 : chunk_frames)))
        return st;
    // the caller's images may still be in flight on its stream
    O3DMI_HIP_CHECK(hipEventRecord(z.ev_enter, s));
    O3DMI_HIP_CHECK(hipStreamWaitEvent(z.side, z.ev_enter, 0));
    return O3DMI_OK;
    };
    st = setup();
    if (comm && world > 1) {
        const int agreed = comm->AgreeStatus(st, s);
        if (agreed) {
            progress->left_in_step = true;
            return agreed;
        }
    } else if (st) {
        return st;
    }

    const int n_chunks = (n + chunk_frames - 1) / chunk_frames;
    std::vector<int> chunk_stamp((size_t)n_chunks, 0);
    progress->n_chunks = n_chunks;

    // The prepare pass of a chunk's frames (front roles without the block
    // touch), kMaxGroup frames per launch, into the chunk set's records.
    auto issue_prepare = [&](int ci) -> int {
        const int set = ci & 1;
        const int f0 = ci * chunk_frames;
        const int nc = n - f0 < chunk_frames ? n - f0 : chunk_frames;
        FrameFrontArgs fa[kMaxGroup];
        for (int b0 = 0; b0 < nc; b0 += kMaxGroup) {
            const int m = nc - b0 < kMaxGroup ? nc - b0 : kMaxGroup;
            for (int k = 0; k < m; ++k) {
                FrameFrontArgs& a = fa[k];
                a = FrameFrontArgs{};
                const StreamFrame& fr = frames[f0 + b0 + k];
                a.depth = (const uint16_t*)fr.depth;
                a.color = c.with_color ? (const uint8_t*)fr.color : nullptr;
                a.rows = c.depth_rows;
                a.cols = c.depth_cols;
                a.color_rows = c.color_rows;
                a.color_cols = c.color_cols;
                a.depth_intrinsic = c.depth_intrinsic;
                a.color_intrinsic = c.color_intrinsic ? c.color_intrinsic
                                                      : c.depth_intrinsic;
                a.extrinsic = fr.extrinsic;
                a.resolution = (int)g->block_resolution;
                a.voxel_size = g->voxel_size;
                a.sdf_trunc = g->voxel_size * c.trunc;
                a.depth_scale = c.depth_scale;
                a.depth_max = c.depth_max;
                a.stride = 4;
                a.group_stamp = 0;
                a.group_bit = k;
                a.touch_plane = 0;
                a.col_lut = g->prep_valid ? g->prep_col : nullptr;
                a.row_lut = g->prep_valid ? g->prep_row : nullptr;
                a.depth_div_short = g->prep_valid && g->prep_div_short;
                a.prep_identity = g->prep_valid && g->prep_identity;
                a.recs = z.chunk_recs[set] +
                         (size_t)(b0 + k) * (size_t)(z.chunk_recs_pixels + 1);
                a.prepare_only = true;
            }
            int st2 = LaunchFrameStep(g->block_hashmap, fa, m, nullptr, z.side);
            if (st2) return st2;
        }
        return O3DMI_OK;
    };

    auto issue_side = [&](int ci, bool touch) -> int {
        const int set = ci & 1;
        const int f0 = ci * chunk_frames;
        const int nc = n - f0 < chunk_frames ? n - f0 : chunk_frames;
        int st2;
        if (touch) {
            // the set's work list was read by chunk ci - 2's launch
            if (ci >= 2)
                O3DMI_HIP_CHECK(hipStreamWaitEvent(z.side, z.ev_main[set], 0));
            if ((st2 = LaunchTouchSlice(shared, z.frames_dev, f0, nc, rank,
                                        world, z.send_table, z.side)))
                return st2;
            if (!raw_form && (st2 = issue_prepare(ci))) return st2;
            if ((st2 = LaunchPackSlice(z.send_table, z.send_seg[set],
                                       z.capacity, z.side)))
                return st2;
            const int64_t seg = SliceSegmentBytes(z.capacity);
            if (comm) {
                if ((st2 = comm->Allgather(z.send_seg[set], z.gathered[set],
                                           seg, z.side)))
                    return st2;
                if (ci >= progress->first_gathers)
                    progress->first_gathers = ci + 1;
            } else {
                if (gathered_in && world > 1)
                    O3DMI_HIP_CHECK(hipMemcpyAsync(
                            z.gathered[set], gathered_in[ci],
                            (size_t)seg * world, hipMemcpyDeviceToDevice,
                            z.side));
                O3DMI_HIP_CHECK(hipMemcpyAsync(
                        (char*)z.gathered[set] + (size_t)seg * rank,
                        z.send_seg[set], (size_t)seg, hipMemcpyDeviceToDevice,
                        z.side));
            }
        }
        g->frame_stamp += 1;
        chunk_stamp[(size_t)ci] = g->frame_stamp;
        if ((st2 = LaunchApplySlice(g->block_hashmap, z.gathered[set], world,
                                    z.capacity, z.recv_table, g->frame_stamp,
                                    z.side)))
            return st2;
        if ((st2 = LaunchBuildChunk(g->block_hashmap, z.recv_table,
                                    z.entries[set], z.entries_cap,
                                    z.entries_count[set],
                                    (int*)g->stream_status + 4, g->frame_stamp,
                                    z.side)))
            return st2;
        O3DMI_HIP_CHECK(hipEventRecord(z.ev_side[set], z.side));
        return O3DMI_OK;
    };

    if ((st = issue_side(0, true))) return st;
    for (int ci = 0; ci < n_chunks; ++ci) {
        const int set = ci & 1;
        const int f0 = ci * chunk_frames;
        const int nc = n - f0 < chunk_frames ? n - f0 : chunk_frames;
        // What the chunk's apply found: normally long there (it was issued a
        // chunk ago).
        ChunkStatus cs = {};
        for (int attempt = 0;; ++attempt) {
            if ((st = WaitChunkStatus(g, chunk_stamp[(size_t)ci], z.side, &cs)))
                return st;
            if (cs.overflow == 0) break;
            if (cs.overflow == -3) {
                // a rank left the call with an error and said so in its wire
                // segment: every rank leaves at this chunk, in step (none has
                // a collective outstanding)
                O3DMI_HIP_CHECK(hipDeviceSynchronize());
                SetLastError("sliced touch: another rank left the call with an "
                             "error");
                return O3DMI_ERR_PEER;
            }
            O3DMI_REQUIRE(attempt < 24, "sliced touch: cannot make room");
            O3DMI_HIP_CHECK(hipDeviceSynchronize());
            z.reapplied += 1;
            g->stream_overflow = 0;  // (PollStreamStatus reads the same words)
            if (cs.overflow > 0) {
                // the block hash ran out of buffer indices: the chunk was
                // dropped; reserve, apply the same records again
                int64_t wanted = 0;
                if ((st = RecoverOverflow(g->block_hashmap, s, &wanted)))
                    return st;
                const int64_t cap = o3dmi_hash_capacity(g->block_hashmap);
                if ((st = o3dmi_hash_reserve(g->block_hashmap,
                                             wanted > 2 * cap ? wanted : 2 * cap,
                                             (o3dmi_stream_t)s)))
                    return st;
                O3DMI_HIP_CHECK(hipStreamSynchronize(s));
                if ((st = issue_side(ci, false))) return st;
            } else if (cs.overflow == -2) {
                // THIS rank's receiver table was too small for the blocks it
                // owns in the chunk: a local matter (the other ranks are not
                // waiting for anything) -- a larger table and work lists, the
                // same gathered records applied again
                FreeChunkTable(&z.recv_table);
                for (int i = 0; i < 2; ++i) {
                    (void)hipFree(z.entries[i]);
                    z.entries[i] = nullptr;
                }
                z.recv_slots *= 2;
                if ((st = AllocChunkTable(&z.recv_table, z.recv_slots, true,
                                          z.side)))
                    return st;
                for (int i = 0; i < 2; ++i)
                    O3DMI_HIP_CHECK(hipMalloc(
                            (void**)&z.entries[i],
                            sizeof(ChunkEntry) * (size_t)(z.recv_slots / 2)));
                z.entries_cap = z.recv_slots / 2;
                O3DMI_HIP_CHECK(hipStreamSynchronize(z.side));
                if ((st = issue_side(ci, false))) return st;
            } else {
                // a SENDER's wire segment or table was too small (the flags
                // travel in the all-gathered headers: every rank takes this
                // branch for the same chunk): double both, touch again
                if (gathered_in) {
                    SetLastError("sliced touch: the given wire segments are "
                                 "too small for this stream "
                                 "(o3dmi_vbg_set_slice_capacity)");
                    return O3DMI_ERR_CAPACITY;
                }
                // (every rank is here for the same chunk with nothing in
                // flight: a rank that cannot grow its buffers says so before
                // the others enter the redone all-gather -- its own send
                // segment is gone by then, no abort segment could be sent)
                st = EnsureSliced(g, world, z.capacity * 2, z.table_slots * 2,
                                  s);
                if (comm && world > 1) {
                    const int agreed = comm->AgreeStatus(st, s);
                    if (agreed) {
                        progress->left_in_step = true;
                        return agreed;
                    }
                } else if (st) {
                    return st;
                }
                if ((st = issue_side(ci, true))) return st;
            }
        }
        if (ci + 1 < n_chunks)
            if ((st = issue_side(ci + 1, true))) return st;
        // the chunk's integrate launch
        O3DMI_HIP_CHECK(hipStreamWaitEvent(s, z.ev_side[set], 0));
        g->frame_stamp += 1;
        const bool prof = g->profiling && g->prof_frames < g->prof_max &&
                          g->prof_stride > 0 &&
                          (g->prof_seen++ % g->prof_stride) == 0;
        hipEvent_t* pe = prof ? &g->prof_events[(size_t)g->prof_frames * 2]
                              : nullptr;
        ChunkIntegrateArgs ia = {};
        ia.n_frames = nc;
        ia.frames = z.iframes_dev + f0;
        ia.entries = z.entries[set];
        ia.count = z.entries_count[set];
        ia.entries_cap = z.entries_cap;
        ia.grid_hint = cs.blocks;
        ia.rows = c.depth_rows;
        ia.cols = c.depth_cols;
        ia.with_color = c.with_color;
        ia.tsdf = (float*)o3dmi_hash_value_buffer(g->block_hashmap, c.ti);
        ia.weight = o3dmi_hash_value_buffer(g->block_hashmap, c.wi);
        ia.color = c.with_color
                           ? o3dmi_hash_value_buffer(g->block_hashmap, c.ci)
                           : nullptr;
        ia.grid_dtype = c.grid_dtype;
        ia.depth_intrinsic = c.depth_intrinsic;
        ia.resolution = (int)g->block_resolution;
        ia.voxel_size = g->voxel_size;
        ia.sdf_trunc = g->voxel_size * c.trunc;
        ia.depth_max = c.depth_max;
        ia.depth_scale = c.depth_scale;
        ia.depth_div_short = g->prep_div_short;
        ia.raw = raw_form;
        ia.size_host = (int*)g->stream_status;
        ia.status_stamp = g->frame_stamp;
        ia.prof_count = prof ? g->prof_counts + g->prof_max + g->prof_frames
                             : nullptr;
        ia.prof_frame_blocks = prof ? g->prof_counts + g->prof_frames : nullptr;
        ia.prof_map_size =
                prof ? g->prof_counts + 2 * g->prof_max + g->prof_frames
                     : nullptr;
        if (pe) O3DMI_HIP_CHECK(hipEventRecord(pe[0], s));
        if ((st = LaunchChunkIntegrate(g->block_hashmap, ia, s))) return st;
        if (pe) {
            O3DMI_HIP_CHECK(hipEventRecord(pe[1], s));
            g->prof_launch_frames += nc;
            g->prof_frames += 1;
        }
        O3DMI_HIP_CHECK(hipEventRecord(z.ev_main[set], s));
        if ((st = PollStreamStatus(g))) return st;  // deferred error flags
        z.chunks_done += 1;
    }
    // the next call's side-stream work must not pass this call's launches
    // (the frame tables, the work lists)
    O3DMI_HIP_CHECK(hipStreamWaitEvent(z.side, z.ev_main[(n_chunks - 1) & 1], 0));
    g->known_valid = false;  // the other paths take the size from the map
    g->stream_overflow = 0;
    g->last_path = 0;
    return O3DMI_OK;
}

// o3dmi_vbg_integrate_frames on a grid with block ownership and a communicator
// is a COLLECTIVE call (one all-gather per chunk of frames). A rank that has to
// leave it with an error of its own (no room after 24 attempts, an allocation
// that failed, a deferred device-side flag) would leave the others waiting in
// the next all-gather: it delivers that one all-gather with an empty segment
// flagged kSliceFlagAbort first, and every rank returns O3DMI_ERR_PEER at the
// same chunk. The map is left usable (a dropped chunk's slots are recovered,
// the stream's overflow state cleared). Best effort after a HIP runtime error:
// the abort segment is then issued on a device that may not execute it.
static int StreamIntegrateSliced(o3dmi_vbg* g, const StreamCommon& c0,
                                 const StreamFrame* frames, int n, int group,
                                 hipStream_t s, o3dmi_comm* comm,
                                 const void* const* gathered_in) {
    SlicedProgress pr;
    const int st = StreamIntegrateSlicedBody(g, c0, frames, n, group, s, comm,
                                             gathered_in, &pr);
    if (st == O3DMI_OK) return st;
    const std::string why = o3dmi_last_error();
    o3dmi_vbg::Sliced& z = g->sliced;
    if (st != O3DMI_ERR_PEER && !pr.left_in_step && comm && z.side &&
        pr.n_chunks > 0 && pr.first_gathers > 0 &&
        pr.first_gathers < pr.n_chunks) {
        // the all-gather the other ranks will wait in next
        const int set = pr.first_gathers & 1;
        SliceHeader hd = {};
        hd.count = 0;
        hd.flags = kSliceFlagAbort;
        hd.capacity = z.capacity;
        if (hipMemcpyAsync(z.send_seg[set], &hd, sizeof(hd),
                           hipMemcpyHostToDevice, z.side) == hipSuccess &&
            hipStreamSynchronize(z.side) == hipSuccess)
            (void)comm->Allgather(z.send_seg[set], z.gathered[set],
                                  SliceSegmentBytes(z.capacity), z.side);
    }
    (void)hipDeviceSynchronize();
    // leave the map usable: a chunk dropped for lack of buffer indices left
    // keys without a block behind (CheckDeferred would refuse every later
    // size / export call)
    int64_t wanted = 0;
    (void)RecoverOverflow(g->block_hashmap, s, &wanted);
    (void)hipStreamSynchronize(s);
    g->stream_overflow = 0;
    g->known_valid = false;
    g->last_path = 0;
    SetLastError(why.c_str());
    return st;
}

extern "C" int o3dmi_internal_estimate_range(
        const int32_t* block_keys_dev, int key_stride, int64_t max_blocks,
        const int32_t* n_blocks_dev, float* range_minmax_map_dev,
        int map_is_clean, const double* intrinsic, const double* extrinsic,
        int h, int w, int down_factor, int64_t block_resolution,
        float voxel_size, float depth_min, float depth_max,
        o3dmi_stream_t stream);
extern "C" int o3dmi_internal_raycast_reset_range(void);
extern "C" int o3dmi_internal_raycast_forget(void);
extern "C" int o3dmi_internal_raycast_tile_order(unsigned long long* cost,
                                                 int* order, int n_tiles,
                                                 unsigned want_seq,
                                                 unsigned seq);

extern "C++" {
namespace o3dmi {
int PreloadBlockHash();
int PreloadIcp();
int PreloadNns();
int PreloadPointcloud();
int PreloadRaycast();
int PreloadStream();
int PreloadTouch();
}  // namespace o3dmi
}  // extern "C++"

// Extension: everything a first frame would otherwise pay for besides its own
// buffers. HIP loads a translation unit's code object at the first launch of
// one of its kernels -- 1.5-2.8 ms each for the large ones (ICP search,
// VoxelDownSample, the search index): of the 8.7 ms a first tracked frame
// took, most was that (profiles/r6k_first_frame.txt). Loads them now; safe to
// call more than once and from any thread.
extern "C" int o3dmi_preload(void) {
    int bad = 0;
    bad += o3dmi::PreloadBlockHash();
    bad += o3dmi::PreloadTouch();
    bad += o3dmi::PreloadStream();
    bad += o3dmi::PreloadRaycast();
    bad += o3dmi::PreloadPointcloud();
    bad += o3dmi::PreloadNns();
    bad += o3dmi::PreloadIcp();
    {
        hipFuncAttributes attr;
        bad += hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(
                                                   &ExportListKeysKernel)) ==
                               hipSuccess
                       ? 0
                       : 1;
    }
    if (bad) {
        SetLastError("o3dmi_preload: a code object could not be loaded");
        return O3DMI_ERR_HIP;
    }
    return O3DMI_OK;
}

static StreamCommon MakeCommon(int depth_rows, int depth_cols, int color_rows,
                               int color_cols, const double* depth_intrinsic,
                               const double* color_intrinsic, float depth_scale,
                               float depth_max, float trunc) {
    StreamCommon c = {};
    c.depth_rows = depth_rows;
    c.depth_cols = depth_cols;
    c.color_rows = color_rows;
    c.color_cols = color_cols;
    c.depth_intrinsic = depth_intrinsic;
    c.color_intrinsic = color_intrinsic;
    c.depth_scale = depth_scale;
    c.depth_max = depth_max;
    c.trunc = trunc;
    return c;
}

int o3dmi_vbg_integrate_frame(o3dmi_vbg_t* g, const void* depth_dev,
                              int depth_rows, int depth_cols,
                              const void* color_dev, int color_rows,
                              int color_cols, int input_dtype,
                              const double* depth_intrinsic,
                              const double* color_intrinsic,
                              const double* extrinsic, float depth_scale,
                              float depth_max, float trunc_voxel_multiplier,
                              o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && depth_dev && depth_intrinsic && extrinsic,
                  "null argument");
    if (!StreamPathApplies(g, input_dtype))
        return IntegrateFrameGeneric(g, depth_dev, depth_rows, depth_cols,
                                     color_dev, color_rows, color_cols,
                                     input_dtype, depth_intrinsic,
                                     color_intrinsic, extrinsic, depth_scale,
                                     depth_max, trunc_voxel_multiplier, stream);
    StreamCommon c = MakeCommon(depth_rows, depth_cols, color_rows, color_cols,
                                depth_intrinsic, color_intrinsic, depth_scale,
                                depth_max, trunc_voxel_multiplier);
    StreamFrame fr = {depth_dev, color_dev, extrinsic};
    return StreamIntegrate(g, c, &fr, 1, 1, (hipStream_t)stream);
}

int o3dmi_vbg_integrate_frames(o3dmi_vbg_t* g, int n_frames,
                               const void* const* depth_devs, int depth_rows,
                               int depth_cols, const void* const* color_devs,
                               int color_rows, int color_cols, int input_dtype,
                               const double* depth_intrinsic,
                               const double* color_intrinsic,
                               const double* extrinsics, float depth_scale,
                               float depth_max, float trunc_voxel_multiplier,
                               int frames_per_launch, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && depth_devs && depth_intrinsic && extrinsics &&
                          n_frames >= 0,
                  "null argument");
    if (!StreamPathApplies(g, input_dtype)) {
        for (int f = 0; f < n_frames; ++f) {
            int st = IntegrateFrameGeneric(
                    g, depth_devs[f], depth_rows, depth_cols,
                    color_devs ? color_devs[f] : nullptr, color_rows,
                    color_cols, input_dtype, depth_intrinsic, color_intrinsic,
                    extrinsics + 16 * (size_t)f, depth_scale, depth_max,
                    trunc_voxel_multiplier, stream);
            if (st) return st;
        }
        return O3DMI_OK;
    }
    StreamCommon c = MakeCommon(depth_rows, depth_cols, color_rows, color_cols,
                                depth_intrinsic, color_intrinsic, depth_scale,
                                depth_max, trunc_voxel_multiplier);
    std::vector<StreamFrame> frames((size_t)n_frames);
    for (int f = 0; f < n_frames; ++f) {
        frames[(size_t)f].depth = depth_devs[f];
        frames[(size_t)f].color = color_devs ? color_devs[f] : nullptr;
        frames[(size_t)f].extrinsic = extrinsics + 16 * (size_t)f;
    }
    // Block ownership with a communicator on this thread: the touch is split
    // over the ranks and the candidate keys all-gathered (sliced_path.h)
    // instead of every rank touching every ray -- when the images allow the
    // raw-image integrate role (else the replicated touch below: same grids).
    if (g->owner_world > 1 && n_frames > 1) {
        o3dmi_comm* comm = ThreadComm();
        static const bool no_slice = std::getenv("O3DMI_NO_SLICED_TOUCH") != nullptr;
        if (comm && !no_slice && comm->world == g->owner_world &&
            comm->rank == g->owner_rank && color_rows == depth_rows &&
            color_cols == depth_cols && (depth_cols % 4) == 0 &&
            (!color_intrinsic ||
             std::memcmp(color_intrinsic, depth_intrinsic,
                         sizeof(double) * 9) == 0 || !color_devs)) {
            // The chunk launch needs the proven short division forms. The
            // proof is a property of the part and the truncation distance,
            // but whether THIS rank has it is rank-local (a failed allocation
            // of the proof's scratch, O3DMI_EXACT_DIV in one rank's
            // environment): the ranks agree once per truncation distance and
            // communicator -- a rank that silently took the
            // non-collective path below would leave its peers in chunk 0's
            // all-gather.
            o3dmi_vbg::Sliced& z = g->sliced;
            const float trunc = g->voxel_size * trunc_voxel_multiplier;
            if (z.agreed_fast_div == 0 || z.agreed_trunc != trunc ||
                z.agreed_comm != (const void*)comm) {
                const int mine = PrefetchFastDivision(trunc, true) == 2
                                         ? O3DMI_OK
                                         : O3DMI_ERR_INTERNAL;
                const int all = comm->AgreeStatus(mine, (hipStream_t)stream);
                z.agreed_fast_div = all == O3DMI_OK ? 1 : -1;
                z.agreed_trunc = trunc;
                z.agreed_comm = (const void*)comm;
            }
            if (z.agreed_fast_div == 1)
                return StreamIntegrateSliced(
                        g, c, frames.data(), n_frames,
                        frames_per_launch <= 0 ? kDefaultGroup
                                               : frames_per_launch,
                        (hipStream_t)stream, comm, nullptr);
        }
    }
    return StreamIntegrate(g, c, frames.data(), n_frames,
                           frames_per_launch <= 0 ? kDefaultGroup
                                                  : frames_per_launch,
                           (hipStream_t)stream);
}

int o3dmi_vbg_set_slice_capacity(o3dmi_vbg_t* g, int records_per_group,
                                 int table_slots) {
    O3DMI_REQUIRE(g != nullptr, "grid is null");
    O3DMI_REQUIRE(records_per_group >= 16 && table_slots >= 64 &&
                          (table_slots & (table_slots - 1)) == 0,
                  "slice capacity: records >= 16, table slots a power of two "
                  ">= 64");
    o3dmi_vbg::Sliced& z = g->sliced;
    if (z.side) {
        O3DMI_HIP_CHECK(hipDeviceSynchronize());
        FreeSliced(g);
    }
    z.capacity = records_per_group;
    z.table_slots = 0;
    g->sliced_slots_wanted = table_slots;
    return O3DMI_OK;
}

int64_t o3dmi_vbg_slice_segment_bytes(const o3dmi_vbg_t* g) {
    if (!g) return 0;
    return SliceSegmentBytes(g->sliced.capacity ? g->sliced.capacity : 4096);
}

int o3dmi_vbg_slice_chunk_frames(int frames_per_launch) {
    int group = frames_per_launch <= 0 ? kDefaultGroup : frames_per_launch;
    if (group > kMaxGroup) group = kMaxGroup;
    return kChunkGroups * group;
}

int o3dmi_vbg_touch_slice(o3dmi_vbg_t* g, int n_frames,
                          const void* const* depth_devs, int depth_rows,
                          int depth_cols, const double* depth_intrinsic,
                          const double* extrinsics, float depth_scale,
                          float depth_max, float trunc_voxel_multiplier,
                          int frames_per_launch, int slice_rank,
                          int slice_world, void* segment_out_dev,
                          o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && depth_devs && depth_intrinsic && extrinsics &&
                          segment_out_dev,
                  "null argument");
    int group = frames_per_launch <= 0 ? kDefaultGroup : frames_per_launch;
    if (group > kMaxGroup) group = kMaxGroup;
    O3DMI_REQUIRE(n_frames >= 1 && n_frames <= kChunkGroups * group,
                  "touch slice: at most one chunk of frames per call");
    hipStream_t s = (hipStream_t)stream;
    o3dmi_vbg::Sliced& z = g->sliced;
    int st = EnsureSliced(g, slice_world > z.world ? slice_world : z.world,
                          z.capacity ? z.capacity : 4096,
                          z.table_slots ? z.table_slots
                                        : g->sliced_slots_wanted,
                          s);
    if (st) return st;
    StreamCommon c = {};
    c.depth_rows = depth_rows;
    c.depth_cols = depth_cols;
    c.depth_intrinsic = depth_intrinsic;
    c.depth_scale = depth_scale;
    c.depth_max = depth_max;
    c.trunc = trunc_voxel_multiplier;
    std::vector<StreamFrame> frames((size_t)n_frames);
    for (int f = 0; f < n_frames; ++f) {
        frames[(size_t)f].depth = depth_devs[f];
        frames[(size_t)f].color = nullptr;
        frames[(size_t)f].extrinsic = extrinsics + 16 * (size_t)f;
    }
    // everything on the side stream, behind the caller's stream and in front
    // of whatever the caller does next on it
    O3DMI_HIP_CHECK(hipEventRecord(z.ev_enter, s));
    O3DMI_HIP_CHECK(hipStreamWaitEvent(z.side, z.ev_enter, 0));
    TouchParams shared;
    O3DMI_HIP_CHECK(hipStreamSynchronize(z.side));
    if ((st = UploadSliceFrames(g, c, frames.data(), n_frames, &shared)))
        return st;
    if ((st = LaunchTouchSlice(shared, z.frames_dev, 0, n_frames, slice_rank,
                               slice_world, z.send_table, z.side)))
        return st;
    if ((st = LaunchPackSlice(z.send_table, segment_out_dev, z.capacity,
                              z.side)))
        return st;
    O3DMI_HIP_CHECK(hipEventRecord(z.ev_enter, z.side));
    O3DMI_HIP_CHECK(hipStreamWaitEvent(s, z.ev_enter, 0));
    return O3DMI_OK;
}

int o3dmi_vbg_integrate_frames_sliced(
        o3dmi_vbg_t* g, int n_frames, const void* const* depth_devs,
        int depth_rows, int depth_cols, const void* const* color_devs,
        int color_rows, int color_cols, int input_dtype,
        const double* depth_intrinsic, const double* color_intrinsic,
        const double* extrinsics, float depth_scale, float depth_max,
        float trunc_voxel_multiplier, int frames_per_launch,
        const void* const* gathered_devs, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && depth_devs && depth_intrinsic && extrinsics &&
                          n_frames >= 0,
                  "null argument");
    O3DMI_REQUIRE(StreamPathApplies(g, input_dtype),
                  "sliced touch: uint16 depth, float32 tsdf and a block "
                  "resolution divisible by 4 are required");
    o3dmi_comm* comm = gathered_devs ? nullptr : ThreadComm();
    O3DMI_REQUIRE(gathered_devs || comm || g->owner_world == 1,
                  "sliced touch: neither wire segments nor a communicator "
                  "(o3dmi_set_comm)");
    if (n_frames == 0) return O3DMI_OK;
    StreamCommon c = MakeCommon(depth_rows, depth_cols, color_rows, color_cols,
                                depth_intrinsic, color_intrinsic, depth_scale,
                                depth_max, trunc_voxel_multiplier);
    std::vector<StreamFrame> frames((size_t)n_frames);
    for (int f = 0; f < n_frames; ++f) {
        frames[(size_t)f].depth = depth_devs[f];
        frames[(size_t)f].color = color_devs ? color_devs[f] : nullptr;
        frames[(size_t)f].extrinsic = extrinsics + 16 * (size_t)f;
    }
    return StreamIntegrateSliced(g, c, frames.data(), n_frames,
                                 frames_per_launch <= 0 ? kDefaultGroup
                                                        : frames_per_launch,
                                 (hipStream_t)stream, comm, gathered_devs);
}

int o3dmi_vbg_sliced_stats(const o3dmi_vbg_t* g, int64_t* chunks,
                           int64_t* reapplied, int* capacity,
                           int* table_slots) {
    O3DMI_REQUIRE(g != nullptr, "grid is null");
    if (chunks) *chunks = g->sliced.chunks_done;
    if (reapplied) *reapplied = g->sliced.reapplied;
    if (capacity) *capacity = g->sliced.capacity;
    if (table_slots) *table_slots = g->sliced.table_slots;
    return O3DMI_OK;
}

int o3dmi_vbg_ray_cast(o3dmi_vbg_t* g, const int32_t* block_coords_dev,
                       int64_t m, const double* intrinsic,
                       const double* extrinsic, int width, int height,
                       float* range_map_dev, float* out_depth,
                       float* out_vertex, float* out_color, float* out_normal,
                       int64_t* out_index, uint8_t* out_mask, float* out_ratio,
                       float* out_ratio_dx, float* out_ratio_dy,
                       float* out_ratio_dz, float depth_scale, float depth_min,
                       float depth_max, float weight_threshold,
                       float trunc_voxel_multiplier, int range_map_down_factor,
                       o3dmi_stream_t stream) {
    return o3dmi_vbg_ray_cast_dev(
            g, block_coords_dev, m, nullptr, intrinsic, extrinsic, width,
            height, range_map_dev, out_depth, out_vertex, out_color, out_normal,
            out_index, out_mask, out_ratio, out_ratio_dx, out_ratio_dy,
            out_ratio_dz, depth_scale, depth_min, depth_max, weight_threshold,
            trunc_voxel_multiplier, range_map_down_factor, stream);
}

// Internal (slam_model.cpp): copies the block keys the most recent
// o3dmi_vbg_integrate_frame touched (its GetUniqueBlockCoordinates result) to
// out_keys_dev {capacity,3} and their number to out_count_dev, on the stream,
// without a host round trip. Must be issued right behind that call.
int o3dmi_vbg_export_last_frame_blocks(o3dmi_vbg_t* g, int32_t* out_keys_dev,
                                       int64_t out_capacity,
                                       int32_t* out_count_dev,
                                       o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && out_keys_dev && out_count_dev, "null argument");
    O3DMI_REQUIRE(g->last_path != 0, "no frame has been integrated");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(64), block(kBlock);
    if (g->last_path == 1) {
        const int64_t cap = out_capacity < g->lists_capacity ? out_capacity
                                                            : g->lists_capacity;
        hipLaunchKernelGGL(ExportListKeysKernel, grid, block, 0, s,
                           g->lists[g->last_seq & 1],
                           g->ring_counters + (g->last_seq & 3), cap,
                           out_keys_dev, out_count_dev);
    } else {
        const int64_t cap = out_capacity < g->frame_indices_capacity
                                    ? out_capacity
                                    : g->frame_indices_capacity;
        hipLaunchKernelGGL(ExportIndexKeysKernel, grid, block, 0, s,
                           g->frame_indices, g->frame_count, cap,
                           o3dmi_hash_key_buffer(g->block_hashmap),
                           out_keys_dev, out_count_dev);
    }
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_vbg_last_frame_block_coordinates(o3dmi_vbg_t* g,
                                           int32_t* out_coords_dev,
                                           int64_t capacity,
                                           int32_t* out_count_dev,
                                           o3dmi_stream_t stream) {
    O3DMI_REQUIRE(capacity > 0, "capacity must be positive");
    return o3dmi_vbg_export_last_frame_blocks(g, out_coords_dev, capacity,
                                              out_count_dev, stream);
}

int o3dmi_vbg_ray_cast_dev(o3dmi_vbg_t* g, const int32_t* block_coords_dev,
                           int64_t max_m, const int32_t* m_dev,
                           const double* intrinsic, const double* extrinsic,
                           int width, int height, float* range_map_dev,
                           float* out_depth, float* out_vertex,
                           float* out_color, float* out_normal,
                           int64_t* out_index, uint8_t* out_mask,
                           float* out_ratio, float* out_ratio_dx,
                           float* out_ratio_dy, float* out_ratio_dz,
                           float depth_scale, float depth_min, float depth_max,
                           float weight_threshold,
                           float trunc_voxel_multiplier,
                           int range_map_down_factor, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && intrinsic && extrinsic, "null argument");
    int ti = g->AttrIndex("tsdf"), wi = g->AttrIndex("weight"),
        ci = g->AttrIndex("color");
    if (ti < 0 || wi < 0) {
        SetLastError(
                "TSDF and/or weight not allocated in blocks, please implement "
                "customized integration.");
        return O3DMI_ERR_INVALID_ARG;
    }
    int grid_dtype;
    int st = GridDtype(g, &grid_dtype);
    if (st) return st;
    O3DMI_REQUIRE(range_map_down_factor > 0 && height >= range_map_down_factor &&
                          width >= range_map_down_factor,
                  "bad image size / down factor");
    // block_coords_dev == NULL: the blocks the last frame-stream integration
    // touched, read straight from the grid's own list (no export launch, no
    // caller-side copy) -- what o3dmi_vbg_last_frame_block_coordinates would
    // hand over.
    int key_stride = 3;
    if (!block_coords_dev) {
        O3DMI_REQUIRE(g->last_path == 1 && g->lists[0] != nullptr,
                      "ray cast without block coordinates: no frame-stream "
                      "integration to take them from "
                      "(o3dmi_vbg_integrate_frame first, or pass the "
                      "coordinates)");
        block_coords_dev = (const int32_t*)g->lists[g->last_seq & 1] + 1;
        m_dev = g->ring_counters + (g->last_seq & 3);
        max_m = g->lists_capacity;
        key_stride = 4;
    }
    // range_map_dev == NULL: the range map is the grid's own scratch (as in
    // the reference, where RayCast allocates it), and the ray cast that
    // consumes it leaves it clean for the next call: no clearing launch per
    // frame.
    int map_is_clean = 0;
    // (what the calls below leave for the range-estimate and the ray-cast
    // launch of this thread does not outlive this call, whichever way it ends)
    struct ForgetSideChannels {
        ~ForgetSideChannels() { (void)o3dmi_internal_raycast_forget(); }
    } forget_side_channels;
    if (!range_map_dev) {
        const int64_t cells = (int64_t)(height / range_map_down_factor) *
                              (width / range_map_down_factor);
        if (g->own_range_cells != cells) {
            if (g->own_range) {
                O3DMI_HIP_CHECK(hipDeviceSynchronize());
                (void)hipFree(g->own_range);
                g->own_range = nullptr;
            }
            // (+ one cell: the clean state's {lo, hi} for the resetting cast)
            O3DMI_HIP_CHECK(hipMalloc((void**)&g->own_range,
                                      sizeof(float) * 2 * (size_t)(cells + 1)));
            g->own_range_cells = cells;
            g->own_range_clean = false;
        }
        range_map_dev = g->own_range;
        map_is_clean = g->own_range_clean && g->own_range_lo == depth_max &&
                       g->own_range_hi == depth_min;
        // the cast below re-cleans what it reads (8-pixel cells only)
        g->own_range_clean = range_map_down_factor == 8 && (height % 8) == 0 &&
                             (width % 8) == 0;
        g->own_range_lo = depth_max;
        g->own_range_hi = depth_min;
        if (g->own_range_clean) {
            if (!map_is_clean) {
                const float lohi[2] = {depth_max, depth_min};
                O3DMI_HIP_CHECK(hipMemcpyAsync(
                        g->own_range + 2 * cells, lohi, sizeof(lohi),
                        hipMemcpyHostToDevice, (hipStream_t)stream));
            }
            (void)o3dmi_internal_raycast_reset_range();
        }
        // An image of more tiles than the chip holds workgroups (1280 x 720:
        // 3600 against 1280) is rendered longest tile first, by the last
        // cast's measured tile times (vbg_raycast.hip TileOrder).
        const int64_t n_tiles =
                (int64_t)((width + 31) / 32) * ((height + 7) / 8);
        if (n_tiles > kCUs * 5 && n_tiles <= kCUs * 16 && max_m > 0) {
            if (g->rc_tiles != n_tiles) {
                if (g->rc_cost) {
                    O3DMI_HIP_CHECK(hipDeviceSynchronize());
                    (void)hipFree(g->rc_cost);
                    (void)hipFree(g->rc_order);
                    g->rc_cost = nullptr;
                    g->rc_order = nullptr;
                    g->rc_tiles = 0;
                }
                O3DMI_HIP_CHECK(hipMalloc((void**)&g->rc_cost,
                                          sizeof(unsigned long long) *
                                                  (size_t)n_tiles));
                O3DMI_HIP_CHECK(hipMalloc((void**)&g->rc_order,
                                          sizeof(int) * (size_t)n_tiles));
                O3DMI_HIP_CHECK(hipMemsetAsync(
                        g->rc_cost, 0,
                        sizeof(unsigned long long) * (size_t)n_tiles,
                        (hipStream_t)stream));
                g->rc_tiles = n_tiles;
                g->rc_seq = 0;
            }
            const unsigned last = g->rc_seq;
            g->rc_seq = last + 1 == 0 ? 1 : last + 1;
            (void)o3dmi_internal_raycast_tile_order(
                    g->rc_cost, g->rc_order, (int)n_tiles, last, g->rc_seq);
        }
    }
    st = o3dmi_internal_estimate_range(
            block_coords_dev, key_stride, max_m, m_dev, range_map_dev,
            map_is_clean, intrinsic, extrinsic, height, width,
            range_map_down_factor, g->block_resolution, g->voxel_size,
            depth_min, depth_max, stream);
    if (st) {
        g->own_range_clean = false;
        return st;
    }
    const void* cbuf = (ci >= 0 && out_color)
                               ? o3dmi_hash_value_buffer(g->block_hashmap, ci)
                               : nullptr;
    st = o3dmi_vbg_raycast(
            g->block_hashmap,
            (const float*)o3dmi_hash_value_buffer(g->block_hashmap, ti),
            o3dmi_hash_value_buffer(g->block_hashmap, wi), cbuf, grid_dtype,
            range_map_dev, out_depth, out_vertex, out_color, out_normal,
            out_index, out_mask, out_ratio, out_ratio_dx, out_ratio_dy,
            out_ratio_dz, intrinsic, extrinsic, height, width,
            (int)g->block_resolution, g->voxel_size, depth_scale, depth_min,
            depth_max, weight_threshold, trunc_voxel_multiplier,
            range_map_down_factor, stream);
    if (st) g->own_range_clean = false;
    return st;
}

int o3dmi_vbg_ray_cast_sharded(
        o3dmi_vbg_t* g, const int32_t* block_coords_dev, int64_t m,
        const double* intrinsic, const double* extrinsic, int width, int height,
        float* range_map_dev, float* out_depth, float* out_vertex,
        float* out_color, float* out_normal, float depth_scale, float depth_min,
        float depth_max, float weight_threshold, float trunc_voxel_multiplier,
        int range_map_down_factor, o3dmi_stream_t stream) {
    o3dmi_comm* comm = ThreadComm();
    if (!comm || comm->world <= 1)
        return o3dmi_vbg_ray_cast(
                g, block_coords_dev, m, intrinsic, extrinsic, width, height,
                range_map_dev, out_depth, out_vertex, out_color, out_normal,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                depth_scale, depth_min, depth_max, weight_threshold,
                trunc_voxel_multiplier, range_map_down_factor, stream);
    // A collective call: what can fail on one rank alone (arguments, the
    // pool, a launch) is the rank-local stage below, whose status the ranks
    // AGREE on before the first all-gather -- a rank whose stage failed does
    // not leave its peers waiting in a collective it never enters; they
    // return O3DMI_ERR_PEER.
    hipStream_t s = (hipStream_t)stream;
    const int world = comm->world, rank = comm->rank;
    struct Map {
        float* out;
        int channels;
        float* staged;
    } maps[4] = {{out_depth, 1, nullptr},
                 {out_vertex, 3, nullptr},
                 {out_color, 3, nullptr},
                 {out_normal, 3, nullptr}};
    struct Staging {
        hipStream_t s;
        void* p = nullptr;
        ~Staging() {
            if (!p) return;
            (void)hipStreamSynchronize(s);
            PoolFree(p);
        }
    } staging{s};
    int band_rows = 0;
    // ---- rank-local stage: the range map and this rank's band of tile rows,
    // rendered into rows [r0, r1) of maps of `band_rows * world` rows (the
    // kernel addresses pixels of the whole image)
    const auto render_band = [&]() -> int {
        O3DMI_REQUIRE(g && range_map_dev && intrinsic && extrinsic &&
                              width > 0 && height > 0,
                      "bad argument");
        int ti = g->AttrIndex("tsdf"), wi = g->AttrIndex("weight"),
            ci = g->AttrIndex("color");
        if (ti < 0 || wi < 0) {
            SetLastError(
                    "TSDF and/or weight not allocated in blocks, please "
                    "implement customized integration.");
            return O3DMI_ERR_INVALID_ARG;
        }
        int grid_dtype;
        int st = GridDtype(g, &grid_dtype);
        if (st) return st;
        // the range map is cheap (one pass over the frustum's block keys)
        // and replicated: every rank needs the cells of its band only, but
        // all of it is an output of the call
        if ((st = o3dmi_vbg_estimate_range_dev(
                     block_coords_dev, m, nullptr, range_map_dev, intrinsic,
                     extrinsic, height, width, range_map_down_factor,
                     g->block_resolution, g->voxel_size, depth_min, depth_max,
                     stream)))
            return st;
        const int tiles = (height + 7) / 8;
        const int band_tiles = (tiles + world - 1) / world;
        band_rows = band_tiles * 8;
        const int padded = band_rows * world;  // rows of the gathered maps
        int r0 = rank * band_rows, r1 = r0 + band_rows;
        if (r1 > height) r1 = height;
        // a rank past the last tile row has no band (rows 0..0: the clipped
        // start `height` is no tile boundary when height % 8 != 0)
        if (r0 >= height) r0 = r1 = 0;
        size_t floats = 0;
        for (Map& mp : maps)
            if (mp.out) floats += (size_t)padded * width * mp.channels;
        if (floats == 0) return O3DMI_OK;
        if ((st = PoolAlloc(&staging.p, floats * sizeof(float)))) return st;
        float* q = (float*)staging.p;
        for (Map& mp : maps)
            if (mp.out) {
                mp.staged = q;
                q += (size_t)padded * width * mp.channels;
            }
        const void* cbuf =
                (ci >= 0 && out_color)
                        ? o3dmi_hash_value_buffer(g->block_hashmap, ci)
                        : nullptr;
        return o3dmi_vbg_raycast_rows(
                g->block_hashmap,
                (const float*)o3dmi_hash_value_buffer(g->block_hashmap, ti),
                o3dmi_hash_value_buffer(g->block_hashmap, wi), cbuf,
                grid_dtype, range_map_dev, maps[0].staged, maps[1].staged,
                maps[2].staged, maps[3].staged, nullptr, nullptr, nullptr,
                nullptr, nullptr, nullptr, intrinsic, extrinsic, height, width,
                r0, r1, (int)g->block_resolution, g->voxel_size, depth_scale,
                depth_min, depth_max, weight_threshold, trunc_voxel_multiplier,
                range_map_down_factor, stream);
    };
    int st = comm->AgreeStatus(render_band(), s);
    if (st) return st;
    // ---- collective stage: a rank's band is a contiguous run of rows, so
    // the maps are gathered in place; the caller's {height, width, C} maps
    // are their first rows
    for (Map& mp : maps) {
        if (!mp.out) continue;
        const int64_t seg = (int64_t)band_rows * width * mp.channels *
                            (int64_t)sizeof(float);
        if ((st = comm->Allgather((char*)mp.staged + (size_t)seg * rank,
                                  mp.staged, seg, s)))
            return st;
        O3DMI_HIP_CHECK(hipMemcpyAsync(
                mp.out, mp.staged,
                (size_t)height * width * mp.channels * sizeof(float),
                hipMemcpyDeviceToDevice, s));
    }
    return O3DMI_OK;
}

int o3dmi_vbg_extract_point_cloud(o3dmi_vbg_t* g, float weight_threshold,
                                  int64_t capacity, float* points_dev,
                                  float* normals_dev, float* colors_dev,
                                  int64_t* total_out, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && total_out, "null argument");
    int ti = g->AttrIndex("tsdf"), wi = g->AttrIndex("weight"),
        ci = g->AttrIndex("color");
    if (ti < 0 || wi < 0) {
        SetLastError(
                "TSDF and/or weight not allocated in blocks, please implement "
                "customized integration.");
        return O3DMI_ERR_INVALID_ARG;
    }
    int grid_dtype;
    int st = GridDtype(g, &grid_dtype);
    if (st) return st;
    // block_hashmap_->GetActiveIndices(active_buf_indices), sorted so that the
    // output order is a function of the grid state only.
    int32_t* active = nullptr;
    const int64_t cap = o3dmi_hash_capacity(g->block_hashmap);
    if ((st = PoolAlloc((void**)&active, sizeof(int32_t) * (size_t)cap)))
        return st;
    int64_t n = 0;
    st = o3dmi_hash_active_indices(g->block_hashmap, active, stream, &n);
    if (!st) st = o3dmi_sort_indices(active, n, stream);
    if (!st)
        st = o3dmi_vbg_extract_points(
                g->block_hashmap, active, n,
                (const float*)o3dmi_hash_value_buffer(g->block_hashmap, ti),
                o3dmi_hash_value_buffer(g->block_hashmap, wi),
                ci >= 0 ? o3dmi_hash_value_buffer(g->block_hashmap, ci)
                        : nullptr,
                grid_dtype, (int)g->block_resolution, g->voxel_size,
                weight_threshold, points_dev, normals_dev,
                ci >= 0 ? colors_dev : nullptr, capacity, total_out, stream);
    // extract_points synchronised the stream (or failed before launching).
    (void)hipStreamSynchronize((hipStream_t)stream);
    PoolFree(active);
    return st;
}

// SURVEY 8(e)(B), the payload step: every active block goes to the rank that
// OWNS it (OwnerOf, the rule of the block-ownership scheme); the owner folds
// the partial blocks of all ranks into one. Afterwards the grids of the ranks
// are disjoint and their union is the model of the whole stream -- the layout
// the block-ownership scheme produces directly.
//   1. active buffer indices, ascending; owner of each; grouped by owner;
//   2. all-gather of the per-owner counts (world x world int64);
//   3. keys and, per attribute, the value rows gathered into owner order and
//      exchanged with ONE all-to-all each (byte ranges; nothing to itself):
//      a rank sends (world - 1) / world of its blocks and receives about as
//      much, instead of world x everything with an all-gather;
//   4. the blocks sent away are erased here and their rows zeroed;
//   5. the received groups are folded in, ascending source rank
//      (o3dmi_vbg_merge_blocks = Integrate's running mean).
int o3dmi_vbg_merge_frame_sharded(o3dmi_vbg_t* g, o3dmi_comm_t* comm,
                                  o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && comm, "null argument");
    const int world = comm->world, me = comm->rank;
    O3DMI_REQUIRE(world >= 1 && world <= kMaxWorld, "world size out of range");
    if (world == 1) return O3DMI_OK;
    O3DMI_REQUIRE(!g->replicated,
                  "merge_frame_sharded after allgather_owned_blocks: the grid "
                  "holds the other ranks' finished blocks, merging again would "
                  "count them twice");
    hipStream_t s = (hipStream_t)stream;
    const size_t n_attr = g->attr_names.size();
    const int64_t res = g->block_resolution;
    std::vector<int64_t> row_bytes(n_attr);
    for (size_t i = 0; i < n_attr; ++i) {
        row_bytes[i] = res * res * res * g->attr_channels[i] *
                       DtypeSize(g->attr_dtypes[i]);
        O3DMI_REQUIRE(row_bytes[i] % 16 == 0, "value rows must be 16-byte "
                                              "multiples");
    }
    struct Scratch {
        hipStream_t s;
        std::vector<void*> p;
        int Alloc(void** out, size_t bytes) {
            int e = PoolAlloc(out, bytes ? bytes : 16);
            if (!e) p.push_back(*out);
            return e;
        }
        ~Scratch() {
            (void)hipStreamSynchronize(s);
            for (void* q : p) PoolFree(q);
        }
    } scratch{s, {}};
    int st;
    // 1. ---------------------------------------------------------------------
    // (a function: run again after a Reserve, which renumbers the buffer
    // indices `grouped` holds)
    int32_t *active = nullptr, *owner = nullptr, *grouped = nullptr;
    int* counters = nullptr;  // counts[world] | cursor[world]
    int64_t n = 0;
    const int* key_buffer = nullptr;
    int host_counts[kMaxWorld] = {0};
    OwnerOffsets off;
    auto group_by_owner = [&]() -> int {
        const int64_t cap = o3dmi_hash_capacity(g->block_hashmap);
        int e;
        if ((e = scratch.Alloc((void**)&active, sizeof(int32_t) * (size_t)cap)) ||
            (e = scratch.Alloc((void**)&owner, sizeof(int32_t) * (size_t)cap)) ||
            (e = scratch.Alloc((void**)&grouped,
                               sizeof(int32_t) * (size_t)cap)) ||
            (e = scratch.Alloc((void**)&counters, sizeof(int) * 2 * kMaxWorld)))
            return e;
        n = 0;
        if ((e = o3dmi_hash_active_indices(g->block_hashmap, active, stream,
                                           &n)))
            return e;
        if (n > 1 && (e = o3dmi_sort_indices(active, n, stream))) return e;
        O3DMI_HIP_CHECK(hipMemsetAsync(counters, 0,
                                       sizeof(int) * 2 * kMaxWorld, s));
        key_buffer = (const int*)o3dmi_hash_key_buffer(g->block_hashmap);
        if (n > 0)
            hipLaunchKernelGGL(OwnerCountKernel, dim3(GridFor(n, kBlock)),
                               dim3(kBlock), 0, s, active, n, key_buffer, world,
                               owner, counters);
        for (int r = 0; r < kMaxWorld; ++r) host_counts[r] = 0;
        O3DMI_HIP_CHECK(hipMemcpyAsync(host_counts, counters,
                                       sizeof(int) * world,
                                       hipMemcpyDeviceToHost, s));
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        int run = 0;
        for (int r = 0; r < kMaxWorld; ++r) {
            off.v[r] = run;
            if (r < world) run += host_counts[r];
        }
        if (n > 0)
            hipLaunchKernelGGL(OwnerGroupKernel, dim3(GridFor(n, kBlock)),
                               dim3(kBlock), 0, s, active, n, owner, off,
                               counters + kMaxWorld, grouped);
        O3DMI_HIP_CHECK(hipGetLastError());
        return O3DMI_OK;
    };
    // (A rank-local failure must not leave the peers waiting in an exchange
    // this rank never enters: step 1's status travels with the counts of step
    // 2 -- a negative first count --, the Reserve's is agreed on before step 3;
    // every rank then returns, its own error or O3DMI_ERR_PEER.)
    const int grouped_st = group_by_owner();
    // 2. ---------------------------------------------------------------------
    int64_t* matrix_dev = nullptr;  // [world][world]: row r = rank r's counts
    if ((st = scratch.Alloc((void**)&matrix_dev,
                            sizeof(int64_t) * (size_t)world * (world + 1))))
        return st;
    std::vector<int64_t> mine((size_t)world), matrix((size_t)world * world);
    for (int r = 0; r < world; ++r) mine[(size_t)r] = host_counts[r];
    if (grouped_st) mine[0] = -(int64_t)grouped_st;
    int64_t* mine_dev = matrix_dev + (size_t)world * world;
    O3DMI_HIP_CHECK(hipMemcpyAsync(mine_dev, mine.data(),
                                   sizeof(int64_t) * world,
                                   hipMemcpyHostToDevice, s));
    if ((st = comm->Allgather(mine_dev, matrix_dev, sizeof(int64_t) * world, s)))
        return st;
    O3DMI_HIP_CHECK(hipMemcpyAsync(matrix.data(), matrix_dev,
                                   sizeof(int64_t) * (size_t)world * world,
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (grouped_st) return grouped_st;
    for (int r = 0; r < world; ++r)
        if (matrix[(size_t)r * world] < 0) {
            SetLastError("merge_frame_sharded: rank " + std::to_string(r) +
                         " could not list its blocks (status " +
                         std::to_string(-matrix[(size_t)r * world]) +
                         "); nothing was exchanged");
            return O3DMI_ERR_PEER;
        }
    // blocks per peer: sent (nothing to itself) and received
    std::vector<int64_t> send_n((size_t)world), recv_n((size_t)world),
            send_first((size_t)world), recv_first((size_t)world);
    int64_t recv_total = 0;
    for (int r = 0; r < world; ++r) {
        send_n[(size_t)r] = r == me ? 0 : host_counts[r];
        send_first[(size_t)r] = off.v[r];
        recv_n[(size_t)r] = r == me ? 0 : matrix[(size_t)r * world + me];
        recv_first[(size_t)r] = recv_total;
        recv_total += recv_n[(size_t)r];
    }
    // Room for every block that will arrive BEFORE anything is sent away or
    // erased (a capacity / allocation failure in step 5 used to
    // leave a grid whose foreign blocks were already gone). The map keeps its
    // own host_counts[me] blocks; a Reserve renumbers the buffer indices, so
    // the grouping is redone (the counts the ranks exchanged do not change).
    int room_st = O3DMI_OK;
    if ((int64_t)host_counts[me] + recv_total >
        o3dmi_hash_capacity(g->block_hashmap)) {
        const int64_t cap0 = o3dmi_hash_capacity(g->block_hashmap);
        const int64_t need = (int64_t)n + recv_total;
        room_st = o3dmi_hash_reserve(g->block_hashmap,
                                     need > 2 * cap0 ? need : 2 * cap0, stream);
        g->known_valid = false;
        if (!room_st) room_st = group_by_owner();
    }
    if ((st = comm->AgreeStatus(room_st, s))) return st;
    // 3. ---------------------------------------------------------------------
    auto exchange = [&](const void* src_rows, int64_t row, void** out) -> int {
        char* send = nullptr;
        char* recv = nullptr;
        int e;
        if ((e = scratch.Alloc((void**)&send, (size_t)(n * row))) ||
            (e = scratch.Alloc((void**)&recv, (size_t)(recv_total * row))))
            return e;
        if ((e = GatherRows(src_rows, grouped, n, row, send, s))) return e;
        std::vector<int64_t> sb((size_t)world), so((size_t)world),
                rb((size_t)world), ro((size_t)world);
        for (int r = 0; r < world; ++r) {
            sb[(size_t)r] = send_n[(size_t)r] * row;
            so[(size_t)r] = send_first[(size_t)r] * row;
            rb[(size_t)r] = recv_n[(size_t)r] * row;
            ro[(size_t)r] = recv_first[(size_t)r] * row;
        }
        *out = recv;
        return comm->Alltoallv(send, sb.data(), so.data(), recv, rb.data(),
                               ro.data(), s);
    };
    void* recv_keys = nullptr;
    std::vector<void*> recv_vals(n_attr, nullptr);
    if ((st = exchange(key_buffer, 12, &recv_keys))) return st;
    for (size_t i = 0; i < n_attr; ++i)
        if ((st = exchange(o3dmi_hash_value_buffer(g->block_hashmap, (int)i),
                           row_bytes[i], &recv_vals[i])))
            return st;
    // 4. ---------------------------------------------------------------------
    // what was sent away: the groups of the other owners = grouped[0 ..
    // first(me)) and grouped[first(me) + count(me) .. n)
    {
        int32_t* gone_keys = nullptr;
        if ((st = scratch.Alloc((void**)&gone_keys,
                                sizeof(int32_t) * 3 * (size_t)(n ? n : 1))))
            return st;
        const int64_t head = off.v[me];
        const int64_t tail_first = head + host_counts[me];
        const int64_t tail = n - tail_first;
        const int64_t spans[2][2] = {{0, head}, {tail_first, tail}};
        for (const auto& sp : spans) {
            const int64_t first = sp[0], cnt = sp[1];
            if (cnt <= 0) continue;
            if ((st = GatherRows(key_buffer, grouped + first, cnt, 12, gone_keys,
                                 s)))
                return st;
            for (size_t i = 0; i < n_attr; ++i) {
                const int grid = (int)(cnt < 65536 ? cnt : 65536);
                hipLaunchKernelGGL(
                        ZeroRowsKernel, dim3(grid), dim3(256), 0, s,
                        (uint8_t*)o3dmi_hash_value_buffer(g->block_hashmap,
                                                          (int)i),
                        grouped + first, cnt, row_bytes[i] / 16);
            }
            O3DMI_HIP_CHECK(hipGetLastError());
            if ((st = o3dmi_hash_erase(g->block_hashmap, gone_keys, cnt, nullptr,
                                       stream)))
                return st;
        }
        g->known_valid = false;
        g->last_path = 0;
    }
    // 5. ---------------------------------------------------------------------
    std::vector<const void*> vals(n_attr);
    for (int r = 0; r < world; ++r) {
        const int64_t cnt = recv_n[(size_t)r];
        if (cnt <= 0) continue;
        const int64_t first = recv_first[(size_t)r];
        for (size_t i = 0; i < n_attr; ++i)
            vals[i] = (const char*)recv_vals[i] + first * row_bytes[i];
        if ((st = o3dmi_vbg_merge_blocks(
                     g, (const int32_t*)recv_keys + 3 * first, vals.data(), cnt,
                     stream)))
            return st;
    }
    return O3DMI_OK;
}

// The step after it when EVERY rank wants the whole model (a ray cast on each
// GPU): all-gather of the owners' finished blocks. Counts first, then keys and
// rows padded to the largest rank; foreign blocks are absent here (erased and
// zeroed above), so folding them in copies them.
int o3dmi_vbg_allgather_owned_blocks(o3dmi_vbg_t* g, o3dmi_comm_t* comm,
                                     o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && comm, "null argument");
    const int world = comm->world, me = comm->rank;
    O3DMI_REQUIRE(world >= 1 && world <= kMaxWorld, "world size out of range");
    if (world == 1) return O3DMI_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t n_attr = g->attr_names.size();
    const int64_t res = g->block_resolution;
    struct Scratch {
        hipStream_t s;
        std::vector<void*> p;
        int Alloc(void** out, size_t bytes) {
            int e = PoolAlloc(out, bytes ? bytes : 16);
            if (!e) p.push_back(*out);
            return e;
        }
        ~Scratch() {
            (void)hipStreamSynchronize(s);
            for (void* q : p) PoolFree(q);
        }
    } scratch{s, {}};
    int st;
    // (Rank-local failures do not leave the peers waiting: the status of the
    // count travels as a negative count, the status of the export is agreed
    // on before the payload all-gathers.)
    int64_t n = 0;
    const int count_st =
            o3dmi_vbg_export_blocks(g, 0, nullptr, nullptr, &n, stream);
    if (count_st) n = -(int64_t)count_st;
    int64_t* counts_dev = nullptr;
    if ((st = scratch.Alloc((void**)&counts_dev,
                            sizeof(int64_t) * (size_t)(world + 1))))
        return st;
    O3DMI_HIP_CHECK(hipMemcpyAsync(counts_dev + world, &n, sizeof(int64_t),
                                   hipMemcpyHostToDevice, s));
    if ((st = comm->Allgather(counts_dev + world, counts_dev, sizeof(int64_t),
                              s)))
        return st;
    std::vector<int64_t> counts((size_t)world);
    O3DMI_HIP_CHECK(hipMemcpyAsync(counts.data(), counts_dev,
                                   sizeof(int64_t) * world,
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (count_st) return count_st;
    int64_t m = 0;
    for (int r = 0; r < world; ++r) {
        const int64_t c = counts[(size_t)r];
        if (c < 0) {
            SetLastError("allgather_owned_blocks: rank " + std::to_string(r) +
                         " could not count its blocks (status " +
                         std::to_string(-c) + "); nothing was exchanged");
            return O3DMI_ERR_PEER;
        }
        m = c > m ? c : m;
    }
    if (m == 0) return O3DMI_OK;
    int32_t* keys = nullptr;
    int32_t* all_keys = nullptr;
    std::vector<void*> rows(n_attr), all_rows(n_attr);
    std::vector<int64_t> row_bytes(n_attr);
    for (size_t i = 0; i < n_attr; ++i)
        row_bytes[i] = res * res * res * g->attr_channels[i] *
                       DtypeSize(g->attr_dtypes[i]);
    // the scratch scales with the LARGEST rank's share: the likely place for
    // one rank alone to run out of memory
    const auto export_mine = [&]() -> int {
        int e;
        if ((e = scratch.Alloc((void**)&keys, (size_t)m * 12)) ||
            (e = scratch.Alloc((void**)&all_keys, (size_t)m * 12 * world)))
            return e;
        for (size_t i = 0; i < n_attr; ++i)
            if ((e = scratch.Alloc(&rows[i], (size_t)(m * row_bytes[i]))) ||
                (e = scratch.Alloc(&all_rows[i],
                                   (size_t)(m * row_bytes[i] * world))))
                return e;
        int64_t n2 = 0;
        return o3dmi_vbg_export_blocks(g, m, keys, rows.data(), &n2, stream);
    };
    if ((st = comm->AgreeStatus(export_mine(), s))) return st;
    if ((st = comm->Allgather(keys, all_keys, m * 12, s))) return st;
    for (size_t i = 0; i < n_attr; ++i)
        if ((st = comm->Allgather(rows[i], all_rows[i], m * row_bytes[i], s)))
            return st;
    std::vector<const void*> vals(n_attr);
    for (int r = 0; r < world; ++r) {
        if (r == me || counts[(size_t)r] == 0) continue;
        for (size_t i = 0; i < n_attr; ++i)
            vals[i] = (const char*)all_rows[i] + (int64_t)r * m * row_bytes[i];
        if ((st = o3dmi_vbg_merge_blocks(g, all_keys + 3 * (int64_t)r * m,
                                         vals.data(), counts[(size_t)r],
                                         stream)))
            return st;
    }
    g->replicated = world > 1;
    return O3DMI_OK;
}

int o3dmi_vbg_attribute_count(const o3dmi_vbg_t* g) {
    return g ? (int)g->attr_names.size() : 0;
}
const char* o3dmi_vbg_attribute_name(const o3dmi_vbg_t* g, int i) {
    if (!g || i < 0 || i >= (int)g->attr_names.size()) return nullptr;
    return g->attr_names[(size_t)i].c_str();
}
float o3dmi_vbg_voxel_size(const o3dmi_vbg_t* g) {
    return g ? g->voxel_size : 0.f;
}
int64_t o3dmi_vbg_block_resolution(const o3dmi_vbg_t* g) {
    return g ? g->block_resolution : 0;
}

int o3dmi_vbg_save(o3dmi_vbg_t* g, const char* file_name,
                   o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && file_name, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t cap = o3dmi_hash_capacity(g->block_hashmap);
    int32_t* active = nullptr;
    int st = PoolAlloc((void**)&active, sizeof(int32_t) * (size_t)cap);
    if (st) return st;
    struct Scratch {
        hipStream_t s;
        std::vector<void*> p;
        ~Scratch() {
            (void)hipStreamSynchronize(s);
            for (void* q : p) PoolFree(q);
        }
    } scratch{s, {active}};
    int64_t n = 0;
    if ((st = o3dmi_hash_active_indices(g->block_hashmap, active, stream, &n)))
        return st;
    if ((st = o3dmi_sort_indices(active, n, stream))) return st;

    o3dmi_npz z;
    auto scalar = [&](const std::string& name, int dtype, const void* v,
                      size_t bytes, bool zero_d) {
        NpzArray a;
        a.name = name;
        a.dtype = dtype;
        if (!zero_d) a.shape = {1};
        a.data.assign((const uint8_t*)v, (const uint8_t*)v + bytes);
        z.arrays.push_back(std::move(a));
    };
    const float vs = g->voxel_size;
    const int64_t res = g->block_resolution;
    const uint8_t zero = 0;
    scalar("voxel_size", O3DMI_F32, &vs, sizeof(vs), false);
    scalar("block_resolution", O3DMI_I64, &res, sizeof(res), false);
    scalar("HIP:0", O3DMI_U8, &zero, 1, true);  // device placeholder
    for (size_t i = 0; i < g->attr_names.size(); ++i) {
        const int32_t id = (int32_t)i;
        scalar("attr_name_" + g->attr_names[i], O3DMI_I32, &id, sizeof(id),
               false);
    }
    // keys.IndexGet(active) / values[i].IndexGet(active) -> host
    auto gathered = [&](const void* src, int64_t row_bytes, NpzArray* a) -> int {
        a->data.resize((size_t)(n * row_bytes));
        if (n == 0) return O3DMI_OK;
        void* tmp = nullptr;
        int e = PoolAlloc(&tmp, (size_t)(n * row_bytes));
        if (e) return e;
        scratch.p.push_back(tmp);
        if ((e = GatherRows(src, active, n, row_bytes, tmp, s))) return e;
        O3DMI_HIP_CHECK(hipMemcpyAsync(a->data.data(), tmp,
                                       (size_t)(n * row_bytes),
                                       hipMemcpyDeviceToHost, s));
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        return O3DMI_OK;
    };
    {
        NpzArray a;
        a.name = "key";
        a.dtype = O3DMI_I32;
        a.shape = {n, 3};
        if ((st = gathered(o3dmi_hash_key_buffer(g->block_hashmap), 12, &a)))
            return st;
        z.arrays.push_back(std::move(a));
    }
    for (size_t i = 0; i < g->attr_names.size(); ++i) {
        NpzArray a;
        char nm[32];
        std::snprintf(nm, sizeof(nm), "value_%03d", (int)i);
        a.name = nm;
        a.dtype = g->attr_dtypes[i];
        a.shape = {n, res, res, res, (int64_t)g->attr_channels[i]};
        const int64_t row = res * res * res * g->attr_channels[i] *
                            DtypeSize(g->attr_dtypes[i]);
        if ((st = gathered(o3dmi_hash_value_buffer(g->block_hashmap, (int)i),
                           row, &a)))
            return st;
        z.arrays.push_back(std::move(a));
    }
    std::string path = file_name;
    std::string ext;
    {
        const size_t dot = path.find_last_of('.');
        if (dot != std::string::npos) ext = path.substr(dot + 1);
        for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
    }
    // "File name for a voxel grid should be with the extension .npz."
    if (ext != "npz") path += ".npz";
    return o3dmi_npz_write(&z, path.c_str());
}

int o3dmi_vbg_export_blocks(o3dmi_vbg_t* g, int64_t capacity,
                            int32_t* keys_dev, void* const* values_dev,
                            int64_t* n_out, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g && n_out, "null argument");
    O3DMI_REQUIRE(keys_dev == nullptr || values_dev != nullptr,
                  "values_dev is null");
    hipStream_t s = (hipStream_t)stream;
    const int64_t cap = o3dmi_hash_capacity(g->block_hashmap);
    int32_t* active = nullptr;
    int st = PoolAlloc((void**)&active, sizeof(int32_t) * (size_t)cap);
    if (st) return st;
    struct Scratch {
        hipStream_t s;
        void* p;
        ~Scratch() {
            (void)hipStreamSynchronize(s);
            PoolFree(p);
        }
    } scratch{s, active};
    int64_t n = 0;
    if ((st = o3dmi_hash_active_indices(g->block_hashmap, active, stream, &n)))
        return st;
    *n_out = n;
    if (!keys_dev || n == 0) return O3DMI_OK;
    if (n > capacity) {
        SetLastError("export_blocks: more active blocks than `capacity`");
        return O3DMI_ERR_CAPACITY;
    }
    if ((st = o3dmi_sort_indices(active, n, stream))) return st;
    if ((st = GatherRows(o3dmi_hash_key_buffer(g->block_hashmap), active, n, 12,
                         keys_dev, s)))
        return st;
    const int64_t res = g->block_resolution;
    for (size_t i = 0; i < g->attr_names.size(); ++i) {
        O3DMI_REQUIRE(values_dev[i] != nullptr, "values_dev[i] is null");
        const int64_t row = res * res * res * g->attr_channels[i] *
                            DtypeSize(g->attr_dtypes[i]);
        if ((st = GatherRows(o3dmi_hash_value_buffer(g->block_hashmap, (int)i),
                             active, n, row, values_dev[i], s)))
            return st;
    }
    return O3DMI_OK;
}

int o3dmi_vbg_merge_blocks(o3dmi_vbg_t* g, const int32_t* keys_dev,
                           const void* const* values_dev, int64_t n,
                           o3dmi_stream_t stream) {
    O3DMI_REQUIRE(g != nullptr, "grid is null");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(keys_dev && values_dev, "null argument");
    const int ti = g->AttrIndex("tsdf"), wi = g->AttrIndex("weight"),
              ci = g->AttrIndex("color");
    if (ti < 0 || wi < 0) {
        SetLastError("TSDF and/or weight not allocated in blocks");
        return O3DMI_ERR_INVALID_ARG;
    }
    O3DMI_REQUIRE(g->attr_dtypes[(size_t)ti] == O3DMI_F32,
                  "tsdf must be Float32");
    O3DMI_REQUIRE((int)g->attr_names.size() ==
                          2 + (ci >= 0 ? 1 : 0),
                  "merge_blocks handles the tsdf / weight / color attributes");
    O3DMI_REQUIRE(ci < 0 || g->attr_channels[(size_t)ci] == 3,
                  "color must have 3 channels");
    int grid_dtype;
    int st = GridDtype(g, &grid_dtype);
    if (st) return st;
    O3DMI_REQUIRE(values_dev[ti] && values_dev[wi] &&
                          (ci < 0 || values_dev[ci]),
                  "values_dev[i] is null");
    g->known_valid = false;
    if ((st = EnsureCapacity(g, n, stream))) return st;
    if ((st = EnsureScratch(g, n))) return st;
    if ((st = o3dmi_hash_activate(g->block_hashmap, keys_dev, n, nullptr,
                                  nullptr, nullptr, stream)))
        return st;
    if ((st = o3dmi_hash_find(g->block_hashmap, keys_dev, n, nullptr,
                              g->scratch_buf_indices, nullptr, stream)))
        return st;
    const int64_t res = g->block_resolution;
    const int vpb = (int)(res * res * res);
    const int64_t total = n * vpb;
    const int block = 256;
    const int64_t want = (total + block - 1) / block;
    const int grid = (int)(want < 65536 ? want : 65536);
    float* tsdf = (float*)o3dmi_hash_value_buffer(g->block_hashmap, ti);
    void* weight = o3dmi_hash_value_buffer(g->block_hashmap, wi);
    void* color = ci >= 0 ? o3dmi_hash_value_buffer(g->block_hashmap, ci)
                          : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (grid_dtype == O3DMI_F32)
        MergeBlocksKernel<float><<<grid, block, 0, s>>>(
                g->scratch_buf_indices, total, vpb, tsdf, (float*)weight,
                (float*)color, (const float*)values_dev[ti],
                (const float*)values_dev[wi],
                ci >= 0 ? (const float*)values_dev[ci] : nullptr);
    else
        MergeBlocksKernel<uint16_t><<<grid, block, 0, s>>>(
                g->scratch_buf_indices, total, vpb, tsdf, (uint16_t*)weight,
                (uint16_t*)color, (const float*)values_dev[ti],
                (const uint16_t*)values_dev[wi],
                ci >= 0 ? (const uint16_t*)values_dev[ci] : nullptr);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

static int VbgLoadImpl(const char* file_name, o3dmi_stream_t stream,
                       o3dmi_vbg_t** out);
int o3dmi_vbg_load(const char* file_name, o3dmi_stream_t stream,
                   o3dmi_vbg_t** out) {
    O3DMI_REQUIRE(file_name && out, "null argument");
    try {  // nothing may throw across the C ABI
        return VbgLoadImpl(file_name, stream, out);
    } catch (const std::exception& e) {
        SetLastError(std::string("o3dmi_vbg_load: ") + e.what());
        return O3DMI_ERR_INVALID_ARG;
    } catch (...) {
        SetLastError("o3dmi_vbg_load: malformed file");
        return O3DMI_ERR_INVALID_ARG;
    }
}
static int VbgLoadImpl(const char* file_name, o3dmi_stream_t stream,
                       o3dmi_vbg_t** out) {
    hipStream_t s = (hipStream_t)stream;
    o3dmi_npz_t* zp = nullptr;
    int st = o3dmi_npz_read(file_name, &zp);
    if (st) return st;
    struct ZFree {
        o3dmi_npz_t* z;
        ~ZFree() { o3dmi_npz_destroy(z); }
    } zfree{zp};
    const std::string prefix = "attr_name_";
    std::vector<std::string> names;
    for (const NpzArray& a : zp->arrays) {
        if (a.name.compare(0, prefix.size(), prefix) == 0) {
            O3DMI_REQUIRE(a.dtype == O3DMI_I32 && a.NumElements() >= 1,
                          "malformed attr_name entry");
            const int id = *(const int32_t*)a.data.data();
            O3DMI_REQUIRE(id >= 0 && id < 8, "attribute index out of range");
            if ((int)names.size() <= id) names.resize((size_t)id + 1);
            names[(size_t)id] = a.name.substr(prefix.size());
        }
    }
    O3DMI_REQUIRE(!names.empty(),
                  "Attribute names not found, not a valid file for voxel block "
                  "grids.");
    const NpzArray* key = zp->Find("key");
    const NpzArray* vsz = zp->Find("voxel_size");
    const NpzArray* bres = zp->Find("block_resolution");
    O3DMI_REQUIRE(key && vsz && bres, "key / voxel_size / block_resolution "
                                      "missing, not a valid voxel block grid "
                                      "file.");
    O3DMI_REQUIRE(key->dtype == O3DMI_I32 && key->shape.size() == 2 &&
                          key->shape[1] == 3,
                  "key must be {n,3} Int32");
    O3DMI_REQUIRE(vsz->dtype == O3DMI_F32 && vsz->NumElements() >= 1 &&
                          bres->dtype == O3DMI_I64 && bres->NumElements() >= 1,
                  "voxel_size must be Float32, block_resolution Int64");
    const float voxel_size = *(const float*)vsz->data.data();
    const int64_t res = *(const int64_t*)bres->data.data();
    const int64_t n = key->shape[0];
    std::vector<const NpzArray*> vals(names.size());
    std::vector<int> dtypes(names.size()), chans(names.size());
    std::vector<const char*> cnames(names.size());
    for (size_t i = 0; i < names.size(); ++i) {
        char nm[32];
        std::snprintf(nm, sizeof(nm), "value_%03d", (int)i);
        vals[i] = zp->Find(nm);
        O3DMI_REQUIRE(vals[i] != nullptr && !names[i].empty(),
                      "value tensor missing for an attribute");
        const auto& sh = vals[i]->shape;
        O3DMI_REQUIRE(sh.size() >= 4 && sh[0] == n && sh[1] == res &&
                              sh[2] == res && sh[3] == res,
                      "value tensor shape mismatch");
        int64_t c = 1;
        for (size_t k = 4; k < sh.size(); ++k) c *= sh[k];
        dtypes[i] = vals[i]->dtype;
        chans[i] = (int)c;
        cnames[i] = names[i].c_str();
    }
    o3dmi_vbg_t* g = nullptr;
    // VoxelBlockGrid(attr_names, attr_dtypes, attr_channels, voxel_size,
    //                block_resolution, keys.GetLength(), device)
    st = o3dmi_vbg_create((int)names.size(), cnames.data(), dtypes.data(),
                          chans.data(), voxel_size, res, n > 0 ? n : 1, stream,
                          &g);
    if (st) return st;
    if (n > 0) {
        // block_hashmap.Insert(keys, soa_value_tensor)
        std::vector<void*> dev;
        auto cleanup = [&]() {
            (void)hipStreamSynchronize(s);
            for (void* p : dev) PoolFree(p);
        };
        auto upload = [&](const std::vector<uint8_t>& h, void** d) -> int {
            int e = PoolAlloc(d, h.size() ? h.size() : 1);
            if (e) return e;
            dev.push_back(*d);
            O3DMI_HIP_CHECK(hipMemcpyAsync(*d, h.data(), h.size(),
                                           hipMemcpyHostToDevice, s));
            return O3DMI_OK;
        };
        void* kd = nullptr;
        std::vector<const void*> vd(names.size());
        st = upload(key->data, &kd);
        for (size_t i = 0; i < names.size() && !st; ++i) {
            void* p = nullptr;
            st = upload(vals[i]->data, &p);
            vd[i] = p;
        }
        if (!st)
            st = o3dmi_hash_insert(g->block_hashmap, (const int32_t*)kd,
                                   vd.data(), n, nullptr, nullptr, stream);
        cleanup();
        if (st) {
            o3dmi_vbg_destroy(g);
            return st;
        }
        g->size_bound = n;
    }
    *out = g;
    return O3DMI_OK;
}

int o3dmi_vbg_profile_begin(o3dmi_vbg_t* g, int max_frames, int stride) {
    O3DMI_REQUIRE(g && max_frames > 0 && stride >= 0, "bad argument");
    g->prof_stride = stride;
    g->prof_seen = 0;
    while ((int)g->prof_events.size() < max_frames * 2) {
        hipEvent_t e;
        O3DMI_HIP_CHECK(hipEventCreate(&e));
        g->prof_events.push_back(e);
    }
    if (g->prof_max < max_frames || !g->prof_counts) {
        (void)hipFree(g->prof_counts);
        g->prof_counts = nullptr;
        O3DMI_HIP_CHECK(hipMalloc((void**)&g->prof_counts,
                                  sizeof(int32_t) * 3 * (size_t)max_frames));
    }
    O3DMI_HIP_CHECK(hipMemset(g->prof_counts, 0,
                              sizeof(int32_t) * 3 * (size_t)max_frames));
    g->prof_max = max_frames;
    g->prof_frames = 0;
    g->prof_launch_frames = 0;
    g->profiling = true;
    return O3DMI_OK;
}

int o3dmi_vbg_profile_end(o3dmi_vbg_t* g, o3dmi_stream_t stream,
                          double* integrate_ms, int64_t* launches,
                          int64_t* block_frames, int64_t* frames) {
    O3DMI_REQUIRE(g && integrate_ms && launches && block_frames && frames,
                  "null argument");
    g->profiling = false;
    O3DMI_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    double ti = 0;
    g->prof_launch_ms.assign((size_t)g->prof_frames, 0.0f);
    for (int f = 0; f < g->prof_frames; ++f) {
        float ms = 0;
        O3DMI_HIP_CHECK(hipEventElapsedTime(
                &ms, g->prof_events[(size_t)f * 2 + 0],
                g->prof_events[(size_t)f * 2 + 1]));
        ti += ms;
        g->prof_launch_ms[(size_t)f] = ms;
    }
    g->prof_launch_counts.assign(3 * (size_t)g->prof_frames, 0);
    for (int k = 0; k < 3 && g->prof_frames > 0; ++k)
        O3DMI_HIP_CHECK(hipMemcpy(
                g->prof_launch_counts.data() + (size_t)k * g->prof_frames,
                g->prof_counts + (size_t)k * g->prof_max,
                sizeof(int32_t) * (size_t)g->prof_frames,
                hipMemcpyDeviceToHost));
    std::vector<int32_t> counts((size_t)g->prof_frames);
    if (g->prof_frames > 0)
        O3DMI_HIP_CHECK(hipMemcpy(counts.data(), g->prof_counts,
                                  sizeof(int32_t) * (size_t)g->prof_frames,
                                  hipMemcpyDeviceToHost));
    int64_t bf = 0;
    for (int32_t c : counts) bf += c;
    if (g->prof_frames > 0)
        O3DMI_HIP_CHECK(hipMemcpy(counts.data(), g->prof_counts + g->prof_max,
                                  sizeof(int32_t) * (size_t)g->prof_frames,
                                  hipMemcpyDeviceToHost));
    g->prof_distinct_blocks = 0;
    for (int32_t c : counts) g->prof_distinct_blocks += c;
    *integrate_ms = ti;
    *launches = g->prof_frames;
    *block_frames = bf;
    *frames = g->prof_launch_frames;
    return O3DMI_OK;
}

int64_t o3dmi_vbg_profile_distinct_blocks(const o3dmi_vbg_t* g) {
    return g ? g->prof_distinct_blocks : 0;
}

int64_t o3dmi_vbg_profile_launches(const o3dmi_vbg_t* g, int64_t capacity,
                                   float* ms, int32_t* block_frames,
                                   int32_t* distinct_blocks,
                                   int32_t* map_size) {
    if (!g) return 0;
    const int64_t n = (int64_t)g->prof_launch_ms.size();
    const int64_t m = n < capacity ? n : capacity;
    for (int64_t i = 0; i < m; ++i) {
        if (ms) ms[i] = g->prof_launch_ms[(size_t)i];
        if (block_frames) block_frames[i] = g->prof_launch_counts[(size_t)i];
        if (distinct_blocks)
            distinct_blocks[i] = g->prof_launch_counts[(size_t)(n + i)];
        if (map_size) map_size[i] = g->prof_launch_counts[(size_t)(2 * n + i)];
    }
    return n;
}

}  // extern "C"
