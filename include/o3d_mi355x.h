/* o3d_mi355x.h -- C ABI of the MI355X (gfx950) backend for Open3D's tensor
 * dense-SLAM hot path: point-to-plane ICP (fixed-radius correspondence search
 * + 6x6 Gauss-Newton reduction) and VoxelBlockGrid TSDF integration /
 * ray-casting over spatially hashed voxel blocks.
 *
 * The entry points sit exactly where Open3D's per-device kernels sit: each
 * `o3dmi_*` kernel function replaces one `...CUDA(...)` function that the
 * device dispatchers in cpp/open3d/t/{geometry,pipelines}/kernel/ *.cpp and
 * cpp/open3d/core/{nns,hashmap} call (file:line cited per function, relative
 * to the Open3D source tree). INTEGRATION.md shows the `...HIP` branch a
 * maintainer adds to each dispatcher.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer named *_dev is device (HBM)
 *    memory of the current HIP device; K (3x3) and T (4x4) are host, row-major
 *    double -- the same placement Open3D enforces (t/geometry/Utility.h
 *    CheckIntrinsicTensor / CheckExtrinsicTensor);
 *  - tensors are contiguous row-major with Open3D's layouts: positions/normals
 *    {N,3}, images {H,W,C}, block keys {M,3} int32, voxel values
 *    {capacity,res,res,res,C} (x fastest; GeometryIndexer.h:244-249);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *    functions return without synchronising unless stated;
 *  - return value: 0 = O3DMI_OK, otherwise an o3dmi_status_t; nothing throws.
 *    o3dmi_last_error() gives a thread-local message.
 *  - arithmetic is IEEE float32 without FMA contraction and with correctly
 *    rounded division/sqrt so that results match Open3D's CPU tensor path.
 */
#ifndef O3D_MI355X_H_
#define O3D_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O3DMI_ABI_VERSION 1

typedef enum {
    O3DMI_OK = 0,
    O3DMI_ERR_INVALID_ARG = 1,
    O3DMI_ERR_HIP = 2,          /* a HIP runtime call failed               */
    O3DMI_ERR_CAPACITY = 3,     /* output buffer / hash capacity too small */
    O3DMI_ERR_KEY_RANGE = 4,    /* |block coordinate| >= 2^20              */
    O3DMI_ERR_SINGULAR = 5,     /* singular 6x6 system                     */
    O3DMI_ERR_NO_BLOCKS = 6,    /* "No block is touched in TSDF volume"    */
    O3DMI_ERR_UNSUPPORTED = 7,
    O3DMI_ERR_NO_INLIERS = 8,   /* "Invalid inlier_count value, must be > 0." */
    O3DMI_ERR_INTERNAL = 9,     /* a device-side consistency check failed  */
    O3DMI_ERR_PEER = 10         /* another rank left a collective call with
                                   an error of its own (the sliced block
                                   touch, the sharded ray cast, the block
                                   exchanges): this rank left it as well,
                                   before any payload moved */
} o3dmi_status_t;

typedef enum {
    O3DMI_F32 = 0,
    O3DMI_F64 = 1,
    O3DMI_U16 = 2,
    O3DMI_U8 = 3,
    O3DMI_I32 = 4,
    O3DMI_I64 = 5,
    /* NPZ interchange only (t/io/NumpyIO.cpp:132-155): */
    O3DMI_I8 = 6,
    O3DMI_I16 = 7,
    O3DMI_U32 = 8,
    O3DMI_U64 = 9,
    O3DMI_BOOL = 10
} o3dmi_dtype_t;

/* RobustKernelMethod, t/pipelines/registration/RobustKernel.h:15-23 */
typedef enum {
    O3DMI_L2_LOSS = 0,
    O3DMI_L1_LOSS = 1,
    O3DMI_HUBER_LOSS = 2,
    O3DMI_CAUCHY_LOSS = 3,
    O3DMI_GM_LOSS = 4,
    O3DMI_TUKEY_LOSS = 5,
    O3DMI_GENERALIZED_LOSS = 6
} o3dmi_robust_kernel_t;

typedef void* o3dmi_stream_t;

int o3dmi_abi_version(void);
const char* o3dmi_status_string(int status);
const char* o3dmi_last_error(void);
/* Fills `name` with the gcnArchName of the current device; fails with
 * O3DMI_ERR_HIP when no HIP device is usable. */
int o3dmi_device_info(char* name, size_t name_len, int* cu_count,
                      int64_t* hbm_bytes);
/* Transient scratch (search indices, pyramid levels, sort temporaries) comes
 * from a caching pool inside the library -- the role MemoryManagerCached plays
 * in the reference (core/MemoryManagerCached.cpp). This returns every cached
 * block to the driver. */
int o3dmi_release_cached_memory(void);

/* ------------------------------------------------------------------------ */
/* Spatial hash of int32x3 block keys -> buffer indices.                     */
/* Replaces a DeviceHashBackend (core/hashmap/DeviceHashBackend.h:20-107;    */
/* CUDA: core/hashmap/CUDA/StdGPUHashBackend.h:178-209,305-383) for          */
/* key dtype Int32, key shape {3}. Owns the key buffer {capacity,3} and      */
/* `n_values` value buffers of `value_dsizes[i]` bytes per entry, zero       */
/* filled at allocation (CPUHashBackendBufferAccessor.hpp:30-37).            */
/* buf_indices are NOT guaranteed dense after Erase.                         */
/* ------------------------------------------------------------------------ */
typedef struct o3dmi_hash o3dmi_hash_t;

int o3dmi_hash_create(int64_t capacity, int n_values,
                      const int64_t* value_dsizes, o3dmi_stream_t stream,
                      o3dmi_hash_t** out);
int o3dmi_hash_destroy(o3dmi_hash_t* h);
/* DeviceHashBackend::Clear */
int o3dmi_hash_clear(o3dmi_hash_t* h, o3dmi_stream_t stream);
/* DeviceHashBackend::Insert with no values (HashMap::Activate,
 * core/hashmap/HashMap.cpp:166-181). Per key i: masks[i] = 1 and
 * buf_indices[i] = new index for exactly one of the duplicates of a new key;
 * otherwise masks[i] = 0, buf_indices[i] = 0 (TBBHashBackend.h:203-247).
 * `n_dev` (optional, may be NULL) is a device int32 holding the live count
 * (<= n); when given, only the first *n_dev keys are processed.
 * Caller guarantees size + n <= capacity (HashMap::Activate reserves first). */
int o3dmi_hash_activate(o3dmi_hash_t* h, const int32_t* keys_dev, int64_t n,
                        const int32_t* n_dev, int32_t* buf_indices_dev,
                        uint8_t* masks_dev, o3dmi_stream_t stream);
/* DeviceHashBackend::Insert with values: value i of the winner of key k is
 * copied from values_soa_dev[j] + k * value_dsizes[j]. */
int o3dmi_hash_insert(o3dmi_hash_t* h, const int32_t* keys_dev,
                      const void* const* values_soa_dev, int64_t n,
                      int32_t* buf_indices_dev, uint8_t* masks_dev,
                      o3dmi_stream_t stream);
/* DeviceHashBackend::Find */
int o3dmi_hash_find(o3dmi_hash_t* h, const int32_t* keys_dev, int64_t n,
                    const int32_t* n_dev, int32_t* buf_indices_dev,
                    uint8_t* masks_dev, o3dmi_stream_t stream);
/* DeviceHashBackend::Erase */
int o3dmi_hash_erase(o3dmi_hash_t* h, const int32_t* keys_dev, int64_t n,
                     uint8_t* masks_dev, o3dmi_stream_t stream);
/* DeviceHashBackend::Size -- synchronises `stream`. Also surfaces deferred
 * device-side errors (key range / capacity). */
int o3dmi_hash_size(o3dmi_hash_t* h, o3dmi_stream_t stream, int64_t* size);
int64_t o3dmi_hash_capacity(const o3dmi_hash_t* h);
int64_t o3dmi_hash_bucket_count(const o3dmi_hash_t* h);
/* DeviceHashBackend::GetActiveIndices: writes Size() indices (unordered);
 * out_dev must hold `capacity` entries; count returned on the host
 * (synchronises). */
int o3dmi_hash_active_indices(o3dmi_hash_t* h, int32_t* out_dev,
                              o3dmi_stream_t stream, int64_t* count);
/* Block-ownership sharding for multi-GPU integration (no counterpart in the
 * reference, which is single-device; SURVEY section 8e scheme A): with world >
 * 1 the block-touch kernels that insert into / list from this map
 * (o3dmi_vbg_depth_touch, o3dmi_vbg_touch_activate, the frame-stream path)
 * only see the blocks whose owner -- o3dmi_block_owner(key, world), a fixed
 * 64-bit mix of the packed key -- equals `rank`. Every rank runs the same
 * frames; the union of the per-rank grids is bit-identical to the single-GPU
 * grid and the per-voxel work is split `world` ways. */
int o3dmi_hash_set_ownership(o3dmi_hash_t* h, int rank, int world);
/* Owner rank of one block key (host int32[3]); -1 for bad arguments. */
int o3dmi_block_owner(const int32_t* key3, int world);
/* HashMap::To(device, copy = true) (core/hashmap/HashMap.cpp:230-255): a new
 * map of the same capacity and value layout on HIP device `device`, holding
 * the same key -> value-row association (active keys and their value rows are
 * gathered, copied device to device and inserted; buffer indices are the new
 * map's own, ownership sharding settings are carried over). `device` may be
 * the map's own device (a deep copy). Synchronises both devices; the calling
 * thread's current device is restored. The new map is driven like any other,
 * with streams of its own device. */
int o3dmi_hash_to_device(o3dmi_hash_t* h, int device, o3dmi_hash_t** out);
/* HashMap::Reserve (core/hashmap/HashMap.cpp:47-77): export active
 * key/values, reallocate at `capacity`, re-insert. buf_indices change. */
int o3dmi_hash_reserve(o3dmi_hash_t* h, int64_t capacity,
                       o3dmi_stream_t stream);
/* HashMap::GetKeyTensor / GetValueTensor(i) storage. */
int32_t* o3dmi_hash_key_buffer(o3dmi_hash_t* h);
void* o3dmi_hash_value_buffer(o3dmi_hash_t* h, int i);

/* ------------------------------------------------------------------------ */
/* VoxelBlockGrid kernels (t/geometry/kernel/VoxelBlockGrid.h:338-410).      */
/* ------------------------------------------------------------------------ */

/* DepthTouchCUDA (t/geometry/kernel/VoxelBlockGridCUDA.cu:106-227): every
 * `stride`-th pixel with 0 < d < depth_max emits floor((o + t*dir)/block_size)
 * at 4 samples t in [max(d-trunc,0), min(d+trunc,depth_max)]; the unique set
 * is written to out_coords_dev ({out_capacity,3} int32, order unspecified)
 * and its size to *out_count_dev (device int32). `frustum_hash` is scratch
 * (capacity >= (rows/stride)*(cols/stride)*4) and is cleared here.
 * depth_dtype: O3DMI_U16 or O3DMI_F32. Asynchronous. */
int o3dmi_vbg_depth_touch(o3dmi_hash_t* frustum_hash, const void* depth_dev,
                          int depth_dtype, int rows, int cols,
                          const double* intrinsic, const double* extrinsic,
                          int32_t* out_coords_dev, int64_t out_capacity,
                          int32_t* out_count_dev, int resolution,
                          float voxel_size, float sdf_trunc, float depth_scale,
                          float depth_max, int stride, o3dmi_stream_t stream);

/* PointCloudTouchCUDA (VoxelBlockGridCUDA.cu:42-104): blocks within
 * +-sdf_trunc of each point. frustum_hash capacity >= n*8 (the reference's
 * estimate); overflow reports O3DMI_ERR_CAPACITY at the next size query. */
int o3dmi_vbg_pointcloud_touch(o3dmi_hash_t* frustum_hash,
                               const float* points_dev, int64_t n,
                               int32_t* out_coords_dev, int64_t out_capacity,
                               int32_t* out_count_dev, int resolution,
                               float voxel_size, float sdf_trunc,
                               o3dmi_stream_t stream);

/* GetVoxelCoordinatesAndFlattenedIndicesCUDA
 * (t/geometry/kernel/VoxelBlockGridImpl.h:43-92): for the n_blocks buffer
 * indices given, the world coordinates {n_blocks * res^3, 3} Float32 of every
 * voxel's origin corner, (key * res + voxel) * voxel_size, and its linear
 * index into the value tensors {n_blocks * res^3} Int64, voxels of a block in
 * x-fastest order. block_keys_dev: the hash map's {capacity, 3} key buffer
 * (o3dmi_hash_key_buffer). Either output may be NULL. The linear index is
 * formed in 64 bits (upstream: int, overflowing past 2^31 voxels; equal below
 * that). Asynchronous. */
int o3dmi_vbg_voxel_coordinates_and_flattened_indices(
        const int32_t* buf_indices_dev, int64_t n_blocks,
        const int32_t* block_keys_dev, int resolution, float voxel_size,
        float* voxel_coords_dev, int64_t* flattened_indices_dev,
        o3dmi_stream_t stream);
/* VoxelBlockGrid::GetVoxelIndices (t/geometry/VoxelBlockGrid.cpp:145-178):
 * {4, n_blocks * res^3} Int64 -- row 0 the buffer index of the voxel's block,
 * rows 1-3 its x, y, z inside the block. Asynchronous. */
int o3dmi_vbg_voxel_indices(const int32_t* buf_indices_dev, int64_t n_blocks,
                            int resolution, int64_t* voxel_indices_dev,
                            o3dmi_stream_t stream);
/* VoxelBlockGrid::GetVoxelCoordinates (VoxelBlockGrid.cpp:130-143): {4, n}
 * voxel indices -> {3, n} Int64 voxel coordinates key * res + (x, y, z). A
 * buffer index outside [0, key_capacity) sets bit 0 of *err_dev (optional)
 * and yields zeros (upstream's IndexGet throws). Asynchronous. */
int o3dmi_vbg_voxel_coordinates(const int64_t* voxel_indices_dev, int64_t n,
                                const int32_t* block_keys_dev,
                                int64_t key_capacity, int resolution,
                                int64_t* voxel_coords_dev, int32_t* err_dev,
                                o3dmi_stream_t stream);

/* IntegrateCUDA<input_depth_t,input_color_t,tsdf_t,weight_t,color_t>
 * (t/geometry/kernel/VoxelBlockGridImpl.h:151-308). input_dtype: O3DMI_U16
 * (u16 depth + u8 colour) or O3DMI_F32 (f32 depth + f32 colour in [0,1]);
 * grid_dtype: O3DMI_U16 (f32 tsdf, u16 weight, u16 colour) or O3DMI_F32.
 * color_dev / color_buf_dev may be NULL (depth-only). `n_indices_dev`
 * optional device-side count as in o3dmi_hash_activate. Asynchronous. */
int o3dmi_vbg_integrate(const void* depth_dev, int depth_rows, int depth_cols,
                        const void* color_dev, int color_rows, int color_cols,
                        int input_dtype, const int32_t* indices_dev,
                        int64_t n_indices, const int32_t* n_indices_dev,
                        const int32_t* block_keys_dev, float* tsdf_dev,
                        void* weight_dev, void* color_buf_dev, int grid_dtype,
                        const double* depth_intrinsic,
                        const double* color_intrinsic, const double* extrinsic,
                        int resolution, float voxel_size, float sdf_trunc,
                        float depth_scale, float depth_max,
                        o3dmi_stream_t stream);

/* Fused per-frame front end used by the frame-stream fast path: DepthTouch
 * candidates are activated directly in `block_hash` and the frame's unique
 * buffer indices are emitted (device-resident, no host round trip). Result
 * (set of activated keys, set of per-frame indices) is identical to
 * DepthTouch -> Activate -> Find. `frame_stamp` must increase by one per call
 * on a given hash. */
int o3dmi_vbg_touch_activate(o3dmi_hash_t* block_hash, const void* depth_dev,
                             int depth_dtype, int rows, int cols,
                             const double* intrinsic, const double* extrinsic,
                             int32_t* out_buf_indices_dev,
                             int64_t out_capacity, int32_t* out_count_dev,
                             int resolution, float voxel_size, float sdf_trunc,
                             float depth_scale, float depth_max, int stride,
                             int32_t frame_stamp, o3dmi_stream_t stream);

/* EstimateRangeCUDA (VoxelBlockGridImpl.h:310-555): range_minmax_map_dev is
 * {h/down, w/down, 2} float32 (min,max). No fragment buffer is needed; the
 * result equals the reference's when its fragment buffer does not overflow. */
int o3dmi_vbg_estimate_range(const int32_t* block_keys_dev, int64_t n_blocks,
                             float* range_minmax_map_dev,
                             const double* intrinsic, const double* extrinsic,
                             int h, int w, int down_factor,
                             int64_t block_resolution, float voxel_size,
                             float depth_min, float depth_max,
                             o3dmi_stream_t stream);

/* The same with the number of keys resident on the device (*n_blocks_dev <=
 * max_blocks): frame-stream callers never read the count back. */
int o3dmi_vbg_estimate_range_dev(const int32_t* block_keys_dev,
                                 int64_t max_blocks,
                                 const int32_t* n_blocks_dev,
                                 float* range_minmax_map_dev,
                                 const double* intrinsic,
                                 const double* extrinsic, int h, int w,
                                 int down_factor, int64_t block_resolution,
                                 float voxel_size, float depth_min,
                                 float depth_max, o3dmi_stream_t stream);

/* RayCastCUDA<tsdf_t,weight_t,color_t> (VoxelBlockGridImpl.h:578-1120).
 * Output maps may be NULL when not requested: depth {h,w,1}, vertex/color/
 * normal {h,w,3} float32; index {h,w,8} int64; mask {h,w,8} bool;
 * interp_ratio{,_dx,_dy,_dz} {h,w,8} float32. h and w must be multiples of
 * range_map_down_factor (O3DMI_ERR_INVALID_ARG otherwise: the {h / down,
 * w / down, 2} range map has no cell for the remainder -- upstream reads past
 * it). */
int o3dmi_vbg_raycast(o3dmi_hash_t* block_hash, const float* tsdf_dev,
                      const void* weight_dev, const void* color_buf_dev,
                      int grid_dtype, const float* range_map_dev,
                      float* out_depth, float* out_vertex, float* out_color,
                      float* out_normal, int64_t* out_index, uint8_t* out_mask,
                      float* out_ratio, float* out_ratio_dx,
                      float* out_ratio_dy, float* out_ratio_dz,
                      const double* intrinsic, const double* extrinsic, int h,
                      int w, int block_resolution, float voxel_size,
                      float depth_scale, float depth_min, float depth_max,
                      float weight_threshold, float trunc_voxel_multiplier,
                      int range_map_down_factor, o3dmi_stream_t stream);

/* The same launch over the pixel rows [row_begin, row_end) of the {h, w}
 * image only (whole 8-row tiles: row_begin % 8 == 0, row_end % 8 == 0 or
 * row_end == h); the output pointers are still those of the WHOLE maps, rows
 * outside the band are not written. A pixel's result does not depend on the
 * band it is rendered in: the bands of several ranks put together are
 * o3dmi_vbg_raycast's maps bit for bit (RayCast row: pixel-tile
 * sharding over a replicated grid; no counterpart in the reference, whose
 * RayCastCUDA renders the whole image on one device). */
int o3dmi_vbg_raycast_rows(
        o3dmi_hash_t* block_hash, const float* tsdf_dev, const void* weight_dev,
        const void* color_buf_dev, int grid_dtype, const float* range_map_dev,
        float* out_depth, float* out_vertex, float* out_color,
        float* out_normal, int64_t* out_index, uint8_t* out_mask,
        float* out_ratio, float* out_ratio_dx, float* out_ratio_dy,
        float* out_ratio_dz, const double* intrinsic, const double* extrinsic,
        int h, int w, int row_begin, int row_end, int block_resolution,
        float voxel_size, float depth_scale, float depth_min, float depth_max,
        float weight_threshold, float trunc_voxel_multiplier,
        int range_map_down_factor, o3dmi_stream_t stream);

/* ExtractPointCloudCUDA<tsdf_t,weight_t,color_t> (VoxelBlockGridImpl.h:
 * 1122-1365) fused with BufferRadiusNeighbors (t/geometry/VoxelBlockGrid.cpp:
 * 22-51): zero crossings of the TSDF along +x/+y/+z between voxels with
 * weight > weight_threshold. indices_dev = active buffer indices. The 27
 * neighbour blocks are looked up in `block_hash` inside the kernel.
 * capacity < 0: count only (the reference's valid_size < 0 pass);
 * otherwise at most `capacity` points are written (points / normals / colors
 * {capacity,3} float32; colors_dev and color_dev may be NULL). Output order is
 * (position in indices_dev, voxel, axis) -- deterministic, where the
 * reference's is its atomic counter's. *total_out = number of crossings
 * (synchronises). */
int o3dmi_vbg_extract_points(o3dmi_hash_t* block_hash,
                             const int32_t* indices_dev, int64_t n_blocks,
                             const float* tsdf_dev, const void* weight_dev,
                             const void* color_dev, int grid_dtype,
                             int resolution, float voxel_size,
                             float weight_threshold, float* points_dev,
                             float* normals_dev, float* colors_dev,
                             int64_t capacity, int64_t* total_out,
                             o3dmi_stream_t stream);
/* Ascending in-place sort of int32 indices (synchronises). */
int o3dmi_sort_indices(int32_t* indices_dev, int64_t n, o3dmi_stream_t stream);

/* UnprojectCUDA (t/geometry/kernel/PointCloudImpl.h:42-143): strided depth ->
 * compacted world points. points_dev holds (rows/stride)*(cols/stride) x 3
 * floats; count to *out_count_dev. Order: the row-major scan order of the
 * strided pixels, the same on every run (upstream's order is its atomic
 * counter's arrival order, i.e. unspecified; this is one instance of it): one
 * launch, every workgroup publishes its chunk's total and reads the chunks
 * before it. Everything downstream of the cloud is then reproducible bit for
 * bit. */
int o3dmi_unproject(const void* depth_dev, int depth_dtype, int rows, int cols,
                    const float* image_colors_dev, float* points_dev,
                    float* colors_dev, int32_t* out_count_dev,
                    const double* intrinsic, const double* extrinsic,
                    float depth_scale, float depth_max, int64_t stride,
                    o3dmi_stream_t stream);
/* Two images of ONE size, intrinsic matrix, scale and stride in one launch --
 * the two clouds a frame-to-model tracking step starts from: the model frame
 * a ray cast rendered (depth F32 + its normal map as image_colors) and the
 * camera's new frame (U16). Each cloud is, to the bit and in the same order,
 * what o3dmi_unproject gives for its arguments (extrinsic_a / extrinsic_b
 * may differ); the outputs must not alias. */
int o3dmi_unproject_pair(
        const void* depth_a_dev, int depth_a_dtype,
        const float* image_colors_a_dev, float* points_a_dev,
        float* colors_a_dev, int32_t* out_count_a_dev,
        const double* extrinsic_a, const void* depth_b_dev, int depth_b_dtype,
        const float* image_colors_b_dev, float* points_b_dev,
        float* colors_b_dev, int32_t* out_count_b_dev,
        const double* extrinsic_b, int rows, int cols, const double* intrinsic,
        float depth_scale, float depth_max, int64_t stride,
        o3dmi_stream_t stream);

/* ------------------------------------------------------------------------ */
/* ICP kernels.                                                              */
/* ------------------------------------------------------------------------ */

/* Fixed-radius index (core/nns/FixedRadiusIndex.h:227-233,364-377;
 * BuildSpatialHashTableCUDA / HybridSearchCUDA, FixedRadiusSearchOps.cu:
 * 20-57). Results follow the CPU path's nanoflann semantics
 * (core/nns/NanoFlannImpl.h:305-370): neighbours with d2 < r2 (strict),
 * nearest first, ties by lower index. dtype O3DMI_F32 or O3DMI_F64.
 * o3dmi_nns_create is stream-ordered: it queues the build (a fill and three
 * launches) on `stream` and returns without waiting, so points_dev must stay
 * valid until that work has run (any later call on the same stream is
 * ordered behind it). At most 2^27 - 1 points per index (records are
 * addressed by 32-bit byte offsets). */
typedef struct o3dmi_nns o3dmi_nns_t;
int o3dmi_nns_create(const void* points_dev, int64_t n, int dtype,
                     double radius, o3dmi_stream_t stream, o3dmi_nns_t** out);
int o3dmi_nns_destroy(o3dmi_nns_t* nns);
/* HybridSearch(max_knn = 1): idx {Q} int32 (-1 when none), dist2 {Q} in the
 * point dtype (0 when none), counts {Q} int32. */
int o3dmi_nns_hybrid_search_k1(const o3dmi_nns_t* nns, const void* queries_dev,
                               int64_t q, int32_t* idx_dev, void* dist2_dev,
                               int32_t* counts_dev, o3dmi_stream_t stream);

/* HybridSearchCUDA<T,TIndex> for general max_knn <= 64 (same nanoflann
 * semantics as the k = 1 form): idx {Q,max_knn} int32 ascending by (d2,
 * index), -1 padded; dist2 {Q,max_knn} in the point dtype, 0 padded (may be
 * NULL); counts {Q} = min(found, max_knn). */
int o3dmi_nns_hybrid_search(const o3dmi_nns_t* nns, const void* queries_dev,
                            int64_t q, int max_knn, int32_t* idx_dev,
                            void* dist2_dev, int32_t* counts_dev,
                            o3dmi_stream_t stream);

/* NearestNeighborSearch::KnnIndex + KnnSearch (core/nns/
 * NearestNeighborSearch.cpp:28-45,93-112; GPU form KnnSearchCUDA, core/nns/
 * KnnIndex.h): the k = min(knn, n) nearest dataset points of every query,
 * ascending by (d2, index); idx {Q,k} int32, dist2 {Q,k} in the point dtype
 * (may be NULL). k <= 64. Builds its own grid over the dataset (cell size from
 * the measured point density) and synchronises. */
int o3dmi_nns_knn_search(const void* points_dev, int64_t n,
                         const void* queries_dev, int64_t q, int dtype, int knn,
                         int32_t* idx_dev, void* dist2_dev,
                         o3dmi_stream_t stream);

/* FixedRadiusSearchCUDA (core/nns/FixedRadiusIndex.h:281-290, FixedRadius
 * SearchImpl.cuh:826-1072) in its two passes, the prefix sum in between left
 * to the caller as in the reference (neighbors_row_splits): counts {Q} int32
 * of the index points with d2 < r2 (r = the index radius); then, given
 * row_splits {Q+1} int64 (exclusive prefix of the counts), the neighbour
 * indices {total} int32 and squared distances {total} (point dtype, may be
 * NULL) of every query, ascending by (d2, index). No cap on the number of
 * neighbours of a query. */
int o3dmi_nns_radius_count(const o3dmi_nns_t* nns, const void* queries_dev,
                           int64_t q, int32_t* counts_dev,
                           o3dmi_stream_t stream);
int o3dmi_nns_radius_search(const o3dmi_nns_t* nns, const void* queries_dev,
                            int64_t q, const int64_t* row_splits_dev,
                            int32_t* idx_dev, void* dist2_dev,
                            o3dmi_stream_t stream);

/* EstimateCovariancesUsingRadiusSearchCUDA (t/geometry/kernel/PointCloudImpl.h:
 * 641-689): covariances {Q,3,3} (point dtype) of ALL index points with
 * d2 < r2 of every query, r = the index radius; fewer than 3 -> identity.
 * The moments are summed by a wave in float64 (parallel order: the last bits
 * of the float64 sums may differ from the reference's one-by-one order). */
int o3dmi_nns_radius_covariances(const o3dmi_nns_t* nns, const void* queries_dev,
                                 int64_t q, void* covariances_dev,
                                 o3dmi_stream_t stream);

/* EstimateCovariancesUsingHybridSearchCUDA after the search
 * (t/geometry/kernel/PointCloudImpl.h:588-638; per-point body :512-585):
 * covariances {n,3,3} in the point dtype from hybrid-search results. */
int o3dmi_pointcloud_estimate_covariances(const void* points_dev,
                                          const int32_t* indices_dev,
                                          const int32_t* counts_dev, int64_t n,
                                          int max_nn, int dtype,
                                          void* covariances_dev,
                                          o3dmi_stream_t stream);
/* EstimateNormalsFromCovariancesCUDA (PointCloudImpl.h:1011-1063). normals
 * {n,3} is in/out when has_normals (orientation is kept consistent with the
 * existing normals). The eigenvector of the smallest eigenvalue comes from
 * this library's own routine (covariance widened to float64, converged cyclic
 * Jacobi), NOT from the reference's closed-form solver (:746-1009): within
 * 1e-4 rad (Float32) / 1e-10 (Float64) of it wherever the two smallest
 * eigenvalues differ by more than 5 % of the largest, any unit vector of the
 * eigenspace otherwise. Sign without prior normals: last non-zero component
 * positive; an all-zero covariance gives +z (with prior normals: the zero
 * vector), an identity covariance (< 3 neighbours) +z. */
int o3dmi_pointcloud_normals_from_covariances(const void* covariances_dev,
                                              int64_t n, int dtype,
                                              void* normals_dev,
                                              int has_normals,
                                              o3dmi_stream_t stream);

/* ComputePosePointToPlaneCUDA up to the reduction
 * (t/pipelines/kernel/RegistrationCUDA.cu:29-117, RegistrationImpl.h:251-287):
 * the 29 sums (21 JtJ lower-triangular row-major, 6 Jtr, sum r, count) are
 * written to sums29_dev as float64. Per-correspondence terms are formed in the
 * point dtype exactly as the reference; accumulation is in float64 with a
 * fixed reduction tree (run-to-run deterministic). corr is int64, -1 = none. */
int o3dmi_icp_p2plane_accumulate(const void* src_dev, const void* tgt_dev,
                                 const void* tgt_normals_dev,
                                 const int64_t* corr_dev, int64_t n, int dtype,
                                 int robust_kernel, double scaling_parameter,
                                 double shape_parameter, double* sums29_dev,
                                 o3dmi_stream_t stream);

/* One fused ICP iteration front end: search (k=1) + fitness/rmse sums +
 * point-to-plane accumulation. sums32_dev (float64[32]): [0..28] as above,
 * [29] = sum of d2 over matches (accumulated in float64), [30] = number of
 * matches, [31] unused. corr_out_dev may be NULL. */
int o3dmi_icp_search_accumulate(const o3dmi_nns_t* nns, const void* src_dev,
                                const void* tgt_normals_dev, int64_t n,
                                int robust_kernel, double scaling_parameter,
                                double shape_parameter, int64_t* corr_out_dev,
                                double* sums32_dev, o3dmi_stream_t stream);

/* TransformationEstimationPointToPoint: the reduction of ComputeRtPointToPoint
 * (t/pipelines/kernel/Registration.cpp:365-404; CPU body Get3x3SxyLinearSystem,
 * RegistrationCPU.cpp:495-617; the reference's own GPU branch is tensor ops
 * with a "TODO: Implement optimized CUDA reduction kernel", Registration.cpp:
 * 383). One pass instead of the reference's two: sums16_dev (float64[16]) =
 * [0..2] sum of source xyz, [3..5] sum of matched target xyz, [6 + 3 j + k] =
 * sum of target_j * source_k, [15] = number of correspondences; products and
 * sums are float64 (exact products for Float32 clouds), fixed reduction tree.
 * o3dmi_compute_rt_p2point turns them into R, t. */
int o3dmi_icp_p2point_accumulate(const void* src_dev, const void* tgt_dev,
                                 const int64_t* corr_dev, int64_t n, int dtype,
                                 double* sums16_dev, o3dmi_stream_t stream);

/* Fused search (k=1) + fitness/rmse sums + point-to-point accumulation:
 * sums32_dev[0..15] as o3dmi_icp_p2point_accumulate, [29] = sum of d2 over
 * matches, [30] = number of matches. The index needs no normals. */
int o3dmi_icp_search_accumulate_p2point(const o3dmi_nns_t* nns,
                                        const void* src_dev, int64_t n,
                                        int64_t* corr_out_dev,
                                        double* sums32_dev,
                                        o3dmi_stream_t stream);

/* Host: mean_s, mean_t, Sxy = sum(t s^T)/n - mean_t mean_s^T, SVD of Sxy
 * (one-sided Jacobi, float64), R = U diag(1,1,det(U)det(V)) V^T,
 * t = mean_t - R mean_s (RegistrationCPU.cpp:640-650). R9 row-major.
 * O3DMI_ERR_NO_INLIERS ("No valid correspondence present.") when count = 0. */
int o3dmi_compute_rt_p2point(const double* sums16_host, double* R9, double* t3);

/* ComputePoseSymmetricCUDA up to the reduction (TransformationEstimation
 * Symmetric; t/pipelines/kernel/RegistrationCUDA.cu, CPU body RegistrationCPU.
 * cpp:124-218, Jacobian RegistrationImpl.h:323-386): 29 sums as for point-to-
 * plane except [27] = sum of squared un-centred residuals. source / target
 * means of the matched points (host float64, rounded to the point dtype) are
 * what ComputeTransformationSymmetric passes (kernel/Registration.cpp:96-97);
 * the moments of o3dmi_icp_p2point_accumulate / the fused p2point search give
 * them. o3dmi_symmetric_pose_to_transformation is PoseToSymmetricTransformation
 * (TransformationConverter.cpp:106-133). */
int o3dmi_icp_symmetric_accumulate(
        const void* src_dev, const void* src_normals_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const int64_t* corr_dev, int64_t n,
        int dtype, const double* source_mean3, const double* target_mean3,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        double* sums29_dev, o3dmi_stream_t stream);
void o3dmi_symmetric_pose_to_transformation(const double* pose6,
                                            const double* source_mean3,
                                            const double* target_mean3,
                                            double* T16);

/* ComputePoseColoredICPCUDA up to the reduction (TransformationEstimationFor
 * ColoredICP; CPU body RegistrationCPU.cpp:220-340, Jacobians RegistrationImpl.
 * h:388-466): 29 sums, [27] = sum r_G^2 + r_I^2. Colours {N,3} in the point
 * dtype, target_color_gradients {Nt,3} from EstimateColorGradients. */
int o3dmi_icp_colored_accumulate(
        const void* src_dev, const void* src_colors_dev, const void* tgt_dev,
        const void* tgt_normals_dev, const void* tgt_colors_dev,
        const void* tgt_color_gradients_dev, const int64_t* corr_dev, int64_t n,
        int dtype, double lambda_geometric, int robust_kernel,
        double scaling_parameter, double shape_parameter, double* sums29_dev,
        o3dmi_stream_t stream);

/* EstimateColorGradientsUsing{Hybrid,KNN}SearchCUDA after the search
 * (t/geometry/kernel/PointCloudImpl.h:1067-1290): per point, least squares of
 * the intensity over its neighbours projected on the tangent plane plus the
 * constraint gradient . normal = 0. The 3x3 normal equations go through the
 * reference's approximate solve_svd3x3 (core/linalg/kernel/SVD3x3.h) restated
 * bit for bit (csrc/approx_svd3.h) -- including its Float64 quirks, which yield NaN
 * on some neighbourhoods; O3DMI_EXACT_COLOR_GRADIENTS=1 selects an exact
 * solve instead. */
int o3dmi_pointcloud_color_gradients_from_neighbors(
        const void* points_dev, const void* normals_dev, const void* colors_dev,
        const int32_t* indices_dev, const int32_t* counts_dev, int64_t n,
        int max_nn, int dtype, void* gradients_dev, o3dmi_stream_t stream);

/* ComputeInformationMatrixCUDA up to the reduction
 * (t/pipelines/kernel/RegistrationCUDA.cu ComputeInformationMatrixKernelCUDA,
 * CPU body RegistrationCPU.cpp:652-735, Jacobians RegistrationImpl.h:686-715):
 * the 21 packed-lower-triangle sums of G^T G over matched target points,
 * float64; sums21[j (j + 1) / 2 + k] = GTG[j][k] = GTG[k][j]. */
int o3dmi_icp_information_accumulate(const void* tgt_dev,
                                     const int64_t* corr_dev, int64_t n,
                                     int dtype, double* sums21_dev,
                                     o3dmi_stream_t stream);

/* TransformPointsCUDA / TransformNormalsCUDA
 * (t/geometry/kernel/Transform.h:42-47, TransformImpl.h:19-60): in place;
 * `transformation` is host float64 4x4, cast to the point dtype first. */
int o3dmi_transform_points(const double* transformation, void* points_dev,
                           int64_t n, int dtype, o3dmi_stream_t stream);
int o3dmi_transform_normals(const double* transformation, void* normals_dev,
                            int64_t n, int dtype, o3dmi_stream_t stream);

/* Host-side helpers with the reference's arithmetic
 * (t/pipelines/kernel/TransformationConverter.cpp:81-104,189-226). */
int o3dmi_decode_and_solve6x6(const double* sums29_host, double* pose6,
                              float* residual, int* inlier_count);
void o3dmi_pose_to_transformation(const double* pose6, double* T16);

/* ------------------------------------------------------------------------ */
/* RGB-D odometry front end                        */
/* ------------------------------------------------------------------------ */
/* odometry::Method, t/pipelines/odometry/RGBDOdometry.h:23-27 */
typedef enum {
    O3DMI_ODOMETRY_POINT_TO_PLANE = 0,
    O3DMI_ODOMETRY_INTENSITY = 1,
    O3DMI_ODOMETRY_HYBRID = 2
} o3dmi_odometry_method_t;

/* Image kernels: all images are contiguous {rows, cols[, C]} device buffers.
 *
 * ClipTransformCUDA (t/geometry/kernel/Image.h:88-93, ImageImpl.h:94-128):
 * out = in / scale; out <= min -> fill; out >= max -> fill. src U16 or F32. */
int o3dmi_image_clip_transform(const void* src_dev, int src_dtype, int rows,
                               int cols, float scale, float min_value,
                               float max_value, float clip_fill, float* dst_dev,
                               o3dmi_stream_t stream);
/* PyrDownDepthCUDA (Image.h:95-98, ImageImpl.h:132-206): dst {rows/2, cols/2};
 * 5x5 Gaussian over the neighbours within depth_diff of the centre. */
int o3dmi_image_pyrdown_depth(const float* src_dev, int rows, int cols,
                              float depth_diff, float invalid_fill,
                              float* dst_dev, o3dmi_stream_t stream);
/* CreateVertexMapCUDA (Image.h:100-103, ImageImpl.h:208-256): dst {rows,cols,3};
 * `intrinsics` host 3x3 float64. */
int o3dmi_image_create_vertex_map(const float* src_dev, int rows, int cols,
                                  const double* intrinsics, float invalid_fill,
                                  float* dst_dev, o3dmi_stream_t stream);
/* CreateNormalMapCUDA (Image.h:105-107, ImageImpl.h:257-322): src and dst
 * {rows,cols,3}. */
int o3dmi_image_create_normal_map(const float* src_dev, int rows, int cols,
                                  float invalid_fill, float* dst_dev,
                                  o3dmi_stream_t stream);
/* ToCUDA (Image.h:83-86, ImageImpl.h:35-85) for U8 / U16 / F32 -> F32:
 * out = in * scale + offset, saturated to the float limits as upstream. */
int o3dmi_image_to_float(const void* src_dev, int src_dtype, int64_t n,
                         double scale, double offset, float* dst_dev,
                         o3dmi_stream_t stream);
/* npp::RGBToGray (NPPImage.h:17; arithmetic of the in-tree tensor-op branch
 * t/geometry/Image.cpp:149-161): {n,3} -> {n}, same dtype (U8 / U16 / F32). */
int o3dmi_image_rgb_to_gray(const void* src_dev, int dtype, int64_t n_pixels,
                            void* dst_dev, o3dmi_stream_t stream);
/* RGBToGray().To(Float32) as RGBDOdometry.cpp:223-224 applies it (U8 colour is
 * scaled by 1/255, F32 colour is not), in one pass. */
int o3dmi_image_rgb_to_intensity(const void* src_dev, int dtype,
                                 int64_t n_pixels, float* dst_dev,
                                 o3dmi_stream_t stream);
/* npp::FilterBilateral / FilterGaussian / FilterSobel / Resize (NPPImage.h:
 * 27-48). The reference has no in-tree arithmetic for these (IPP / NPP);
 * semantics: replicate border; bilateral over the circular neighbourhood of
 * radius kernel_size/2 with w = exp(-dv^2/(2 sv^2)) exp(-d^2/(2 sd^2));
 * Gaussian = outer product of normalised exp(-d^2/(2 s^2)) taps; Sobel 3x3
 * unnormalised; Resize = nearest, factor 0.5. dst must not alias src. */
int o3dmi_image_filter_bilateral(const float* src_dev, int rows, int cols,
                                 int kernel_size, float value_sigma,
                                 float distance_sigma, float* dst_dev,
                                 o3dmi_stream_t stream);
int o3dmi_image_filter_gaussian(const float* src_dev, int rows, int cols,
                                int kernel_size, float sigma, float* dst_dev,
                                o3dmi_stream_t stream);
int o3dmi_image_filter_sobel(const float* src_dev, int rows, int cols,
                             float* dx_dev, float* dy_dev,
                             o3dmi_stream_t stream);
int o3dmi_image_resize_half_nearest(const float* src_dev, int rows, int cols,
                                    float* dst_dev, o3dmi_stream_t stream);
/* Image::PyrDown (t/geometry/Image.cpp:404-407) = FilterGaussian(5, 1) +
 * Resize(0.5, Nearest), evaluated at the kept pixels only. */
int o3dmi_image_pyrdown(const float* src_dev, int rows, int cols,
                        float* dst_dev, o3dmi_stream_t stream);

/* One pyramid level of the point-to-plane method in a single pass
 * (RGBDOdometry.cpp:124-153): source / target vertex maps and the target
 * normal map of the bilateral-smoothed (5, 5, 10) target depth; NaN = invalid.
 * When source_depth_next_dev / target_depth_next_dev are given ({rows/2,
 * cols/2}), the same launch also writes PyrDownDepth(depth_diff, NaN) of both
 * depth images for the next coarser level. Output identical to the
 * CreateVertexMap / FilterBilateral / CreateNormalMap / PyrDownDepth sequence
 * above. */
int o3dmi_odometry_p2plane_level(const float* source_depth_dev,
                                 const float* target_depth_dev, int rows,
                                 int cols, const double* intrinsics,
                                 float* source_vertex_dev,
                                 float* target_vertex_dev,
                                 float* target_normal_dev,
                                 float* source_depth_next_dev,
                                 float* target_depth_next_dev,
                                 float depth_diff, o3dmi_stream_t stream);
/* ClipTransform of two images (source and target depth, dtypes independent)
 * in one launch. */
int o3dmi_image_clip_transform_pair(const void* src0_dev, int src0_dtype,
                                    const void* src1_dev, int src1_dtype,
                                    int rows, int cols, float scale,
                                    float min_value, float max_value,
                                    float clip_fill, float* dst0_dev,
                                    float* dst1_dev, o3dmi_stream_t stream);

/* ComputeOdometryResult{PointToPlane,Intensity,Hybrid}CUDA up to the reduction
 * (t/pipelines/kernel/RGBDOdometryImpl.h:74-118; per-pixel terms
 * RGBDOdometryJacobianImpl.h:106-336, RGBDOdometryCPU.cpp:98-364): the 29
 * sums (21 JtJ lower-triangular, 6 Jt*huber'(r), sum huber(r), count) as
 * float64 in sums29_dev; per-pixel terms are float32 exactly as the
 * reference, running sums float64 with a fixed tree. Maps a method does not
 * read may be NULL. scratch_dev: NULL or o3dmi_odometry_sums_scratch_doubles()
 * float64 (NULL allocates internally and synchronises). Solve on the host
 * with o3dmi_decode_and_solve6x6 / o3dmi_pose_to_transformation. */
int o3dmi_odometry_sums_scratch_doubles(void);
int o3dmi_odometry_sums(int method, int rows, int cols,
                        const float* source_depth_dev,
                        const float* target_depth_dev,
                        const float* source_intensity_dev,
                        const float* target_intensity_dev,
                        const float* target_depth_dx_dev,
                        const float* target_depth_dy_dev,
                        const float* target_intensity_dx_dev,
                        const float* target_intensity_dy_dev,
                        const float* source_vertex_dev,
                        const float* target_vertex_dev,
                        const float* target_normal_dev,
                        const double* intrinsics,
                        const double* init_source_to_target,
                        float depth_outlier_trunc, float depth_huber_delta,
                        float intensity_huber_delta, double* scratch_dev,
                        double* sums29_dev, o3dmi_stream_t stream);
/* ComputeOdometryInformationMatrixCUDA (RGBDOdometryImpl.h:120-125,
 * RGBDOdometryCPU.cpp:26-96): information_host = 6x6 float64 on the host
 * (synchronises). */
int o3dmi_odometry_information(int rows, int cols,
                               const float* source_vertex_dev,
                               const float* target_vertex_dev,
                               const double* intrinsics,
                               const double* source_to_target,
                               float square_dist_thr, double* information_host,
                               o3dmi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* O3D_MI355X_H_ */
