/* o3d_mi355x_host.h -- host-side mirror (C++ implementation, C linkage) of the
 * Open3D classes that drive the hot path, built on top of the kernel C ABI in
 * o3d_mi355x.h. These are what a language binding (pybind / ctypes / cgo)
 * calls when it wants the whole operator rather than one kernel:
 *
 *   o3dmi_registration_multiscale_icp  <- t::pipelines::registration::MultiScaleICP / ICP
 *   (_ex: estimator choice)               (t/pipelines/registration/Registration.cpp:93-106,362-444)
 *   o3dmi_registration_evaluate,       <- EvaluateRegistration, GetInformationMatrix
 *   o3dmi_registration_information_matrix (Registration.cpp:64-91,446-486)
 *   o3dmi_voxel_down_sample,           <- t::geometry::PointCloud::{VoxelDownSample, EstimateNormals,
 *   o3dmi_pointcloud_estimate_*           EstimateColorGradients} (t/geometry/PointCloud.cpp:496-567,856-1060)
 *   o3dmi_vbg_*                        <- t::geometry::VoxelBlockGrid (+ Save / Load)
 *                                         (t/geometry/VoxelBlockGrid.cpp:65-117,212-602)
 *   o3dmi_rgbd_odometry_multiscale     <- t::pipelines::odometry::RGBDOdometryMultiScale
 *                                         (t/pipelines/odometry/RGBDOdometry.cpp:56-513)
 *   o3dmi_slam_model_*                 <- t::pipelines::slam::Model (t/pipelines/slam/Model.cpp:23-118)
 *   o3dmi_npz_*                        <- t::io::WriteNpz / ReadNpz (t/io/NumpyIO.cpp:157-789)
 *
 * Same argument meaning, defaults and error behaviour as the reference
 * (errors are status codes + o3dmi_last_error() instead of exceptions).
 */
#ifndef O3D_MI355X_HOST_H_
#define O3D_MI355X_HOST_H_

#include "o3d_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ICPConvergenceCriteria (t/pipelines/registration/Registration.h:31-58):
 * defaults relative_fitness 1e-6, relative_rmse 1e-6, max_iteration 30. */
typedef struct {
    double relative_fitness;
    double relative_rmse;
    int max_iteration;
} o3dmi_icp_criteria_t;

/* RegistrationResult (Registration.h:61-98). */
typedef struct {
    double transformation[16]; /* 4x4 float64, row-major, host            */
    double inlier_rmse;
    double fitness;
    int converged;
    int num_iterations;
    int64_t num_correspondences; /* rows written to correspondences_dev   */
} o3dmi_registration_result_t;

/* callback_after_iteration (Registration.cpp:330-345): iteration_index,
 * scale_index, scale_iteration_index, inlier_rmse, fitness, transformation. */
typedef void (*o3dmi_icp_callback_t)(int64_t iteration_index,
                                     int64_t scale_index,
                                     int64_t scale_iteration_index,
                                     double inlier_rmse, double fitness,
                                     const double* transformation, void* user);

/* Cross-GPU hook: in-place SUM over all ranks of `n` host doubles (the 29
 * Gauss-Newton sums, sum d2, match count and the local source-point count).
 * NULL = single GPU. Each rank holds a shard of the source cloud and the whole
 * target; every rank then solves the same 6x6 system. */
typedef int (*o3dmi_allreduce_sum_t)(double* host_buf, int n, void* user);

/* The same exchange on the DEVICE: the hook enqueues an in-place sum
 * all-reduce of dev_buf[0..n) (float64) over all ranks on `stream`
 * (asynchronously -- e.g. RCCL's ncclAllReduce, or torch.distributed's
 * all_reduce under that stream) and returns 0. When set for the calling host
 * thread it takes precedence over the host hook of the driver functions: the
 * per-iteration sums then stay on the device from the final reduction kernel
 * through the collective to the kernel that posts them to the host mailbox
 * (one host wait per iteration, no staging copies). fn = NULL restores the
 * host path. Thread-local: one rank per process, or one host thread per
 * device. */
typedef int (*o3dmi_allreduce_device_t)(double* dev_buf, int n,
                                        o3dmi_stream_t stream, void* user);
int o3dmi_set_device_allreduce(o3dmi_allreduce_device_t fn, void* user);

/* ---- Collectives owned by the library (multi-GPU) -------
 * One process (or host thread) per GPU. An o3dmi_comm_t carries the three
 * exchanges the sharded hot path needs -- sum all-reduce of float64 (the ICP
 * sums), all-gather of fixed-size records and all-to-all of byte ranges (block
 * IDs and voxel rows to their owners) -- all enqueued on the caller's stream.
 * Transports:
 *   RCCL    librccl.so resolved with dlopen at first use (the instance the
 *           process already holds, e.g. PyTorch's, else the system one;
 *           O3DMI_RCCL_LIB overrides): ncclAllReduce / ncclAllGather / grouped
 *           ncclSend + ncclRecv over xGMI. o3dmi_rccl_unique_id +
 *           o3dmi_comm_create_rccl wrap ncclGetUniqueId / ncclCommInitRank on
 *           the current device (the 128-byte id travels by the caller's own
 *           means: MPI, a file, torch.distributed); o3dmi_comm_adopt_rccl
 *           wraps an existing ncclComm_t (not destroyed with the wrapper).
 *   custom  a table of functions (tests over gloo; other runtimes). Each
 *           entry enqueues (or completes) the exchange for the given stream
 *           and returns 0.
 * o3dmi_set_comm(c) makes c the communicator of the calling host thread: the
 * registration drivers (every estimator) then sum their per-iteration values
 * through it ON THE DEVICE, between the final reduction kernel and the kernel that
 * posts them to the host mailbox (takes precedence over both hooks below;
 * NULL restores them). o3dmi_set_rccl_comm(ncclComm) = adopt + set in one
 * call (NULL clears). The reference has no counterpart. */
typedef struct o3dmi_comm o3dmi_comm_t;
typedef struct {
    int (*allreduce_sum_f64)(void* user, double* dev_buf, int64_t n,
                             o3dmi_stream_t stream);
    int (*allgather)(void* user, const void* send_dev, void* recv_dev,
                     int64_t bytes_per_rank, o3dmi_stream_t stream);
    int (*alltoallv)(void* user, const void* send_dev,
                     const int64_t* send_bytes, const int64_t* send_offsets,
                     void* recv_dev, const int64_t* recv_bytes,
                     const int64_t* recv_offsets, o3dmi_stream_t stream);
} o3dmi_transport_t;
int o3dmi_rccl_available(void);
int o3dmi_rccl_unique_id(void* id128 /* 128 bytes */);
int o3dmi_comm_create_rccl(const void* id128, int rank, int world,
                           o3dmi_comm_t** out);
int o3dmi_comm_adopt_rccl(void* nccl_comm, o3dmi_comm_t** out);
int o3dmi_comm_create_custom(const o3dmi_transport_t* table, void* user,
                             int rank, int world, o3dmi_comm_t** out);
int o3dmi_comm_destroy(o3dmi_comm_t* c);
int o3dmi_comm_rank(const o3dmi_comm_t* c);
int o3dmi_comm_world(const o3dmi_comm_t* c);
/* ncclCommCount of the RCCL communicator behind `c`, asked of RCCL at the
 * time of the call; 0 for a custom transport (or a NULL handle): lets a
 * caller record which transport carried its collectives and over how many
 * ranks (bench.py config.transport / config.rccl_ranks). */
int o3dmi_comm_rccl_ranks(const o3dmi_comm_t* c);
int o3dmi_set_comm(o3dmi_comm_t* c);
int o3dmi_set_rccl_comm(void* nccl_comm);
/* Who shards the source cloud of an ICP call made with a communicator
 * installed (thread-local, default 0):
 *   0  the caller: each rank passes ITS shard of the source (the semantics of
 *      the two hooks). A voxel pyramid is then built per shard, i.e. the
 *      coarse levels differ from the unsharded run's.
 *   1  the driver: every rank passes the WHOLE source; the pyramid is built
 *      from it on every rank (it IS the unsharded pyramid), and each rank
 *      searches and accumulates a contiguous slice of every level. The poses
 *      equal the unsharded run's to the rounding of the float64 sums for any
 *      number of scales. correspondences_dev: a rank fills its slice of the
 *      finest level's rows, the other rows read -1. */
int o3dmi_set_icp_level_sharding(int on);
/* The exchanges themselves (what the drivers call). Counts / offsets of the
 * all-to-all are host arrays of `world` entries, in bytes. */
int o3dmi_comm_allreduce_sum_f64(o3dmi_comm_t* c, double* dev_buf, int64_t n,
                                 o3dmi_stream_t stream);
int o3dmi_comm_allgather(o3dmi_comm_t* c, const void* send_dev, void* recv_dev,
                         int64_t bytes_per_rank, o3dmi_stream_t stream);
int o3dmi_comm_alltoallv(o3dmi_comm_t* c, const void* send_dev,
                         const int64_t* send_bytes,
                         const int64_t* send_offsets, void* recv_dev,
                         const int64_t* recv_bytes,
                         const int64_t* recv_offsets, o3dmi_stream_t stream);

/* MultiScaleICP with TransformationEstimationPointToPlane(kernel).
 * source/target/normals: device, {N,3}, dtype O3DMI_F32 or O3DMI_F64.
 * voxel_sizes[i] <= 0 means "no down-sampling" for the finest level, as in
 * the reference (Registration.cpp:233-236).
 * correspondences_dev: optional device int64 buffer of ns entries receiving
 * the final correspondence set of the finest scale (-1 = none).
 * Status O3DMI_ERR_SINGULAR mirrors the reference's "Singular 6x6 linear
 * system detected, tracking failed." exception. */
int o3dmi_registration_multiscale_icp(
        const void* source_dev, int64_t ns, const void* target_dev,
        const void* target_normals_dev, int64_t nt, int dtype, int num_scales,
        const double* voxel_sizes, const o3dmi_icp_criteria_t* criterias,
        const double* max_correspondence_distances,
        const double* init_source_to_target /* 4x4, may be NULL = identity */,
        int robust_kernel, double scaling_parameter, double shape_parameter,
        o3dmi_icp_callback_t callback, void* callback_user,
        o3dmi_allreduce_sum_t allreduce, void* allreduce_user,
        int64_t* correspondences_dev, o3dmi_registration_result_t* result,
        o3dmi_stream_t stream);

/* Sizes that are still on the device. A tracking loop produces its clouds
 * with o3dmi_unproject, which leaves the point counts in device words; reading
 * them back costs the loop a stream drain per frame. After this call the NEXT
 * o3dmi_registration_multiscale_icp[_ex] of the calling host thread takes the
 * live source / target sizes from ns_dev / nt_dev (int32, written by work
 * queued earlier on the call's stream; NULL = the host argument as usual) and
 * reads its ns / nt arguments as the capacities of the buffers. With a
 * down-sampled finest level (voxel_sizes[last] > 0, the tracking
 * configuration) nothing waits for them: the pyramid launches bound themselves
 * by the device words. Without one the driver fetches them first. A live size
 * of zero is then reported through the usual "0 correspondence" result. The
 * reference has no counterpart (its Tensor shapes live on the host). */
int o3dmi_registration_set_device_counts(const int32_t* ns_dev,
                                         const int32_t* nt_dev);

/* TransformationEstimation choice for o3dmi_registration_multiscale_icp_ex
 * (t/pipelines/registration/TransformationEstimation.h:28-34). */
typedef enum {
    O3DMI_ICP_POINT_TO_PLANE = 0,
    O3DMI_ICP_POINT_TO_POINT = 1,
    O3DMI_ICP_SYMMETRIC = 2,
    O3DMI_ICP_COLORED = 3
} o3dmi_icp_estimation_t;

/* Point attributes beyond positions / target normals that some estimators
 * read (device, {N,3}, point dtype). Unused members may be NULL. */
typedef struct {
    const void* source_normals;         /* SYMMETRIC                          */
    const void* source_colors;          /* COLORED                            */
    const void* target_colors;          /* COLORED                            */
    const void* target_color_gradients; /* COLORED, optional: estimated on the
                                           finest level when NULL, as
                                           Registration.cpp:243-262 does     */
    double lambda_geometric;            /* COLORED; outside [0,1] -> 0.968    */
} o3dmi_icp_attributes_t;

/* MultiScaleICP with a selectable estimator. POINT_TO_PLANE is exactly
 * o3dmi_registration_multiscale_icp; POINT_TO_POINT
 * (TransformationEstimationPointToPoint, TransformationEstimation.cpp:101-160)
 * ignores the normals (may be NULL) and the robust kernel; SYMMETRIC
 * (TransformationEstimationSymmetric, :229-292) needs attrs->source_normals
 * and target_normals_dev -- the source normals are carried through the pyramid
 * and rotated with the source, as PointCloud::Transform does; COLORED
 * (TransformationEstimationForColoredICP, :296-440) needs target normals and
 * both colour sets, every attribute is averaged through the VoxelDownSample
 * pyramid, and missing colour gradients are estimated on the finest level
 * with EstimateColorGradients(30, 4 voxel_size or 2 max_distance).
 * attrs may be NULL for the first two estimators. */
int o3dmi_registration_multiscale_icp_ex(
        const void* source_dev, int64_t ns, const void* target_dev,
        const void* target_normals_dev, int64_t nt, int dtype, int num_scales,
        const double* voxel_sizes, const o3dmi_icp_criteria_t* criterias,
        const double* max_correspondence_distances,
        const double* init_source_to_target, int estimation,
        const o3dmi_icp_attributes_t* attrs, int robust_kernel,
        double scaling_parameter, double shape_parameter,
        o3dmi_icp_callback_t callback, void* callback_user,
        o3dmi_allreduce_sum_t allreduce, void* allreduce_user,
        int64_t* correspondences_dev, o3dmi_registration_result_t* result,
        o3dmi_stream_t stream);

/* TransformationEstimation*::ComputeRMSE (TransformationEstimation.cpp:
 * 101-130 point-to-point, 160-193 point-to-plane, 229-274 symmetric, 296-378
 * coloured) on given correspondences (int64, -1 = none). The reference's
 * definitions are kept as they are: point-to-plane squares every component
 * of (s - t) * n; the coloured estimator returns the SUM of squared geometric
 * and photometric residuals, not a root mean. *rmse_out is a host double; the
 * call synchronises. No correspondence: 0 for the symmetric estimator (as the
 * reference), O3DMI_ERR_NO_INLIERS otherwise (the reference divides by 0). */
int o3dmi_registration_compute_rmse(
        int estimation, const void* source_dev, int64_t ns,
        const void* target_dev, const void* target_normals_dev, int dtype,
        const o3dmi_icp_attributes_t* attrs, const int64_t* correspondences_dev,
        double* rmse_out, o3dmi_stream_t stream);

/* registration::EvaluateRegistration (Registration.cpp:64-91): fitness,
 * inlier_rmse and the correspondence set of `source` moved by `transformation`
 * (host 4x4 float64, NULL = identity) against `target`. result->transformation
 * is `transformation` (identity when nothing matches, Registration.cpp:56-58),
 * converged = 0, num_iterations = 0. correspondences_dev: optional int64[ns]. */
int o3dmi_registration_evaluate(const void* source_dev, int64_t ns,
                                const void* target_dev, int64_t nt, int dtype,
                                double max_correspondence_distance,
                                const double* transformation,
                                int64_t* correspondences_dev,
                                o3dmi_registration_result_t* result,
                                o3dmi_stream_t stream);

/* registration::GetInformationMatrix (Registration.cpp:446-486): 6x6 float64
 * (host, row-major). O3DMI_ERR_NO_INLIERS mirrors "0 correspondence present
 * between the pointclouds. Try increasing the max_correspondence_distance
 * parameter.". */
int o3dmi_registration_information_matrix(
        const void* source_dev, int64_t ns, const void* target_dev, int64_t nt,
        int dtype, double max_correspondence_distance,
        const double* transformation, double* information36,
        o3dmi_stream_t stream);

/* PointCloud::VoxelDownSample (t/geometry/PointCloud.cpp:496-567) for
 * positions (+ optional normals): mean per voxel in float32, voxel order =
 * order of first occurrence. Outputs must hold n rows; *m_out receives the
 * number of voxels (synchronises). */
int o3dmi_voxel_down_sample(const void* positions_dev, const void* normals_dev,
                            int64_t n, int dtype, double voxel_size,
                            void* out_positions_dev, void* out_normals_dev,
                            int64_t* m_out, o3dmi_stream_t stream);

/* PointCloud::EstimateNormals(max_nn, radius) (t/geometry/PointCloud.cpp:
 * 856-976): normals {n,3} (in/out when has_normals). radius > 0 and max_nn > 0
 * = hybrid search; radius <= 0 = KNN search (the reference's default
 * max_nn = 30, radius = nullopt); max_nn <= 0 = radius search (every
 * neighbour within radius); max_nn <= 64 otherwise. Synchronises.
 * Parity: neighbour sets and covariances are the reference's bit for bit;
 * the eigenvector is this library's own float64 routine (see
 * o3dmi_pointcloud_normals_from_covariances): a TOLERANCE against the
 * reference's closed form, with a pinned sign. */
int o3dmi_pointcloud_estimate_normals(const void* points_dev, int64_t n,
                                      int dtype, int max_nn, double radius,
                                      void* normals_dev, int has_normals,
                                      o3dmi_stream_t stream);

/* PointCloud::EstimateColorGradients(max_nn, radius) (t/geometry/PointCloud.
 * cpp:987-1060): hybrid search when both are given, KNN search when
 * radius <= 0, radius search (every neighbour within radius) when
 * max_nn <= 0; gradients {n,3} in the point dtype. max_nn <= 64 otherwise.
 * Synchronises.
 * Parity is a TOLERANCE, not bits: the reference solves each point's 3x3
 * normal equations with an approximate SVD (core/linalg/kernel/SVD3x3.h, four
 * fixed sweeps; its Float64 instantiation is broken), this library with a
 * converged pseudo-inverse. Against the reference's Float32 body: median
 * 1e-7 of the gradient scale, a percent-level tail on ill-conditioned
 * neighbourhoods (tests bound the 99th percentile at 0.15); against numpy's
 * pseudo-inverse: exact. The reference-arithmetic routine lives on as test
 * infrastructure only (oracle/approx_svd3_oracle.h). */
int o3dmi_pointcloud_estimate_color_gradients(
        const void* points_dev, const void* normals_dev, const void* colors_dev,
        int64_t n, int dtype, int max_nn, double radius, void* gradients_dev,
        o3dmi_stream_t stream);

/* ------------------------------------------------------------------------ */
/* VoxelBlockGrid                                                            */
/* ------------------------------------------------------------------------ */
typedef struct o3dmi_vbg o3dmi_vbg_t;

/* VoxelBlockGrid(attr_names, attr_dtypes, attr_channels, voxel_size,
 * block_resolution, block_count) (VoxelBlockGrid.cpp:65-117). Attribute i has
 * dtype attr_dtypes[i] (O3DMI_F32 / O3DMI_U16 / ...) and attr_channels[i]
 * channels; "tsdf" and "weight" are required by Integrate/RayCast. */
int o3dmi_vbg_create(int n_attrs, const char* const* attr_names,
                     const int* attr_dtypes, const int* attr_channels,
                     float voxel_size, int64_t block_resolution,
                     int64_t block_count, o3dmi_stream_t stream,
                     o3dmi_vbg_t** out);
/* VoxelBlockGrid::To(device, copy = true) (through HashMap::To,
 * core/hashmap/HashMap.cpp:230-255): the grid on HIP device `device` (may be
 * its own: a deep copy) with the same attributes, voxel size, block keys and
 * voxel values; buffer indices are the new grid's own. The caller's current
 * device must be the source grid's; it is restored. The new grid is used with
 * `device` current and streams of that device. */
int o3dmi_vbg_to_device(o3dmi_vbg_t* g, int device, o3dmi_vbg_t** out);
int o3dmi_vbg_destroy(o3dmi_vbg_t* g);
o3dmi_hash_t* o3dmi_vbg_hashmap(o3dmi_vbg_t* g);
/* GetAttribute(name): device pointer of the {capacity,res,res,res,C} buffer
 * (NULL + warning semantics: returns NULL when absent). */
void* o3dmi_vbg_attribute(o3dmi_vbg_t* g, const char* name, int* dtype,
                          int* channels);

/* Block-ownership sharding of a grid across `world` GPUs (one process per
 * GPU): this grid only activates and integrates the blocks it owns
 * (o3dmi_hash_set_ownership); applies to GetUniqueBlockCoordinates and to the
 * frame(s) paths. Explicit block lists given to o3dmi_vbg_integrate_blocks
 * are taken as they are. */
int o3dmi_vbg_set_block_ownership(o3dmi_vbg_t* g, int rank, int world);

/* GetUniqueBlockCoordinates(depth, intrinsic, extrinsic, depth_scale,
 * depth_max, trunc_voxel_multiplier) (VoxelBlockGrid.cpp:212-245).
 * out_coords_dev must hold (rows/4)*(cols/4)*4 rows; *m_out = number of
 * unique blocks (synchronises; O3DMI_ERR_NO_BLOCKS when zero). */
int o3dmi_vbg_get_unique_block_coordinates(
        o3dmi_vbg_t* g, const void* depth_dev, int depth_dtype, int rows,
        int cols, const double* intrinsic, const double* extrinsic,
        float depth_scale, float depth_max, float trunc_voxel_multiplier,
        int32_t* out_coords_dev, int64_t* m_out, o3dmi_stream_t stream);

/* GetUniqueBlockCoordinates(pcd, trunc_voxel_multiplier)
 * (VoxelBlockGrid.cpp:246-267): the blocks within +-voxel_size *
 * trunc_voxel_multiplier of each of the n points {n, 3} Float32. The frustum
 * map is (re)created for n * 8 entries, upstream's estimate; out_coords_dev
 * holds out_capacity rows (n * 8 always suffices); *m_out = number of unique
 * blocks (synchronises). */
int o3dmi_vbg_get_unique_block_coordinates_pcd(
        o3dmi_vbg_t* g, const float* points_dev, int64_t n,
        float trunc_voxel_multiplier, int32_t* out_coords_dev,
        int64_t out_capacity, int64_t* m_out, o3dmi_stream_t stream);

/* GetVoxelIndices(buf_indices) / GetVoxelCoordinates(voxel_indices) /
 * GetVoxelCoordinatesAndFlattenedIndices(buf_indices) of this grid
 * (VoxelBlockGrid.cpp:130-211): the kernel-seam functions of the same names
 * (o3d_mi355x.h) with the grid's key buffer, block resolution and voxel size.
 * The forms without an argument upstream take GetActiveIndices():
 * o3dmi_hash_active_indices(o3dmi_vbg_hashmap(g), ...). get_voxel_coordinates
 * synchronises (a buffer index outside the map is O3DMI_ERR_INVALID_ARG, as
 * upstream's IndexGet throws); the other two are asynchronous. */
int o3dmi_vbg_get_voxel_indices(o3dmi_vbg_t* g, const int32_t* buf_indices_dev,
                                int64_t n_blocks, int64_t* voxel_indices_dev,
                                o3dmi_stream_t stream);
int o3dmi_vbg_get_voxel_coordinates(o3dmi_vbg_t* g,
                                    const int64_t* voxel_indices_dev,
                                    int64_t n_voxels, int64_t* voxel_coords_dev,
                                    o3dmi_stream_t stream);
int o3dmi_vbg_get_voxel_coordinates_and_flattened_indices(
        o3dmi_vbg_t* g, const int32_t* buf_indices_dev, int64_t n_blocks,
        float* voxel_coords_dev, int64_t* flattened_indices_dev,
        o3dmi_stream_t stream);

/* Integrate(block_coords, depth, color, depth_intrinsic, color_intrinsic,
 * extrinsic, depth_scale, depth_max, trunc_voxel_multiplier)
 * (VoxelBlockGrid.cpp:292-326): Activate + Find + per-voxel update. May
 * Reserve (rehash) when size + m exceeds the capacity, as the reference. */
int o3dmi_vbg_integrate_blocks(o3dmi_vbg_t* g, const int32_t* block_coords_dev,
                               int64_t m, const void* depth_dev,
                               int depth_rows, int depth_cols,
                               const void* color_dev, int color_rows,
                               int color_cols, int input_dtype,
                               const double* depth_intrinsic,
                               const double* color_intrinsic,
                               const double* extrinsic, float depth_scale,
                               float depth_max, float trunc_voxel_multiplier,
                               o3dmi_stream_t stream);

/* Frame-stream fast path: GetUniqueBlockCoordinates + Integrate of one frame
 * with every count kept on the device (no host round trip unless the hash map
 * must grow). Results are identical to the two calls above. */
int o3dmi_vbg_integrate_frame(o3dmi_vbg_t* g, const void* depth_dev,
                              int depth_rows, int depth_cols,
                              const void* color_dev, int color_rows,
                              int color_cols, int input_dtype,
                              const double* depth_intrinsic,
                              const double* color_intrinsic,
                              const double* extrinsic, float depth_scale,
                              float depth_max, float trunc_voxel_multiplier,
                              o3dmi_stream_t stream);

/* The same for a batch of frames sharing intrinsics and image sizes:
 * depth_devs / color_devs are host arrays of n_frames device pointers,
 * extrinsics is n_frames x 16 host doubles. Frames are integrated strictly in
 * order (frame f sees the grid left by frame f-1), so the result is identical
 * to n_frames calls of o3dmi_vbg_integrate_frame.
 * frames_per_launch (1..16, <= 0 = 8): consecutive frames are grouped; one
 * launch applies the frames of a group, in order, to each touched block while
 * its voxel state stays in registers (state is read / written once per group
 * instead of once per frame; a block only receives the frames that touched
 * it), and the same launch already carries the touch / prepare work of the
 * next group. All work is issued on `stream`. */
int o3dmi_vbg_integrate_frames(o3dmi_vbg_t* g, int n_frames,
                               const void* const* depth_devs, int depth_rows,
                               int depth_cols, const void* const* color_devs,
                               int color_rows, int color_cols, int input_dtype,
                               const double* depth_intrinsic,
                               const double* color_intrinsic,
                               const double* extrinsics, float depth_scale,
                               float depth_max, float trunc_voxel_multiplier,
                               int frames_per_launch, o3dmi_stream_t stream);

/* ---- block-ownership sharding with a SLICED block touch (scheme
 * A as specified: "every GPU runs DepthTouch on a 1/G pixel slice, all-gathers
 * the candidate keys, keeps hash(key) mod G == rank"; the loop being sliced is
 * DepthTouchCPU, VoxelBlockGridCPU.cpp:117-201).
 *
 * With o3dmi_vbg_set_block_ownership(rank, world > 1) AND a communicator on the
 * calling thread (o3dmi_set_comm), o3dmi_vbg_integrate_frames takes this path
 * by itself whenever depth and colour images share size and intrinsics:
 * frames go in chunks of 16 x frames_per_launch (<= 256); on a side stream
 * rank r touches only its band of ray tiles, the candidate {block key, one bit
 * per frame of the chunk} records of all ranks are all-gathered (ONE
 * collective per chunk, fixed-size segments) and the keys a rank owns are
 * activated; on the caller's stream ONE launch per chunk applies all its
 * frames, in order, to the owned blocks' register-resident voxels, straight
 * from the raw depth / colour images (no per-pixel prepare pass). Each rank's
 * grid is bit-identical to what the replicated touch produces: the blocks it
 * owns of the single-GPU grid.
 *
 * In this mode o3dmi_vbg_integrate_frames is
 *  - COLLECTIVE: every rank calls it with the same frames. A rank that has to
 *    leave it with an error of its own first delivers the all-gather the
 *    others will wait in next with an empty segment flagged "abort"; every
 *    rank then returns O3DMI_ERR_PEER at the same chunk (nobody is left in a
 *    collective), the failing rank its own status, with its map left usable;
 *  - NOT purely stream-ordered on the host: it waits for the previous call's
 *    side-stream work, uploads the call's frame tables synchronously and
 *    follows each chunk's status word, so back-to-back calls serialise on the
 *    host (hand over long batches: bench.py passes 8000 frames per call).
 * The functions below expose the two halves for callers with their own
 * exchange, for tests and for bench.py --emulate-world. */

/* Frames per chunk for a frames_per_launch setting (16 launches). */
int o3dmi_vbg_slice_chunk_frames(int frames_per_launch);
/* Wire format sizes: a rank's segment of a chunk holds `records` 48-byte
 * records {key, 256 frame bits} + a 128-byte header. Default 4096 records and
 * chunk tables of 8192 slots; with a communicator both double by themselves
 * when a chunk does not fit (every rank reads the same headers and takes the
 * same decision), with caller-provided segments an overflow is
 * O3DMI_ERR_CAPACITY. */
int o3dmi_vbg_set_slice_capacity(o3dmi_vbg_t* g, int records,
                                 int table_slots);
int64_t o3dmi_vbg_slice_segment_bytes(const o3dmi_vbg_t* g);
/* Rank slice_rank's band of the ray tiles of up to one chunk of frames ->
 * its wire segment (device buffer of o3dmi_vbg_slice_segment_bytes bytes).
 * The grid only lends its scratch tables; its map is not touched. */
int o3dmi_vbg_touch_slice(o3dmi_vbg_t* g, int n_frames,
                          const void* const* depth_devs, int depth_rows,
                          int depth_cols, const double* depth_intrinsic,
                          const double* extrinsics, float depth_scale,
                          float depth_max, float trunc_voxel_multiplier,
                          int frames_per_launch, int slice_rank,
                          int slice_world, void* segment_out_dev,
                          o3dmi_stream_t stream);
/* o3dmi_vbg_integrate_frames through the sliced path. gathered_devs: for each
 * chunk of the call a device buffer with the `world` wire segments of all
 * ranks (rank-major; the own segment is recomputed and replaced), or NULL: the
 * all-gather runs over the calling thread's communicator. */
int o3dmi_vbg_integrate_frames_sliced(
        o3dmi_vbg_t* g, int n_frames, const void* const* depth_devs,
        int depth_rows, int depth_cols, const void* const* color_devs,
        int color_rows, int color_cols, int input_dtype,
        const double* depth_intrinsic, const double* color_intrinsic,
        const double* extrinsics, float depth_scale, float depth_max,
        float trunc_voxel_multiplier, int frames_per_launch,
        const void* const* gathered_devs, o3dmi_stream_t stream);
/* Chunks integrated through the sliced path, chunks applied a second time
 * (Reserve of the block map, or grown segments), current sizes. */
int o3dmi_vbg_sliced_stats(const o3dmi_vbg_t* g, int64_t* chunks,
                           int64_t* reapplied, int* capacity,
                           int* table_slots);

/* Measurement hook for bench.py: while profiling is on, every stride-th
 * launch that carries integrate work (o3dmi_vbg_integrate_frame(s)) is
 * bracketed with HIP events on the stream it is launched on (stride 0 = none).
 * o3dmi_vbg_profile_end synchronises and returns, over the bracketed launches:
 * their summed duration (ms), their number, the number of block-frames they
 * integrated (sum over touched blocks of the frames applied to each -- the
 * roofline's unit) and the number of frames they carried. */
int o3dmi_vbg_profile_begin(o3dmi_vbg_t* g, int max_launches, int stride);
int o3dmi_vbg_profile_end(o3dmi_vbg_t* g, o3dmi_stream_t stream,
                          double* integrate_ms, int64_t* launches,
                          int64_t* block_frames, int64_t* frames);
/* Sum over the launches bracketed by the last profile_begin / profile_end pair
 * of the DISTINCT blocks each of them worked on (the length of the group's
 * block list): what the voxel state costs in memory traffic when the frames of
 * a group are applied to register-resident blocks -- once in, once out per
 * launch, however many frames the group has. */
int64_t o3dmi_vbg_profile_distinct_blocks(const o3dmi_vbg_t* g);
/* Diagnostics: which of the frame stream's exact short division forms are in
 * use for the truncation distance voxel_size * trunc_voxel_multiplier on the
 * current device -- 0 = IEEE divisions only (the on-device proof is still
 * running, failed, or O3DMI_EXACT_DIV is set), 1 = sdf / trunc and 1 / (w + 1),
 * 2 / 3 = also 1 / z with one / two Newton steps. The proof runs asynchronously
 * (started by o3dmi_vbg_create for multiplier 8 and by the first integration
 * with any other); launches take the IEEE forms until it has finished, the
 * results are the same either way. wait != 0 blocks until it has. */
int o3dmi_vbg_division_forms(float voxel_size, float trunc_voxel_multiplier,
                             int wait);
/* The same measurement per bracketed launch (any output may be null): HIP-event
 * duration (ms), block-frames, distinct blocks, and the map size (blocks
 * active) the launch's integrate role saw when it started. Returns the number
 * of launches bracketed; at most `capacity` entries are written. */
int64_t o3dmi_vbg_profile_launches(const o3dmi_vbg_t* g, int64_t capacity,
                                   float* ms, int32_t* block_frames,
                                   int32_t* distinct_blocks,
                                   int32_t* map_size);

/* RayCast(block_coords, intrinsic, extrinsic, width, height, attrs, ...)
 * (VoxelBlockGrid.cpp:328-402). Output pointers follow o3dmi_vbg_raycast;
 * range_map_dev {h/down, w/down, 2} is also an output ("range"). */
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
                       o3dmi_stream_t stream);

/* RayCast of a REPLICATED grid sharded by pixel rows over the ranks of the
 * calling thread's communicator (o3dmi_set_comm; SURVEY 8(e), RayCast row --
 * the tracking frame of a multi-GPU loop, whose model every rank holds): rank
 * r renders rows [r B, (r + 1) B) (B = whole 8-row tiles, ceil(tiles / world)
 * of them) of depth / vertex / colour / normal, one all-gather per requested
 * map delivers every rank's band, and every rank ends with the maps
 * o3dmi_vbg_ray_cast produces, bit for bit. COLLECTIVE: every rank calls it
 * with the same arguments. What can fail on one rank alone (its arguments,
 * scratch memory, the band's launch) happens before the first all-gather and
 * the ranks agree on its status (one 4-byte all-gather, one host wait): the
 * failing rank returns its error, every other rank O3DMI_ERR_PEER, and no
 * rank waits in an exchange. The call blocks the host until the maps are
 * complete. Without a communicator (or with one rank) it is
 * o3dmi_vbg_ray_cast. */
int o3dmi_vbg_ray_cast_sharded(
        o3dmi_vbg_t* g, const int32_t* block_coords_dev, int64_t m,
        const double* intrinsic, const double* extrinsic, int width, int height,
        float* range_map_dev, float* out_depth, float* out_vertex,
        float* out_color, float* out_normal, float depth_scale, float depth_min,
        float depth_max, float weight_threshold, float trunc_voxel_multiplier,
        int range_map_down_factor, o3dmi_stream_t stream);

/* ------------------------------------------------------------------------ */
/* RGB-D odometry (t/pipelines/odometry/RGBDOdometry.h)                      */
/* ------------------------------------------------------------------------ */
/* OdometryConvergenceCriteria (RGBDOdometry.h:30-57): defaults relative_rmse
 * 1e-6, relative_fitness 1e-6. */
typedef struct {
    int max_iteration;
    double relative_rmse;
    double relative_fitness;
} o3dmi_odometry_criteria_t;

/* OdometryResult (RGBDOdometry.h:59-86) + the number of iterations run. */
typedef struct {
    double transformation[16]; /* 4x4 float64, row-major, host */
    double inlier_rmse;
    double fitness;
    int num_iterations;
} o3dmi_odometry_result_t;

/* RGBDOdometryMultiScale(source, target, intrinsics, init_source_to_target,
 * depth_scale, depth_max, criteria_list, method, params)
 * (RGBDOdometry.cpp:56-108; drivers :110-380). depth {rows,cols} U16 or F32;
 * colour {rows,cols,3} U8 or F32, may be NULL for point-to-plane; source and
 * target dtypes are independent (slam::Model tracks a U16 / U8 input frame
 * against the F32 ray-cast frame, slam/Model.cpp:72-92); criteria are
 * ordered coarse to fine, one per pyramid level; OdometryLossParams defaults
 * (RGBDOdometry.h:88-120): depth_outlier_trunc 0.07, depth_huber_delta 0.05,
 * intensity_huber_delta 0.1. init NULL = identity.
 * Status: O3DMI_ERR_NO_INLIERS ("Invalid inlier_count value ..., must be > 0."),
 * O3DMI_ERR_SINGULAR -- the reference's two exceptions on this path. */
int o3dmi_rgbd_odometry_multiscale(
        const void* source_depth_dev, const void* source_color_dev,
        const void* target_depth_dev, const void* target_color_dev,
        int source_depth_dtype, int source_color_dtype, int target_depth_dtype,
        int target_color_dtype, int rows, int cols, const double* intrinsics,
        const double* init_source_to_target, float depth_scale,
        float depth_max, int n_levels,
        const o3dmi_odometry_criteria_t* criteria, int method,
        float depth_outlier_trunc, float depth_huber_delta,
        float intensity_huber_delta, o3dmi_odometry_result_t* result,
        o3dmi_stream_t stream);

/* ComputeOdometryInformationMatrix(source_depth, target_depth, intrinsic,
 * source_to_target, dist_thr, depth_scale, depth_max)
 * (RGBDOdometry.cpp:488-513): 6x6 float64 on the host. */
int o3dmi_rgbd_odometry_information_matrix(
        const void* source_depth_dev, const void* target_depth_dev,
        int depth_dtype, int rows, int cols, const double* intrinsics,
        const double* source_to_target, float dist_thr, float depth_scale,
        float depth_max, double* information_host, o3dmi_stream_t stream);

/* Extension: loads the kernels of the tracking / integration path now. HIP
 * loads a translation unit's device code at the first launch of one of its
 * kernels (1.5-2.8 ms each for the large ones); an application that cares about
 * the latency of its FIRST frame calls this once at start-up. */
int o3dmi_preload(void);

/* Extension (no counterpart in the reference's API): the block coordinates
 * the most recent o3dmi_vbg_integrate_frame (or the last frame of an
 * o3dmi_vbg_integrate_frames call with frames_per_launch = 1) touched -- the
 * set GetUniqueBlockCoordinates returns for that frame's (depth, intrinsic,
 * extrinsic), in the order the touch found them -- copied on the stream to
 * out_coords_dev {capacity,3}, their number to *out_count_dev (device int32):
 * no second block touch and no host round trip for the ray cast that follows
 * an integration (slam::Model uses it the same way). Issue it right behind
 * the integrate call; pass both to o3dmi_vbg_ray_cast_dev. */
int o3dmi_vbg_last_frame_block_coordinates(o3dmi_vbg_t* g,
                                           int32_t* out_coords_dev,
                                           int64_t capacity,
                                           int32_t* out_count_dev,
                                           o3dmi_stream_t stream);

/* RayCast with the number of block coordinates resident on the device
 * (*m_dev <= max_m; m_dev NULL = max_m): no host round trip between the
 * integration that produced the block list and the ray cast that uses it.
 * block_coords_dev NULL: the blocks the most recent frame-stream integration
 * touched (what o3dmi_vbg_last_frame_block_coordinates would copy out), read
 * from the grid's own list -- no export launch; max_m / m_dev are ignored.
 * range_map_dev NULL: the range map is the grid's own scratch, as in the
 * reference (VoxelBlockGrid.cpp:357-360 allocates it inside RayCast); with
 * the default down factor of 8 the ray cast that consumes it leaves it clean
 * for the next call, which then needs no clearing launch. */
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
                           int range_map_down_factor, o3dmi_stream_t stream);

/* ExtractPointCloud(weight_threshold, estimated_point_number)
 * (VoxelBlockGrid.cpp:404-434). capacity < 0: only counts (the reference's
 * 2-pass estimation) -> *total_out; otherwise writes up to `capacity` points:
 * points / normals {capacity,3} float32, colors {capacity,3} float32 in [0,1]
 * when the grid has a "color" attribute (colors_dev may be NULL).
 * *total_out = number of surface points found (synchronises). Order: active
 * blocks by ascending buffer index, then voxel, then axis. */
int o3dmi_vbg_extract_point_cloud(o3dmi_vbg_t* g, float weight_threshold,
                                  int64_t capacity, float* points_dev,
                                  float* normals_dev, float* colors_dev,
                                  int64_t* total_out, o3dmi_stream_t stream);

/* ------------------------------------------------------------------------ */
/* slam::Model (t/pipelines/slam/Model.h:24-137, Model.cpp:23-118)           */
/* ------------------------------------------------------------------------ */
typedef struct o3dmi_slam_model o3dmi_slam_model_t;

/* Model(voxel_size, block_resolution = 16, block_count = 1000, T_init = I):
 * grid attributes ("tsdf" F32 x1, "weight" U16 x1, "color" U16 x3)
 * (Model.cpp:23-38). T_init host 4x4 float64 or NULL. */
int o3dmi_slam_model_create(float voxel_size, int block_resolution,
                            int64_t block_count, const double* T_init,
                            o3dmi_stream_t stream, o3dmi_slam_model_t** out);
int o3dmi_slam_model_destroy(o3dmi_slam_model_t* m);
o3dmi_vbg_t* o3dmi_slam_model_voxel_grid(o3dmi_slam_model_t* m);
/* GetCurrentFramePose / UpdateFramePose (Model.h:45-54); frame ids that do
 * not advance by one only warn in the reference, here they are accepted. */
int o3dmi_slam_model_get_current_frame_pose(const o3dmi_slam_model_t* m,
                                            double* T_frame_to_world);
int o3dmi_slam_model_update_frame_pose(o3dmi_slam_model_t* m, int frame_id,
                                       const double* T_frame_to_world);
int o3dmi_slam_model_frame_id(const o3dmi_slam_model_t* m);
/* SynthesizeModelFrame (Model.cpp:40-68): ray-casts the frustum blocks of the
 * last Integrate at the current pose into depth {h,w,1} F32 (raw units) and,
 * if color_out_dev != NULL, colour {h,w,3} F32 in [0,1]. weight_threshold < 0
 * = min(frame_id, 3). */
int o3dmi_slam_model_synthesize_model_frame(
        o3dmi_slam_model_t* m, const double* intrinsics, int width, int height,
        float depth_scale, float depth_min, float depth_max,
        float trunc_voxel_multiplier, float weight_threshold,
        float* depth_out_dev, float* color_out_dev, o3dmi_stream_t stream);
/* TrackFrameToModel (Model.cpp:70-92): RGBDOdometryMultiScale(input frame,
 * ray-cast frame, intrinsics, identity, depth_scale, depth_max, criteria,
 * method, OdometryLossParams(depth_diff)). Defaults: depth_diff 0.07, method
 * point-to-plane, criteria {6, 3, 1}. raycast_* are Float32. */
int o3dmi_slam_model_track_frame_to_model(
        o3dmi_slam_model_t* m, const void* input_depth_dev,
        int input_depth_dtype, const void* input_color_dev,
        int input_color_dtype, const float* raycast_depth_dev,
        const float* raycast_color_dev, int rows, int cols,
        const double* intrinsics, float depth_scale, float depth_max,
        float depth_diff, int method, int n_levels,
        const o3dmi_odometry_criteria_t* criteria,
        o3dmi_odometry_result_t* result, o3dmi_stream_t stream);
/* Integrate (Model.cpp:94-108): GetUniqueBlockCoordinates + Integrate at
 * InverseTransformation(current pose); remembers the frustum blocks. Colour
 * may be NULL (depth-only integration). */
int o3dmi_slam_model_integrate(o3dmi_slam_model_t* m, const void* depth_dev,
                               int depth_dtype, const void* color_dev,
                               int rows, int cols, const double* intrinsics,
                               float depth_scale, float depth_max,
                               float trunc_voxel_multiplier,
                               o3dmi_stream_t stream);
/* Number of frustum blocks of the last Integrate (reads the device-resident
 * count back: synchronises) and their keys (device, {n,3} int32; valid until
 * the next Integrate). */
int64_t o3dmi_slam_model_frustum_block_count(const o3dmi_slam_model_t* m);
const int32_t* o3dmi_slam_model_frustum_block_coords(const o3dmi_slam_model_t* m);
/* ExtractPointCloud (Model.cpp:110-113). */
int o3dmi_slam_model_extract_point_cloud(o3dmi_slam_model_t* m,
                                         float weight_threshold,
                                         int64_t capacity, float* points_dev,
                                         float* normals_dev, float* colors_dev,
                                         int64_t* total_out,
                                         o3dmi_stream_t stream);

/* ------------------------------------------------------------------------ */
/* NPZ interchange (t/io/NumpyIO.cpp) and VoxelBlockGrid::Save / Load        */
/* ------------------------------------------------------------------------ */
/* An in-memory set of named host arrays. o3dmi_npz_write = t::io::WriteNpz
 * (NumpyIO.cpp:759-789; NPY 1.0 headers and stored zip entries formatted as
 * the reference does, ZIP64 where its 32-bit fields would wrap);
 * o3dmi_npz_read = t::io::ReadNpz (:675-757; also reads numpy's own savez /
 * savez_compressed output). Little-endian C-order arrays only. */
typedef struct o3dmi_npz o3dmi_npz_t;
int o3dmi_npz_create(o3dmi_npz_t** out);
int o3dmi_npz_destroy(o3dmi_npz_t* z);
/* Copies `data_host`; an existing entry of the same name is replaced. */
int o3dmi_npz_add(o3dmi_npz_t* z, const char* name, int dtype, int ndim,
                  const int64_t* shape, const void* data_host);
int o3dmi_npz_count(const o3dmi_npz_t* z);
const char* o3dmi_npz_name(const o3dmi_npz_t* z, int i);
/* shape8 must hold 8 entries; *data_host points into the set. */
int o3dmi_npz_get(const o3dmi_npz_t* z, const char* name, int* dtype, int* ndim,
                  int64_t* shape8, const void** data_host);
int o3dmi_npz_write(const o3dmi_npz_t* z, const char* file_name);
int o3dmi_npz_read(const char* file_name, o3dmi_npz_t** out);

/* VoxelBlockGrid::Save(file_name) (VoxelBlockGrid.cpp:474-524): entries
 * "voxel_size" {1} f32, "block_resolution" {1} i64, a 0-d u8 placeholder named
 * after the device ("HIP:0" -- stock Open3D ignores unknown prefixes and loads
 * on CPU:0), "attr_name_<name>" {1} i32 = value index, "key" {n,3} i32 and
 * "value_%03d" {n,res,res,res,C} of the ACTIVE blocks (ascending buffer
 * index). ".npz" is appended when missing, as the reference. */
int o3dmi_vbg_save(o3dmi_vbg_t* g, const char* file_name,
                   o3dmi_stream_t stream);
/* VoxelBlockGrid::Load(file_name) (VoxelBlockGrid.cpp:538-596): capacity =
 * number of stored keys; keys and value rows are inserted into a new grid on
 * the current device. */
int o3dmi_vbg_load(const char* file_name, o3dmi_stream_t stream,
                   o3dmi_vbg_t** out);
/* Frame-sharded multi-GPU integration, the merge step (SURVEY section 8e
 * scheme B; no counterpart in the reference, which is single-device). Each
 * rank integrates its own frames into a private grid; the grids are then
 * combined block by block.
 * o3dmi_vbg_export_blocks writes the ACTIVE blocks in ascending buffer index
 * (the order of Save): keys {n,3} int32 and, per attribute i, the value rows
 * {n,res,res,res,C_i} into values_dev[i] (device, caller-allocated for
 * `capacity` blocks). keys_dev == NULL only counts. *n_out = number of active
 * blocks (synchronises); O3DMI_ERR_CAPACITY if it exceeds `capacity`.
 * o3dmi_vbg_merge_blocks folds n foreign blocks of the same attribute layout
 * into `g`: missing blocks are activated (HashMap::Activate's capacity policy),
 * then per voxel, with w1 / w2 the weights of `g` / of the foreign block:
 *   w2 == 0: unchanged;  w1 == 0: the foreign voxel is copied;
 *   else inv = 1 / (w1 + w2), tsdf = (w1 tsdf1 + w2 tsdf2) inv,
 *        colour likewise per channel, weight = w1 + w2 (uint16 grids:
 *        saturating at 65535)
 * in float32 with Integrate's store conversions -- the running mean Integrate
 * itself computes (VoxelBlockGridImpl.h:258-300), so folding a one-frame grid
 * in is bit-identical to integrating that frame. Precondition (not checked):
 * the n keys are pairwise distinct -- two rows with one key would race on
 * the same voxels. */
int o3dmi_vbg_export_blocks(o3dmi_vbg_t* g, int64_t capacity,
                            int32_t* keys_dev, void* const* values_dev,
                            int64_t* n_out, o3dmi_stream_t stream);
int o3dmi_vbg_merge_blocks(o3dmi_vbg_t* g, const int32_t* keys_dev,
                           const void* const* values_dev, int64_t n,
                           o3dmi_stream_t stream);

/* The payload exchange of the frame-sharded scheme,
 * owner-partitioned: every active block of this rank's private grid travels
 * to the rank that OWNS it (the fixed key hash of the block-ownership scheme,
 * o3dmi_hash_set_ownership) with one all-to-all per tensor -- keys, then each
 * attribute's rows -- and the owner folds the partial blocks of all ranks in
 * (o3dmi_vbg_merge_blocks, own partial first, then ascending source rank).
 * Blocks sent away are erased and their value rows zeroed, so afterwards the
 * ranks hold DISJOINT grids whose union is the model of the whole stream: the
 * same layout block-ownership sharding produces. A rank moves (world - 1) /
 * world of its blocks once (an all-gather of everything would move world x
 * as much to every rank). Collective: every rank of `comm` must call it. A
 * rank that cannot list its blocks, or cannot reserve room for the arriving
 * ones, says so in the count exchange / in a status all-gather before the
 * first payload all-to-all: it returns its own error, the others
 * O3DMI_ERR_PEER, and every grid still holds what it held (same for the
 * counting and export stages of o3dmi_vbg_allgather_owned_blocks). Room for
 * the arriving blocks is reserved BEFORE anything is erased; after
 * o3dmi_vbg_allgather_owned_blocks a further merge is refused (the replicated
 * blocks would be counted again).
 * o3dmi_vbg_allgather_owned_blocks then replicates the finished blocks on
 * every rank (when each GPU is to ray-cast the whole model). */
int o3dmi_vbg_merge_frame_sharded(o3dmi_vbg_t* g, o3dmi_comm_t* comm,
                                  o3dmi_stream_t stream);
int o3dmi_vbg_allgather_owned_blocks(o3dmi_vbg_t* g, o3dmi_comm_t* comm,
                                     o3dmi_stream_t stream);

/* Introspection of a grid (needed after Load): attribute count / i-th name,
 * voxel size, block resolution. */
int o3dmi_vbg_attribute_count(const o3dmi_vbg_t* g);
const char* o3dmi_vbg_attribute_name(const o3dmi_vbg_t* g, int i);
float o3dmi_vbg_voxel_size(const o3dmi_vbg_t* g);
int64_t o3dmi_vbg_block_resolution(const o3dmi_vbg_t* g);

#ifdef __cplusplus
}
#endif
#endif /* O3D_MI355X_HOST_H_ */
