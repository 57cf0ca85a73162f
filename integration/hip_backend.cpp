// The reference-side binding of libo3d_mi355x.so, COMPILED:
// the `...HIP` functions a maintainer adds next to Open3D's `...CUDA` ones --
// INTEGRATION.md sections 0-5 as one translation unit. Nothing here is
// product code and nothing here runs in the product: tests/test_integration_
// binding.py type-checks this file against
//   * the reference's OWN headers, included from where they lie under
//     /root/reference/cpp (t/geometry/kernel/VoxelBlockGrid.h,
//     t/pipelines/kernel/RegistrationImpl.h, t/geometry/kernel/Transform.h,
//     core/nns/FixedRadiusIndex.h, core/hashmap/DeviceHashBackend.h), over the
//     stand-in Tensor / TensorMap of oracle/ref_shim, and
//   * include/o3d_mi355x.h,
// and links it against the built library, so that the documented drop-in
// cannot rot: every `...HIP` function below is static_assert-ed to have the
// type of the reference's per-device function it stands beside.
//
//   g++ -std=c++17 -fsyntax-only -I integration/shim -I oracle/ref_shim \
//       -I /root/reference/cpp -I include integration/hip_backend.cpp
#include <memory>
#include <tuple>
#include <type_traits>
#include <vector>

#include "open3d/core/HIPBackend.h"
#include "open3d/core/Tensor.h"
#include "open3d/core/hashmap/DeviceHashBackend.h"
#include "open3d/core/hashmap/HashMap.h"
#include "open3d/core/nns/FixedRadiusIndex.h"
#include "open3d/core/nns/NeighborSearchCommon.h"
#include "open3d/t/geometry/kernel/Transform.h"
#include "open3d/t/geometry/kernel/VoxelBlockGrid.h"
#include "open3d/t/pipelines/kernel/RegistrationImpl.h"
#include "open3d/t/pipelines/registration/RobustKernel.h"

namespace open3d {

// ---------------------------------------------------------------------------
// 3. Hash map -- core/hashmap/DeviceHashBackend.cpp:18-49
//    (CreateDeviceHashBackend gains `else if (device.IsHIP()) return
//    std::make_shared<HIPHashBackend>(...)`).
// ---------------------------------------------------------------------------
namespace core {

class HIPHashBackend : public DeviceHashBackend {  // DeviceHashBackend.h:20-107
public:
    HIPHashBackend(int64_t init_capacity,
                   int64_t key_dsize,
                   const std::vector<int64_t>& value_dsizes,
                   const Device& device)
        : DeviceHashBackend(init_capacity, key_dsize, value_dsizes, device) {
        if (key_dsize != 12)
            utility::LogError("HIP backend supports Int32x3 keys");
        O3DMI_CALL(o3dmi_hash_create(init_capacity, (int)value_dsizes.size(),
                                     value_dsizes.data(), HipStream(), &h_));
    }
    ~HIPHashBackend() override { o3dmi_hash_destroy(h_); }

    void Reserve(int64_t capacity) override {
        O3DMI_CALL(o3dmi_hash_reserve(h_, capacity, HipStream()));
    }
    void Insert(const void* input_keys,
                const std::vector<const void*>& input_values_soa,
                buf_index_t* output_buf_indices,
                bool* output_masks,
                int64_t count) override {
        static_assert(sizeof(buf_index_t) == sizeof(int32_t) &&
                              sizeof(bool) == sizeof(uint8_t),
                      "the ABI's int32 indices / uint8 masks are the "
                      "reference's buf_index_t / bool");
        if (input_values_soa.empty())  // HashMap::Activate
            O3DMI_CALL(o3dmi_hash_activate(
                    h_, (const int32_t*)input_keys, count, nullptr,
                    (int32_t*)output_buf_indices, (uint8_t*)output_masks,
                    HipStream()));
        else
            O3DMI_CALL(o3dmi_hash_insert(
                    h_, (const int32_t*)input_keys, input_values_soa.data(),
                    count, (int32_t*)output_buf_indices,
                    (uint8_t*)output_masks, HipStream()));
    }
    void Find(const void* input_keys,
              buf_index_t* output_buf_indices,
              bool* output_masks,
              int64_t count) override {
        O3DMI_CALL(o3dmi_hash_find(h_, (const int32_t*)input_keys, count,
                                   nullptr, (int32_t*)output_buf_indices,
                                   (uint8_t*)output_masks, HipStream()));
    }
    void Erase(const void* input_keys,
               bool* output_masks,
               int64_t count) override {
        O3DMI_CALL(o3dmi_hash_erase(h_, (const int32_t*)input_keys, count,
                                    (uint8_t*)output_masks, HipStream()));
    }
    int64_t GetActiveIndices(buf_index_t* output_buf_indices) override {
        int64_t c = 0;
        O3DMI_CALL(o3dmi_hash_active_indices(h_, (int32_t*)output_buf_indices,
                                             HipStream(), &c));
        return c;
    }
    void Clear() override { O3DMI_CALL(o3dmi_hash_clear(h_, HipStream())); }
    int64_t Size() const override {
        int64_t s = 0;
        O3DMI_CALL(o3dmi_hash_size(h_, HipStream(), &s));
        return s;
    }
    int64_t GetBucketCount() const override {
        return o3dmi_hash_bucket_count(h_);
    }
    float LoadFactor() const override {
        return float(Size()) / float(GetBucketCount());
    }
    std::vector<int64_t> BucketSizes() const override {
        // open addressing: a "bucket" is a slot (0 or 1 entries); the
        // reference's TBB backend returns its own buckets' sizes, tests only
        // read the sum
        return std::vector<int64_t>((size_t)GetBucketCount(), 0);
    }
    // The buffers live in the o3dmi_hash_t (o3dmi_hash_key_buffer /
    // o3dmi_hash_value_buffer(i), adopted as Tensors through core::Blob with a
    // no-op deleter): nothing to allocate or free here.
    void Allocate(int64_t capacity) override { (void)capacity; }
    void Free() override {}

    o3dmi_hash_t* Handle() const { return h_; }

private:
    o3dmi_hash_t* h_ = nullptr;
};

// CreateDeviceHashBackend's HIP branch (DeviceHashBackend.cpp:18-49)
std::shared_ptr<DeviceHashBackend> CreateHIPHashBackend(
        int64_t init_capacity,
        const Dtype& key_dtype,
        const SizeVector& key_element_shape,
        const std::vector<Dtype>& value_dtypes,
        const std::vector<SizeVector>& value_element_shapes,
        const Device& device) {
    const int64_t key_dsize =
            key_dtype.ByteSize() * key_element_shape.NumElements();
    std::vector<int64_t> value_dsizes;
    for (size_t i = 0; i < value_dtypes.size(); ++i)
        value_dsizes.push_back(value_dtypes[i].ByteSize() *
                               value_element_shapes[i].NumElements());
    return std::make_shared<HIPHashBackend>(init_capacity, key_dsize,
                                            value_dsizes, device);
}

inline o3dmi_hash_t* HipHandle(const std::shared_ptr<HashMap>& hashmap) {
    auto backend = std::dynamic_pointer_cast<HIPHashBackend>(
            hashmap->GetDeviceHashBackend());
    if (!backend) utility::LogError("hash map is not on a HIP device");
    return backend->Handle();
}

}  // namespace core

// ---------------------------------------------------------------------------
// 1, 2, 4. VoxelBlockGrid kernels -- t/geometry/kernel/VoxelBlockGrid.cpp
//    dispatchers gain `else if (x.IsHIP()) { ...HIP(...); }`
// ---------------------------------------------------------------------------
namespace t {
namespace geometry {
namespace kernel {
namespace voxel_grid {

static core::Tensor HostF64(const core::Tensor& m) {
    return m.To(core::Device("CPU:0")).To(core::Float64).Contiguous();
}

// beside DepthTouchCUDA (kernel/VoxelBlockGrid.h:345-355,
// VoxelBlockGridCUDA.cu:106-227)
void DepthTouchHIP(std::shared_ptr<core::HashMap>& hashmap,
                   const core::Tensor& depth,
                   const core::Tensor& intrinsic,
                   const core::Tensor& extrinsic,
                   core::Tensor& voxel_block_coords,
                   index_t voxel_grid_resolution,
                   float voxel_size,
                   float sdf_trunc,
                   float depth_scale,
                   float depth_max,
                   index_t stride) {
    const core::Tensor K = HostF64(intrinsic), T = HostF64(extrinsic);
    const int64_t cap =
            (depth.GetShape(0) / stride) * (depth.GetShape(1) / stride) * 4;
    voxel_block_coords = core::Tensor({cap, 3}, core::Int32, depth.GetDevice());
    core::Tensor count({1}, core::Int32, depth.GetDevice());
    O3DMI_CALL(o3dmi_vbg_depth_touch(
            core::HipHandle(hashmap), depth.GetDataPtr(),
            core::ToO3dmi(depth.GetDtype()), (int)depth.GetShape(0),
            (int)depth.GetShape(1), K.GetDataPtr<double>(),
            T.GetDataPtr<double>(), voxel_block_coords.GetDataPtr<int32_t>(),
            cap, count.GetDataPtr<int32_t>(), voxel_grid_resolution, voxel_size,
            sdf_trunc, depth_scale, depth_max, stride, core::HipStream()));
    // the one synchronisation the reference also has (Item())
    const int64_t m = count.To(core::Device("CPU:0")).Item<int32_t>();
    if (m == 0)
        utility::LogError(
                "No block is touched in TSDF volume, abort integration. Please "
                "check specified parameters, especially depth_scale and "
                "voxel_size");
    voxel_block_coords = voxel_block_coords.Slice(0, 0, m);
}
static_assert(std::is_same<decltype(&DepthTouchHIP),
                           decltype(&DepthTouchCPU)>::value,
              "DepthTouchHIP has the dispatcher's per-device signature");

// beside PointCloudTouchCUDA (kernel/VoxelBlockGrid.h:338-343)
void PointCloudTouchHIP(std::shared_ptr<core::HashMap>& hashmap,
                        const core::Tensor& points,
                        core::Tensor& voxel_block_coords,
                        index_t voxel_grid_resolution,
                        float voxel_size,
                        float sdf_trunc) {
    const int64_t cap = points.GetLength() * 27;
    voxel_block_coords =
            core::Tensor({cap, 3}, core::Int32, points.GetDevice());
    core::Tensor count({1}, core::Int32, points.GetDevice());
    O3DMI_CALL(o3dmi_vbg_pointcloud_touch(
            core::HipHandle(hashmap), points.GetDataPtr<float>(),
            points.GetLength(), voxel_block_coords.GetDataPtr<int32_t>(), cap,
            count.GetDataPtr<int32_t>(), voxel_grid_resolution, voxel_size,
            sdf_trunc, core::HipStream()));
    const int64_t m = count.To(core::Device("CPU:0")).Item<int32_t>();
    voxel_block_coords = voxel_block_coords.Slice(0, 0, m);
}
static_assert(std::is_same<decltype(&PointCloudTouchHIP),
                           decltype(&PointCloudTouchCPU)>::value,
              "PointCloudTouchHIP has the dispatcher's per-device signature");

// beside IntegrateCUDA<input_depth_t, input_color_t, tsdf_t, weight_t,
// color_t> (kernel/VoxelBlockGrid.h:364-381, VoxelBlockGridImpl.h:151-308).
// Not a template: the 2 x 2 dtype dispatch (kernel/VoxelBlockGrid.cpp:107-146)
// happens inside the library.
void IntegrateHIP(const core::Tensor& depth,
                  const core::Tensor& color,
                  const core::Tensor& block_indices,
                  const core::Tensor& block_keys,
                  TensorMap& block_value_map,
                  const core::Tensor& depth_intrinsic,
                  const core::Tensor& color_intrinsic,
                  const core::Tensor& extrinsic,
                  index_t resolution,
                  float voxel_size,
                  float sdf_trunc,
                  float depth_scale,
                  float depth_max) {
    if (!block_value_map.Contains("tsdf") ||
        !block_value_map.Contains("weight"))
        utility::LogError(
                "TSDF and/or weight not allocated in blocks, please implement "
                "customized integration.");
    const bool with_color =
            block_value_map.Contains("color") && color.NumElements() > 0;
    const core::Tensor Kd = HostF64(depth_intrinsic),
                       Kc = HostF64(color_intrinsic), T = HostF64(extrinsic);
    core::Tensor w = block_value_map.at("weight");
    O3DMI_CALL(o3dmi_vbg_integrate(
            depth.GetDataPtr(), (int)depth.GetShape(0), (int)depth.GetShape(1),
            with_color ? color.GetDataPtr() : nullptr,
            with_color ? (int)color.GetShape(0) : 0,
            with_color ? (int)color.GetShape(1) : 0,
            depth.GetDtype() == core::Float32 ? O3DMI_F32 : O3DMI_U16,
            block_indices.GetDataPtr<int32_t>(), block_indices.GetLength(),
            /*n_indices_dev=*/nullptr, block_keys.GetDataPtr<int32_t>(),
            block_value_map.at("tsdf").GetDataPtr<float>(), w.GetDataPtr(),
            with_color ? block_value_map.at("color").GetDataPtr() : nullptr,
            w.GetDtype() == core::Float32 ? O3DMI_F32 : O3DMI_U16,
            Kd.GetDataPtr<double>(), Kc.GetDataPtr<double>(),
            T.GetDataPtr<double>(), resolution, voxel_size, sdf_trunc,
            depth_scale, depth_max, core::HipStream()));
}
static_assert(
        std::is_same<decltype(&IntegrateHIP),
                     decltype(&IntegrateCPU<uint16_t, uint8_t, float, uint16_t,
                                            uint16_t>)>::value,
        "IntegrateHIP takes IntegrateCPU/CUDA's arguments");

// beside EstimateRangeCUDA (kernel/VoxelBlockGrid.h:383-394)
void EstimateRangeHIP(const core::Tensor& block_keys,
                      core::Tensor& range_minmax_map,
                      const core::Tensor& intrinsics,
                      const core::Tensor& extrinsics,
                      int h,
                      int w,
                      int down_factor,
                      int64_t block_resolution,
                      float voxel_size,
                      float depth_min,
                      float depth_max,
                      core::Tensor& fragment_buffer) {
    (void)fragment_buffer;  // no fragment stage on this backend
    const core::Tensor K = HostF64(intrinsics), T = HostF64(extrinsics);
    range_minmax_map = core::Tensor({h / down_factor, w / down_factor, 2},
                                    core::Float32, block_keys.GetDevice());
    O3DMI_CALL(o3dmi_vbg_estimate_range(
            block_keys.GetDataPtr<int32_t>(), block_keys.GetLength(),
            range_minmax_map.GetDataPtr<float>(), K.GetDataPtr<double>(),
            T.GetDataPtr<double>(), h, w, down_factor, block_resolution,
            voxel_size, depth_min, depth_max, core::HipStream()));
}
static_assert(std::is_same<decltype(&EstimateRangeHIP),
                           decltype(&EstimateRangeCPU)>::value,
              "EstimateRangeHIP has the dispatcher's per-device signature");

// beside RayCastCUDA<tsdf_t, weight_t, color_t> (kernel/VoxelBlockGrid.h:
// 396-413, VoxelBlockGridImpl.h:578-1120); dtype dispatch inside the library
void RayCastHIP(std::shared_ptr<core::HashMap>& hashmap,
                const TensorMap& block_value_map,
                const core::Tensor& range_map,
                TensorMap& renderings_map,
                const core::Tensor& intrinsic,
                const core::Tensor& extrinsic,
                index_t h,
                index_t w,
                index_t block_resolution,
                float voxel_size,
                float depth_scale,
                float depth_min,
                float depth_max,
                float weight_threshold,
                float trunc_voxel_multiplier,
                int range_map_down_factor) {
    const core::Tensor K = HostF64(intrinsic), T = HostF64(extrinsic);
    auto P = [&](const char* n) -> void* {  // NULL: map not requested
        return renderings_map.Contains(n) ? renderings_map.at(n).GetDataPtr()
                                          : nullptr;
    };
    const core::Tensor& weight = block_value_map.at("weight");
    O3DMI_CALL(o3dmi_vbg_raycast(
            core::HipHandle(hashmap),
            block_value_map.at("tsdf").GetDataPtr<float>(),
            weight.GetDataPtr(),
            block_value_map.Contains("color")
                    ? block_value_map.at("color").GetDataPtr()
                    : nullptr,
            weight.GetDtype() == core::Float32 ? O3DMI_F32 : O3DMI_U16,
            range_map.GetDataPtr<float>(), (float*)P("depth"),
            (float*)P("vertex"), (float*)P("color"), (float*)P("normal"),
            (int64_t*)P("index"), (uint8_t*)P("mask"),
            (float*)P("interp_ratio"), (float*)P("interp_ratio_dx"),
            (float*)P("interp_ratio_dy"), (float*)P("interp_ratio_dz"),
            K.GetDataPtr<double>(), T.GetDataPtr<double>(), h, w,
            block_resolution, voxel_size, depth_scale, depth_min, depth_max,
            weight_threshold, trunc_voxel_multiplier, range_map_down_factor,
            core::HipStream()));
}
static_assert(
        std::is_same<decltype(&RayCastHIP),
                     decltype(&RayCastCPU<float, uint16_t, uint16_t>)>::value,
        "RayCastHIP takes RayCastCPU/CUDA's arguments");

}  // namespace voxel_grid

// 5. beside TransformPointsCUDA / TransformNormalsCUDA
//    (t/geometry/kernel/Transform.h:42-47)
namespace transform {
void TransformPointsHIP(const core::Tensor& transformation,
                        core::Tensor& points) {
    const core::Tensor T = transformation.To(core::Device("CPU:0"))
                                   .To(core::Float64)
                                   .Contiguous();
    O3DMI_CALL(o3dmi_transform_points(T.GetDataPtr<double>(),
                                      points.GetDataPtr(), points.GetLength(),
                                      core::ToO3dmi(points.GetDtype()),
                                      core::HipStream()));
}
void TransformNormalsHIP(const core::Tensor& transformation,
                         core::Tensor& normals) {
    const core::Tensor T = transformation.To(core::Device("CPU:0"))
                                   .To(core::Float64)
                                   .Contiguous();
    O3DMI_CALL(o3dmi_transform_normals(T.GetDataPtr<double>(),
                                       normals.GetDataPtr(),
                                       normals.GetLength(),
                                       core::ToO3dmi(normals.GetDtype()),
                                       core::HipStream()));
}
static_assert(std::is_same<decltype(&TransformPointsHIP),
                           decltype(&TransformPointsCPU)>::value &&
                      std::is_same<decltype(&TransformNormalsHIP),
                                   decltype(&TransformNormalsCPU)>::value,
              "Transform*HIP have the dispatcher's per-device signature");
}  // namespace transform
}  // namespace kernel
}  // namespace geometry

// ---------------------------------------------------------------------------
// 5. ICP -- t/pipelines/kernel/Registration.cpp:35-78 (ComputePosePointToPlane
//    dispatcher) gains `else if (source_points.IsHIP())`
// ---------------------------------------------------------------------------
namespace pipelines {
namespace kernel {

// beside ComputePosePointToPlaneCUDA (RegistrationImpl.h:93-102,
// RegistrationCUDA.cu:81-117)
void ComputePosePointToPlaneHIP(const core::Tensor& source_points,
                                const core::Tensor& target_points,
                                const core::Tensor& target_normals,
                                const core::Tensor& correspondence_indices,
                                core::Tensor& pose,
                                float& residual,
                                int& inlier_count,
                                const core::Dtype& dtype,
                                const core::Device& device,
                                const registration::RobustKernel& kernel) {
    core::Tensor sums = core::Tensor::Empty({29}, core::Float64, device);
    O3DMI_CALL(o3dmi_icp_p2plane_accumulate(
            source_points.Contiguous().GetDataPtr(),
            target_points.Contiguous().GetDataPtr(),
            target_normals.Contiguous().GetDataPtr(),
            correspondence_indices.Contiguous().GetDataPtr<int64_t>(),
            source_points.GetLength(), core::ToO3dmi(dtype), (int)kernel.type_,
            kernel.scaling_parameter_, kernel.shape_parameter_,
            sums.GetDataPtr<double>(), core::HipStream()));
    // the one device-to-host copy of the iteration; DecodeAndSolve6x6
    // (TransformationConverter.cpp:189-226) on the host copy
    const core::Tensor host = sums.To(core::Device("CPU:0"));
    pose = core::Tensor::Empty({6}, core::Float64, core::Device("CPU:0"));
    O3DMI_CALL(o3dmi_decode_and_solve6x6(host.GetDataPtr<double>(),
                                         pose.GetDataPtr<double>(), &residual,
                                         &inlier_count));
}
static_assert(std::is_same<decltype(&ComputePosePointToPlaneHIP),
                           decltype(&ComputePosePointToPlaneCPU)>::value,
              "ComputePosePointToPlaneHIP has the dispatcher's per-device "
              "signature");

}  // namespace kernel
}  // namespace pipelines
}  // namespace t

// ---------------------------------------------------------------------------
// 5. Neighbour search -- core/nns/FixedRadiusIndex.cpp:58-136 (SetTensorData)
//    and :229-299 (SearchHybrid). The reference builds a CSR hash table
//    (BuildSpatialHashTableCUDA<T>) and hands its tensors to
//    HybridSearchCUDA<T, TIndex> (FixedRadiusIndex.h:227-233,364-377); this
//    backend's index is an opaque handle built once per SetTensorData, so the
//    HIP branches of the two members replace the pair of calls.
// ---------------------------------------------------------------------------
namespace core {
namespace nns {

struct HIPFixedRadiusIndex {
    o3dmi_nns_t* handle = nullptr;
    Tensor dataset_points;  // kept alive: the build is stream-ordered
    ~HIPFixedRadiusIndex() {
        if (handle) o3dmi_nns_destroy(handle);
    }

    // FixedRadiusIndex::SetTensorData, HIP branch
    bool SetTensorData(const Tensor& points, double radius) {
        if (handle) o3dmi_nns_destroy(handle);
        handle = nullptr;
        dataset_points = points.Contiguous();
        O3DMI_CALL(o3dmi_nns_create(dataset_points.GetDataPtr(),
                                    dataset_points.GetLength(),
                                    ToO3dmi(dataset_points.GetDtype()), radius,
                                    HipStream(), &handle));
        return true;
    }

    // FixedRadiusIndex::SearchHybrid, HIP branch: {indices {Q,max_knn},
    // distances {Q,max_knn}, counts {Q}} -- NNSIndex::SearchHybrid's tuple
    std::tuple<Tensor, Tensor, Tensor> SearchHybrid(const Tensor& query_points,
                                                    double radius,
                                                    int max_knn) const {
        (void)radius;  // fixed at SetTensorData, as FixedRadiusIndex checks
        const Tensor q = query_points.Contiguous();
        const int64_t n = q.GetLength();
        Tensor indices = Tensor::Empty({n, max_knn}, Int32, q.GetDevice());
        Tensor distances =
                Tensor::Empty({n, max_knn}, q.GetDtype(), q.GetDevice());
        Tensor counts = Tensor::Empty({n}, Int32, q.GetDevice());
        if (max_knn == 1)
            O3DMI_CALL(o3dmi_nns_hybrid_search_k1(
                    handle, q.GetDataPtr(), n, indices.GetDataPtr<int32_t>(),
                    distances.GetDataPtr(), counts.GetDataPtr<int32_t>(),
                    HipStream()));
        else
            O3DMI_CALL(o3dmi_nns_hybrid_search(
                    handle, q.GetDataPtr(), n, max_knn,
                    indices.GetDataPtr<int32_t>(), distances.GetDataPtr(),
                    counts.GetDataPtr<int32_t>(), HipStream()));
        return std::make_tuple(indices, distances, counts);
    }
};

// The seam-level form for a caller that keeps the reference's two free
// functions: the build output tensors stay empty (the handle carries the
// index), the search ignores them.
template <class T>
void BuildSpatialHashTableHIP(const Tensor& points,
                              double radius,
                              const Tensor& points_row_splits,
                              const Tensor& hash_table_splits,
                              Tensor& hash_table_index,
                              Tensor& hash_table_cell_splits,
                              HIPFixedRadiusIndex& index) {
    (void)points_row_splits;
    (void)hash_table_splits;
    (void)hash_table_index;
    (void)hash_table_cell_splits;
    index.SetTensorData(points, radius);
}

template <class T, class TIndex>
void HybridSearchHIP(const Tensor& points,
                     const Tensor& queries,
                     double radius,
                     int max_knn,
                     const Tensor& points_row_splits,
                     const Tensor& queries_row_splits,
                     const Tensor& hash_table_splits,
                     const Tensor& hash_table_index,
                     const Tensor& hash_table_cell_splits,
                     const Metric metric,
                     Tensor& neighbors_index,
                     Tensor& neighbors_count,
                     Tensor& neighbors_distance,
                     const HIPFixedRadiusIndex& index) {
    static_assert(std::is_same<TIndex, int32_t>::value,
                  "the ABI returns Int32 indices (NearestNeighborSearch's "
                  "default index dtype; the ICP driver casts to Int64)");
    (void)points;
    (void)points_row_splits;
    (void)queries_row_splits;
    (void)hash_table_splits;
    (void)hash_table_index;
    (void)hash_table_cell_splits;
    if (metric != L2) utility::LogError("HIP hybrid search: L2 metric only");
    std::tie(neighbors_index, neighbors_distance, neighbors_count) =
            index.SearchHybrid(queries, radius, max_knn);
}
template void BuildSpatialHashTableHIP<float>(const Tensor&, double,
                                              const Tensor&, const Tensor&,
                                              Tensor&, Tensor&,
                                              HIPFixedRadiusIndex&);
template void HybridSearchHIP<float, int32_t>(
        const Tensor&, const Tensor&, double, int, const Tensor&,
        const Tensor&, const Tensor&, const Tensor&, const Tensor&,
        const Metric, Tensor&, Tensor&, Tensor&, const HIPFixedRadiusIndex&);

}  // namespace nns
}  // namespace core
}  // namespace open3d
