// Stand-in for open3d/pipelines/registration/SymmetricICPImpl.h (legacy Eigen
// code, off the point-to-plane path).
#pragma once
#include "open3d/core/EigenConverter.h"
namespace open3d {
namespace pipelines {
namespace registration {
inline Eigen::Matrix4d TransformSymmetricPoseToMatrix4d(
        const Eigen::Vector6d&, const Eigen::Vector3d&,
        const Eigen::Vector3d&) {
    utility::LogError("shim: symmetric ICP is not available");
}
}  // namespace registration
}  // namespace pipelines
}  // namespace open3d
