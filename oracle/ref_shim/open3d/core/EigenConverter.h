// Stand-in for open3d/core/EigenConverter.h. Eigen is not available here;
// the only user in the files compiled through this shim is
// PoseToSymmetricTransformation (TransformationConverter.cpp:106-132), which
// is not on the point-to-plane path: the types below exist so that it
// compiles, and calling it reports an error.
#pragma once
#include <cstring>
#include <type_traits>
#include "open3d/core/Tensor.h"
namespace Eigen {
struct Vector3d { double v[3]; };
struct Vector6d { double v[6]; };
struct Matrix4d { double v[16]; };
template <typename T>
struct Map {
    using Plain = std::remove_const_t<T>;
    explicit Map(const double* p) { std::memcpy(&m_, p, sizeof(Plain)); }
    operator Plain() const { return m_; }
    Plain m_;
};
}  // namespace Eigen
namespace open3d {
namespace core {
namespace eigen_converter {
inline Tensor EigenMatrixToTensor(const Eigen::Matrix4d&) {
    utility::LogError("shim: Eigen is not available");
}
}  // namespace eigen_converter
}  // namespace core
}  // namespace open3d
