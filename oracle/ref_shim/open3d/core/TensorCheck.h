// Stand-in for open3d/core/TensorCheck.h.
#pragma once
#include <vector>
#include "open3d/core/Tensor.h"
namespace open3d {
namespace core {
inline void AssertTensorShape(const Tensor& t, const SizeVector& s) {
    if (t.GetShape() != s)
        utility::LogError("Tensor has shape {}, but is expected to have {}.");
}
inline void AssertTensorDtype(const Tensor& t, const Dtype& d) {
    if (t.GetDtype() != d)
        utility::LogError("Tensor has dtype {}, but is expected to have {}.");
}
inline void AssertTensorDtypes(const Tensor& t, const std::vector<Dtype>& ds) {
    for (const Dtype& d : ds)
        if (t.GetDtype() == d) return;
    utility::LogError("Tensor has dtype {}, but is expected to be one of {}.");
}
inline void AssertTensorDevice(const Tensor&, const Device&) {}
}  // namespace core
}  // namespace open3d
