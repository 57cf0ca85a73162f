// Stand-in for open3d/t/geometry/TensorMap.h: name -> Tensor.
#pragma once
#include <string>
#include <unordered_map>
#include "open3d/core/Tensor.h"
namespace open3d {
namespace t {
namespace geometry {
class TensorMap : public std::unordered_map<std::string, core::Tensor> {
public:
    TensorMap() = default;
    explicit TensorMap(const std::string& primary) : primary_(primary) {}
    bool Contains(const std::string& k) const { return count(k) != 0; }
private:
    std::string primary_;
};
}  // namespace geometry
}  // namespace t
}  // namespace open3d
