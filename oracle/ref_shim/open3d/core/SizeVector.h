// Stand-in for open3d/core/SizeVector.h.
#pragma once
#include <cstdint>
#include <initializer_list>
#include <string>
#include <vector>
namespace open3d {
namespace core {
class SizeVector : public std::vector<int64_t> {
public:
    using std::vector<int64_t>::vector;
    SizeVector() = default;
    SizeVector(const std::vector<int64_t>& v) : std::vector<int64_t>(v) {}
    int64_t NumElements() const {
        int64_t n = 1;
        for (int64_t d : *this) n *= d;
        return n;
    }
    int64_t GetLength() const { return empty() ? 0 : (*this)[0]; }
    std::string ToString() const {
        std::string s = "{";
        for (size_t i = 0; i < size(); ++i)
            s += (i ? ", " : "") + std::to_string((*this)[i]);
        return s + "}";
    }
};
}  // namespace core
}  // namespace open3d
