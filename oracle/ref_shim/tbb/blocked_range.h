// Stand-in for <tbb/blocked_range.h>.
#pragma once
#include <cstddef>
namespace tbb {
template <typename T>
class blocked_range {
public:
    blocked_range(T b, T e, std::size_t grain = 1) : b_(b), e_(e), g_(grain) {}
    T begin() const { return b_; }
    T end() const { return e_; }
    std::size_t grainsize() const { return g_; }
private:
    T b_, e_;
    std::size_t g_;
};
}  // namespace tbb
