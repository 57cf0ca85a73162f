// Stand-in for open3d/core/ParallelFor.h: every index in [0, n) visited
// exactly once; OpenMP instead of tbb::parallel_for (ParallelFor.h:87-104).
#pragma once
#include <cstdint>
#include "open3d/core/Device.h"
#include "open3d/utility/Logging.h"
#include "open3d/utility/Parallel.h"
namespace open3d {
namespace core {
template <typename func_t>
void ParallelFor(const Device&, int64_t n, const func_t& func) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) func(i);
}
template <typename vec_func_t, typename func_t>
void ParallelFor(const Device& d, int64_t n, const func_t& func,
                 const vec_func_t&) {
    ParallelFor(d, n, func);
}
}  // namespace core
}  // namespace open3d
