// Stand-in for open3d/utility/Parallel.h.
#pragma once
#ifdef _OPENMP
#include <omp.h>
#endif
namespace open3d {
namespace utility {
inline int EstimateMaxThreads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
inline bool InParallel() { return false; }
constexpr int DefaultGrainSizeTBB() { return 256; }
constexpr int DefaultGrainSizeTBB2D() { return 32; }
}  // namespace utility
}  // namespace open3d
