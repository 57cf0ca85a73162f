// Stand-in for <tbb/parallel_for.h>: ranges are split into grain-sized chunks
// distributed with OpenMP.
#pragma once
#include <algorithm>
#include <cstdint>
#include "tbb/blocked_range.h"
namespace tbb {
template <typename T, typename Body>
void parallel_for(const blocked_range<T>& r, const Body& body) {
    const int64_t b = (int64_t)r.begin(), e = (int64_t)r.end();
    const int64_t g = std::max<int64_t>(1, (int64_t)r.grainsize());
    const int64_t chunks = (e - b + g - 1) / g;
#pragma omp parallel for schedule(dynamic)
    for (int64_t c = 0; c < chunks; ++c)
        body(blocked_range<T>((T)(b + c * g), (T)std::min(e, b + (c + 1) * g), g));
}
template <typename T, typename Body>
void parallel_for(T b, T e, const Body& body) {
#pragma omp parallel for schedule(static)
    for (int64_t i = (int64_t)b; i < (int64_t)e; ++i) body((T)i);
}
}  // namespace tbb
