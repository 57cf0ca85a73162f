// Stand-in for <tbb/parallel_reduce.h> (functional form).
//
// The real scheduler's split points are unspecified, so results that depend on
// them (float rounding of sums) are "parity unpinned" in the oracle's header.
// Two valid TBB schedules are offered:
//   * default: the whole range as a single chunk, reduced left to right;
//   * shim::SetSchedule(seed, min_chunk) with seed != 0: the range is split
//     recursively at seeded random points down to chunks of at most min_chunk
//     elements, every chunk starts from the identity, and the partial results
//     are joined left-before-right with the caller's reduction -- what TBB does
//     with some partition of the range, one draw per seed. Successive calls
//     continue the same random stream, so one seed fixes a whole run.
// tests/test_icp_gpu.py runs the reference's own float32 body under many seeds
// to measure the pose spread of the reference itself.
#pragma once
#include <cstdint>

#include "tbb/blocked_range.h"
namespace tbb {
namespace shim {
struct Schedule {
    std::uint64_t state = 0;  // 0 = single sequential chunk
    long long min_chunk = 1024;
};
inline Schedule& TheSchedule() {
    static Schedule s;
    return s;
}
inline void SetSchedule(std::uint64_t seed, long long min_chunk) {
    TheSchedule().state = seed ? seed * 0x9E3779B97F4A7C15ull + 1 : 0;
    TheSchedule().min_chunk = min_chunk < 1 ? 1 : min_chunk;
}
inline std::uint64_t Next() {  // xorshift64*
    std::uint64_t& x = TheSchedule().state;
    x ^= x >> 12;
    x ^= x << 25;
    x ^= x >> 27;
    return x * 0x2545F4914F6CDD1Dull;
}
template <typename T, typename Value, typename RealBody, typename Reduction>
Value Reduce(T b, T e, const Value& identity, const RealBody& body,
             const Reduction& reduction) {
    const long long n = (long long)(e - b);
    if (n <= TheSchedule().min_chunk)
        return body(blocked_range<T>(b, e), identity);
    const T m = b + (T)(1 + (long long)(Next() % (std::uint64_t)(n - 1)));
    Value left = Reduce(b, m, identity, body, reduction);
    Value right = Reduce(m, e, identity, body, reduction);
    return reduction(left, right);
}
}  // namespace shim

template <typename Range, typename Value, typename RealBody, typename Reduction>
Value parallel_reduce(const Range& range, const Value& identity,
                      const RealBody& body, const Reduction& reduction) {
    if (shim::TheSchedule().state == 0) return body(range, identity);
    return shim::Reduce(range.begin(), range.end(), identity, body, reduction);
}
}  // namespace tbb
