// Stand-in for <tbb/parallel_reduce.h> (functional form): one valid TBB
// schedule -- the whole range as a single chunk, reduced left to right.
// The real scheduler's split points are unspecified; results that depend on
// them (float rounding of sums) are "parity unpinned" in the oracle's header.
#pragma once
#include "tbb/blocked_range.h"
namespace tbb {
template <typename Range, typename Value, typename RealBody, typename Reduction>
Value parallel_reduce(const Range& range, const Value& identity,
                      const RealBody& body, const Reduction&) {
    return body(range, identity);
}
}  // namespace tbb
