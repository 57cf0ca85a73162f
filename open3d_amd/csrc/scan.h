// Device-wide prefix sum (scan.hip): int32 in, int64 out, two launches on the
// caller's stream, no host wait.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace o3dmi {

// Bytes of scratch PrefixSumAsync needs for n elements (one int64 per tile).
size_t ScanScratchBytes(int64_t n);

// out[i] = sum of in[0..i] (inclusive) or in[0..i-1]; *grand_dev (optional)
// receives the sum of all n elements. in / out may not alias.
int PrefixSumAsync(const int32_t* in_dev, int64_t n, bool inclusive,
                   int64_t* out_dev, int64_t* grand_dev, void* scratch_dev,
                   hipStream_t s);

}  // namespace o3dmi
