// Device-wide prefix sum and a counting sort of bounded integers -- the two
// primitives the "next" rows need (CSR splits of the radius search behind
// EstimateNormals / EstimateColorGradients; ascending buffer indices behind
// ExtractPointCloud / Save). Written for this use: a string of launches on the
// caller's stream, no host wait, no temporary beyond one int64 per tile.
//
// Scan = two launches over 8192-element tiles. Launch 1 leaves each tile's
// total; launch 2 has every workgroup add up the totals of the tiles before
// it (wave 0, 64 totals in flight per step -- 123 tiles for a million
// elements) and scans its own tile in registers: thread t owns elements
// [8t, 8t + 8) of the tile, so loads and stores are 32-byte runs per lane.
// HBM-bound: n x (4 B read twice + 8 B written); neither caller is on the
// per-frame path.

#include "common.h"
#include "scan.h"

namespace o3dmi {
namespace {

constexpr int kScanBlock = 1024;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;

__global__ void __launch_bounds__(kScanBlock)
TileTotalsKernel(const int32_t* __restrict__ in, int64_t n,
                 long long* __restrict__ totals) {
    __shared__ long long part[kScanBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanTile +
                         (int64_t)threadIdx.x * kScanItems;
    long long s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) s += in[base + k];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
#pragma unroll
        for (int w = 0; w < kScanBlock / 64; ++w) t += part[w];
        totals[blockIdx.x] = t;
    }
}

// out[i] = in[0] + ... + in[i] (kInclusive) or in[0] + ... + in[i - 1].
// *grand (optional) receives the sum of everything.
template <bool kInclusive>
__global__ void __launch_bounds__(kScanBlock)
TileScanKernel(const int32_t* __restrict__ in, int64_t n,
               const long long* __restrict__ totals,
               long long* __restrict__ out, long long* __restrict__ grand) {
    __shared__ long long part[kScanBlock / 64];
    __shared__ long long before;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanTile +
                         (int64_t)threadIdx.x * kScanItems;
    int32_t v[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = base + k < n ? in[base + k] : 0;
    if (wave == 0) {
        long long b = 0, all = 0;
        const int tiles = gridDim.x;
        for (int t = lane; t < tiles; t += 64) {
            const long long c = totals[t];
            all += c;
            b += t < (int)blockIdx.x ? c : 0;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            b += __shfl_xor(b, m);
            all += __shfl_xor(all, m);
        }
        if (lane == 0) {
            before = b;
            if (grand && blockIdx.x == 0) *grand = all;
        }
    }
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) mine += v[k];
    // inclusive scan of the lanes' sums within the wave, then across waves
    long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    long long run = before + incl - mine;
#pragma unroll
    for (int w = 0; w < kScanBlock / 64; ++w)
        if (w < wave) run += part[w];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (kInclusive) run += v[k];
        if (base + k < n) out[base + k] = run;
        if (!kInclusive) run += v[k];
    }
}

// ---- counting sort of integers in [0, bound) ---------------------------------
__global__ void CountValuesKernel(const int32_t* __restrict__ values, int64_t n,
                                  int32_t bound, int32_t* __restrict__ counts,
                                  int* __restrict__ err) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t v = values[i];
        if (v < 0 || v >= bound) {
            *err = 1;
            continue;
        }
        atomicAdd(&counts[v], 1);
    }
}

__global__ void EmitValuesKernel(const int32_t* __restrict__ counts,
                                 const long long* __restrict__ offsets,
                                 int32_t bound, int32_t* __restrict__ out) {
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < bound;
         v += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = counts[v];
        if (c == 0) continue;
        const long long o = offsets[v];
        for (int32_t k = 0; k < c; ++k) out[o + k] = (int32_t)v;
    }
}

__global__ void MaxValueKernel(const int32_t* __restrict__ values, int64_t n,
                               int32_t* __restrict__ result) {
    int32_t m = -1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t v = values[i];
        m = v > m ? v : m;
        if (v < 0) result[1] = 1;
    }
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) {
        const int32_t o = __shfl_xor(m, k);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m >= 0) atomicMax(&result[0], m);
}

}  // namespace

size_t ScanScratchBytes(int64_t n) {
    const int64_t tiles = n > 0 ? (n + kScanTile - 1) / kScanTile : 1;
    return sizeof(long long) * (size_t)tiles;
}

int PrefixSumAsync(const int32_t* in_dev, int64_t n, bool inclusive,
                   int64_t* out_dev, int64_t* grand_dev, void* scratch_dev,
                   hipStream_t s) {
    if (n <= 0) {
        if (grand_dev)
            O3DMI_HIP_CHECK(hipMemsetAsync(grand_dev, 0, sizeof(int64_t), s));
        return O3DMI_OK;
    }
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    O3DMI_REQUIRE(tiles < (1ll << 31), "prefix sum: too many elements");
    long long* totals = (long long*)scratch_dev;
    hipLaunchKernelGGL(TileTotalsKernel, dim3((unsigned)tiles),
                       dim3(kScanBlock), 0, s, in_dev, n, totals);
    if (inclusive)
        hipLaunchKernelGGL(TileScanKernel<true>, dim3((unsigned)tiles),
                           dim3(kScanBlock), 0, s, in_dev, n, totals,
                           (long long*)out_dev, (long long*)grand_dev);
    else
        hipLaunchKernelGGL(TileScanKernel<false>, dim3((unsigned)tiles),
                           dim3(kScanBlock), 0, s, in_dev, n, totals,
                           (long long*)out_dev, (long long*)grand_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

}  // namespace o3dmi

using namespace o3dmi;

// Ascending sort of non-negative 32-bit integers (buffer indices:
// GetActiveIndices returns them in slot order, which differs from run to run).
// Buffer indices are small (< capacity) and mostly distinct, so this is a
// counting sort with one bin per value: largest value -> bins -> histogram ->
// prefix sum -> every value writes itself `count` times. Duplicates survive.
extern "C" int o3dmi_sort_indices(int32_t* indices_dev, int64_t n,
                                  o3dmi_stream_t stream) {
    O3DMI_REQUIRE(n >= 0 && n < (1ll << 31), "n out of range");
    if (n <= 1) return O3DMI_OK;
    O3DMI_REQUIRE(indices_dev != nullptr, "null argument");
    hipStream_t s = (hipStream_t)stream;
    int32_t* head = nullptr;  // {max, negative seen, out of range}
    int st = PoolAlloc((void**)&head, 256);
    if (st) return st;
    int32_t host[2] = {-1, 0};
    hipError_t e = hipMemsetAsync(head, 0xFF, sizeof(int32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(head + 1, 0, 2 * sizeof(int32_t), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(MaxValueKernel, dim3(GridFor(n, kBlock)),
                           dim3(kBlock), 0, s, indices_dev, n, head);
        e = hipMemcpyAsync(host, head, sizeof(host), hipMemcpyDeviceToHost, s);
    }
    hipError_t e2 = hipStreamSynchronize(s);
    if (e != hipSuccess || e2 != hipSuccess) {
        PoolFree(head);
        O3DMI_HIP_CHECK(e);
        O3DMI_HIP_CHECK(e2);
    }
    if (host[1] != 0 || host[0] < 0) {
        PoolFree(head);
        SetLastError("sort_indices: negative index");
        return O3DMI_ERR_INVALID_ARG;
    }
    const int64_t bound = (int64_t)host[0] + 1;
    const size_t cnt_bytes = (sizeof(int32_t) * (size_t)bound + 255) & ~(size_t)255;
    const size_t off_bytes = (sizeof(int64_t) * (size_t)bound + 255) & ~(size_t)255;
    char* scratch = nullptr;
    st = PoolAlloc((void**)&scratch,
                   cnt_bytes + off_bytes + ScanScratchBytes(bound));
    if (st) {
        PoolFree(head);
        return st;
    }
    int32_t* counts = (int32_t*)scratch;
    int64_t* offsets = (int64_t*)(scratch + cnt_bytes);
    e = hipMemsetAsync(counts, 0, cnt_bytes, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(CountValuesKernel, dim3(GridFor(n, kBlock)),
                           dim3(kBlock), 0, s, indices_dev, n, (int32_t)bound,
                           counts, head + 2);
        st = PrefixSumAsync(counts, bound, false, offsets, nullptr,
                            scratch + cnt_bytes + off_bytes, s);
        if (!st)
            hipLaunchKernelGGL(EmitValuesKernel, dim3(GridFor(bound, kBlock)),
                               dim3(kBlock), 0, s, counts,
                               (const long long*)offsets, (int32_t)bound,
                               indices_dev);
        e = hipGetLastError();
    }
    e2 = hipStreamSynchronize(s);  // pooled blocks: stream drained
    PoolFree(scratch);
    PoolFree(head);
    if (st) return st;
    O3DMI_HIP_CHECK(e);
    O3DMI_HIP_CHECK(e2);
    return O3DMI_OK;
}
