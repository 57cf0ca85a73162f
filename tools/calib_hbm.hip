// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// shapes of the frame-stream integrate kernel (MI355X_MICROARCH.md, "HBM":
// FETCH_SIZE is known to report half the bytes of a 16 B/lane coalesced read
// stream; other widths and WRITE_SIZE are uncalibrated).
//
// Every kernel moves a KNOWN number of bytes over buffers far larger than the
// 256 MiB Infinity Cache; tools/profile_step_pmc.sh runs this binary under
// `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and
// tools/summarize_calib.py divides the expected bytes by the counter values.
//
//   read8 / read16 / read24    lane i reads 8 / 16 / 24 contiguous bytes
//   write8 / write16 / write24 lane i writes them
//   state_rw                   the integrate role's state access: per lane
//                              16 B (tsdf) + 8 B (weight) + 24 B (colour)
//                              read and written back, three arrays
//   gather8                    8-byte gathers out of a 2.4 MB image with the
//                              locality of projected voxels (L2-resident)
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib_hbm.hip -o tools/calib_hbm
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                               \
        }                                                               \
    } while (0)

template <int N>
struct alignas(8) Words {
    uint2 w[N];
};

template <int N>  // N x 8 bytes per lane
__global__ void ReadKernel(const Words<N>* __restrict__ src, int64_t n,
                           unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const Words<N> v = src[i];
#pragma unroll
        for (int k = 0; k < N; ++k) acc += v.w[k].x ^ v.w[k].y;
    }
    if (acc == 0x12345678u) *sink = acc;  // keeps the loads alive
}

template <int N>
__global__ void WriteKernel(Words<N>* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        Words<N> v;
#pragma unroll
        for (int k = 0; k < N; ++k) v.w[k] = make_uint2((unsigned)i, k);
        dst[i] = v;
    }
}

__global__ void StateRwKernel(float4* __restrict__ tsdf,
                              uint2* __restrict__ weight,
                              Words<3>* __restrict__ color, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 t = tsdf[i];
        uint2 w = weight[i];
        Words<3> c = color[i];
        t.x += 1.0f;
        w.x += 1u;
        c.w[0].x += 1u;
        tsdf[i] = t;
        weight[i] = w;
        color[i] = c;
    }
}

__global__ void GatherKernel(const uint2* __restrict__ image, int cols,
                             int rows, int64_t n, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        // a 16 x 16 patch of "voxels" per workgroup-iteration, 3 pixels apart
        const int lane = (int)(i & 255);
        const int64_t tile = i >> 8;
        const int u0 = (int)((tile * 37) % (cols - 64));
        const int v0 = (int)((tile * 101) % (rows - 64));
        const int u = u0 + 3 * (lane & 15), v = v0 + 3 * (lane >> 4);
        const uint2 r = image[(int64_t)v * cols + u];
        acc += r.x ^ r.y;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const int64_t bytes = 2ll << 30;  // 2 GiB per stream
    void *a = nullptr, *b = nullptr, *c = nullptr;
    unsigned* sink = nullptr;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&c, bytes));
    CHECK(hipMalloc((void**)&sink, 4));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 2, bytes));
    CHECK(hipMemset(c, 3, bytes));
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(ReadKernel<1>, grid, block, 0, 0,
                           (const Words<1>*)a, bytes / 8, sink);
        hipLaunchKernelGGL(ReadKernel<2>, grid, block, 0, 0,
                           (const Words<2>*)a, bytes / 16, sink);
        hipLaunchKernelGGL(ReadKernel<3>, grid, block, 0, 0,
                           (const Words<3>*)a, bytes / 24, sink);
        hipLaunchKernelGGL(WriteKernel<1>, grid, block, 0, 0, (Words<1>*)b,
                           bytes / 8);
        hipLaunchKernelGGL(WriteKernel<2>, grid, block, 0, 0, (Words<2>*)b,
                           bytes / 16);
        hipLaunchKernelGGL(WriteKernel<3>, grid, block, 0, 0, (Words<3>*)b,
                           bytes / 24);
        // 48 bytes of state per lane over three arrays: n lanes
        const int64_t n = (bytes / 24);  // colour array is the largest
        hipLaunchKernelGGL(StateRwKernel, grid, block, 0, 0, (float4*)a,
                           (uint2*)b, (Words<3>*)c, n);
        hipLaunchKernelGGL(GatherKernel, grid, block, 0, 0, (const uint2*)a,
                           640, 480, (int64_t)11600000, sink);
    }
    CHECK(hipDeviceSynchronize());
    // expected bytes per launch, for the summariser
    const int64_t n = bytes / 24;
    std::printf("{\"ReadKernel<1>\": {\"read\": %lld, \"write\": 0},\n",
                (long long)(bytes / 8 * 8));
    std::printf(" \"ReadKernel<2>\": {\"read\": %lld, \"write\": 0},\n",
                (long long)(bytes / 16 * 16));
    std::printf(" \"ReadKernel<3>\": {\"read\": %lld, \"write\": 0},\n",
                (long long)(bytes / 24 * 24));
    std::printf(" \"WriteKernel<1>\": {\"read\": 0, \"write\": %lld},\n",
                (long long)(bytes / 8 * 8));
    std::printf(" \"WriteKernel<2>\": {\"read\": 0, \"write\": %lld},\n",
                (long long)(bytes / 16 * 16));
    std::printf(" \"WriteKernel<3>\": {\"read\": 0, \"write\": %lld},\n",
                (long long)(bytes / 24 * 24));
    std::printf(" \"StateRwKernel\": {\"read\": %lld, \"write\": %lld},\n",
                (long long)(n * 48), (long long)(n * 48));
    std::printf(" \"GatherKernel\": {\"read\": %lld, \"write\": 0, "
                "\"note\": \"2.4 MB image, cache-resident: expected fabric "
                "traffic ~ the image once\"}}\n",
                (long long)(640 * 480 * 8));
    return 0;
}
