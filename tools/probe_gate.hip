// Probe: how fast can the HOST release a kernel that is already resident and
// polling a flag -- with the flag in host memory (every poll crosses PCIe:
// round 3's gated ICP launch, 2x slower than relaunching) or in DEVICE memory
// that the host writes through the BAR (polls stay on the device, only the
// release crosses PCIe)? Prints, per variant, the host-observed time from the
// releasing store to the kernel's own "done" post (a host-mapped word), for a
// grid of `wgs` polling workgroups.
//
//   tools/probe_gate <variant> [wgs]     variant: host | fine | managed
// (one variant per process: a host store to memory that is not host-visible
// faults)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_gate.hip -o tools/probe_gate
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(2);                                               \
        }                                                               \
    } while (0)

__global__ void GateKernel(const int* gate, int seq, int* tickets,
                           int* done_host, float* sink) {
    __shared__ int go;
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(gate, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (++spins > (1 << 22)) break;  // ~ a second: never hang the box
            __builtin_amdgcn_s_sleep(2);
        }
        go = 1;
    }
    __syncthreads();
    // the 16 floats behind the flag: what a gated search launch would read
    float acc = 0;
    if (go) {
        const float* m = (const float*)(gate + 16);
        for (int k = 0; k < 16; ++k) acc += m[k];
    }
    if (acc == 12345.f) *sink = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (__hip_atomic_fetch_add(tickets, 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) ==
            (int)gridDim.x - 1) {
            *tickets = 0;
            __hip_atomic_store(done_host, seq, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int main(int argc, char** argv) {
    const char* variant = argc > 1 ? argv[1] : "host";
    const int wgs = argc > 2 ? std::atoi(argv[2]) : 256;
    int* gate = nullptr;  // 64 ints: flag + 16 floats at +16
    if (!std::strcmp(variant, "host")) {
        CHECK(hipHostMalloc((void**)&gate, 256,
                            hipHostMallocMapped | hipHostMallocCoherent));
    } else if (!std::strcmp(variant, "fine")) {
        CHECK(hipExtMallocWithFlags((void**)&gate, 256,
                                    hipDeviceMallocFinegrained));
    } else if (!std::strcmp(variant, "uncached")) {
        CHECK(hipExtMallocWithFlags((void**)&gate, 256,
                                    hipDeviceMallocUncached));
    } else {
        CHECK(hipMallocManaged((void**)&gate, 256));
        CHECK(hipMemAdvise(gate, 256, hipMemAdviseSetPreferredLocation, 0));
        CHECK(hipMemPrefetchAsync(gate, 256, 0, 0));
    }
    hipPointerAttribute_t at;
    CHECK(hipPointerGetAttributes(&at, gate));
    std::printf("{\"variant\": \"%s\", \"wgs\": %d, \"memory_type\": %d, "
                "\"host_ptr\": %d, ", variant, wgs, (int)at.type,
                at.hostPointer != nullptr);
    std::fflush(stdout);
    int *tickets, *done;
    float* sink;
    CHECK(hipMalloc((void**)&tickets, 4));
    CHECK(hipMemset(tickets, 0, 4));
    CHECK(hipMalloc((void**)&sink, 4));
    CHECK(hipHostMalloc((void**)&done, 64,
                        hipHostMallocMapped | hipHostMallocCoherent));
    *done = 0;
    // first host store (faults here if the memory is not host-visible)
    volatile int* g = gate;
    g[0] = 0;
    for (int k = 0; k < 16; ++k) ((volatile float*)(gate + 16))[k] = 1.0f;
    CHECK(hipDeviceSynchronize());
    std::vector<double> us;
    for (int it = 1; it <= 60; ++it) {
        hipLaunchKernelGGL(GateKernel, dim3(wgs), dim3(64), 0, 0, gate, it,
                           tickets, done, sink);
        // let it become resident and start polling
        std::this_thread::sleep_for(std::chrono::microseconds(150));
        const auto t0 = std::chrono::steady_clock::now();
        ((volatile float*)(gate + 16))[it & 15] = (float)it;
        _mm_sfence();  // BAR memory is write-combining: push the data out
        __atomic_store_n((int*)gate, it, __ATOMIC_RELEASE);
        _mm_sfence();  // ... and the flag
        while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != it) {
        }
        const auto t1 = std::chrono::steady_clock::now();
        us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        CHECK(hipDeviceSynchronize());
    }
    std::sort(us.begin(), us.end());
    // for scale: launch + completion of an EMPTY grid of the same size
    std::vector<double> lu;
    for (int it = 61; it <= 100; ++it) {
        __atomic_store_n((int*)gate, it, __ATOMIC_RELEASE);
        CHECK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(GateKernel, dim3(wgs), dim3(64), 0, 0, gate, it,
                           tickets, done, sink);
        while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != it) {
        }
        const auto t1 = std::chrono::steady_clock::now();
        lu.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(lu.begin(), lu.end());
    std::printf("\"release_to_done_us_median\": %.2f, \"min\": %.2f, "
                "\"p90\": %.2f, \"launch_to_done_us_median\": %.2f}\n",
                us[us.size() / 2], us[0], us[us.size() * 9 / 10],
                lu[lu.size() / 2]);
    return 0;
}
