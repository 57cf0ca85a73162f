// Host mailbox for the per-iteration sums of the Gauss-Newton drivers (ICP,
// RGB-D odometry). The final-sum kernel writes its <= 32 float64 results
// straight into host-mapped pinned memory and then release-stores a sequence
// number; the host spins on that word. This replaces hipMemcpyAsync +
// hipStreamSynchronize per iteration (two runtime calls and an interrupt-driven
// wait, ~25 us) with a PCIe write the host observes within ~2 us of the kernel
// finishing -- the iteration loop is latency bound, not bandwidth bound.
#pragma once

#include <hip/hip_runtime.h>

namespace o3dmi {

struct Mailbox {
    double* data = nullptr;  // [32], host-mapped (device-visible address)
    int* flag = nullptr;     // host-mapped sequence word
    int seq = 0;             // last sequence number handed out
};

// One mailbox per host thread, allocated on first use and kept for the life of
// the process (a driver call is synchronous on its host thread, so a thread
// never has two iterations in flight). nullptr if the allocation failed.
// `which`: 0 = the per-iteration sums of a driver; 1, 2 = the level counts of
// the source / target pyramid chain of the ICP driver (three independent
// sequences: the chains run on two streams while nothing else is waited for).
constexpr int kThreadMailboxes = 3;
Mailbox* ThreadMailbox(int which = 0);

// Blocks until the kernel that was given `seq` has posted. Returns hipSuccess,
// or the stream's error if the stream finished / failed without posting.
hipError_t MailboxWait(Mailbox* mb, int seq, hipStream_t s);

// SEALED post (the final-sum tail of the ICP search launch, icp.hip
// RowSumTail): 32 float64 + data[32] = XOR of their bit patterns ^
// MailSeal(seq), written with write-through system-scope stores and NO release
// fence, then the sequence word. MailboxWaitSealed accepts the block only when
// the seal fits the values it read, so the order in which the stores land in
// host memory does not matter (a torn read fails the seal and is read again).
__host__ __device__ inline unsigned long long MailSeal(int seq) {
    return 0x9E3779B97F4A7C15ull * (unsigned long long)(unsigned)(seq + 1);
}
// Blocks until launch `seq` has posted a sealed block; copies its 32 values.
hipError_t MailboxWaitSealed(Mailbox* mb, int seq, hipStream_t s,
                             double* out32);

// Device side of a sealed post by ONE FULL WAVE (64 lanes, all active): lane
// k < 32 hands in value k. For a post from inside a launch that is still
// running (no fence, no workgroup barrier).
__device__ __forceinline__ void MailboxPostSealedWave(double* data, int* flag,
                                                      int seq, double value) {
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long b =
            lane < 32 ? (unsigned long long)__double_as_longlong(value) : 0ull;
    unsigned long long x = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) x ^= __shfl_xor(x, d, 64);
    if (lane < 32)
        __hip_atomic_store((unsigned long long*)data + lane, b,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane == 32)
        __hip_atomic_store((unsigned long long*)data + 32, x ^ MailSeal(seq),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0)
        __hip_atomic_store(flag, seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// Device side: called by the threads of the single final workgroup after they
// wrote data[0..n); publishes `seq`.
__device__ __forceinline__ void MailboxPublish(int* flag, int seq) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace o3dmi
