#include "mailbox.h"

#include <chrono>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace o3dmi {

Mailbox* ThreadMailbox(int which) {
    thread_local Mailbox mbs[kThreadMailboxes];
    thread_local bool tries[kThreadMailboxes] = {};
    if (which < 0 || which >= kThreadMailboxes) return nullptr;
    Mailbox& mb = mbs[which];
    bool& tried = tries[which];
    if (!tried) {
        tried = true;
        void* p = nullptr;
        if (hipHostMalloc(&p, 512, hipHostMallocMapped | hipHostMallocCoherent) ==
            hipSuccess) {
            mb.data = (double*)p;
            mb.flag = (int*)((char*)p + 384);
            *mb.flag = 0;
            mb.seq = 0;
        }
    }
    return mb.data ? &mb : nullptr;
}

namespace {
hipError_t WaitWord(const int* flag, int seq, hipStream_t s);
}

hipError_t MailboxWait(Mailbox* mb, int seq, hipStream_t s) {
    return WaitWord(mb->flag, seq, s);
}

hipError_t MailboxWaitSealed(Mailbox* mb, int seq, hipStream_t s,
                             double* out32) {
    using clock = std::chrono::steady_clock;
    hipError_t e = WaitWord(mb->flag, seq, s);
    if (e != hipSuccess) return e;
    const auto give_up = clock::now() + std::chrono::seconds(2);
    for (;;) {
        unsigned long long v[33], x = 0;
        const volatile unsigned long long* src =
                (const volatile unsigned long long*)mb->data;
        for (int k = 0; k < 33; ++k) v[k] = src[k];
        for (int k = 0; k < 32; ++k) x ^= v[k];
        if ((x ^ MailSeal(seq)) == v[32]) {
            std::memcpy(out32, v, sizeof(double) * 32);
            return hipSuccess;
        }
        // the sequence word overtook part of the block: read again
        if (clock::now() > give_up) return hipErrorUnknown;
#if defined(__x86_64__)
        _mm_pause();
#endif
    }
}

namespace {
hipError_t WaitWord(const int* flag, int seq, hipStream_t s) {
    using clock = std::chrono::steady_clock;
    auto next_query = clock::now() + std::chrono::milliseconds(2);
    for (;;) {
        for (int spin = 0; spin < 256; ++spin) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq)
                return hipSuccess;
#if defined(__x86_64__)
            _mm_pause();
#endif
        }
        if (clock::now() >= next_query) {
            // A failed launch / device fault never posts: ask the stream.
            hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq)
                    return hipSuccess;
                return hipErrorUnknown;  // finished without posting
            }
            if (e != hipErrorNotReady) return e;
            next_query = clock::now() + std::chrono::milliseconds(2);
        }
    }
}
}  // namespace

}  // namespace o3dmi
