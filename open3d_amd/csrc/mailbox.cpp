#include "mailbox.h"

#include <chrono>

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

MailRing* ThreadMailRing() {
    thread_local MailRing ring;
    thread_local bool tried = false;
    if (!tried) {
        tried = true;
        void* p = nullptr;
        const size_t data_bytes = sizeof(double) * kMailSlots * kMailDoubles;
        if (hipHostMalloc(&p, data_bytes + 64 * kMailSlots,
                          hipHostMallocMapped | hipHostMallocCoherent) ==
            hipSuccess) {
            ring.data = (double*)p;
            ring.flags = (int*)((char*)p + data_bytes);
            for (int k = 0; k < kMailSlots; ++k) ring.flags[k * 16] = 0;
            ring.seq = 0;
        }
    }
    return ring.data ? &ring : nullptr;
}

GateInbox* ThreadGateInbox() {
    thread_local GateInbox ib;
    thread_local bool tried = false;
    if (!tried) {
        tried = true;
        void* p = nullptr;
        if (hipHostMalloc(&p, 256, hipHostMallocMapped | hipHostMallocCoherent) ==
            hipSuccess) {
            ib.words = (int*)p;
            for (int k = 0; k < 64; ++k) ib.words[k] = 0;
            ib.seq = 0;
        }
    }
    return ib.words ? &ib : nullptr;
}

void GateInbox::Release(int s, const double* m) const {
    float* m32 = (float*)(words + 16);
    double* m64 = (double*)(words + 32);
    for (int k = 0; k < 16; ++k) {
        m32[k] = (float)m[k];
        m64[k] = m[k];
    }
    __atomic_store_n(&words[0], s, __ATOMIC_RELEASE);
}

void GateInbox::Cancel(int s) const {
    __atomic_store_n(&words[1], s, __ATOMIC_RELAXED);
    __atomic_store_n(&words[0], s, __ATOMIC_RELEASE);
}

namespace {
hipError_t WaitWord(const int* flag, int seq, hipStream_t s);
}

hipError_t MailboxWait(Mailbox* mb, int seq, hipStream_t s) {
    return WaitWord(mb->flag, seq, s);
}

hipError_t MailRingWait(MailRing* ring, int seq, hipStream_t s) {
    return WaitWord(ring->Flag(seq), seq, s);
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
