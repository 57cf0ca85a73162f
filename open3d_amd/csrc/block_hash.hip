// Spatial hash of voxel-block keys for MI355X: open addressing with linear
// probing over packed 64-bit keys (one CAS decides slot ownership), buffer
// indices handed out from a free-index heap exactly as Open3D's
// HashBackendBuffer does (core/hashmap/HashBackendBuffer.cpp:16-78,
// CPUHashBackendBufferAccessor.hpp:39-42: heap initialised to identity,
// allocate = heap[top++], free = heap[--top] = idx).
//
// Semantics follow DeviceHashBackend (core/hashmap/DeviceHashBackend.h:20-107)
// as exercised through HashMap::{Activate,Insert,Find,Erase,GetActiveIndices,
// Reserve,Clear} (core/hashmap/HashMap.cpp:47-216); the probing scheme itself
// is new (the reference's GPU backends are stdgpu / slab hash).
//
// Load factor <= 0.5 (n_slots = next_pow2(2 * capacity)), so probe sequences
// stay within one or two 64-byte lines.

#include <vector>

#include "common.h"

namespace o3dmi {

static thread_local std::string g_last_error;
void SetLastError(const std::string& msg) { g_last_error = msg; }

namespace {

// Clear(): every slot empty, the free-index heap the identity, the counters
// zero -- one launch (three fills and a heap launch until round 5: 19 us in
// front of every frame's DepthTouch, whose own kernel takes 14;
// profiles/r5r_api_legs.txt). `also_zero`: a caller's counter cleared in the
// same launch (the touch kernels' output count), may be null.
__global__ void ClearKernel(HashView v, long long n_slots, int capacity,
                            int* also_zero) {
    const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long i = i0; i < n_slots; i += step) v.slot_keys[i] = ~0ull;
    for (long long i = i0; i < capacity; i += step) v.heap[i] = (int)i;
    if (i0 < 4) v.counters[i0] = 0;
    if (i0 == 4 && also_zero) *also_zero = 0;
}

// Insert-if-absent. One thread per input key; duplicates in the same launch
// resolve through the CAS: exactly one thread per distinct new key wins.
template <bool kHasValues>
__global__ void ActivateKernel(HashView hv, const int* __restrict__ keys,
                               int64_t n, const int* __restrict__ n_dev,
                               int* __restrict__ buf_indices,
                               uint8_t* __restrict__ masks, int n_values,
                               const void* const* values_src,
                               void* const* values_dst,
                               const int64_t* value_dsizes) {
    if (n_dev) {
        int64_t live = *n_dev;
        n = live < n ? live : n;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int x = keys[3 * i + 0], y = keys[3 * i + 1], z = keys[3 * i + 2];
        int out_idx = 0;
        uint8_t out_mask = 0;
        if (!KeyInRange(x, y, z)) {
            atomicOr(&hv.counters[1], kErrKeyRange);
        } else {
            unsigned h = 0;
            // 1 = this thread created the entry (duplicates in one launch:
            // exactly one thread per distinct new key)
            if (ClaimSlot(hv, PackKey(x, y, z), h) == 1) {
                int top = atomicAdd(&hv.counters[0], 1);
                if (top >= hv.capacity) {
                    atomicOr(&hv.counters[1], kErrCapacity);
                } else {
                    int idx = hv.heap[top];
                    hv.key_buffer[3 * idx + 0] = x;
                    hv.key_buffer[3 * idx + 1] = y;
                    hv.key_buffer[3 * idx + 2] = z;
                    hv.slot_vals[h] = idx;
                    if (kHasValues) {
                        for (int j = 0; j < n_values; ++j) {
                            int64_t sz = value_dsizes[j];
                            const uint8_t* s =
                                    (const uint8_t*)values_src[j] + sz * i;
                            uint8_t* d = (uint8_t*)values_dst[j] +
                                         sz * (int64_t)idx;
                            for (int64_t b = 0; b < sz; ++b) d[b] = s[b];
                        }
                    }
                    out_idx = idx;
                    out_mask = 1;
                }
            }
        }
        if (buf_indices) buf_indices[i] = out_idx;
        if (masks) masks[i] = out_mask;
    }
}

__global__ void FindKernel(HashView hv, const int* __restrict__ keys,
                           int64_t n, const int* __restrict__ n_dev,
                           int* __restrict__ buf_indices,
                           uint8_t* __restrict__ masks) {
    if (n_dev) {
        int64_t live = *n_dev;
        n = live < n ? live : n;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int idx = hv.Find(keys[3 * i + 0], keys[3 * i + 1], keys[3 * i + 2]);
        if (buf_indices) buf_indices[i] = idx < 0 ? 0 : idx;
        if (masks) masks[i] = idx >= 0;
    }
}

__global__ void EraseKernel(HashView hv, const int* __restrict__ keys,
                            int64_t n, uint8_t* __restrict__ masks) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int x = keys[3 * i + 0], y = keys[3 * i + 1], z = keys[3 * i + 2];
        uint8_t ok = 0;
        if (KeyInRange(x, y, z)) {
            unsigned long long k = PackKey(x, y, z);
            unsigned h = HashKey(k) & hv.mask;
            for (unsigned step = 0; step <= hv.mask; ++step) {
                unsigned long long cur = hv.slot_keys[h];
                if (cur == kEmptyKey) break;
                if (cur == k) {
                    // One winner among duplicate erase requests.
                    if (atomicCAS(&hv.slot_keys[h], k, kTombKey) == k) {
                        int idx = hv.slot_vals[h];
                        int top = atomicSub(&hv.counters[0], 1);
                        hv.heap[top - 1] = idx;
                        ok = 1;
                    }
                    break;
                }
                h = (h + 1) & hv.mask;
            }
        }
        if (masks) masks[i] = ok;
    }
}

// Compacts the buffer indices of all occupied slots (unordered), one
// wave-aggregated atomic per wavefront.
__global__ void ActiveIndicesKernel(HashView hv, int64_t n_slots, int* out,
                                    int* count) {
    for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
         s < ((n_slots + 63) / 64) * 64; s += (int64_t)gridDim.x * blockDim.x) {
        bool occ = false;
        if (s < n_slots) {
            unsigned long long k = hv.slot_keys[s];
            occ = (k != kEmptyKey) && (k != kTombKey);
        }
        unsigned long long ballot = __ballot(occ);
        int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == 0 && ballot) base = atomicAdd(count, __popcll(ballot));
        base = __shfl(base, 0);
        if (occ) {
            int off = __popcll(ballot & ((1ull << lane) - 1ull));
            out[base + off] = hv.slot_vals[s];
        }
    }
}

// Slot-table rebuild (same capacity, buffer indices unchanged): puts the keys
// of the listed buffer indices back into a cleared table.
__global__ void ReinsertKernel(HashView hv, const int* __restrict__ active,
                               int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int idx = active[i];
        unsigned h = 0;
        if (ClaimSlot(hv,
                      PackKey(hv.key_buffer[3 * idx + 0],
                              hv.key_buffer[3 * idx + 1],
                              hv.key_buffer[3 * idx + 2]),
                      h) == 1)
            hv.slot_vals[h] = idx;
    }
}

// RecoverOverflow (stream_path.h): slots whose insert found no buffer index
// (marker -1) become tombstones; one thread settles the counters.
__global__ void RecoverOverflowKernel(HashView hv, int64_t n_slots,
                                      int* __restrict__ wanted) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
         i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = hv.slot_keys[i];
        if (k != kEmptyKey && k != kTombKey && hv.slot_vals[i] == -1)
            hv.slot_keys[i] = kTombKey;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *wanted = hv.counters[0];
        if (hv.counters[0] > hv.capacity) hv.counters[0] = hv.capacity;
        hv.counters[3] = 0;
    }
}

template <typename T>
int DevAlloc(T** p, int64_t n) {
    O3DMI_HIP_CHECK(hipMalloc((void**)p, (size_t)(n > 0 ? n : 1) * sizeof(T)));
    return O3DMI_OK;
}

int64_t NextPow2(int64_t v) {
    int64_t p = 64;
    while (p < v) p <<= 1;
    return p;
}

int AllocateStorage(o3dmi_hash* h, int64_t capacity, hipStream_t s) {
    h->capacity = capacity;
    h->n_slots = NextPow2(2 * capacity);
    HashView& v = h->view;
    int st;
    if ((st = DevAlloc(&v.slot_keys, h->n_slots))) return st;
    if ((st = DevAlloc(&v.slot_vals, h->n_slots))) return st;
    if ((st = DevAlloc(&v.slot_touch, 2 * h->n_slots))) return st;
    if ((st = DevAlloc(&v.heap, capacity))) return st;
    if ((st = DevAlloc(&v.counters, 4))) return st;
    if ((st = DevAlloc(&v.key_buffer, capacity * 3))) return st;
    if (!h->scratch_count)
        if ((st = DevAlloc(&h->scratch_count, 4))) return st;
    v.mask = (unsigned)(h->n_slots - 1);
    v.capacity = (int)capacity;
    for (int j = 0; j < h->n_values; ++j) {
        size_t bytes = (size_t)h->value_dsizes[j] * (size_t)capacity;
        O3DMI_HIP_CHECK(hipMalloc(&h->value_buffers[j], bytes ? bytes : 1));
        O3DMI_HIP_CHECK(hipMemsetAsync(h->value_buffers[j], 0, bytes, s));
    }
    O3DMI_HIP_CHECK(hipMemsetAsync(v.key_buffer, 0,
                                   sizeof(int) * 3 * (size_t)capacity, s));
    O3DMI_HIP_CHECK(hipMemsetAsync(v.slot_touch, 0,
                                   sizeof(unsigned long long) * 2 *
                                           (size_t)h->n_slots, s));
    return O3DMI_OK;
}

void FreeStorage(o3dmi_hash* h) {
    HashView& v = h->view;
    (void)hipFree(v.slot_keys);
    (void)hipFree(v.slot_vals);
    (void)hipFree(v.slot_touch);
    (void)hipFree(v.heap);
    (void)hipFree(v.counters);
    (void)hipFree(v.key_buffer);
    for (int j = 0; j < h->n_values; ++j) {
        (void)hipFree(h->value_buffers[j]);
        h->value_buffers[j] = nullptr;
    }
    // Storage is gone; the sharding role of the map is not (Reserve re-uses
    // the object).
    const int owner_rank = v.owner_rank, owner_world = v.owner_world;
    v = HashView{};
    v.owner_rank = owner_rank;
    v.owner_world = owner_world;
}

int ClearImpl(o3dmi_hash* h, hipStream_t s, int* also_zero = nullptr) {
    // 8 slots per thread: a 50 000-block map (2^17 slots) is 64 workgroups
    int64_t groups = (h->n_slots / 8 + kBlock - 1) / kBlock;
    if (groups < 1) groups = 1;
    if (groups > 4096) groups = 4096;
    hipLaunchKernelGGL(ClearKernel, dim3((unsigned)groups), dim3(kBlock), 0, s,
                       h->view, (long long)h->n_slots, (int)h->capacity,
                       also_zero);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

// Erase leaves tombstones; inserts reuse the ones on their probe path, the
// others stay. Once live + tombstone slots pass 3/4 of the table, the table is
// rebuilt in place from the key buffer (live keys <= capacity <= half the
// slots), so walks always end at an empty slot.
bool SlotsCrowded(const o3dmi_hash* h, int slots_taken) {
    return (int64_t)slots_taken * 4 >= h->n_slots * 3;
}

int RebuildSlots(o3dmi_hash* h, hipStream_t s) {
    int* active = nullptr;
    O3DMI_HIP_CHECK(hipMalloc((void**)&active, sizeof(int) * h->capacity));
    O3DMI_HIP_CHECK(hipMemsetAsync(h->scratch_count, 0, sizeof(int), s));
    hipLaunchKernelGGL(ActiveIndicesKernel, dim3(GridFor(h->n_slots, kBlock)),
                       dim3(kBlock), 0, s, h->view, h->n_slots, active,
                       h->scratch_count);
    int n = 0;
    O3DMI_HIP_CHECK(hipMemcpyAsync(&n, h->scratch_count, sizeof(int),
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    HashView& v = h->view;
    O3DMI_HIP_CHECK(hipMemsetAsync(v.slot_keys, 0xFF,
                                   sizeof(unsigned long long) *
                                           (size_t)h->n_slots, s));
    O3DMI_HIP_CHECK(hipMemsetAsync(v.slot_touch, 0,
                                   sizeof(unsigned long long) * 2 *
                                           (size_t)h->n_slots, s));
    O3DMI_HIP_CHECK(hipMemsetAsync(v.counters + 2, 0, sizeof(int), s));
    if (n > 0)
        hipLaunchKernelGGL(ReinsertKernel, dim3(GridFor(n, kBlock)),
                           dim3(kBlock), 0, s, v, active, (int64_t)n);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(active);
    return O3DMI_OK;
}

int RebuildSlotsIfCrowded(o3dmi_hash* h, hipStream_t s) {
    int host[3] = {0, 0, 0};
    O3DMI_HIP_CHECK(hipMemcpyAsync(host, h->view.counters, sizeof(host),
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (!SlotsCrowded(h, host[2])) return O3DMI_OK;
    return RebuildSlots(h, s);
}

// Reads the deferred error flags (and the size). Inserts after an Erase take
// empty slots whenever their probe path holds no tombstone, so live +
// tombstone slots can pass the 3/4 mark without another Erase: every caller
// that waits for the counters anyway (Size, and through it the frame stream's
// capacity policy and Reserve) also rebuilds a crowded table here.
int CheckDeferred(o3dmi_hash* h, hipStream_t s, int* top_out) {
    // into pinned memory: one asynchronous copy and one wait (a copy into
    // pageable memory is staged by the runtime, a second round trip)
    if (!h->counters_host &&
        hipHostMalloc((void**)&h->counters_host, sizeof(int) * 4) !=
                hipSuccess) {
        (void)hipGetLastError();
        h->counters_host = nullptr;
    }
    int stack[4] = {0, 0, 0, 0};
    int* host = h->counters_host ? h->counters_host : stack;
    O3DMI_HIP_CHECK(hipMemcpyAsync(host, h->view.counters, sizeof(int) * 4,
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (top_out) *top_out = host[0];
    if (host[3] != 0) {
        // a frame-stream group ran out of buffer indices and the stream has
        // not been recovered yet (RecoverOverflow): the slot table holds keys
        // without a block; nothing may size or export the map before
        SetLastError("hash map: frame-stream overflow not recovered");
        return O3DMI_ERR_INTERNAL;
    }
    if (host[1] & kErrKeyRange) {
        SetLastError("block coordinate outside +-2^20");
        return O3DMI_ERR_KEY_RANGE;
    }
    if (host[1] & kErrCapacity) {
        SetLastError("hash map capacity exceeded (caller must Reserve first)");
        return O3DMI_ERR_CAPACITY;
    }
    if (host[1] & (kErrProbe | kErrTouchStamp)) {
        SetLastError(host[1] & kErrProbe
                             ? "hash map probe sequence wrapped (table full)"
                             : "frame-stream touch word of another group");
        return O3DMI_ERR_INTERNAL;
    }
    if (SlotsCrowded(h, host[2])) return RebuildSlots(h, s);
    return O3DMI_OK;
}


}  // namespace

int ClearHashAndCounter(o3dmi_hash* h, int* counter_dev, hipStream_t s) {
    return ClearImpl(h, s, counter_dev);
}

int RecoverOverflow(o3dmi_hash* h, hipStream_t s, int64_t* wanted) {
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(RecoverOverflowKernel,
                       dim3(GridFor(h->n_slots, kBlock)), dim3(kBlock), 0, s,
                       h->view, h->n_slots, h->scratch_count);
    O3DMI_HIP_CHECK(hipGetLastError());
    int w = 0;
    O3DMI_HIP_CHECK(hipMemcpyAsync(&w, h->scratch_count, sizeof(int),
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    if (wanted) *wanted = w;
    return O3DMI_OK;
}

// o3dmi_preload: HIP loads this translation unit's code object at the first
// launch of one of its kernels; asking for a kernel's attributes does it now.
int PreloadBlockHash() {
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(
                                               &ClearKernel)) == hipSuccess
                   ? 0
                   : 1;
}

}  // namespace o3dmi

using namespace o3dmi;

extern "C" {

int o3dmi_abi_version(void) { return O3DMI_ABI_VERSION; }

const char* o3dmi_status_string(int status) {
    switch (status) {
        case O3DMI_OK: return "ok";
        case O3DMI_ERR_INVALID_ARG: return "invalid argument";
        case O3DMI_ERR_HIP: return "HIP runtime error";
        case O3DMI_ERR_CAPACITY: return "capacity exceeded";
        case O3DMI_ERR_KEY_RANGE: return "block key out of range";
        case O3DMI_ERR_SINGULAR:
            return "Singular 6x6 linear system detected, tracking failed.";
        case O3DMI_ERR_NO_BLOCKS:
            return "No block is touched in TSDF volume, abort integration. "
                   "Please check specified parameters, especially depth_scale "
                   "and voxel_size";
        case O3DMI_ERR_UNSUPPORTED: return "unsupported";
        case O3DMI_ERR_NO_INLIERS:
            return "Invalid inlier_count value, must be > 0.";
        case O3DMI_ERR_INTERNAL: return "device-side consistency check failed";
        case O3DMI_ERR_PEER:
            return "another rank left the collective call with an error";
        default: return "unknown status";
    }
}

const char* o3dmi_last_error(void) { return g_last_error.c_str(); }

int o3dmi_device_info(char* name, size_t name_len, int* cu_count,
                      int64_t* hbm_bytes) {
    int dev = 0;
    O3DMI_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    O3DMI_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (name && name_len) {
        std::strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return O3DMI_OK;
}

int o3dmi_hash_create(int64_t capacity, int n_values,
                      const int64_t* value_dsizes, o3dmi_stream_t stream,
                      o3dmi_hash_t** out) {
    O3DMI_REQUIRE(out != nullptr, "out is null");
    O3DMI_REQUIRE(capacity > 0 && capacity < (1ll << 30),
                  "capacity must be in (0, 2^30)");
    O3DMI_REQUIRE(n_values >= 0 && n_values <= 8, "n_values must be in [0,8]");
    auto* h = new o3dmi_hash();
    h->n_values = n_values;
    for (int j = 0; j < n_values; ++j) h->value_dsizes[j] = value_dsizes[j];
    hipStream_t s = (hipStream_t)stream;
    int st = AllocateStorage(h, capacity, s);
    if (st == O3DMI_OK) st = ClearImpl(h, s);
    if (st != O3DMI_OK) {
        FreeStorage(h);
        delete h;
        return st;
    }
    *out = h;
    return O3DMI_OK;
}

int o3dmi_hash_destroy(o3dmi_hash_t* h) {
    if (!h) return O3DMI_OK;
    FreeStorage(h);
    (void)hipFree(h->scratch_count);
    if (h->counters_host) (void)hipHostFree(h->counters_host);
    delete h;
    return O3DMI_OK;
}

int o3dmi_hash_clear(o3dmi_hash_t* h, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(h != nullptr, "hash is null");
    return ClearImpl(h, (hipStream_t)stream);
}

int o3dmi_hash_activate(o3dmi_hash_t* h, const int32_t* keys_dev, int64_t n,
                        const int32_t* n_dev, int32_t* buf_indices_dev,
                        uint8_t* masks_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(h != nullptr, "hash is null");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    hipLaunchKernelGGL(ActivateKernel<false>, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, h->view, keys_dev,
                       n, n_dev, buf_indices_dev, masks_dev, 0, nullptr,
                       nullptr, nullptr);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_hash_insert(o3dmi_hash_t* h, const int32_t* keys_dev,
                      const void* const* values_soa_dev, int64_t n,
                      int32_t* buf_indices_dev, uint8_t* masks_dev,
                      o3dmi_stream_t stream) {
    O3DMI_REQUIRE(h != nullptr, "hash is null");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    hipStream_t s = (hipStream_t)stream;
    if (h->n_values == 0 || values_soa_dev == nullptr) {
        return o3dmi_hash_activate(h, keys_dev, n, nullptr, buf_indices_dev,
                                   masks_dev, stream);
    }
    // Small argument block on the device: src ptrs, dst ptrs, sizes.
    struct Args {
        const void* src[8];
        void* dst[8];
        int64_t sz[8];
    } host_args;
    for (int j = 0; j < h->n_values; ++j) {
        host_args.src[j] = values_soa_dev[j];
        host_args.dst[j] = h->value_buffers[j];
        host_args.sz[j] = h->value_dsizes[j];
    }
    Args* dev_args = nullptr;
    O3DMI_HIP_CHECK(hipMalloc((void**)&dev_args, sizeof(Args)));
    O3DMI_HIP_CHECK(hipMemcpyAsync(dev_args, &host_args, sizeof(Args),
                                   hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(ActivateKernel<true>, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, s, h->view, keys_dev, n,
                       (const int*)nullptr, buf_indices_dev, masks_dev,
                       h->n_values, (const void* const*)dev_args->src,
                       (void* const*)dev_args->dst,
                       (const int64_t*)dev_args->sz);
    O3DMI_HIP_CHECK(hipGetLastError());
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(dev_args);
    return O3DMI_OK;
}

int o3dmi_hash_find(o3dmi_hash_t* h, const int32_t* keys_dev, int64_t n,
                    const int32_t* n_dev, int32_t* buf_indices_dev,
                    uint8_t* masks_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(h != nullptr, "hash is null");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    hipLaunchKernelGGL(FindKernel, dim3(GridFor(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, h->view, keys_dev, n, n_dev,
                       buf_indices_dev, masks_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_hash_erase(o3dmi_hash_t* h, const int32_t* keys_dev, int64_t n,
                     uint8_t* masks_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(h != nullptr, "hash is null");
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    hipLaunchKernelGGL(EraseKernel, dim3(GridFor(n, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, h->view, keys_dev, n, masks_dev);
    O3DMI_HIP_CHECK(hipGetLastError());
    return RebuildSlotsIfCrowded(h, (hipStream_t)stream);
}

int o3dmi_hash_size(o3dmi_hash_t* h, o3dmi_stream_t stream, int64_t* size) {
    O3DMI_REQUIRE(h != nullptr && size != nullptr, "null argument");
    int top = 0;
    int st = CheckDeferred(h, (hipStream_t)stream, &top);
    *size = top;
    return st;
}

int64_t o3dmi_hash_capacity(const o3dmi_hash_t* h) {
    return h ? h->capacity : 0;
}
int64_t o3dmi_hash_bucket_count(const o3dmi_hash_t* h) {
    return h ? h->n_slots : 0;
}

int o3dmi_hash_active_indices(o3dmi_hash_t* h, int32_t* out_dev,
                              o3dmi_stream_t stream, int64_t* count) {
    O3DMI_REQUIRE(h != nullptr && out_dev != nullptr && count != nullptr,
                  "null argument");
    hipStream_t s = (hipStream_t)stream;
    O3DMI_HIP_CHECK(hipMemsetAsync(h->scratch_count, 0, sizeof(int), s));
    hipLaunchKernelGGL(ActiveIndicesKernel, dim3(GridFor(h->n_slots, kBlock)),
                       dim3(kBlock), 0, s, h->view, h->n_slots, out_dev,
                       h->scratch_count);
    O3DMI_HIP_CHECK(hipGetLastError());
    int c = 0;
    O3DMI_HIP_CHECK(hipMemcpyAsync(&c, h->scratch_count, sizeof(int),
                                   hipMemcpyDeviceToHost, s));
    O3DMI_HIP_CHECK(hipStreamSynchronize(s));
    *count = c;
    return O3DMI_OK;
}

int o3dmi_hash_reserve(o3dmi_hash_t* h, int64_t capacity,
                       o3dmi_stream_t stream) {
    O3DMI_REQUIRE(h != nullptr, "hash is null");
    hipStream_t s = (hipStream_t)stream;
    int64_t count = 0;
    int st = o3dmi_hash_size(h, stream, &count);
    if (st != O3DMI_OK) return st;
    if (capacity <= count) return O3DMI_OK;  // HashMap.cpp:49-52

    // Export active keys/values (HashMap.cpp:54-66).
    int* active = nullptr;
    int* keys = nullptr;
    void* vals[8] = {nullptr};
    if (count > 0) {
        O3DMI_HIP_CHECK(hipMalloc((void**)&active, sizeof(int) * h->capacity));
        int64_t c2 = 0;
        st = o3dmi_hash_active_indices(h, active, stream, &c2);
        if (st != O3DMI_OK) return st;
        O3DMI_HIP_CHECK(hipMalloc((void**)&keys, sizeof(int) * 3 * c2));
        for (int j = 0; j < h->n_values; ++j)
            O3DMI_HIP_CHECK(hipMalloc(&vals[j], h->value_dsizes[j] * c2));
        // Gather the active rows (rows.hip).
        if ((st = GatherRows(h->view.key_buffer, active, c2, sizeof(int) * 3,
                             keys, s)))
            return st;
        for (int j = 0; j < h->n_values; ++j)
            if ((st = GatherRows(h->value_buffers[j], active, c2,
                                 h->value_dsizes[j], vals[j], s)))
                return st;
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        count = c2;
    }
    FreeStorage(h);
    st = AllocateStorage(h, capacity, s);
    if (st == O3DMI_OK) st = ClearImpl(h, s);
    if (st != O3DMI_OK) return st;
    if (count > 0) {
        // Re-insert: winners copy their value rows (HashMap.cpp:72-76).
        int* buf = nullptr;
        O3DMI_HIP_CHECK(hipMalloc((void**)&buf, sizeof(int) * count));
        hipLaunchKernelGGL(ActivateKernel<false>,
                           dim3(GridFor(count, kBlock)), dim3(kBlock), 0, s,
                           h->view, keys, count, (const int*)nullptr, buf,
                           (uint8_t*)nullptr, 0, nullptr, nullptr, nullptr);
        O3DMI_HIP_CHECK(hipGetLastError());
        for (int j = 0; j < h->n_values; ++j)
            if ((st = ScatterRows(vals[j], buf, count, h->value_dsizes[j],
                                  h->value_buffers[j], s)))
                return st;
        O3DMI_HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(buf);
    }
    (void)hipFree(active);
    (void)hipFree(keys);
    for (int j = 0; j < h->n_values; ++j) (void)hipFree(vals[j]);
    return O3DMI_OK;
}

int o3dmi_hash_to_device(o3dmi_hash_t* h, int device, o3dmi_hash_t** out) {
    O3DMI_REQUIRE(h != nullptr && out != nullptr, "null argument");
    int n_dev = 0, src_dev = 0;
    O3DMI_HIP_CHECK(hipGetDeviceCount(&n_dev));
    O3DMI_REQUIRE(device >= 0 && device < n_dev, "no such HIP device");
    O3DMI_HIP_CHECK(hipGetDevice(&src_dev));
    struct Restore {
        int dev;
        ~Restore() { (void)hipSetDevice(dev); }
    } restore{src_dev};
    // Export the active keys / value rows on the source device
    // (HashMap.cpp:238-247: GetActiveIndices + IndexGet).
    int64_t count = 0;
    int st = o3dmi_hash_size(h, nullptr, &count);
    if (st != O3DMI_OK) return st;
    int* active = nullptr;
    int* keys = nullptr;
    void* vals[8] = {nullptr};
    int* keys_dst = nullptr;
    void* vals_dst[8] = {nullptr};
    o3dmi_hash_t* nh = nullptr;
    auto cleanup = [&](int code) {
        (void)hipSetDevice(src_dev);
        (void)hipFree(active);
        (void)hipFree(keys);
        for (int j = 0; j < h->n_values; ++j) (void)hipFree(vals[j]);
        (void)hipSetDevice(device);
        (void)hipFree(keys_dst);
        for (int j = 0; j < h->n_values; ++j) (void)hipFree(vals_dst[j]);
        if (code != O3DMI_OK && nh) o3dmi_hash_destroy(nh);
        return code;
    };
    if (count > 0) {
        if (hipMalloc((void**)&active, sizeof(int) * h->capacity) != hipSuccess)
            return cleanup(O3DMI_ERR_HIP);
        int64_t c2 = 0;
        if ((st = o3dmi_hash_active_indices(h, active, nullptr, &c2)))
            return cleanup(st);
        count = c2;
        if (hipMalloc((void**)&keys, sizeof(int) * 3 * count) != hipSuccess)
            return cleanup(O3DMI_ERR_HIP);
        if ((st = GatherRows(h->view.key_buffer, active, count, sizeof(int) * 3,
                             keys, nullptr)))
            return cleanup(st);
        for (int j = 0; j < h->n_values; ++j) {
            if (hipMalloc(&vals[j], h->value_dsizes[j] * count) != hipSuccess)
                return cleanup(O3DMI_ERR_HIP);
            if ((st = GatherRows(h->value_buffers[j], active, count,
                                 h->value_dsizes[j], vals[j], nullptr)))
                return cleanup(st);
        }
        if (hipStreamSynchronize(nullptr) != hipSuccess)
            return cleanup(O3DMI_ERR_HIP);
    }
    // The new map on the target device (HashMap.cpp:249-253).
    if (hipSetDevice(device) != hipSuccess) return cleanup(O3DMI_ERR_HIP);
    if ((st = o3dmi_hash_create(h->capacity, h->n_values, h->value_dsizes,
                                nullptr, &nh)))
        return cleanup(st);
    nh->view.owner_rank = h->view.owner_rank;
    nh->view.owner_world = h->view.owner_world;
    if (count > 0) {
        // device -> device: peer copy across devices, plain copy on one
        auto copy = [&](void* dst, const void* src, size_t bytes) {
            return device == src_dev
                           ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice)
                           : hipMemcpyPeer(dst, device, src, src_dev, bytes);
        };
        if (hipMalloc((void**)&keys_dst, sizeof(int) * 3 * count) !=
                    hipSuccess ||
            copy(keys_dst, keys, sizeof(int) * 3 * count) != hipSuccess)
            return cleanup(O3DMI_ERR_HIP);
        for (int j = 0; j < h->n_values; ++j)
            if (hipMalloc(&vals_dst[j], h->value_dsizes[j] * count) !=
                        hipSuccess ||
                copy(vals_dst[j], vals[j], h->value_dsizes[j] * count) !=
                        hipSuccess)
                return cleanup(O3DMI_ERR_HIP);
        // Insert: keys first, then the value rows land in the winners' rows
        // (one scatter per attribute -- the same association
        // DeviceHashBackend::Insert produces, without its per-byte copy).
        int* buf = nullptr;
        if (hipMalloc((void**)&buf, sizeof(int) * count) != hipSuccess)
            return cleanup(O3DMI_ERR_HIP);
        st = o3dmi_hash_activate(nh, keys_dst, count, nullptr, buf, nullptr,
                                 nullptr);
        for (int j = 0; st == O3DMI_OK && j < h->n_values; ++j)
            st = ScatterRows(vals_dst[j], buf, count, h->value_dsizes[j],
                             nh->value_buffers[j], nullptr);
        if (st == O3DMI_OK && hipStreamSynchronize(nullptr) != hipSuccess)
            st = O3DMI_ERR_HIP;
        (void)hipFree(buf);
        if (st != O3DMI_OK) return cleanup(st);
    }
    *out = nh;
    return cleanup(O3DMI_OK);
}

int o3dmi_hash_set_ownership(o3dmi_hash_t* h, int rank, int world) {
    O3DMI_REQUIRE(h != nullptr, "hash is null");
    O3DMI_REQUIRE(world >= 1 && rank >= 0 && rank < world,
                  "ownership: need 0 <= rank < world");
    h->view.owner_rank = rank;
    h->view.owner_world = world;
    return O3DMI_OK;
}

int o3dmi_block_owner(const int32_t* key3, int world) {
    if (!key3 || world < 1 || !KeyInRange(key3[0], key3[1], key3[2])) return -1;
    return OwnerOf(PackKey(key3[0], key3[1], key3[2]), world);
}

int32_t* o3dmi_hash_key_buffer(o3dmi_hash_t* h) {
    return h ? h->view.key_buffer : nullptr;
}
void* o3dmi_hash_value_buffer(o3dmi_hash_t* h, int i) {
    if (!h || i < 0 || i >= h->n_values) return nullptr;
    return h->value_buffers[i];
}

}  // extern "C"
