// Caching device-memory pool for the library's transient scratch (index
// builds, pyramid levels, sort / scan temporaries). hipMalloc / hipFree cost
// tens of microseconds each and hipFree synchronises the device; a tracking
// loop that rebuilds a 3-level pyramid and three search indices per frame
// would otherwise spend more time in the allocator than in its kernels.
//
// Blocks are rounded up to a power of two (>= 512 B), kept on per-size free
// lists and never returned to the driver until o3dmi_release_cached_memory().
// Callers return a block only after the work using it has completed on its
// stream (every user below synchronises before PoolFree), so a block can be
// handed to any stream next.

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace o3dmi {
namespace {
std::mutex g_mu;
// Free lists are per HIP device: a process may drive several GPUs (one host
// thread per device), and a block must only be handed out on the device it
// was allocated on.
using DevClass = std::pair<int, size_t>;           // (device, size class)
std::map<DevClass, std::vector<void*>> g_free;     // -> cached blocks
std::unordered_map<void*, DevClass> g_size;        // live or cached -> class
size_t g_cached_bytes = 0;

size_t SizeClass(size_t bytes) {
    size_t c = 512;
    while (c < bytes) c <<= 1;
    return c;
}
}  // namespace

int PoolAlloc(void** out, size_t bytes) {
    const size_t c = SizeClass(bytes ? bytes : 1);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_free.find(DevClass(dev, c));
        if (it != g_free.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            g_cached_bytes -= c;
            return O3DMI_OK;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, c);
    if (e != hipSuccess) {
        // Out of memory: drop the cache and retry once.
        o3dmi_release_cached_memory();
        e = hipMalloc(&p, c);
    }
    if (e != hipSuccess) {
        SetLastError(std::string("hipMalloc: ") + hipGetErrorString(e));
        return O3DMI_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_size[p] = DevClass(dev, c);
    *out = p;
    return O3DMI_OK;
}

void PoolFree(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_size.find(p);
    if (it == g_size.end()) {  // not ours
        (void)hipFree(p);
        return;
    }
    g_free[it->second].push_back(p);
    g_cached_bytes += it->second.second;
}

}  // namespace o3dmi

extern "C" int o3dmi_release_cached_memory(void) {
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lk(o3dmi::g_mu);
        for (auto& kv : o3dmi::g_free) {
            for (void* p : kv.second) {
                blocks.push_back(p);
                o3dmi::g_size.erase(p);
            }
            kv.second.clear();
        }
        o3dmi::g_cached_bytes = 0;
    }
    for (void* p : blocks) (void)hipFree(p);
    return O3DMI_OK;
}
