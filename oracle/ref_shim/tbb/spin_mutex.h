// Stand-in for <tbb/spin_mutex.h>.
#pragma once
#include <atomic>
namespace tbb {
class spin_mutex {
public:
    class scoped_lock {
    public:
        explicit scoped_lock(spin_mutex& m) : m_(m) {
            while (m_.flag_.test_and_set(std::memory_order_acquire)) {}
        }
        ~scoped_lock() { m_.flag_.clear(std::memory_order_release); }
    private:
        spin_mutex& m_;
    };
private:
    std::atomic_flag flag_ = ATOMIC_FLAG_INIT;
};
namespace profiling {
template <typename T> inline void set_name(T&, const char*) {}
}  // namespace profiling
}  // namespace tbb
