// Stand-in for <tbb/concurrent_unordered_set.h>: std::unordered_set under a
// spin lock (only emplace / iteration / size are used, VoxelBlockGridCPU.cpp).
#pragma once
#include <atomic>
#include <cstddef>
#include <unordered_set>
#include <utility>
namespace tbb {
template <typename Key, typename Hash = std::hash<Key>,
          typename Eq = std::equal_to<Key>>
class concurrent_unordered_set {
    using Set = std::unordered_set<Key, Hash, Eq>;
public:
    using iterator = typename Set::iterator;
    using const_iterator = typename Set::const_iterator;
    template <typename... Args>
    void emplace(Args&&... args) {
        Key k(std::forward<Args>(args)...);
        while (flag_.test_and_set(std::memory_order_acquire)) {}
        set_.insert(k);
        flag_.clear(std::memory_order_release);
    }
    void insert(const Key& k) { emplace(k); }
    std::size_t size() const { return set_.size(); }
    iterator begin() { return set_.begin(); }
    iterator end() { return set_.end(); }
    const_iterator begin() const { return set_.begin(); }
    const_iterator end() const { return set_.end(); }
private:
    Set set_;
    std::atomic_flag flag_ = ATOMIC_FLAG_INIT;
};
}  // namespace tbb
