// Stand-in for <tbb/concurrent_unordered_set.h> (only emplace / iteration /
// size are used, VoxelBlockGridCPU.cpp:69-110,141-198).
//
// tbb's container inserts and looks up without a global lock; one
// std::unordered_set behind one spin lock (the first stand-in) serialised every
// emplace of DepthTouchCPU -- ~77 k per VGA frame, almost all duplicates -- and
// made the reference's CPU baseline look slower than it is. This one is
// sharded 256 ways by the key's hash (a lock and a set per shard), so threads
// only meet on the same shard; iteration walks a snapshot taken after the
// parallel phase (the reference iterates only then).
#pragma once
#include <atomic>
#include <cstddef>
#include <unordered_set>
#include <utility>
#include <vector>
namespace tbb {
template <typename Key, typename Hash = std::hash<Key>,
          typename Eq = std::equal_to<Key>>
class concurrent_unordered_set {
    static constexpr std::size_t kShards = 256;
    struct alignas(64) Shard {
        std::unordered_set<Key, Hash, Eq> set;
        std::atomic_flag flag = ATOMIC_FLAG_INIT;
    };
public:
    using iterator = typename std::vector<Key>::const_iterator;
    using const_iterator = iterator;
    concurrent_unordered_set() : shards_(kShards) {}
    template <typename... Args>
    void emplace(Args&&... args) {
        Key k(std::forward<Args>(args)...);
        // spread the hash's low bits (they also pick the bucket inside a set)
        std::size_t h = Hash()(k);
        h ^= h >> 17;
        Shard& s = shards_[(h * 0x9E3779B97F4A7C15ull >> 32) % kShards];
        while (s.flag.test_and_set(std::memory_order_acquire)) {}
        s.set.insert(k);
        s.flag.clear(std::memory_order_release);
    }
    void insert(const Key& k) { emplace(k); }
    std::size_t size() const {
        std::size_t n = 0;
        for (const Shard& s : shards_) n += s.set.size();
        return n;
    }
    iterator begin() const {
        Snapshot();
        return flat_.begin();
    }
    iterator end() const {
        Snapshot();
        return flat_.end();
    }
private:
    void Snapshot() const {
        const std::size_t n = size();
        if (flat_.size() == n) return;
        flat_.clear();
        flat_.reserve(n);
        for (const Shard& s : shards_)
            for (const Key& k : s.set) flat_.push_back(k);
    }
    std::vector<Shard> shards_;
    mutable std::vector<Key> flat_;
};
}  // namespace tbb
