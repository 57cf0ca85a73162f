// Stand-in for open3d/core/hashmap/CPU/TBBHashBackend.h: the key -> buffer
// index map the reference's RayCastCPU looks blocks up in
// (TBBHashBackend.h:59-60 holds a tbb::concurrent_unordered_map; lookups only
// here, so a std::unordered_map with the reference's own hash functor is
// equivalent).
#pragma once
#include <memory>
#include <unordered_map>
#include "open3d/core/hashmap/HashMap.h"
namespace open3d {
namespace core {
template <typename Key, typename Hash, typename Eq>
class TBBHashBackend : public DeviceHashBackend {
public:
    using Map = std::unordered_map<Key, buf_index_t, Hash, Eq>;
    TBBHashBackend() : impl_(std::make_shared<Map>()) {}
    std::shared_ptr<Map> GetImpl() const { return impl_; }
private:
    std::shared_ptr<Map> impl_;
};
}  // namespace core
}  // namespace open3d
