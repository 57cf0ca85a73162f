// Stand-in for open3d/core/Indexer.h: element-wise iteration over contiguous
// tensors of identical shape -- the only use the hot-path kernel files make of
// it (t/geometry/kernel/ImageImpl.h `To...`). No broadcasting, no reductions.
#pragma once
#include <cstdint>
#include <vector>

#include "open3d/core/Tensor.h"

namespace open3d {
namespace core {

enum class DtypePolicy { NONE, ALL_SAME, INPUT_SAME, INPUT_SAME_OUTPUT_BOOL };

class Indexer {
public:
    Indexer(const std::vector<Tensor>& inputs, const Tensor& output,
            DtypePolicy = DtypePolicy::ALL_SAME)
        : inputs_(inputs), output_(output) {
        for (const Tensor& t : inputs_) {
            if (t.NumElements() != output_.NumElements())
                utility::LogError("shim Indexer: shapes must match");
        }
    }
    int64_t NumWorkloads() const { return output_.NumElements(); }

    template <typename T>
    T* GetInputPtr(int64_t input_idx, int64_t workload_idx) const {
        return const_cast<T*>(inputs_[input_idx].template GetDataPtr<T>()) +
               workload_idx;
    }
    template <typename T>
    T* GetOutputPtr(int64_t workload_idx) const {
        return const_cast<T*>(output_.template GetDataPtr<T>()) + workload_idx;
    }

private:
    std::vector<Tensor> inputs_;
    Tensor output_;
};

}  // namespace core
}  // namespace open3d
