// In-memory set of named host arrays and its NPZ (zip of .npy) codec: the
// role of t::io::{WriteNpz, ReadNpz} (cpp/open3d/t/io/NumpyIO.cpp) for
// VoxelBlockGrid::Save / Load.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

struct NpzArray {
    std::string name;
    int dtype = 0;  // o3dmi_dtype_t
    std::vector<int64_t> shape;
    std::vector<uint8_t> data;
    int64_t NumElements() const {
        int64_t n = 1;
        for (int64_t s : shape) n *= s;
        return n;
    }
};

struct o3dmi_npz {
    std::vector<NpzArray> arrays;
    const NpzArray* Find(const std::string& name) const {
        for (const auto& a : arrays)
            if (a.name == name) return &a;
        return nullptr;
    }
};

namespace o3dmi {
int NpzDtypeSize(int dtype);  // 0 when unsupported
}
