// Stand-in for open3d/core/Dtype.h.
#pragma once
#include <cstdint>
#include <string>
namespace open3d {
namespace core {
class Dtype {
public:
    enum Code { kUndefined, kFloat32, kFloat64, kInt8, kInt16, kInt32, kInt64,
                kUInt8, kUInt16, kUInt32, kUInt64, kBool };
    constexpr Dtype() : code_(kUndefined), size_(0) {}
    constexpr Dtype(Code c, int64_t s) : code_(c), size_(s) {}
    int64_t ByteSize() const { return size_; }
    bool operator==(const Dtype& o) const { return code_ == o.code_; }
    bool operator!=(const Dtype& o) const { return code_ != o.code_; }
    std::string ToString() const {
        static const char* n[] = {"Undefined", "Float32", "Float64", "Int8",
                                  "Int16", "Int32", "Int64", "UInt8", "UInt16",
                                  "UInt32", "UInt64", "Bool"};
        return n[code_];
    }
    Code code() const { return code_; }
    template <typename T> static inline Dtype FromType();
    static const Dtype Undefined, Float32, Float64, Int8, Int16, Int32, Int64,
            UInt8, UInt16, UInt32, UInt64, Bool;
private:
    Code code_;
    int64_t size_;
};
inline constexpr Dtype Undefined{Dtype::kUndefined, 0};
inline constexpr Dtype Float32{Dtype::kFloat32, 4};
inline constexpr Dtype Float64{Dtype::kFloat64, 8};
inline constexpr Dtype Int8{Dtype::kInt8, 1};
inline constexpr Dtype Int16{Dtype::kInt16, 2};
inline constexpr Dtype Int32{Dtype::kInt32, 4};
inline constexpr Dtype Int64{Dtype::kInt64, 8};
inline constexpr Dtype UInt8{Dtype::kUInt8, 1};
inline constexpr Dtype UInt16{Dtype::kUInt16, 2};
inline constexpr Dtype UInt32{Dtype::kUInt32, 4};
inline constexpr Dtype UInt64{Dtype::kUInt64, 8};
inline constexpr Dtype Bool{Dtype::kBool, 1};
inline const Dtype Dtype::Undefined{Dtype::kUndefined, 0};
inline const Dtype Dtype::Float32{Dtype::kFloat32, 4};
inline const Dtype Dtype::Float64{Dtype::kFloat64, 8};
inline const Dtype Dtype::Int8{Dtype::kInt8, 1};
inline const Dtype Dtype::Int16{Dtype::kInt16, 2};
inline const Dtype Dtype::Int32{Dtype::kInt32, 4};
inline const Dtype Dtype::Int64{Dtype::kInt64, 8};
inline const Dtype Dtype::UInt8{Dtype::kUInt8, 1};
inline const Dtype Dtype::UInt16{Dtype::kUInt16, 2};
inline const Dtype Dtype::UInt32{Dtype::kUInt32, 4};
inline const Dtype Dtype::UInt64{Dtype::kUInt64, 8};
inline const Dtype Dtype::Bool{Dtype::kBool, 1};
template <> inline Dtype Dtype::FromType<float>() { return core::Float32; }
template <> inline Dtype Dtype::FromType<double>() { return core::Float64; }
template <> inline Dtype Dtype::FromType<int8_t>() { return core::Int8; }
template <> inline Dtype Dtype::FromType<int16_t>() { return core::Int16; }
template <> inline Dtype Dtype::FromType<int32_t>() { return core::Int32; }
template <> inline Dtype Dtype::FromType<int64_t>() { return core::Int64; }
template <> inline Dtype Dtype::FromType<uint8_t>() { return core::UInt8; }
template <> inline Dtype Dtype::FromType<uint16_t>() { return core::UInt16; }
template <> inline Dtype Dtype::FromType<uint32_t>() { return core::UInt32; }
template <> inline Dtype Dtype::FromType<uint64_t>() { return core::UInt64; }
template <> inline Dtype Dtype::FromType<bool>() { return core::Bool; }
}  // namespace core
}  // namespace open3d
