// Stand-in for open3d/core/Tensor.h: a contiguous, row-major, host-only
// n-d array with the slice of the Tensor interface the hot-path kernel files
// use. Owns its memory through a shared_ptr or wraps foreign memory.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "open3d/core/Device.h"
#include "open3d/core/Dtype.h"
#include "open3d/core/SizeVector.h"
#include "open3d/utility/Logging.h"

namespace open3d {
namespace core {

/// Stand-in for core::TensorKey (slices only).
class TensorKey {
public:
    static TensorKey Slice(int64_t start, int64_t stop, int64_t step) {
        TensorKey k;
        k.start = start;
        k.stop = stop;
        k.step = step;
        return k;
    }
    int64_t start = 0, stop = 0, step = 1;
};

/// Test hook: the values of the last std::vector a Tensor was constructed
/// from, as doubles (lets ref_entry read the reference's reduction result,
/// e.g. RGBDOdometryCPU.cpp's A_1x29, before it is decoded and solved).
inline std::vector<double>& LastVectorInit() {
    static thread_local std::vector<double> v;
    return v;
}

class Tensor {
public:
    Tensor() = default;
    Tensor(const SizeVector& shape, Dtype dtype,
           const Device& = Device("CPU:0"))
        : shape_(shape), dtype_(dtype) {
        Allocate();
    }
    template <typename T>
    Tensor(const std::vector<T>& init_vals, const SizeVector& shape,
           Dtype dtype, const Device& = Device("CPU:0"))
        : shape_(shape), dtype_(dtype) {
        if (dtype != Dtype::FromType<T>())
            utility::LogError("init_vals type mismatch");
        Allocate();
        if ((int64_t)init_vals.size() != NumElements())
            utility::LogError("init_vals size mismatch");
        std::memcpy(data_, init_vals.data(), sizeof(T) * init_vals.size());
        LastVectorInit().assign(init_vals.begin(), init_vals.end());
    }
    /// Wraps foreign memory (no ownership).
    static Tensor FromPtr(const void* ptr, const SizeVector& shape,
                          Dtype dtype) {
        Tensor t;
        t.shape_ = shape;
        t.dtype_ = dtype;
        t.data_ = const_cast<void*>(ptr);
        return t;
    }
    static Tensor Empty(const SizeVector& shape, Dtype dtype,
                        const Device& d = Device("CPU:0")) {
        return Tensor(shape, dtype, d);
    }
    static Tensor Zeros(const SizeVector& shape, Dtype dtype,
                        const Device& d = Device("CPU:0")) {
        return Tensor(shape, dtype, d);  // Allocate() zero-fills
    }
    static Tensor Eye(int64_t n, Dtype dtype,
                      const Device& d = Device("CPU:0")) {
        Tensor t({n, n}, dtype, d);
        for (int64_t i = 0; i < n; ++i) t.SetDouble(i * n + i, 1.0);
        return t;
    }

    template <typename T> T* GetDataPtr() { return static_cast<T*>(data_); }
    template <typename T> const T* GetDataPtr() const {
        return static_cast<const T*>(data_);
    }
    void* GetDataPtr() { return data_; }
    const void* GetDataPtr() const { return data_; }

    SizeVector GetShape() const { return shape_; }
    const SizeVector& GetShapeRef() const { return shape_; }
    int64_t GetShape(int64_t i) const {
        return shape_[(size_t)(i < 0 ? i + (int64_t)shape_.size() : i)];
    }
    int64_t NumDims() const { return (int64_t)shape_.size(); }
    int64_t NumElements() const { return shape_.NumElements(); }
    int64_t GetLength() const { return shape_.GetLength(); }
    Dtype GetDtype() const { return dtype_; }
    Device GetDevice() const { return Device("CPU:0"); }
    bool IsContiguous() const { return true; }
    bool IsCPU() const { return true; }
    bool IsCUDA() const { return false; }
    bool IsSYCL() const { return false; }
    Tensor Contiguous() const { return *this; }
    Tensor Clone() const {
        Tensor t(shape_, dtype_);
        std::memcpy(t.data_, data_, (size_t)(NumElements() * dtype_.ByteSize()));
        return t;
    }
    Tensor To(const Device&, bool = false) const { return *this; }
    Tensor To(Dtype dtype, bool = false) const {
        if (dtype == dtype_) return *this;
        Tensor t(shape_, dtype);
        const int64_t n = NumElements();
        for (int64_t i = 0; i < n; ++i) t.SetDouble(i, GetDouble(i));
        return t;
    }
    Tensor To(const Device&, Dtype dtype, bool = false) const {
        return To(dtype);
    }
    Tensor Reshape(const SizeVector& shape) const {
        Tensor t = *this;
        SizeVector s = shape;
        int64_t known = 1, infer = -1;
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i] == -1) infer = (int64_t)i; else known *= s[i];
        }
        if (infer >= 0) s[(size_t)infer] = NumElements() / known;
        t.shape_ = s;
        return t;
    }
    Tensor View(const SizeVector& shape) const { return Reshape(shape); }
    /// Only dim 0, unit step (the kernels slice result rows).
    Tensor Slice(int64_t dim, int64_t start, int64_t stop,
                 int64_t step = 1) const {
        if (dim != 0 || step != 1) utility::LogError("shim: Slice dim/step");
        Tensor t = *this;
        int64_t row = 1;
        for (size_t i = 1; i < shape_.size(); ++i) row *= shape_[i];
        t.shape_[0] = stop - start;
        t.data_ = static_cast<char*>(data_) + start * row * dtype_.ByteSize();
        return t;
    }
    template <typename T> T Item() const {
        if (NumElements() != 1) utility::LogError("Item on non-scalar");
        return *static_cast<const T*>(data_);
    }
    template <typename T> void Fill(T v) {
        const int64_t n = NumElements();
        for (int64_t i = 0; i < n; ++i) SetDouble(i, (double)v);
    }
    Tensor Neg() const {
        Tensor t = Clone();
        const int64_t n = NumElements();
        for (int64_t i = 0; i < n; ++i) t.SetDouble(i, -GetDouble(i));
        return t;
    }
    /// x with (*this) x = rhs: LU with partial pivoting, as LAPACK gesv
    /// (core/linalg/Solve.cpp:22-97, SolveCPU.cpp:15-30). Float64 only.
    Tensor Solve(const Tensor& rhs) const {
        if (dtype_ != core::Float64 || rhs.dtype_ != core::Float64 ||
            NumDims() != 2 || shape_[0] != shape_[1])
            utility::LogError("shim Solve: Float64 square only");
        const int64_t n = shape_[0];
        const int64_t k = rhs.NumDims() == 1 ? 1 : rhs.shape_[1];
        std::vector<double> A(GetDataPtr<double>(),
                              GetDataPtr<double>() + n * n);
        Tensor X = rhs.Clone();
        double* B = X.GetDataPtr<double>();
        for (int64_t c = 0; c < n; ++c) {
            int64_t p = c;
            double best = std::fabs(A[c * n + c]);
            for (int64_t r = c + 1; r < n; ++r)
                if (std::fabs(A[r * n + c]) > best) {
                    best = std::fabs(A[r * n + c]);
                    p = r;
                }
            if (best == 0.0)
                utility::LogError("Singular condition detected.");
            if (p != c) {
                for (int64_t j = 0; j < n; ++j)
                    std::swap(A[c * n + j], A[p * n + j]);
                for (int64_t j = 0; j < k; ++j)
                    std::swap(B[c * k + j], B[p * k + j]);
            }
            for (int64_t r = c + 1; r < n; ++r) {
                double f = A[r * n + c] / A[c * n + c];
                A[r * n + c] = f;
                for (int64_t j = c + 1; j < n; ++j)
                    A[r * n + j] -= f * A[c * n + j];
                for (int64_t j = 0; j < k; ++j) B[r * k + j] -= f * B[c * k + j];
            }
        }
        for (int64_t j = 0; j < k; ++j)
            for (int64_t r = n - 1; r >= 0; --r) {
                double s = B[r * k + j];
                for (int64_t c = r + 1; c < n; ++c)
                    s -= A[r * n + c] * B[c * k + j];
                B[r * k + j] = s / A[r * n + r];
            }
        return X;
    }
    std::string ToString() const { return "Tensor" + shape_.ToString(); }

    /// Row view (shares memory).
    Tensor operator[](int64_t i) const {
        if (shape_.empty()) utility::LogError("operator[] on 0-d tensor");
        Tensor t = *this;
        t.shape_ = SizeVector(shape_.begin() + 1, shape_.end());
        t.data_ = static_cast<char*>(data_) +
                  i * t.shape_.NumElements() * dtype_.ByteSize();
        return t;
    }
    /// `t[i][j] = value` on a 0-d view.
    template <typename T,
              typename = std::enable_if_t<std::is_arithmetic<T>::value>>
    Tensor& operator=(T v) {
        const int64_t n = NumElements();
        for (int64_t i = 0; i < n; ++i) SetDouble(i, (double)v);
        return *this;
    }
    /// Unit-step slices of a 1-d or 2-d tensor, returned as a copy.
    Tensor GetItem(const std::vector<TensorKey>& keys) const {
        if (keys.size() == 1 && NumDims() == 1) {
            Tensor out({keys[0].stop - keys[0].start}, dtype_);
            for (int64_t i = keys[0].start; i < keys[0].stop; ++i)
                out.SetDouble(i - keys[0].start, GetDouble(i));
            return out;
        }
        if (keys.size() == 2 && NumDims() == 2) {
            const int64_t r0 = keys[0].start, r1 = keys[0].stop;
            const int64_t c0 = keys[1].start, c1 = keys[1].stop;
            Tensor out({r1 - r0, c1 - c0}, dtype_);
            for (int64_t r = r0; r < r1; ++r)
                for (int64_t c = c0; c < c1; ++c)
                    out.SetDouble((r - r0) * (c1 - c0) + (c - c0),
                                  GetDouble(r * shape_[1] + c));
            return out;
        }
        utility::LogError("shim GetItem: unsupported keys");
    }
    void SetItem(const std::vector<TensorKey>& keys, const Tensor& value) {
        if (keys.size() == 1 && NumDims() == 1) {
            for (int64_t i = keys[0].start; i < keys[0].stop; ++i)
                SetDouble(i, value.GetDouble(i - keys[0].start));
            return;
        }
        if (keys.size() == 2 && NumDims() == 2) {
            const int64_t r0 = keys[0].start, r1 = keys[0].stop;
            const int64_t c0 = keys[1].start, c1 = keys[1].stop;
            if (value.NumElements() != (r1 - r0) * (c1 - c0))
                utility::LogError("shim SetItem: size mismatch");
            for (int64_t r = r0; r < r1; ++r)
                for (int64_t c = c0; c < c1; ++c)
                    SetDouble(r * shape_[1] + c,
                              value.GetDouble((r - r0) * (c1 - c0) + (c - c0)));
            return;
        }
        utility::LogError("shim SetItem: unsupported keys");
    }
    void SetItem(const TensorKey& key, const Tensor& value) {
        SetItem(std::vector<TensorKey>{key}, value);
    }
    Tensor GetItem(const TensorKey& key) const {
        return GetItem(std::vector<TensorKey>{key});
    }
    Tensor T() const {
        if (NumDims() != 2) utility::LogError("shim T: 2-d only");
        Tensor t({shape_[1], shape_[0]}, dtype_);
        for (int64_t r = 0; r < shape_[0]; ++r)
            for (int64_t c = 0; c < shape_[1]; ++c)
                t.SetDouble(c * shape_[0] + r, GetDouble(r * shape_[1] + c));
        return t;
    }
    Tensor Matmul(const Tensor& rhs) const {
        const int64_t m = shape_[0], k = shape_[1];
        const int64_t n = rhs.NumDims() == 1 ? 1 : rhs.shape_[1];
        Tensor out(rhs.NumDims() == 1 ? SizeVector{m} : SizeVector{m, n},
                   dtype_);
        for (int64_t i = 0; i < m; ++i)
            for (int64_t j = 0; j < n; ++j) {
                double s = 0;
                for (int64_t l = 0; l < k; ++l)
                    s += GetDouble(i * k + l) * rhs.GetDouble(l * n + j);
                out.SetDouble(i * n + j, s);
            }
        return out;
    }
    Tensor operator-(const Tensor& o) const {
        Tensor t = Clone();
        const int64_t n = NumElements();
        for (int64_t i = 0; i < n; ++i)
            t.SetDouble(i, GetDouble(i) - o.GetDouble(i));
        return t;
    }
    template <typename T> Tensor Div(T v) const {
        Tensor t = Clone();
        const int64_t n = NumElements();
        for (int64_t i = 0; i < n; ++i) t.SetDouble(i, GetDouble(i) / (double)v);
        return t;
    }
    Tensor Transpose(int64_t, int64_t) const { return T(); }
    template <typename T>
    static Tensor Full(const SizeVector& shape, T v, Dtype dtype,
                       const Device& d = Device("CPU:0")) {
        Tensor t(shape, dtype, d);
        t.Fill(v);
        return t;
    }
    Tensor Flatten(int64_t = 0, int64_t = -1) const {
        return Reshape({NumElements()});
    }
    double Det() const {
        if (NumDims() != 2 || shape_[0] != 3 || shape_[1] != 3)
            utility::LogError("shim Det: 3x3 only");
        auto a = [&](int r, int c) { return GetDouble(r * 3 + c); };
        return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) -
               a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) +
               a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
    }
    /// Not available (only the point-to-point estimator, off the hot path,
    /// needs it).
    std::tuple<Tensor, Tensor, Tensor> SVD() const {
        utility::LogError("shim: SVD is not available");
    }

    double GetDouble(int64_t i) const {
        switch (dtype_.code()) {
            case Dtype::kFloat32: return ((const float*)data_)[i];
            case Dtype::kFloat64: return ((const double*)data_)[i];
            case Dtype::kInt8: return ((const int8_t*)data_)[i];
            case Dtype::kInt16: return ((const int16_t*)data_)[i];
            case Dtype::kInt32: return ((const int32_t*)data_)[i];
            case Dtype::kInt64: return (double)((const int64_t*)data_)[i];
            case Dtype::kUInt8: return ((const uint8_t*)data_)[i];
            case Dtype::kUInt16: return ((const uint16_t*)data_)[i];
            case Dtype::kUInt32: return ((const uint32_t*)data_)[i];
            case Dtype::kUInt64: return (double)((const uint64_t*)data_)[i];
            case Dtype::kBool: return ((const bool*)data_)[i];
            default: return 0;
        }
    }
    void SetDouble(int64_t i, double v) {
        switch (dtype_.code()) {
            case Dtype::kFloat32: ((float*)data_)[i] = (float)v; break;
            case Dtype::kFloat64: ((double*)data_)[i] = v; break;
            case Dtype::kInt8: ((int8_t*)data_)[i] = (int8_t)v; break;
            case Dtype::kInt16: ((int16_t*)data_)[i] = (int16_t)v; break;
            case Dtype::kInt32: ((int32_t*)data_)[i] = (int32_t)v; break;
            case Dtype::kInt64: ((int64_t*)data_)[i] = (int64_t)v; break;
            case Dtype::kUInt8: ((uint8_t*)data_)[i] = (uint8_t)v; break;
            case Dtype::kUInt16: ((uint16_t*)data_)[i] = (uint16_t)v; break;
            case Dtype::kUInt32: ((uint32_t*)data_)[i] = (uint32_t)v; break;
            case Dtype::kUInt64: ((uint64_t*)data_)[i] = (uint64_t)v; break;
            case Dtype::kBool: ((bool*)data_)[i] = v != 0; break;
            default: break;
        }
    }

private:
    void Allocate() {
        size_t bytes = (size_t)(NumElements() * dtype_.ByteSize());
        blob_ = std::shared_ptr<void>(std::calloc(bytes ? bytes : 1, 1),
                                      std::free);
        data_ = blob_.get();
    }
    SizeVector shape_;
    Dtype dtype_ = Dtype::Undefined;
    std::shared_ptr<void> blob_;
    void* data_ = nullptr;
};

}  // namespace core
}  // namespace open3d
