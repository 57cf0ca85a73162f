// RGB-D odometry front end on MI355X: the depth /
// intensity image pyramid ops and the per-pixel Jacobian + 29-sum reductions
// that t::pipelines::odometry::RGBDOdometryMultiScale runs per frame pair.
//
// Replaces (paths relative to cpp/open3d):
//   t/geometry/kernel/Image.h:83-107       ToCUDA, ClipTransformCUDA,
//                                          PyrDownDepthCUDA, CreateVertexMapCUDA,
//                                          CreateNormalMapCUDA (bodies ImageImpl.h:35-322)
//   t/geometry/kernel/NPPImage.h:17-48     npp::RGBToGray, Resize, FilterBilateral,
//                                          FilterGaussian, FilterSobel
//   t/pipelines/kernel/RGBDOdometryImpl.h:74-125
//                                          ComputeOdometryResult{PointToPlane,Intensity,
//                                          Hybrid}CUDA, ComputeOdometryInformationMatrixCUDA
//                                          (Jacobians RGBDOdometryJacobianImpl.h:29-343)
//
// Everything here is a streaming or small-stencil pass over {H,W} images: HBM /
// launch bound, no MFMA. Per-pixel float32 arithmetic follows the reference
// operation for operation (no FMA contraction, correctly rounded div/sqrt);
// the 29 running sums are float64 with a fixed reduction tree.
//
// The filters the reference delegates to IPP / NPP have no in-tree arithmetic
// (parity unpinned); they follow the published IPP semantics stated
// in include/o3d_mi355x.h and are checked against the reference's own golden
// vectors for those filters (cpp/tests/t/geometry/Image.cpp:239-686).

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "reduce_sums.h"

namespace o3dmi {
namespace {

__device__ __forceinline__ bool IsNan(float v) { return v != v; }

__device__ __forceinline__ bool IsInvalid(float v, float invalid_fill,
                                          int fill_class) {
    // ImageImpl.h:214-222: isinf / isnan / == depending on the fill value.
    if (fill_class == 1) return isinf(v);
    if (fill_class == 2) return IsNan(v);
    return v == invalid_fill;
}

int FillClass(float invalid_fill) {
    if (std::isinf(invalid_fill)) return 1;
    if (std::isnan(invalid_fill)) return 2;
    return 0;
}

// ---- ImageImpl.h:94-128 -----------------------------------------------------
template <typename S>
__global__ void ClipTransformKernel(const S* __restrict__ src,
                                    float* __restrict__ dst, int64_t n,
                                    float scale, float min_value,
                                    float max_value, float clip_fill) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float in = static_cast<float>(src[i]);
        float out = in / scale;
        out = out <= min_value ? clip_fill : out;
        out = out >= max_value ? clip_fill : out;
        dst[i] = out;
    }
}

// ---- ImageImpl.h:132-206 ----------------------------------------------------
__device__ __forceinline__ float PyrDownDepthAt(const float* __restrict__ src,
                                                int rows, int cols, int x,
                                                int y, float depth_diff,
                                                float invalid_fill) {
    const float gweights[3] = {0.375f, 0.25f, 0.0625f};
    const int y_src = 2 * y, x_src = 2 * x;
    const float v_center = src[(int64_t)y_src * cols + x_src];
    if (v_center == invalid_fill) return invalid_fill;
    const int x_min = max(0, x_src - 2), y_min = max(0, y_src - 2);
    const int x_max = min(cols - 1, x_src + 2);
    const int y_max = min(rows - 1, y_src + 2);
    float v_sum = 0, w_sum = 0;
    for (int yk = y_min; yk <= y_max; ++yk) {
        for (int xk = x_min; xk <= x_max; ++xk) {
            const float v = src[(int64_t)yk * cols + xk];
            const int dy = abs(yk - y_src), dx = abs(xk - x_src);
            if (v != invalid_fill && fabsf(v - v_center) < depth_diff) {
                const float wt = gweights[dx] * gweights[dy];
                v_sum += wt * v;
                w_sum += wt;
            }
        }
    }
    return w_sum == 0 ? invalid_fill : v_sum / w_sum;
}

__global__ void PyrDownDepthKernel(const float* __restrict__ src,
                                   float* __restrict__ dst, int rows, int cols,
                                   float depth_diff, float invalid_fill) {
    const int rows_down = rows / 2, cols_down = cols / 2;
    const int64_t n = (int64_t)rows_down * cols_down;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(w / cols_down), x = (int)(w % cols_down);
        dst[w] = PyrDownDepthAt(src, rows, cols, x, y, depth_diff,
                                invalid_fill);
    }
}

// ClipTransform of the source and the target depth in one launch
// (blockIdx.y selects the image; their dtypes may differ).
__global__ void ClipTransformPairKernel(const void* __restrict__ src0,
                                        int src0_is_f32,
                                        const void* __restrict__ src1,
                                        int src1_is_f32,
                                        float* __restrict__ dst0,
                                        float* __restrict__ dst1, int64_t n,
                                        float scale, float min_value,
                                        float max_value, float clip_fill) {
    const void* src = blockIdx.y == 0 ? src0 : src1;
    const int is_f32 = blockIdx.y == 0 ? src0_is_f32 : src1_is_f32;
    float* dst = blockIdx.y == 0 ? dst0 : dst1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float in = is_f32 ? ((const float*)src)[i]
                                : static_cast<float>(((const uint16_t*)src)[i]);
        float out = in / scale;
        out = out <= min_value ? clip_fill : out;
        out = out >= max_value ? clip_fill : out;
        dst[i] = out;
    }
}

// ---- ImageImpl.h:208-256 (identity extrinsic) --------------------------------
__global__ void CreateVertexMapKernel(const float* __restrict__ src,
                                      float* __restrict__ dst, int rows,
                                      int cols, Camera cam, float invalid_fill,
                                      int fill_class) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(w / cols), x = (int)(w % cols);
        const float d = src[w];
        float vx = invalid_fill, vy = invalid_fill, vz = invalid_fill;
        if (!IsInvalid(d, invalid_fill, fill_class))
            cam.Unproject((float)x, (float)y, d, vx, vy, vz);
        dst[3 * w + 0] = vx;
        dst[3 * w + 1] = vy;
        dst[3 * w + 2] = vz;
    }
}

// ---- ImageImpl.h:257-322 ----------------------------------------------------
__device__ __forceinline__ void NormalFromVertices(
        const float* v00, const float* v10, const float* v01,
        float invalid_fill, float* normal) {
    if ((v00[0] == invalid_fill && v00[1] == invalid_fill &&
         v00[2] == invalid_fill) ||
        (v01[0] == invalid_fill && v01[1] == invalid_fill &&
         v01[2] == invalid_fill) ||
        (v10[0] == invalid_fill && v10[1] == invalid_fill &&
         v10[2] == invalid_fill)) {
        normal[0] = invalid_fill;
        normal[1] = invalid_fill;
        normal[2] = invalid_fill;
        return;
    }
    const float dx0 = v01[0] - v00[0];
    const float dy0 = v01[1] - v00[1];
    const float dz0 = v01[2] - v00[2];
    const float dx1 = v10[0] - v00[0];
    const float dy1 = v10[1] - v00[1];
    const float dz1 = v10[2] - v00[2];
    normal[0] = dy0 * dz1 - dz0 * dy1;
    normal[1] = dz0 * dx1 - dx0 * dz1;
    normal[2] = dx0 * dy1 - dy0 * dx1;
    constexpr float EPSILON = 1e-5f;
    float normal_norm = sqrtf(normal[0] * normal[0] + normal[1] * normal[1] +
                              normal[2] * normal[2]);
    // std::max(normal_norm, EPSILON): (a < b) ? b : a -- a NaN norm stays NaN.
    normal_norm = (normal_norm < EPSILON) ? EPSILON : normal_norm;
    normal[0] /= normal_norm;
    normal[1] /= normal_norm;
    normal[2] /= normal_norm;
}

__global__ void CreateNormalMapKernel(const float* __restrict__ src,
                                      float* __restrict__ dst, int rows,
                                      int cols, float invalid_fill) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(w / cols), x = (int)(w % cols);
        float normal[3] = {invalid_fill, invalid_fill, invalid_fill};
        if (y < rows - 1 && x < cols - 1) {
            float v00[3], v10[3], v01[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                v00[c] = src[3 * w + c];
                v10[c] = src[3 * (w + 1) + c];
                v01[c] = src[3 * (w + cols) + c];
            }
            NormalFromVertices(v00, v10, v01, invalid_fill, normal);
        }
        dst[3 * w + 0] = normal[0];
        dst[3 * w + 1] = normal[1];
        dst[3 * w + 2] = normal[2];
    }
}

// ---- ImageImpl.h:35-85, elem_t = float ---------------------------------------
template <typename S>
__global__ void ToFloatKernel(const S* __restrict__ src,
                              float* __restrict__ dst, int64_t n, float c_scale,
                              float c_offset) {
    // numeric_limits<float>::min() is the smallest positive normal: values
    // below it (zero, negatives) are clamped up to it, as written upstream.
    const float lo = 1.17549435e-38f, hi = 3.40282347e+38f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float out = static_cast<float>(src[i]) * c_scale + c_offset;
        out = out < lo ? lo : out;
        out = out > hi ? hi : out;
        dst[i] = out;
    }
}

// ---- Image.cpp:149-161 -------------------------------------------------------
__device__ __forceinline__ float GrayF(float R, float G, float B) {
    const float r = R * 0.299f;
    const float g = G * 0.587f;
    const float b = B * 0.114f;
    return (r + g) + b;
}
template <typename S>
__global__ void RGBToGrayKernel(const S* __restrict__ src, S* __restrict__ dst,
                                int64_t n, float hi) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float gray = GrayF((float)src[3 * i + 0], (float)src[3 * i + 1],
                                 (float)src[3 * i + 2]);
        if (hi <= 0.f) {
            dst[i] = (S)gray;
        } else {
            float rounded = (float)round((double)gray);
            rounded = rounded < 0.f ? 0.f : (rounded > hi ? hi : rounded);
            dst[i] = (S)rounded;
        }
    }
}
// RGBToGray().To(Float32, 1/255) of a u8 colour image in one pass
// (RGBDOdometry.cpp:223-224); same per-pixel arithmetic as the two kernels.
__global__ void RGB8ToIntensityKernel(const uint8_t* __restrict__ src,
                                      float* __restrict__ dst, int64_t n,
                                      float c_scale) {
    const float lo = 1.17549435e-38f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float gray = GrayF((float)src[3 * i + 0], (float)src[3 * i + 1],
                                 (float)src[3 * i + 2]);
        float rounded = (float)round((double)gray);
        rounded = rounded < 0.f ? 0.f : (rounded > 255.f ? 255.f : rounded);
        const uint8_t g8 = (uint8_t)rounded;
        float out = (float)g8 * c_scale + 0.0f;
        out = out < lo ? lo : out;
        dst[i] = out;
    }
}

// ---- IPP / NPP semantics filters ----------------------------------------------
__device__ __forceinline__ int Clampi(int v, int lo, int hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}
__device__ __forceinline__ float ExpCR(float x) {
    return (float)exp((double)x);
}

struct BilateralParams {
    int radius;
    float two_val_sqr;    // 2 * sigma_v^2
    float space_w[9];     // exp(-d2 / (2 sigma_d^2)) for d2 = 0..8 (radius <= 2)
};

__device__ __forceinline__ float BilateralAt(const float* __restrict__ src,
                                             int rows, int cols, int x, int y,
                                             const BilateralParams& bp) {
    const float c = src[(int64_t)y * cols + x];
    float v_sum = 0, w_sum = 0;
    const int r = bp.radius;
    for (int dy = -r; dy <= r; ++dy) {
        for (int dx = -r; dx <= r; ++dx) {
            const int d2 = dx * dx + dy * dy;
            if (d2 > r * r) continue;
            const int yy = Clampi(y + dy, 0, rows - 1);
            const int xx = Clampi(x + dx, 0, cols - 1);
            const float v = src[(int64_t)yy * cols + xx];
            const float dv = v - c;
            const float w = ExpCR(-(dv * dv) / bp.two_val_sqr) * bp.space_w[d2];
            v_sum += w * v;
            w_sum += w;
        }
    }
    return v_sum / w_sum;
}

__global__ void FilterBilateralKernel(const float* __restrict__ src,
                                      float* __restrict__ dst, int rows,
                                      int cols, BilateralParams bp) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(w / cols), x = (int)(w % cols);
        dst[w] = BilateralAt(src, rows, cols, x, y, bp);
    }
}

struct GaussTaps {
    int radius;
    float w[7];
};

__device__ __forceinline__ float GaussianAt(const float* __restrict__ src,
                                            int rows, int cols, int x, int y,
                                            const GaussTaps& gt) {
    float acc = 0;
    const int r = gt.radius;
    for (int dy = -r; dy <= r; ++dy) {
        for (int dx = -r; dx <= r; ++dx) {
            const int yy = Clampi(y + dy, 0, rows - 1);
            const int xx = Clampi(x + dx, 0, cols - 1);
            acc += (gt.w[dy + r] * gt.w[dx + r]) * src[(int64_t)yy * cols + xx];
        }
    }
    return acc;
}

// step = 1: FilterGaussian; step = 2: Image::PyrDown (Image.cpp:404-407) =
// FilterGaussian + Resize(0.5, Nearest), evaluated only at the kept pixels.
__global__ void FilterGaussianKernel(const float* __restrict__ src,
                                     float* __restrict__ dst, int rows,
                                     int cols, int out_rows, int out_cols,
                                     int step, GaussTaps gt) {
    const int64_t n = (int64_t)out_rows * out_cols;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(w / out_cols), x = (int)(w % out_cols);
        dst[w] = GaussianAt(src, rows, cols, x * step, y * step, gt);
    }
}

__global__ void ResizeHalfNearestKernel(const float* __restrict__ src,
                                        float* __restrict__ dst, int cols,
                                        int out_rows, int out_cols) {
    const int64_t n = (int64_t)out_rows * out_cols;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(w / out_cols), x = (int)(w % out_cols);
        dst[w] = src[(int64_t)(2 * y) * cols + 2 * x];
    }
}

__global__ void FilterSobel3Kernel(const float* __restrict__ src,
                                   float* __restrict__ out_dx,
                                   float* __restrict__ out_dy, int rows,
                                   int cols) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(w / cols), x = (int)(w % cols);
        const int y0 = Clampi(y - 1, 0, rows - 1), y2 = Clampi(y + 1, 0, rows - 1);
        const int x0 = Clampi(x - 1, 0, cols - 1), x2 = Clampi(x + 1, 0, cols - 1);
#define O3DMI_S(yy, xx) src[(int64_t)(yy) * cols + (xx)]
        const float right = O3DMI_S(y0, x2) + 2.0f * O3DMI_S(y, x2) + O3DMI_S(y2, x2);
        const float left = O3DMI_S(y0, x0) + 2.0f * O3DMI_S(y, x0) + O3DMI_S(y2, x0);
        const float bottom = O3DMI_S(y2, x0) + 2.0f * O3DMI_S(y2, x) + O3DMI_S(y2, x2);
        const float top = O3DMI_S(y0, x0) + 2.0f * O3DMI_S(y0, x) + O3DMI_S(y0, x2);
#undef O3DMI_S
        out_dx[w] = right - left;
        out_dy[w] = bottom - top;
    }
}

// Pyramid level of the point-to-plane method in one pass over the level
// (RGBDOdometry.cpp:124-153): source vertex map, target vertex map, and the
// target normal map = CreateNormalMap(CreateVertexMap(FilterBilateral(target
// depth, 5, 5, 10))) without materialising the smoothed depth or its vertex
// map in HBM. A workgroup owns a 32 x 8 pixel tile: the bilateral-smoothed
// depth of the tile plus one halo column / row (what the forward differences
// of the normal need) is evaluated once into LDS, then every lane forms its
// normal from three LDS values. Same per-pixel arithmetic as the separate
// kernels, hence identical output.
constexpr int kTileW = 32, kTileH = 8;

__global__ void __launch_bounds__(kTileW* kTileH)
P2PlaneLevelKernel(const float* __restrict__ src_depth,
                   const float* __restrict__ tgt_depth,
                   float* __restrict__ src_vertex,
                   float* __restrict__ tgt_vertex,
                   float* __restrict__ tgt_normal, int rows, int cols,
                   Camera cam, BilateralParams bp,
                   float* __restrict__ src_depth_next,
                   float* __restrict__ tgt_depth_next, float depth_diff) {
    __shared__ float smooth[kTileH + 1][kTileW + 1];
    const float kNan = __builtin_nanf("");
    const int tx = threadIdx.x % kTileW, ty = threadIdx.x / kTileW;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int x = x0 + tx, y = y0 + ty;
    const bool inside = x < cols && y < rows;
    // PyrDownDepth of both images for the next (coarser) level rides along:
    // this tile covers a 16 x 4 patch of it; lanes 0..63 take the source,
    // lanes 64..127 the target (RGBDOdometry.cpp:146-153).
    if (src_depth_next != nullptr && threadIdx.x < 2 * (kTileW / 2) * (kTileH / 2)) {
        const int k = threadIdx.x % ((kTileW / 2) * (kTileH / 2));
        const int x2 = x0 / 2 + k % (kTileW / 2), y2 = y0 / 2 + k / (kTileW / 2);
        const int rows2 = rows / 2, cols2 = cols / 2;
        if (x2 < cols2 && y2 < rows2) {
            const bool is_src = threadIdx.x < (kTileW / 2) * (kTileH / 2);
            const float v = PyrDownDepthAt(is_src ? src_depth : tgt_depth, rows,
                                           cols, x2, y2, depth_diff, kNan);
            (is_src ? src_depth_next : tgt_depth_next)[(int64_t)y2 * cols2 + x2] = v;
        }
    }
    if (inside) smooth[ty][tx] = BilateralAt(tgt_depth, rows, cols, x, y, bp);
    // halo: right column (kTileH + 1 values, corner included), bottom row
    if (threadIdx.x < kTileH + 1 + kTileW) {
        int hx, hy;
        if (threadIdx.x <= kTileH) {
            hx = kTileW;
            hy = threadIdx.x;
        } else {
            hx = threadIdx.x - (kTileH + 1);
            hy = kTileH;
        }
        const int gx = x0 + hx, gy = y0 + hy;
        if (gx < cols && gy < rows)
            smooth[hy][hx] = BilateralAt(tgt_depth, rows, cols, gx, gy, bp);
    }
    __syncthreads();
    if (!inside) return;
    const int64_t w = (int64_t)y * cols + x;
    float v[3];
    const float ds = src_depth[w];
    v[0] = v[1] = v[2] = kNan;
    if (!IsNan(ds)) cam.Unproject((float)x, (float)y, ds, v[0], v[1], v[2]);
    src_vertex[3 * w + 0] = v[0];
    src_vertex[3 * w + 1] = v[1];
    src_vertex[3 * w + 2] = v[2];
    const float dt = tgt_depth[w];
    v[0] = v[1] = v[2] = kNan;
    if (!IsNan(dt)) cam.Unproject((float)x, (float)y, dt, v[0], v[1], v[2]);
    tgt_vertex[3 * w + 0] = v[0];
    tgt_vertex[3 * w + 1] = v[1];
    tgt_vertex[3 * w + 2] = v[2];
    float normal[3] = {kNan, kNan, kNan};
    if (y < rows - 1 && x < cols - 1) {
        float v00[3], v10[3], v01[3];
        auto smooth_vertex = [&](int dx, int dy, float* out) {
            const float d = smooth[ty + dy][tx + dx];
            out[0] = out[1] = out[2] = kNan;
            if (!IsNan(d))
                cam.Unproject((float)(x + dx), (float)(y + dy), d, out[0],
                              out[1], out[2]);
        };
        smooth_vertex(0, 0, v00);
        smooth_vertex(1, 0, v10);
        smooth_vertex(0, 1, v01);
        NormalFromVertices(v00, v10, v01, kNan, normal);
    }
    tgt_normal[3 * w + 0] = normal[0];
    tgt_normal[3 * w + 1] = normal[1];
    tgt_normal[3 * w + 2] = normal[2];
}

// ---- Jacobians, RGBDOdometryJacobianImpl.h ------------------------------------
// :29-32. Sign takes an int (GeometryMacros.h:92-94): r is truncated first.
__device__ __forceinline__ float HuberDeriv(float r, float delta) {
    const float abs_r = fabsf(r);
    const int ri = (int)r;
    const int sgn = (ri > 0) - (ri < 0);
    return abs_r < delta ? r : delta * (float)sgn;
}
// :34-37 (the 0.5 literals are double)
__device__ __forceinline__ float HuberLoss(float r, float delta) {
    const float abs_r = fabsf(r);
    return abs_r < delta
                   ? (float)(0.5 * (double)r * (double)r)
                   : (float)((double)(delta * abs_r) -
                             0.5 * (double)delta * (double)delta);
}
// float -> int as the reference's x86-64 CPU build converts (cvttss2si):
// out of range / NaN -> INT_MIN.
__device__ __forceinline__ int ToIntX86(float v) {
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
    return (int)v;
}

struct OdoMaps {
    const float* source_depth;
    const float* target_depth;
    const float* source_intensity;
    const float* target_intensity;
    const float* target_depth_dx;
    const float* target_depth_dy;
    const float* target_intensity_dx;
    const float* target_intensity_dy;
    const float* source_vertex;
    const float* target_vertex;
    const float* target_normal;
    int rows, cols;
};

// :106-162
__device__ __forceinline__ bool JacobianPointToPlane(
        const OdoMaps& m, const Camera& ti, int64_t w, float trunc, float* J,
        float& r) {
    const float sx = m.source_vertex[3 * w + 0];
    if (IsNan(sx)) return false;
    const float sy = m.source_vertex[3 * w + 1];
    const float sz = m.source_vertex[3 * w + 2];
    float Tx, Ty, Tz, u, v;
    ti.RigidTransform(sx, sy, sz, Tx, Ty, Tz);
    ti.Project(Tx, Ty, Tz, u, v);
    u = roundf(u);
    v = roundf(v);
    if (Tz < 0 || !InBoundary2D(u, v, m.rows, m.cols)) return false;
    const int ui = (int)u, vi = (int)v;
    const int64_t t = (int64_t)vi * m.cols + ui;
    const float tx = m.target_vertex[3 * t + 0];
    const float nx = m.target_normal[3 * t + 0];
    if (IsNan(tx) || IsNan(nx)) return false;
    const float ty = m.target_vertex[3 * t + 1];
    const float tz = m.target_vertex[3 * t + 2];
    const float ny = m.target_normal[3 * t + 1];
    const float nz = m.target_normal[3 * t + 2];
    r = (Tx - tx) * nx + (Ty - ty) * ny + (Tz - tz) * nz;
    if (fabsf(r) > trunc) return false;
    J[0] = -Tz * ny + Ty * nz;
    J[1] = Tz * nx - Tx * nz;
    J[2] = -Ty * nx + Tx * ny;
    J[3] = nx;
    J[4] = ny;
    J[5] = nz;
    return true;
}

// Shared head of :164-233 and :235-336.
__device__ __forceinline__ bool ProjectToTargetDepth(
        const OdoMaps& m, const Camera& ti, int64_t w, float trunc, float& Tx,
        float& Ty, float& Tz, int64_t& t, float& diff_D) {
    const float sx = m.source_vertex[3 * w + 0];
    if (IsNan(sx)) return false;
    const float sy = m.source_vertex[3 * w + 1];
    const float sz = m.source_vertex[3 * w + 2];
    float u_tf, v_tf;
    ti.RigidTransform(sx, sy, sz, Tx, Ty, Tz);
    ti.Project(Tx, Ty, Tz, u_tf, v_tf);
    const int u_t = ToIntX86(roundf(u_tf));
    const int v_t = ToIntX86(roundf(v_tf));
    if (Tz < 0 || !InBoundary2D((float)u_t, (float)v_t, m.rows, m.cols))
        return false;
    t = (int64_t)v_t * m.cols + u_t;
    const float depth_t = m.target_depth[t];
    diff_D = depth_t - Tz;
    if (IsNan(depth_t) || fabsf(diff_D) > trunc) return false;
    return true;
}

// :164-233
__device__ __forceinline__ bool JacobianIntensity(const OdoMaps& m,
                                                  const Camera& ti, int64_t w,
                                                  float trunc, float* J_I,
                                                  float& r_I) {
    const float sobel_scale = 0.125f;
    float Tx, Ty, Tz, diff_D;
    int64_t t;
    if (!ProjectToTargetDepth(m, ti, w, trunc, Tx, Ty, Tz, t, diff_D))
        return false;
    const float diff_I = m.target_intensity[t] - m.source_intensity[w];
    const float dIdx = sobel_scale * m.target_intensity_dx[t];
    const float dIdy = sobel_scale * m.target_intensity_dy[t];
    const float invz = 1 / Tz;
    const float c0 = dIdx * ti.fx * invz;
    const float c1 = dIdy * ti.fy * invz;
    const float c2 = -(c0 * Tx + c1 * Ty) * invz;
    J_I[0] = (-Tz * c1 + Ty * c2);
    J_I[1] = (Tz * c0 - Tx * c2);
    J_I[2] = (-Ty * c0 + Tx * c1);
    J_I[3] = c0;
    J_I[4] = c1;
    J_I[5] = c2;
    r_I = diff_I;
    return true;
}

// :235-336
__device__ __forceinline__ bool JacobianHybrid(const OdoMaps& m,
                                               const Camera& ti, int64_t w,
                                               float trunc, float* J_I,
                                               float* J_D, float& r_I,
                                               float& r_D) {
    const float sqrt_lambda_intensity = 0.707f;
    const float sqrt_lambda_depth = 0.707f;
    const float sobel_scale = 0.125f;
    float Tx, Ty, Tz, diff_D;
    int64_t t;
    if (!ProjectToTargetDepth(m, ti, w, trunc, Tx, Ty, Tz, t, diff_D))
        return false;
    const float dDdx = sobel_scale * m.target_depth_dx[t];
    const float dDdy = sobel_scale * m.target_depth_dy[t];
    if (IsNan(dDdx) || IsNan(dDdy)) return false;
    const float diff_I = m.target_intensity[t] - m.source_intensity[w];
    const float dIdx = sobel_scale * m.target_intensity_dx[t];
    const float dIdy = sobel_scale * m.target_intensity_dy[t];
    const float invz = 1 / Tz;
    const float c0 = dIdx * ti.fx * invz;
    const float c1 = dIdy * ti.fy * invz;
    const float c2 = -(c0 * Tx + c1 * Ty) * invz;
    const float d0 = dDdx * ti.fx * invz;
    const float d1 = dDdy * ti.fy * invz;
    const float d2 = -(d0 * Tx + d1 * Ty) * invz;
    J_I[0] = sqrt_lambda_intensity * (-Tz * c1 + Ty * c2);
    J_I[1] = sqrt_lambda_intensity * (Tz * c0 - Tx * c2);
    J_I[2] = sqrt_lambda_intensity * (-Ty * c0 + Tx * c1);
    J_I[3] = sqrt_lambda_intensity * (c0);
    J_I[4] = sqrt_lambda_intensity * (c1);
    J_I[5] = sqrt_lambda_intensity * (c2);
    r_I = sqrt_lambda_intensity * diff_I;
    J_D[0] = sqrt_lambda_depth * ((-Tz * d1 + Ty * d2) - Ty);
    J_D[1] = sqrt_lambda_depth * ((Tz * d0 - Tx * d2) + Tx);
    J_D[2] = sqrt_lambda_depth * ((-Ty * d0 + Tx * d1));
    J_D[3] = sqrt_lambda_depth * (d0);
    J_D[4] = sqrt_lambda_depth * (d1);
    J_D[5] = sqrt_lambda_depth * (d2 - 1.0f);
    r_D = sqrt_lambda_depth * diff_D;
    return true;
}

constexpr int kOdoSums = 29;

// RGBDOdometryCPU.cpp:98-364 (the three reductions): the terms of one pixel.
template <int METHOD>
__device__ __forceinline__ void AccumulatePixel(double (&A)[kOdoSums],
                                                const OdoMaps& m,
                                                const Camera& ti, int64_t w,
                                                float trunc, float depth_delta,
                                                float intensity_delta) {
    if (METHOD == 0 || METHOD == 1) {
        float J[6], r;
        const bool valid = METHOD == 0
                                   ? JacobianPointToPlane(m, ti, w, trunc, J, r)
                                   : JacobianIntensity(m, ti, w, trunc, J, r);
        if (!valid) return;
        const float delta = METHOD == 0 ? depth_delta : intensity_delta;
        const float d_huber = HuberDeriv(r, delta);
        const float r_huber = HuberLoss(r, delta);
        int i = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int k = 0; k <= j; ++k) {
                A[i] += (double)(J[j] * J[k]);
                ++i;
            }
            A[21 + j] += (double)(J[j] * d_huber);
        }
        A[27] += (double)r_huber;
        A[28] += 1.0;
    } else {
        float J_I[6], J_D[6], r_I, r_D;
        if (!JacobianHybrid(m, ti, w, trunc, J_I, J_D, r_I, r_D)) return;
        const float d_huber_I = HuberDeriv(r_I, intensity_delta);
        const float d_huber_D = HuberDeriv(r_D, depth_delta);
        const float r_huber_I = HuberLoss(r_I, intensity_delta);
        const float r_huber_D = HuberLoss(r_D, depth_delta);
        int i = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int k = 0; k <= j; ++k) {
                A[i] += (double)(J_I[j] * J_I[k] + J_D[j] * J_D[k]);
                ++i;
            }
            A[21 + j] += (double)(J_I[j] * d_huber_I + J_D[j] * d_huber_D);
        }
        A[27] += (double)(r_huber_I + r_huber_D);
        A[28] += 1.0;
    }
}

template <int METHOD>
__global__ void __launch_bounds__(kSumsBlock)
OdometrySumsKernel(OdoMaps m, Camera ti, float trunc, float depth_delta,
                   float intensity_delta, double* __restrict__ partials) {
    double A[kOdoSums];
#pragma unroll
    for (int k = 0; k < kOdoSums; ++k) A[k] = 0;
    const int64_t n = (int64_t)m.rows * m.cols;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x)
        AccumulatePixel<METHOD>(A, m, ti, w, trunc, depth_delta,
                                intensity_delta);
    BlockSumAndStore<kOdoSums>(A, partials);
}

constexpr int kInfoSums = 21;

// ComputeOdometryInformationMatrixCPU (RGBDOdometryCPU.cpp:26-96) with
// GetJacobianPointToPoint (RGBDOdometryJacobianImpl.h:39-104).
__global__ void __launch_bounds__(kSumsBlock)
InformationKernel(const float* __restrict__ source_vertex,
                  const float* __restrict__ target_vertex, int rows, int cols,
                  Camera ti, float square_dist_thr,
                  double* __restrict__ partials) {
    double A[kInfoSums];
#pragma unroll
    for (int k = 0; k < kInfoSums; ++k) A[k] = 0;
    const int64_t n = (int64_t)rows * cols;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n;
         w += (int64_t)gridDim.x * blockDim.x) {
        const float sx = source_vertex[3 * w + 0];
        if (IsNan(sx)) continue;
        float Tx, Ty, Tz, u, v;
        ti.RigidTransform(sx, source_vertex[3 * w + 1], source_vertex[3 * w + 2],
                          Tx, Ty, Tz);
        ti.Project(Tx, Ty, Tz, u, v);
        u = roundf(u);
        v = roundf(v);
        if (Tz < 0 || !InBoundary2D(u, v, rows, cols)) continue;
        const int64_t t = (int64_t)(int)v * cols + (int)u;
        const float tx = target_vertex[3 * t + 0];
        if (IsNan(tx)) continue;
        const float rx = Tx - tx;
        const float ry = Ty - target_vertex[3 * t + 1];
        const float rz = Tz - target_vertex[3 * t + 2];
        const float r2 = rx * rx + ry * ry + rz * rz;
        if (r2 > square_dist_thr) continue;
        const float J_x[6] = {0.0f, Tz, -Ty, 1.0f, 0.0f, 0.0f};
        const float J_y[6] = {-Tz, 0.0f, Tx, 0.0f, 1.0f, 0.0f};
        const float J_z[6] = {Ty, -Tx, 0.0f, 0.0f, 0.0f, 1.0f};
        int i = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int k = 0; k <= j; ++k) {
                A[i] += (double)(J_x[j] * J_x[k]);
                A[i] += (double)(J_y[j] * J_y[k]);
                A[i] += (double)(J_z[j] * J_z[k]);
                ++i;
            }
        }
    }
    BlockSumAndStore<kInfoSums>(A, partials);
}

int CheckImage(const void* p, int rows, int cols) {
    O3DMI_REQUIRE(rows >= 0 && cols >= 0, "negative image size");
    O3DMI_REQUIRE(p != nullptr || (int64_t)rows * cols == 0, "image is null");
    return O3DMI_OK;
}

BilateralParams MakeBilateral(int kernel_size, float value_sigma,
                              float distance_sigma) {
    BilateralParams bp;
    bp.radius = kernel_size / 2;
    const float val_sqr = value_sigma * value_sigma;
    const float pos_sqr = distance_sigma * distance_sigma;
    bp.two_val_sqr = 2.0f * val_sqr;
    for (int d2 = 0; d2 <= 8; ++d2)
        bp.space_w[d2] =
                (float)std::exp((double)(-(float)d2 / (2.0f * pos_sqr)));
    return bp;
}

GaussTaps MakeGauss(int kernel_size, float sigma) {
    GaussTaps gt;
    gt.radius = kernel_size / 2;
    float sum = 0;
    for (int i = 0; i < kernel_size; ++i) {
        const float d = static_cast<float>(i - kernel_size / 2);
        gt.w[i] = (float)std::exp((double)((d * d) * (-0.5f / (sigma * sigma))));
        sum += gt.w[i];
    }
    for (int i = 0; i < kernel_size; ++i) gt.w[i] = gt.w[i] / sum;
    return gt;
}

const double kEye4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

}  // namespace
}  // namespace o3dmi

using namespace o3dmi;

extern "C" {

int o3dmi_image_clip_transform(const void* src_dev, int src_dtype, int rows,
                               int cols, float scale, float min_value,
                               float max_value, float clip_fill, float* dst_dev,
                               o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    O3DMI_REQUIRE(src_dtype == O3DMI_U16 || src_dtype == O3DMI_F32,
                  "ClipTransform: depth must be UInt16 or Float32");
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst_dev != nullptr, "dst is null");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    if (src_dtype == O3DMI_U16)
        hipLaunchKernelGGL(ClipTransformKernel<uint16_t>, grid, block, 0, s,
                           (const uint16_t*)src_dev, dst_dev, n, scale,
                           min_value, max_value, clip_fill);
    else
        hipLaunchKernelGGL(ClipTransformKernel<float>, grid, block, 0, s,
                           (const float*)src_dev, dst_dev, n, scale, min_value,
                           max_value, clip_fill);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_pyrdown_depth(const float* src_dev, int rows, int cols,
                              float depth_diff, float invalid_fill,
                              float* dst_dev, o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    const int64_t n = (int64_t)(rows / 2) * (cols / 2);
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst_dev != nullptr, "dst is null");
    hipLaunchKernelGGL(PyrDownDepthKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, src_dev, dst_dev,
                       rows, cols, depth_diff, invalid_fill);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_create_vertex_map(const float* src_dev, int rows, int cols,
                                  const double* intrinsics, float invalid_fill,
                                  float* dst_dev, o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    O3DMI_REQUIRE(intrinsics != nullptr, "intrinsics is null");
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst_dev != nullptr, "dst is null");
    hipLaunchKernelGGL(CreateVertexMapKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, src_dev, dst_dev,
                       rows, cols, Camera::Make(intrinsics, kEye4),
                       invalid_fill, FillClass(invalid_fill));
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_create_normal_map(const float* src_dev, int rows, int cols,
                                  float invalid_fill, float* dst_dev,
                                  o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst_dev != nullptr, "dst is null");
    hipLaunchKernelGGL(CreateNormalMapKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, src_dev, dst_dev,
                       rows, cols, invalid_fill);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_to_float(const void* src_dev, int src_dtype, int64_t n,
                         double scale, double offset, float* dst_dev,
                         o3dmi_stream_t stream) {
    O3DMI_REQUIRE(n >= 0, "n < 0");
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(src_dev && dst_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n, kBlock)), block(kBlock);
    const float cs = (float)scale, co = (float)offset;
    if (src_dtype == O3DMI_U8)
        hipLaunchKernelGGL(ToFloatKernel<uint8_t>, grid, block, 0, s,
                           (const uint8_t*)src_dev, dst_dev, n, cs, co);
    else if (src_dtype == O3DMI_U16)
        hipLaunchKernelGGL(ToFloatKernel<uint16_t>, grid, block, 0, s,
                           (const uint16_t*)src_dev, dst_dev, n, cs, co);
    else if (src_dtype == O3DMI_F32)
        hipLaunchKernelGGL(ToFloatKernel<float>, grid, block, 0, s,
                           (const float*)src_dev, dst_dev, n, cs, co);
    else
        O3DMI_REQUIRE(false, "To: unsupported source dtype");
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_rgb_to_gray(const void* src_dev, int dtype, int64_t n_pixels,
                            void* dst_dev, o3dmi_stream_t stream) {
    O3DMI_REQUIRE(n_pixels >= 0, "n < 0");
    if (n_pixels == 0) return O3DMI_OK;
    O3DMI_REQUIRE(src_dev && dst_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n_pixels, kBlock)), block(kBlock);
    if (dtype == O3DMI_U8)
        hipLaunchKernelGGL(RGBToGrayKernel<uint8_t>, grid, block, 0, s,
                           (const uint8_t*)src_dev, (uint8_t*)dst_dev, n_pixels,
                           255.f);
    else if (dtype == O3DMI_U16)
        hipLaunchKernelGGL(RGBToGrayKernel<uint16_t>, grid, block, 0, s,
                           (const uint16_t*)src_dev, (uint16_t*)dst_dev,
                           n_pixels, 65535.f);
    else if (dtype == O3DMI_F32)
        hipLaunchKernelGGL(RGBToGrayKernel<float>, grid, block, 0, s,
                           (const float*)src_dev, (float*)dst_dev, n_pixels,
                           0.f);
    else
        O3DMI_REQUIRE(false, "RGBToGray: unsupported dtype");
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_rgb_to_intensity(const void* src_dev, int dtype,
                                 int64_t n_pixels, float* dst_dev,
                                 o3dmi_stream_t stream) {
    O3DMI_REQUIRE(n_pixels >= 0, "n < 0");
    if (n_pixels == 0) return O3DMI_OK;
    O3DMI_REQUIRE(src_dev && dst_dev, "null argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(GridFor(n_pixels, kBlock)), block(kBlock);
    if (dtype == O3DMI_U8)
        hipLaunchKernelGGL(RGB8ToIntensityKernel, grid, block, 0, s,
                           (const uint8_t*)src_dev, dst_dev, n_pixels,
                           (float)(1. / 255));
    else if (dtype == O3DMI_F32)
        hipLaunchKernelGGL(RGBToGrayKernel<float>, grid, block, 0, s,
                           (const float*)src_dev, dst_dev, n_pixels, 0.f);
    else
        O3DMI_REQUIRE(false, "colour must be UInt8 or Float32");
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_filter_bilateral(const float* src_dev, int rows, int cols,
                                 int kernel_size, float value_sigma,
                                 float distance_sigma, float* dst_dev,
                                 o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    O3DMI_REQUIRE(kernel_size == 3 || kernel_size == 5,
                  "FilterBilateral: kernel size must be 3 or 5");
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst_dev != nullptr && dst_dev != src_dev,
                  "dst is null or aliases src");
    hipLaunchKernelGGL(FilterBilateralKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, src_dev, dst_dev,
                       rows, cols,
                       MakeBilateral(kernel_size, value_sigma, distance_sigma));
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_filter_gaussian(const float* src_dev, int rows, int cols,
                                int kernel_size, float sigma, float* dst_dev,
                                o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    O3DMI_REQUIRE(kernel_size == 3 || kernel_size == 5 || kernel_size == 7,
                  "FilterGaussian: kernel size must be 3, 5 or 7");
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst_dev != nullptr && dst_dev != src_dev,
                  "dst is null or aliases src");
    hipLaunchKernelGGL(FilterGaussianKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, src_dev, dst_dev,
                       rows, cols, rows, cols, 1, MakeGauss(kernel_size, sigma));
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_filter_sobel(const float* src_dev, int rows, int cols,
                             float* dx_dev, float* dy_dev,
                             o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dx_dev && dy_dev, "null output");
    hipLaunchKernelGGL(FilterSobel3Kernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, src_dev, dx_dev,
                       dy_dev, rows, cols);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_resize_half_nearest(const float* src_dev, int rows, int cols,
                                    float* dst_dev, o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    const int out_rows = (int)(rows * 0.5f), out_cols = (int)(cols * 0.5f);
    const int64_t n = (int64_t)out_rows * out_cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst_dev != nullptr, "dst is null");
    hipLaunchKernelGGL(ResizeHalfNearestKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, src_dev, dst_dev,
                       cols, out_rows, out_cols);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_pyrdown(const float* src_dev, int rows, int cols,
                        float* dst_dev, o3dmi_stream_t stream) {
    int st = CheckImage(src_dev, rows, cols);
    if (st) return st;
    const int out_rows = (int)(rows * 0.5f), out_cols = (int)(cols * 0.5f);
    const int64_t n = (int64_t)out_rows * out_cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst_dev != nullptr, "dst is null");
    hipLaunchKernelGGL(FilterGaussianKernel, dim3(GridFor(n, kBlock)),
                       dim3(kBlock), 0, (hipStream_t)stream, src_dev, dst_dev,
                       rows, cols, out_rows, out_cols, 2, MakeGauss(5, 1.0f));
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_odometry_p2plane_level(const float* source_depth_dev,
                                 const float* target_depth_dev, int rows,
                                 int cols, const double* intrinsics,
                                 float* source_vertex_dev,
                                 float* target_vertex_dev,
                                 float* target_normal_dev,
                                 float* source_depth_next_dev,
                                 float* target_depth_next_dev,
                                 float depth_diff, o3dmi_stream_t stream) {
    int st = CheckImage(source_depth_dev, rows, cols);
    if (st) return st;
    O3DMI_REQUIRE(intrinsics != nullptr, "intrinsics is null");
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(target_depth_dev && source_vertex_dev && target_vertex_dev &&
                          target_normal_dev,
                  "null argument");
    O3DMI_REQUIRE((source_depth_next_dev == nullptr) ==
                          (target_depth_next_dev == nullptr),
                  "next-level depth outputs come in pairs");
    const dim3 tiles((cols + kTileW - 1) / kTileW, (rows + kTileH - 1) / kTileH);
    hipLaunchKernelGGL(P2PlaneLevelKernel, tiles, dim3(kTileW * kTileH), 0,
                       (hipStream_t)stream, source_depth_dev,
                       target_depth_dev, source_vertex_dev, target_vertex_dev,
                       target_normal_dev, rows, cols,
                       Camera::Make(intrinsics, kEye4),
                       MakeBilateral(5, 5.0f, 10.0f), source_depth_next_dev,
                       target_depth_next_dev, depth_diff);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_image_clip_transform_pair(const void* src0_dev, int src0_dtype,
                                    const void* src1_dev, int src1_dtype,
                                    int rows, int cols, float scale,
                                    float min_value, float max_value,
                                    float clip_fill, float* dst0_dev,
                                    float* dst1_dev, o3dmi_stream_t stream) {
    int st = CheckImage(src0_dev, rows, cols);
    if (st) return st;
    if ((st = CheckImage(src1_dev, rows, cols))) return st;
    O3DMI_REQUIRE((src0_dtype == O3DMI_U16 || src0_dtype == O3DMI_F32) &&
                          (src1_dtype == O3DMI_U16 || src1_dtype == O3DMI_F32),
                  "ClipTransform: depth must be UInt16 or Float32");
    const int64_t n = (int64_t)rows * cols;
    if (n == 0) return O3DMI_OK;
    O3DMI_REQUIRE(dst0_dev && dst1_dev, "dst is null");
    hipLaunchKernelGGL(ClipTransformPairKernel, dim3(GridFor(n, kBlock), 2),
                       dim3(kBlock), 0, (hipStream_t)stream, src0_dev,
                       (int)(src0_dtype == O3DMI_F32), src1_dev,
                       (int)(src1_dtype == O3DMI_F32), dst0_dev, dst1_dev, n,
                       scale, min_value, max_value, clip_fill);
    O3DMI_HIP_CHECK(hipGetLastError());
    return O3DMI_OK;
}

int o3dmi_odometry_sums_scratch_doubles(void) {
    return kSumsMaxGrid * kOdoSums;
}

// Internal (not in the public header): the same reduction, with the result
// also posted to a host mailbox (mailbox.h) when mail_data != NULL.
int o3dmi_odometry_sums_post(int method, int rows, int cols,
                             const float* const* maps11,
                             const double* intrinsics,
                             const double* init_source_to_target,
                             float depth_outlier_trunc, float depth_huber_delta,
                             float intensity_huber_delta, double* scratch_dev,
                             double* sums29_dev, double* mail_data,
                             int* mail_flag, int mail_seq,
                             o3dmi_stream_t stream);

int o3dmi_odometry_sums(int method, int rows, int cols,
                        const float* source_depth_dev,
                        const float* target_depth_dev,
                        const float* source_intensity_dev,
                        const float* target_intensity_dev,
                        const float* target_depth_dx_dev,
                        const float* target_depth_dy_dev,
                        const float* target_intensity_dx_dev,
                        const float* target_intensity_dy_dev,
                        const float* source_vertex_dev,
                        const float* target_vertex_dev,
                        const float* target_normal_dev,
                        const double* intrinsics,
                        const double* init_source_to_target,
                        float depth_outlier_trunc, float depth_huber_delta,
                        float intensity_huber_delta, double* scratch_dev,
                        double* sums29_dev, o3dmi_stream_t stream) {
    const float* maps[11] = {source_depth_dev,        target_depth_dev,
                             source_intensity_dev,    target_intensity_dev,
                             target_depth_dx_dev,     target_depth_dy_dev,
                             target_intensity_dx_dev, target_intensity_dy_dev,
                             source_vertex_dev,       target_vertex_dev,
                             target_normal_dev};
    O3DMI_REQUIRE(sums29_dev != nullptr, "null argument");
    return o3dmi_odometry_sums_post(method, rows, cols, maps, intrinsics,
                                    init_source_to_target, depth_outlier_trunc,
                                    depth_huber_delta, intensity_huber_delta,
                                    scratch_dev, sums29_dev, nullptr, nullptr,
                                    0, stream);
}

int o3dmi_odometry_sums_post(int method, int rows, int cols,
                             const float* const* maps11,
                             const double* intrinsics,
                             const double* init_source_to_target,
                             float depth_outlier_trunc, float depth_huber_delta,
                             float intensity_huber_delta, double* scratch_dev,
                             double* sums29_dev, double* mail_data,
                             int* mail_flag, int mail_seq,
                             o3dmi_stream_t stream) {
    const float* source_depth_dev = maps11[0];
    const float* target_depth_dev = maps11[1];
    const float* source_intensity_dev = maps11[2];
    const float* target_intensity_dev = maps11[3];
    const float* target_depth_dx_dev = maps11[4];
    const float* target_depth_dy_dev = maps11[5];
    const float* target_intensity_dx_dev = maps11[6];
    const float* target_intensity_dy_dev = maps11[7];
    const float* source_vertex_dev = maps11[8];
    const float* target_vertex_dev = maps11[9];
    const float* target_normal_dev = maps11[10];
    O3DMI_REQUIRE(method >= 0 && method <= 2, "Odometry method not implemented.");
    O3DMI_REQUIRE(rows > 0 && cols > 0, "empty image");
    O3DMI_REQUIRE(intrinsics && init_source_to_target && source_vertex_dev &&
                          (sums29_dev || mail_data),
                  "null argument");
    if (method == O3DMI_ODOMETRY_POINT_TO_PLANE) {
        O3DMI_REQUIRE(target_vertex_dev && target_normal_dev,
                      "point-to-plane needs target vertex and normal maps");
    } else {
        O3DMI_REQUIRE(target_depth_dev && source_intensity_dev &&
                              target_intensity_dev && target_intensity_dx_dev &&
                              target_intensity_dy_dev,
                      "intensity terms need depth / intensity / gradient maps");
        if (method == O3DMI_ODOMETRY_HYBRID)
            O3DMI_REQUIRE(target_depth_dx_dev && target_depth_dy_dev,
                          "hybrid needs target depth gradients");
    }
    hipStream_t s = (hipStream_t)stream;
    OdoMaps m;
    m.source_depth = source_depth_dev;
    m.target_depth = target_depth_dev;
    m.source_intensity = source_intensity_dev;
    m.target_intensity = target_intensity_dev;
    m.target_depth_dx = target_depth_dx_dev;
    m.target_depth_dy = target_depth_dy_dev;
    m.target_intensity_dx = target_intensity_dx_dev;
    m.target_intensity_dy = target_intensity_dy_dev;
    m.source_vertex = source_vertex_dev;
    m.target_vertex = target_vertex_dev;
    m.target_normal = target_normal_dev;
    m.rows = rows;
    m.cols = cols;
    const Camera ti = Camera::Make(intrinsics, init_source_to_target);
    const int g = SumsGrid((int64_t)rows * cols);
    double* partials = scratch_dev;
    bool own = false;
    if (!partials) {
        int st = PoolAlloc((void**)&partials,
                           sizeof(double) * kSumsMaxGrid * kOdoSums);
        if (st) return st;
        own = true;
    }
    if (method == 0)
        hipLaunchKernelGGL(OdometrySumsKernel<0>, dim3(g), dim3(kSumsBlock), 0,
                           s, m, ti, depth_outlier_trunc, depth_huber_delta,
                           intensity_huber_delta, partials);
    else if (method == 1)
        hipLaunchKernelGGL(OdometrySumsKernel<1>, dim3(g), dim3(kSumsBlock), 0,
                           s, m, ti, depth_outlier_trunc, depth_huber_delta,
                           intensity_huber_delta, partials);
    else
        hipLaunchKernelGGL(OdometrySumsKernel<2>, dim3(g), dim3(kSumsBlock), 0,
                           s, m, ti, depth_outlier_trunc, depth_huber_delta,
                           intensity_huber_delta, partials);
    hipLaunchKernelGGL(FinalSumKernel<kOdoSums>, dim3(1), dim3(kFinalThreads), 0, s,
                       partials, g, sums29_dev, mail_data, mail_flag, mail_seq);
    hipError_t e = hipGetLastError();
    if (own) {
        // The pool hands the block out again only to later work; drain first.
        (void)hipStreamSynchronize(s);
        PoolFree(partials);
    }
    O3DMI_HIP_CHECK(e);
    return O3DMI_OK;
}

int o3dmi_odometry_information(int rows, int cols,
                               const float* source_vertex_dev,
                               const float* target_vertex_dev,
                               const double* intrinsics,
                               const double* source_to_target,
                               float square_dist_thr, double* information_host,
                               o3dmi_stream_t stream) {
    O3DMI_REQUIRE(rows > 0 && cols > 0, "empty image");
    O3DMI_REQUIRE(source_vertex_dev && target_vertex_dev && intrinsics &&
                          source_to_target && information_host,
                  "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int g = SumsGrid((int64_t)rows * cols);
    double* partials = nullptr;
    int st = PoolAlloc((void**)&partials,
                       sizeof(double) * ((size_t)kSumsMaxGrid * kInfoSums + 32));
    if (st) return st;
    double* out_dev = partials + (size_t)kSumsMaxGrid * kInfoSums;
    hipLaunchKernelGGL(InformationKernel, dim3(g), dim3(kSumsBlock), 0, s,
                       source_vertex_dev, target_vertex_dev, rows, cols,
                       Camera::Make(intrinsics, source_to_target),
                       square_dist_thr, partials);
    hipLaunchKernelGGL(FinalSumKernel<kInfoSums>, dim3(1), dim3(kFinalThreads), 0, s,
                       partials, g, out_dev, (double*)nullptr, (int*)nullptr, 0);
    double A[kInfoSums];
    hipError_t e = hipMemcpyAsync(A, out_dev, sizeof(A), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    else (void)hipStreamSynchronize(s);
    PoolFree(partials);
    O3DMI_HIP_CHECK(e);
    for (int j = 0; j < 6; j++) {
        const int reduction_idx = (j * (j + 1)) / 2;
        for (int k = 0; k <= j; k++) {
            information_host[j * 6 + k] = A[reduction_idx + k];
            information_host[k * 6 + j] = A[reduction_idx + k];
        }
    }
    return O3DMI_OK;
}

}  // extern "C"
