// TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the Open3D CPU
// tensor path for the dense-SLAM hot loop. Nothing under oracle/ is shipped or
// measured as product; only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.
//
// Shared helpers: float copies of the camera model exactly as the reference
// keeps them (cpp/open3d/t/geometry/kernel/GeometryIndexer.h:25-144) and the
// rigid inverse (cpp/open3d/t/geometry/Utility.h:77-120).
//
// Build flags (oracle/Makefile): -O2 -ffp-contract=off, no -march, no
// fast-math -- same as the reference's defaults
// (cmake/Open3DSetGlobalProperties.cmake sets none of these).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

// GeometryIndexer.h:25-144. Members are float copies of the F64 inputs.
struct TransformIndexer {
    float extrinsic_[3][4];
    float fx_, fy_, cx_, cy_;
    float scale_;

    TransformIndexer(const double* intrinsics /*3x3*/,
                     const double* extrinsics /*4x4*/,
                     float scale = 1.0f) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j)
                extrinsic_[i][j] = (float)extrinsics[i * 4 + j];
        fx_ = (float)intrinsics[0 * 3 + 0];
        fy_ = (float)intrinsics[1 * 3 + 1];
        cx_ = (float)intrinsics[0 * 3 + 2];
        cy_ = (float)intrinsics[1 * 3 + 2];
        scale_ = scale;
    }

    // GeometryIndexer.h:62-78
    void RigidTransform(float x_in, float y_in, float z_in, float* x_out,
                        float* y_out, float* z_out) const {
        x_in *= scale_;
        y_in *= scale_;
        z_in *= scale_;
        *x_out = x_in * extrinsic_[0][0] + y_in * extrinsic_[0][1] +
                 z_in * extrinsic_[0][2] + extrinsic_[0][3];
        *y_out = x_in * extrinsic_[1][0] + y_in * extrinsic_[1][1] +
                 z_in * extrinsic_[1][2] + extrinsic_[1][3];
        *z_out = x_in * extrinsic_[2][0] + y_in * extrinsic_[2][1] +
                 z_in * extrinsic_[2][2] + extrinsic_[2][3];
    }

    // GeometryIndexer.h:81-97
    void Rotate(float x_in, float y_in, float z_in, float* x_out, float* y_out,
                float* z_out) const {
        x_in *= scale_;
        y_in *= scale_;
        z_in *= scale_;
        *x_out = x_in * extrinsic_[0][0] + y_in * extrinsic_[0][1] +
                 z_in * extrinsic_[0][2];
        *y_out = x_in * extrinsic_[1][0] + y_in * extrinsic_[1][1] +
                 z_in * extrinsic_[1][2];
        *z_out = x_in * extrinsic_[2][0] + y_in * extrinsic_[2][1] +
                 z_in * extrinsic_[2][2];
    }

    // GeometryIndexer.h:100-108
    void Project(float x_in, float y_in, float z_in, float* u_out,
                 float* v_out) const {
        float inv_z = 1.0f / z_in;
        *u_out = fx_ * x_in * inv_z + cx_;
        *v_out = fy_ * y_in * inv_z + cy_;
    }

    // GeometryIndexer.h:111-120
    void Unproject(float u_in, float v_in, float d_in, float* x_out,
                   float* y_out, float* z_out) const {
        *x_out = (u_in - cx_) * d_in / fx_;
        *y_out = (v_in - cy_) * d_in / fy_;
        *z_out = d_in;
    }

    void GetCameraPosition(float* x, float* y, float* z) const {
        *x = extrinsic_[0][3];
        *y = extrinsic_[1][3];
        *z = extrinsic_[2][3];
    }
};

// t/geometry/Utility.h:77-120 (double precision, R^T and -R^T t).
inline void InverseTransformation(const double* T, double* Tinv) {
    Tinv[0] = T[0];  Tinv[1] = T[4];  Tinv[2] = T[8];
    Tinv[4] = T[1];  Tinv[5] = T[5];  Tinv[6] = T[9];
    Tinv[8] = T[2];  Tinv[9] = T[6];  Tinv[10] = T[10];
    Tinv[3] = -(Tinv[0] * T[3] + Tinv[1] * T[7] + Tinv[2] * T[11]);
    Tinv[7] = -(Tinv[4] * T[3] + Tinv[5] * T[7] + Tinv[6] * T[11]);
    Tinv[11] = -(Tinv[8] * T[3] + Tinv[9] * T[7] + Tinv[10] * T[11]);
    Tinv[12] = 0; Tinv[13] = 0; Tinv[14] = 0; Tinv[15] = 1;
}

// 2-D image bounds test, TArrayIndexer::InBoundary (GeometryIndexer.h:294-297)
// with shape_[0]=rows, shape_[1]=cols.
inline bool InBoundary2D(float x, float y, int rows, int cols) {
    return y >= 0 && x >= 0 && y <= rows - 1.0f && x <= cols - 1.0f;
}

inline int Sign(int x) { return x < 0 ? -1 : (x > 0 ? 1 : 0); }

}  // namespace orc
