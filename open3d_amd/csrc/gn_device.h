// Device-side pieces of a Gauss-Newton step shared by the RGB-D odometry's
// persistent kernel (odometry.hip) and the ICP search kernel's in-launch tail
// (icp.hip): the 6x6 solve by one wave and the pose -> transformation map.
#pragma once

#include <hip/hip_runtime.h>

namespace o3dmi {

// o3dmi_decode_and_solve6x6 (TransformationConverter.cpp:189-226 + partial-
// pivot LU) by ONE WAVE: lane (i, j) = (lane / 8, lane % 8) holds entry (i, j)
// of the 6 x 7 augmented matrix [M | b] in a register, row operations are
// shuffles. Every entry sees exactly the multiply / subtract sequence of the
// sequential host code (a single lane running that code in float64 takes
// ~12 us -- longer than the rest of an iteration). All lanes return the same
// status (0 ok, 2 singular) and the same pose.
__device__ inline int GnSolveWave(const double* A, int lane, double (&x)[6]) {
    const int i = lane >> 3, j = lane & 7;
    double m = 0;
    if (i < 6 && j < 6) {
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        m = A[(hi * (hi + 1)) / 2 + lo];
    } else if (i < 6 && j == 6) {
        m = -A[21 + i];
    }
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double mx = fabs(__shfl(m, k * 8 + k, 64));
        for (int r = k + 1; r < 6; ++r) {
            const double v = fabs(__shfl(m, r * 8 + k, 64));
            if (v > mx) {
                mx = v;
                p = r;
            }
        }
        if (mx == 0.0 || !(mx == mx)) return 2;
        // swap rows k and p (all columns, b included)
        const int src = (i == k) ? p * 8 + j : ((i == p) ? k * 8 + j : lane);
        m = __shfl(m, src, 64);
        const double pivot = __shfl(m, k * 8 + k, 64);
        const double mik = __shfl(m, i * 8 + k, 64);
        const double mkj = __shfl(m, k * 8 + j, 64);
        if (i > k && i < 6) {
            const double l = mik / pivot;
            if (j == k) m = l;
            else if (j > k && j <= 6) m = m - l * mkj;
        }
    }
    // back substitution (forward substitution of b happened with the
    // elimination steps, in the same order)
    for (int r = 5; r >= 0; --r) {
        double br = __shfl(m, r * 8 + 6, 64);
        for (int c = r + 1; c < 6; ++c) br -= __shfl(m, r * 8 + c, 64) * x[c];
        x[r] = br / __shfl(m, r * 8 + r, 64);
    }
    return 0;
}

// PoseToTransformationImpl (t/pipelines/kernel/TransformationConverterImpl.h:
// 23-42): R = Rz(gamma) Ry(beta) Rx(alpha) from pose[0..2], t = pose[3..5],
// row-major 4x4. sc[0..2] = sin, sc[3..5] = cos of (alpha, beta, gamma).
__device__ __forceinline__ void PoseToTransformationDevice(const double* pose,
                                                           const double* sc,
                                                           double* T) {
    const double s0 = sc[0], s1 = sc[1], s2 = sc[2];
    const double c0 = sc[3], c1 = sc[4], c2 = sc[5];
    for (int q = 0; q < 16; ++q) T[q] = 0;
    T[0] = c2 * c1;
    T[1] = -1 * s2 * c0 + c2 * s1 * s0;
    T[2] = s2 * s0 + c2 * s1 * c0;
    T[4] = s2 * c1;
    T[5] = c2 * c0 + s2 * s1 * s0;
    T[6] = -1 * c2 * s0 + s2 * s1 * c0;
    T[8] = -1 * s1;
    T[9] = c1 * s0;
    T[10] = c1 * c0;
    T[3] = pose[3];
    T[7] = pose[4];
    T[11] = pose[5];
    T[15] = 1;
}

}  // namespace o3dmi
